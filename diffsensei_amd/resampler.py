"""Host mirror of reference src/models/resampler.py `Resampler` (perceiver resampler that turns CLIP-H tokens +
one Magi token per character into 16 tokens per character, and prepends 16 learned dummy tokens).

Same constructor arguments, state-dict keys (`latents`, `proj_in.*`, `proj_in_magi.*`, `layers.{i}.0.{norm1,norm2,
to_q,to_kv,to_out}.*`, `layers.{i}.1.{0,1,3}.*`, `proj_out.*`, `norm_out.*`, `dummy_tokens`), `forward(x, magi)` and
`dtype()`.  Arithmetic = HIP kernels only (GEMM / LayerNorm / small attention); torch is used for buffer assembly.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops

Tensor = torch.Tensor


class Resampler:
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=4, num_dummy_tokens=4, embedding_dim=768,
                 magi_embedding_dim=512, output_dim=1024, ff_mult=4, device="cuda"):
        self.dim, self.depth, self.dim_head, self.heads = dim, depth, dim_head, heads
        self.num_queries, self.num_dummy_tokens = num_queries, num_dummy_tokens
        self.embedding_dim, self.magi_embedding_dim = embedding_dim, magi_embedding_dim
        self.output_dim, self.ff_mult = output_dim, ff_mult
        self.device = torch.device(device)
        self._sd: Dict[str, Tensor] = {}

    # ---- weights
    def param_shapes(self) -> Dict[str, tuple]:
        d, inner, ffi = self.dim, self.dim_head * self.heads, int(self.dim * self.ff_mult)
        p = {"latents": (1, self.num_queries, d), "proj_in.weight": (d, self.embedding_dim), "proj_in.bias": (d,),
             "proj_in_magi.weight": (d, self.magi_embedding_dim), "proj_in_magi.bias": (d,),
             "proj_out.weight": (self.output_dim, d), "proj_out.bias": (self.output_dim,),
             "norm_out.weight": (self.output_dim,), "norm_out.bias": (self.output_dim,),
             "dummy_tokens": (self.num_dummy_tokens, self.output_dim)}
        for i in range(self.depth):
            a, f = f"layers.{i}.0.", f"layers.{i}.1."
            p.update({a + "norm1.weight": (d,), a + "norm1.bias": (d,), a + "norm2.weight": (d,), a + "norm2.bias": (d,),
                      a + "to_q.weight": (inner, d), a + "to_kv.weight": (2 * inner, d), a + "to_out.weight": (d, inner),
                      f + "0.weight": (d,), f + "0.bias": (d,), f + "1.weight": (ffi, d), f + "3.weight": (d, ffi)})
        return p

    def load_state_dict(self, sd: Dict[str, Tensor], strict: bool = True):
        shapes = self.param_shapes()
        missing = [k for k in shapes if k not in sd]
        if strict and missing:
            raise RuntimeError(f"Resampler.load_state_dict: missing {missing[:5]}")
        for k, shp in shapes.items():
            if k in sd:
                if tuple(sd[k].shape) != tuple(shp):
                    raise RuntimeError(f"Resampler: {k} has shape {tuple(sd[k].shape)}, expected {shp}")
                self._sd[k] = sd[k].detach().to(self.device, torch.float16).contiguous()
        return self

    def state_dict(self):
        return dict(self._sd)

    def weights_changed(self) -> None:
        """Called after `tensors()` were rewritten in place / re-homed (the RCCL start-up broadcast).  The kernels read the
        listed tensors themselves at every call - nothing derived from them is cached here - so there is nothing to drop; the
        hook exists so that `distributed.broadcast_pipeline` can require it of every engine."""

    def tensors(self):
        """Frozen weights (the multi-GPU weight broadcast list)."""
        return list(self._sd.values())

    def init_random(self, seed: int = 0) -> "Resampler":
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        sd = {}
        for k, shp in self.param_shapes().items():
            if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith(".0.weight") or k == "norm_out.weight":
                t = 1.0 + 0.1 * torch.randn(shp, generator=g)
            elif k.endswith(".bias"):
                t = 0.05 * torch.randn(shp, generator=g)
            elif k == "latents":
                t = torch.randn(shp, generator=g) / self.dim ** 0.5
            elif k == "dummy_tokens":
                t = torch.randn(shp, generator=g)
            else:
                t = torch.randn(shp, generator=g) / shp[1] ** 0.5
            sd[k] = t
        return self.load_state_dict(sd)

    def to(self, device=None, dtype=None, **kw):
        if dtype is not None and dtype != torch.float16:
            raise ValueError("Resampler (HIP) computes in fp16")
        if device is not None:
            self.device = torch.device(device)
            self._sd = {k: v.to(self.device) for k, v in self._sd.items()}
        return self

    def dtype(self):
        return torch.float16

    # ---- forward (reference src/models/resampler.py:119-141)
    def forward(self, x: Tensor, magi_image_embeds: Tensor) -> Tensor:
        sd = self._sd
        bsz, n_ips, seq, _ = x.shape
        bn = bsz * n_ips
        d, inner = self.dim, self.dim_head * self.heads
        x = x.to(self.device, torch.float16).reshape(bn * seq, -1).contiguous()
        m = magi_image_embeds.to(self.device, torch.float16).reshape(bn, -1).contiguous()
        tokens = torch.empty((bn, seq + 1, d), dtype=torch.float16, device=self.device)
        tokens[:, :seq] = ops.gemm(x, sd["proj_in.weight"], sd["proj_in.bias"]).reshape(bn, seq, d)
        tokens[:, seq] = ops.gemm(m, sd["proj_in_magi.weight"], sd["proj_in_magi.bias"])
        lat = sd["latents"].repeat(bn, 1, 1).reshape(bn * self.num_queries, d).contiguous()
        nq = self.num_queries
        scale = 1.0 / self.dim_head ** 0.5   # (q*s)(k*s)^T with s = dim_head^-1/4  (reference :67-68)
        kv_in = torch.empty((bn, seq + 1 + nq, d), dtype=torch.float16, device=self.device)
        for i in range(self.depth):
            a, f = f"layers.{i}.0.", f"layers.{i}.1."
            kv_in[:, :seq + 1] = ops.layernorm(tokens, sd[a + "norm1.weight"], sd[a + "norm1.bias"])
            ln = ops.layernorm(lat, sd[a + "norm2.weight"], sd[a + "norm2.bias"])
            kv_in[:, seq + 1:] = ln.reshape(bn, nq, d)
            q = ops.gemm(ln, sd[a + "to_q.weight"]).reshape(bn, nq, inner)
            kv = ops.gemm(kv_in.reshape(-1, d), sd[a + "to_kv.weight"]).reshape(bn, seq + 1 + nq, 2 * inner)
            o = ops.small_attention(q, kv[:, :, :inner], kv[:, :, inner:], self.heads, scale)
            lat = ops.gemm(o.reshape(bn * nq, inner), sd[a + "to_out.weight"], residual=lat)
            h = ops.layernorm(lat, sd[f + "0.weight"], sd[f + "0.bias"])
            h = ops.gemm(h, sd[f + "1.weight"], act="gelu")
            lat = ops.gemm(h, sd[f + "3.weight"], residual=lat)
        lat = ops.gemm(lat, sd["proj_out.weight"], sd["proj_out.bias"])
        lat = ops.layernorm(lat, sd["norm_out.weight"], sd["norm_out.bias"])
        out = torch.empty((bsz, self.num_dummy_tokens + n_ips * nq, self.output_dim), dtype=torch.float16,
                          device=self.device)
        out[:, :self.num_dummy_tokens] = sd["dummy_tokens"].unsqueeze(0)
        out[:, self.num_dummy_tokens:] = lat.reshape(bsz, n_ips * nq, self.output_dim)
        return out

    __call__ = forward
