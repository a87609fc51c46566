"""Character-encoder forwards on the HIP kernels: CLIP ViT-H/14 vision tower (penultimate hidden state) and the
Magi ViT-MAE crop encoder (CLS of the last hidden state) — what reference
src/pipelines/pipeline_diffsensei.py:127-128 calls through `transformers` [3P].

The engines are built FROM a `transformers` model (or its state dict + config): weights are copied to fp16 device
tensors in kernel layout; the forward is GEMM / LayerNorm / small-attention launches only.  Once per panel, <0.2 %
of the FLOPs — correctness and zero host syncs matter here, not MFMA peak.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import ops

Tensor = torch.Tensor


def _f16(t: Tensor, device) -> Tensor:
    return t.detach().to(device=device, dtype=torch.float16).contiguous()


def _pad_k(w2d: Tensor, mult: int = 8) -> Tensor:
    k = w2d.shape[1]
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return w2d.contiguous()
    out = torch.zeros((w2d.shape[0], kp), dtype=w2d.dtype, device=w2d.device)
    out[:, :k] = w2d
    return out


class _ViTLayer:
    __slots__ = ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "o_w", "o_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")


class ViTEncoderEngine:
    """Pre-LN transformer encoder over patch tokens (shared by CLIP vision and ViT-MAE)."""

    def __init__(self, hidden: int, heads: int, patch: int, image_size: int, act: str, eps: float, device):
        self.hidden, self.heads, self.patch, self.image_size = hidden, heads, patch, image_size
        self.act, self.eps, self.device = act, eps, torch.device(device)
        self.layers: List[_ViTLayer] = []
        self.patch_w: Optional[Tensor] = None      # [hidden, pad8(3*p*p)]
        self.patch_b: Optional[Tensor] = None
        self.cls_row: Optional[Tensor] = None      # class token + its position embedding, [hidden]
        self.pos_patches: Optional[Tensor] = None  # position embeddings of the patch tokens, [n_patches, hidden]
        self.pre_ln = None                          # (w, b) or None
        self.post_ln = None
        self.dtype = torch.float16
        self.config = None

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch) ** 2

    def weights_changed(self) -> None:
        """Called after `tensors()` were rewritten in place / re-homed (the RCCL start-up broadcast).  The kernels read the
        listed tensors themselves at every call - nothing derived from them is cached here - so there is nothing to drop; the
        hook exists so that `distributed.broadcast_pipeline` can require it of every engine."""

    def tensors(self) -> List[Tensor]:
        """Frozen weights in kernel layout (the multi-GPU weight broadcast list)."""
        out = [getattr(L, s) for L in self.layers for s in L.__slots__]
        out += [t for t in (self.patch_w, self.patch_b, self.cls_row, self.pos_patches) if t is not None]
        for pair in (self.pre_ln, self.post_ln):
            if pair is not None:
                out += list(pair)
        return out

    def embed(self, pixel_values: Tensor) -> Tensor:
        """[B,3,S,S] -> tokens [B, 1+n_patches, hidden] (non-overlapping patch conv as one GEMM)."""
        B = pixel_values.shape[0]
        p, g = self.patch, self.image_size // self.patch
        x = pixel_values.to(self.device, torch.float16)
        # im2col of non-overlapping patches is a pure re-indexing (no arithmetic): [B, g*g, 3*p*p]
        cols = x.reshape(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, 3 * p * p)
        kp = self.patch_w.shape[1]
        if kp != cols.shape[1]:
            padded = torch.zeros((cols.shape[0], kp), dtype=torch.float16, device=self.device)
            padded[:, :cols.shape[1]] = cols
            cols = padded
        pos = self.pos_patches.unsqueeze(0).expand(B, -1, -1).reshape(B * g * g, self.hidden).contiguous()
        tok = ops.gemm(cols.contiguous(), self.patch_w, self.patch_b, residual=pos)
        out = torch.empty((B, 1 + g * g, self.hidden), dtype=torch.float16, device=self.device)
        out[:, 0] = self.cls_row
        out[:, 1:] = tok.reshape(B, g * g, self.hidden)
        return out

    def run_layers(self, h: Tensor, n_layers: Optional[int] = None) -> Tensor:
        B, N, D = h.shape
        hd = D // self.heads
        scale = hd ** -0.5
        h = h.reshape(B * N, D).contiguous()
        for layer in self.layers[:n_layers if n_layers is not None else len(self.layers)]:
            n1 = ops.layernorm(h, layer.ln1_w, layer.ln1_b, self.eps)
            qkv = ops.gemm(n1, layer.qkv_w, layer.qkv_b).reshape(B, N, 3 * D)
            o = ops.small_attention(qkv[:, :, :D].contiguous(), qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], self.heads, scale)
            h = ops.gemm(o.reshape(B * N, D), layer.o_w, layer.o_b, residual=h)
            n2 = ops.layernorm(h, layer.ln2_w, layer.ln2_b, self.eps)
            m = ops.gemm(n2, layer.fc1_w, layer.fc1_b, act=self.act)
            h = ops.gemm(m, layer.fc2_w, layer.fc2_b, residual=h)
        return h.reshape(B, N, D)


class _Keys:
    """State-dict lookup tolerant to the transformers 4.x (the reference's checkpoints) and 5.x key spellings."""

    def __init__(self, sd: Dict[str, Tensor], prefixes=("",)):
        self.sd, self.prefixes = sd, prefixes

    def __call__(self, *cands: str) -> Tensor:
        for pre in self.prefixes:
            for c in cands:
                if pre + c in self.sd:
                    return self.sd[pre + c]
        raise KeyError(f"none of {cands} (prefixes {self.prefixes}) in the encoder state dict")


class ClipVisionEngine(ViTEncoderEngine):
    """`image_encoder(pixels, output_hidden_states=True).hidden_states[-2]` (reference :127)."""

    @classmethod
    def from_transformers(cls, model, device="cuda") -> "ClipVisionEngine":
        cfg = model.config if not hasattr(model.config, "vision_config") else model.config.vision_config
        act = {"gelu": "gelu", "quick_gelu": "quick_gelu"}[cfg.hidden_act]
        eng = cls(cfg.hidden_size, cfg.num_attention_heads, cfg.patch_size, cfg.image_size, act, cfg.layer_norm_eps, device)
        eng.config = cfg
        K = _Keys(dict(model.state_dict()), ("vision_model.", ""))
        pw = K("embeddings.patch_embedding.weight")
        eng.patch_w = _pad_k(_f16(pw.reshape(pw.shape[0], -1), device))
        eng.patch_b = None
        pos = K("embeddings.position_embedding.weight").float()
        eng.cls_row = _f16(K("embeddings.class_embedding").float() + pos[0], device)   # weight folding at load
        eng.pos_patches = _f16(pos[1:], device)
        # transformers spells it "pre_layrnorm"
        eng.pre_ln = (_f16(K("pre_layrnorm.weight", "pre_layernorm.weight"), device),
                      _f16(K("pre_layrnorm.bias", "pre_layernorm.bias"), device))
        for i in range(cfg.num_hidden_layers):
            L, p = _ViTLayer(), f"encoder.layers.{i}."
            L.ln1_w, L.ln1_b = _f16(K(p + "layer_norm1.weight"), device), _f16(K(p + "layer_norm1.bias"), device)
            L.qkv_w = _f16(torch.cat([K(p + "self_attn.q_proj.weight"), K(p + "self_attn.k_proj.weight"),
                                      K(p + "self_attn.v_proj.weight")], 0), device)
            L.qkv_b = _f16(torch.cat([K(p + "self_attn.q_proj.bias"), K(p + "self_attn.k_proj.bias"),
                                      K(p + "self_attn.v_proj.bias")], 0), device)
            L.o_w, L.o_b = _f16(K(p + "self_attn.out_proj.weight"), device), _f16(K(p + "self_attn.out_proj.bias"), device)
            L.ln2_w, L.ln2_b = _f16(K(p + "layer_norm2.weight"), device), _f16(K(p + "layer_norm2.bias"), device)
            L.fc1_w, L.fc1_b = _f16(K(p + "mlp.fc1.weight"), device), _f16(K(p + "mlp.fc1.bias"), device)
            L.fc2_w, L.fc2_b = _f16(K(p + "mlp.fc2.weight"), device), _f16(K(p + "mlp.fc2.bias"), device)
            eng.layers.append(L)
        return eng

    def penultimate_hidden(self, pixel_values: Tensor) -> Tensor:
        tok = self.embed(pixel_values)
        B, N, D = tok.shape
        h = ops.layernorm(tok.reshape(B * N, D), self.pre_ln[0], self.pre_ln[1], self.eps).reshape(B, N, D)
        return self.run_layers(h, len(self.layers) - 1)


class ViTMAEEngine(ViTEncoderEngine):
    """`magi_image_encoder(pixels).last_hidden_state[:, 0]` (reference :128).

    ViT-MAE shuffles the patch tokens with random noise before the encoder; attention is permutation-equivariant
    and the CLS token is prepended after the shuffle, so with mask_ratio = 0 (the Magi crop encoder's setting) the
    CLS output does not depend on the shuffle and is computed here on the unshuffled sequence."""

    @classmethod
    def from_transformers(cls, model, device="cuda") -> "ViTMAEEngine":
        cfg = model.config
        if getattr(cfg, "mask_ratio", 0.0) != 0.0:
            raise ValueError("ViTMAEEngine needs mask_ratio == 0 (random masking changes the CLS output)")
        eng = cls(cfg.hidden_size, cfg.num_attention_heads, cfg.patch_size, cfg.image_size, "gelu", cfg.layer_norm_eps,
                  device)
        eng.config = cfg
        K = _Keys(dict(model.state_dict()), ("", "vit."))
        pw = K("embeddings.patch_embeddings.projection.weight")
        eng.patch_w = _pad_k(_f16(pw.reshape(pw.shape[0], -1), device))
        eng.patch_b = _f16(K("embeddings.patch_embeddings.projection.bias"), device)
        pos = K("embeddings.position_embeddings").float()[0]
        eng.cls_row = _f16(K("embeddings.cls_token").float().reshape(-1) + pos[0], device)
        eng.pos_patches = _f16(pos[1:], device)
        eng.post_ln = (_f16(K("layernorm.weight"), device), _f16(K("layernorm.bias"), device))
        for i in range(cfg.num_hidden_layers):
            L = _ViTLayer()
            a, b = f"encoder.layer.{i}.", f"layers.{i}."          # transformers 4.x / 5.x
            L.ln1_w = _f16(K(a + "layernorm_before.weight", b + "layernorm_before.weight"), device)
            L.ln1_b = _f16(K(a + "layernorm_before.bias", b + "layernorm_before.bias"), device)
            qw = [K(a + f"attention.attention.{n}.weight", b + f"attention.{m}_proj.weight")
                  for n, m in (("query", "q"), ("key", "k"), ("value", "v"))]
            qb = [K(a + f"attention.attention.{n}.bias", b + f"attention.{m}_proj.bias")
                  for n, m in (("query", "q"), ("key", "k"), ("value", "v"))]
            L.qkv_w, L.qkv_b = _f16(torch.cat(qw, 0), device), _f16(torch.cat(qb, 0), device)
            L.o_w = _f16(K(a + "attention.output.dense.weight", b + "attention.o_proj.weight"), device)
            L.o_b = _f16(K(a + "attention.output.dense.bias", b + "attention.o_proj.bias"), device)
            L.ln2_w = _f16(K(a + "layernorm_after.weight", b + "layernorm_after.weight"), device)
            L.ln2_b = _f16(K(a + "layernorm_after.bias", b + "layernorm_after.bias"), device)
            L.fc1_w = _f16(K(a + "intermediate.dense.weight", b + "mlp.fc1.weight"), device)
            L.fc1_b = _f16(K(a + "intermediate.dense.bias", b + "mlp.fc1.bias"), device)
            L.fc2_w = _f16(K(a + "output.dense.weight", b + "mlp.fc2.weight"), device)
            L.fc2_b = _f16(K(a + "output.dense.bias", b + "mlp.fc2.bias"), device)
            eng.layers.append(L)
        return eng

    def cls_embedding(self, pixel_values: Tensor) -> Tensor:
        h = self.run_layers(self.embed(pixel_values))
        cls_tok = h[:, 0].contiguous()
        return ops.layernorm(cls_tok, self.post_ln[0], self.post_ln[1], self.eps)


class ClipTextEngine:
    """SDXL prompt encoders on the HIP kernels: `text_encoder(ids, output_hidden_states=True)` of
    `StableDiffusionXLPipeline.encode_prompt` [3P], called at reference src/pipelines/pipeline_diffsensei.py:237-245.

    Returns what `encode_prompt` consumes: `hidden_states[-2]` (output of the penultimate layer, no final LayerNorm) and
    `out[0]` (CLIPTextModel: last_hidden_state after the final LayerNorm; CLIPTextModelWithProjection: the EOS token's
    final-LayerNorm state times `text_projection`).  Pre-LN blocks with CAUSAL attention; token + position embedding
    gather-add, GEMMs, LayerNorms and attention are HIP launches; locating the EOS token is index plumbing."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.layers: List[_ViTLayer] = []
        self.dtype = torch.float16

    @classmethod
    def from_transformers(cls, model, device="cuda") -> "ClipTextEngine":
        cfg = model.config
        eng = cls(device)
        eng.config = cfg
        eng.hidden, eng.heads, eng.eps = cfg.hidden_size, cfg.num_attention_heads, cfg.layer_norm_eps
        eng.act = {"gelu": "gelu", "quick_gelu": "quick_gelu"}[cfg.hidden_act]
        eng.eos_token_id = getattr(cfg, "eos_token_id", None)
        K = _Keys(dict(model.state_dict()), ("text_model.", ""))
        eng.tok_emb = _f16(K("embeddings.token_embedding.weight"), device)
        eng.pos_emb = _f16(K("embeddings.position_embedding.weight"), device)
        eng.final_ln = (_f16(K("final_layer_norm.weight"), device), _f16(K("final_layer_norm.bias"), device))
        sd = model.state_dict()
        eng.text_projection = _f16(sd["text_projection.weight"], device) if "text_projection.weight" in sd else None
        for i in range(cfg.num_hidden_layers):
            L, p = _ViTLayer(), f"encoder.layers.{i}."
            L.ln1_w, L.ln1_b = _f16(K(p + "layer_norm1.weight"), device), _f16(K(p + "layer_norm1.bias"), device)
            L.qkv_w = _f16(torch.cat([K(p + "self_attn.q_proj.weight"), K(p + "self_attn.k_proj.weight"),
                                      K(p + "self_attn.v_proj.weight")], 0), device)
            L.qkv_b = _f16(torch.cat([K(p + "self_attn.q_proj.bias"), K(p + "self_attn.k_proj.bias"),
                                      K(p + "self_attn.v_proj.bias")], 0), device)
            L.o_w, L.o_b = _f16(K(p + "self_attn.out_proj.weight"), device), _f16(K(p + "self_attn.out_proj.bias"), device)
            L.ln2_w, L.ln2_b = _f16(K(p + "layer_norm2.weight"), device), _f16(K(p + "layer_norm2.bias"), device)
            L.fc1_w, L.fc1_b = _f16(K(p + "mlp.fc1.weight"), device), _f16(K(p + "mlp.fc1.bias"), device)
            L.fc2_w, L.fc2_b = _f16(K(p + "mlp.fc2.weight"), device), _f16(K(p + "mlp.fc2.bias"), device)
            eng.layers.append(L)
        return eng

    def parameters(self):
        yield self.tok_emb

    def weights_changed(self) -> None:
        """Called after `tensors()` were rewritten in place / re-homed (the RCCL start-up broadcast).  The kernels read the
        listed tensors themselves at every call - nothing derived from them is cached here - so there is nothing to drop; the
        hook exists so that `distributed.broadcast_pipeline` can require it of every engine."""

    def tensors(self) -> List[Tensor]:
        """Frozen weights in kernel layout (the multi-GPU weight broadcast list)."""
        out = [getattr(L, s) for L in self.layers for s in L.__slots__]
        out += [self.tok_emb, self.pos_emb, *self.final_ln]
        if self.text_projection is not None:
            out.append(self.text_projection)
        return out

    def _block(self, h: Tensor, L: _ViTLayer, B: int, N: int) -> Tensor:
        D = self.hidden
        scale = (D // self.heads) ** -0.5
        n1 = ops.layernorm(h, L.ln1_w, L.ln1_b, self.eps)
        qkv = ops.gemm(n1, L.qkv_w, L.qkv_b).reshape(B, N, 3 * D)
        o = ops.causal_attention(qkv[:, :, :D].contiguous(), qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], self.heads, scale)
        h = ops.gemm(o.reshape(B * N, D), L.o_w, L.o_b, residual=h)
        n2 = ops.layernorm(h, L.ln2_w, L.ln2_b, self.eps)
        m = ops.gemm(n2, L.fc1_w, L.fc1_b, act=self.act)
        return ops.gemm(m, L.fc2_w, L.fc2_b, residual=h)

    def encode(self, input_ids: Tensor):
        """input_ids: [B,T] integer tokens -> (penultimate hidden states [B,T,D], pooled-or-last output)."""
        ids = input_ids.to(self.device, torch.int32).contiguous()
        B, T = ids.shape
        D = self.hidden
        h = ops.embed_tokens(ids, self.tok_emb, self.pos_emb).reshape(B * T, D)
        penultimate = None
        for i, L in enumerate(self.layers):
            if i == len(self.layers) - 1:
                penultimate = h.reshape(B, T, D).clone()
            h = self._block(h, L, B, T)
        if len(self.layers) == 1:
            penultimate = penultimate if penultimate is not None else h.reshape(B, T, D)
        last = ops.layernorm(h, self.final_ln[0], self.final_ln[1], self.eps).reshape(B, T, D)
        if self.text_projection is None:
            return penultimate, last
        # pooled = state at the EOS token (transformers: first eos_token_id position, legacy: argmax of the ids)
        if self.eos_token_id is not None and self.eos_token_id != 2:
            pos = (ids == self.eos_token_id).int().argmax(dim=-1)
        else:
            pos = ids.argmax(dim=-1)
        pooled = last[torch.arange(B, device=self.device), pos.long()].contiguous()
        return penultimate, ops.gemm(pooled, self.text_projection)
