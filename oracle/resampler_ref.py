"""Oracle (test infrastructure only): fp32 restatement of the reference perceiver Resampler.

reference: src/models/resampler.py — PerceiverAttention.forward :47-76, FeedForward :11-18, Resampler.forward :119-141.
State-dict keys are the reference module's own (`latents`, `proj_in.*`, `proj_in_magi.*`, `layers.{i}.0.*`
for the attention, `layers.{i}.1.{0,1,3}.*` for the FF Sequential, `proj_out.*`, `norm_out.*`, `dummy_tokens`).
Checked against the reference module itself in tests/test_oracle_golden.py (fixtures tests/golden/resampler_*.npz).
"""
from __future__ import annotations

import math
from typing import Callable, Dict

import torch
import torch.nn.functional as F

_id = lambda t: t


def resampler_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, magi: torch.Tensor, heads: int, dim_head: int,
                      q: Callable = _id) -> torch.Tensor:
    sd = {k: v.float() for k, v in sd.items()}
    bsz, n_ips, seq, _ = x.shape
    x = x.float().reshape(bsz * n_ips, seq, -1)
    x = q(x @ sd["proj_in.weight"].t() + sd["proj_in.bias"])
    m = q(magi.float() @ sd["proj_in_magi.weight"].t() + sd["proj_in_magi.bias"]).reshape(bsz * n_ips, 1, -1)
    x = torch.cat([x, m], dim=1)
    lat = sd["latents"].repeat(x.shape[0], 1, 1)
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))
    dim = lat.shape[-1]
    scale = 1 / math.sqrt(math.sqrt(dim_head))
    for i in range(depth):
        a = f"layers.{i}.0."
        xn = q(F.layer_norm(x, (dim,), sd[a + "norm1.weight"], sd[a + "norm1.bias"]))
        ln = q(F.layer_norm(lat, (dim,), sd[a + "norm2.weight"], sd[a + "norm2.bias"]))
        b, l, _ = ln.shape
        qq = q(ln @ sd[a + "to_q.weight"].t())
        kv = q(torch.cat([xn, ln], dim=-2) @ sd[a + "to_kv.weight"].t())
        k, v = kv.chunk(2, dim=-1)

        def hs(t):
            return t.reshape(b, t.shape[1], heads, -1).transpose(1, 2)

        w = (hs(qq) * scale) @ (hs(k) * scale).transpose(-2, -1)
        w = torch.softmax(w.float(), dim=-1)
        o = q((w @ hs(v)).permute(0, 2, 1, 3).reshape(b, l, -1))
        lat = q(q(o @ sd[a + "to_out.weight"].t()) + lat)
        f = f"layers.{i}.1."
        h = q(F.layer_norm(lat, (dim,), sd[f + "0.weight"], sd[f + "0.bias"]))
        h = q(F.gelu(q(h @ sd[f + "1.weight"].t())))
        lat = q(q(h @ sd[f + "3.weight"].t()) + lat)
    out_dim = sd["proj_out.weight"].shape[0]
    lat = q(lat @ sd["proj_out.weight"].t() + sd["proj_out.bias"])
    lat = q(F.layer_norm(lat, (out_dim,), sd["norm_out.weight"], sd["norm_out.bias"]))
    n_q = sd["latents"].shape[1]
    lat = lat.reshape(bsz, n_ips * n_q, out_dim)
    dummy = sd["dummy_tokens"].unsqueeze(0).repeat(bsz, 1, 1)
    return torch.cat([dummy, lat], dim=1)
