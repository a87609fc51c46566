"""Oracle (test infrastructure only): attention arithmetic of the reference processors.

Restates reference src/models/attention_processor.py with explicit softmax math (no SDPA call):
  * `self_attention`            -> AttnProcessor2_0.__call__            (:19-96)
  * `ip_region_mask`            -> MaskedIPAttnProcessor2_0.prepare_attention_mask_ip (:115-169)
  * `masked_ip_cross_attention` -> MaskedIPAttnProcessor2_0.__call__    (:171-273)
All tensors fp32 (or whatever dtype is handed in); `q` is an optional rounding hook used to emulate
the reference's fp16 storage between ops.
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Tuple

import torch

Tensor = torch.Tensor
_id = lambda t: t


def _heads(x: Tensor, heads: int) -> Tensor:
    b, n, c = x.shape
    return x.view(b, n, heads, c // heads).transpose(1, 2)  # [b, h, n, d]


def sdpa(qh: Tensor, kh: Tensor, vh: Tensor, mask: Optional[Tensor] = None, q: Callable = _id) -> Tensor:
    """softmax(q k^T / sqrt(d) + mask) v — what F.scaled_dot_product_attention computes
    (reference :76-78, :235-237, :251-253), fp32 softmax, probabilities rounded like the fused kernels do."""
    d = qh.shape[-1]
    s = torch.matmul(qh.float(), kh.float().transpose(-1, -2)) * (1.0 / math.sqrt(d))
    if mask is not None:
        s = s + mask.float()
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, vh.float()).to(qh.dtype)


def self_attention(x: Tensor, wq: Tensor, wk: Tensor, wv: Tensor, wo: Tensor, bo: Tensor, heads: int,
                   q: Callable = _id) -> Tensor:
    """AttnProcessor2_0 for a BasicTransformerBlock.attn1 (no mask, no group/spatial norm, no residual inside).

    reference: src/models/attention_processor.py:56-84 (to_q/to_k/to_v without bias, SDPA, to_out[0] with bias).
    """
    b, n, c = x.shape
    query = q(x @ wq.t())
    key = q(x @ wk.t())
    value = q(x @ wv.t())
    o = sdpa(_heads(query, heads), _heads(key, heads), _heads(value, heads))
    o = q(o.transpose(1, 2).reshape(b, n, c))
    return q(o @ wo.t() + bo)


def mask_grid_size(sequence_length: int, aspect_ratio: float) -> Tuple[int, int]:
    """(height, width) the reference infers from N and H/W — reference :131-139, restated literally."""
    width = int((sequence_length / aspect_ratio) ** 0.5)
    height = sequence_length // width
    while width * height != sequence_length:
        if width * height < sequence_length:
            width += 1
        else:
            width -= 1
        height = sequence_length // width
    return height, width


def ip_region_mask(bbox: Tensor, sequence_length: int, heads: int, aspect_ratio: float,
                   num_ip_tokens: int, num_dummy_tokens: int, dtype=torch.float32) -> Tensor:
    """Additive mask [B, heads, N, num_dummy + num_ip]  (0 = attend, -10000 = masked).

    reference :141-169.  Token (row i, col j) sits at (x, y) = (linspace(0,1,W)[j], linspace(0,1,H)[i]);
    it is inside box k iff x1<=x<=x2 and y1<=y<=y2 (inclusive both ends).  Character k's tokens are
    open inside box k; the dummy tokens are open where NO box covers the token.
    """
    batch, max_num_ips, _ = bbox.shape
    height, width = mask_grid_size(sequence_length, aspect_ratio)
    xs = torch.linspace(0, 1, steps=width)
    ys = torch.linspace(0, 1, steps=height)
    x_grid = xs.repeat(height)                   # idx = i*W + j  -> xs[j]
    y_grid = ys.repeat_interleave(width)         #               -> ys[i]
    bb = bbox.float().cpu()
    inside = ((x_grid[None, None] >= bb[:, :, 0, None]) & (x_grid[None, None] <= bb[:, :, 2, None]) &
              (y_grid[None, None] >= bb[:, :, 1, None]) & (y_grid[None, None] <= bb[:, :, 3, None]))  # [B, K, N]
    ip = torch.where(inside, 0.0, -10000.0).permute(0, 2, 1)                          # [B, N, K]
    dummy = torch.where(inside.any(dim=1), -10000.0, 0.0)[:, :, None]                 # [B, N, 1]
    ip = ip.repeat_interleave(num_ip_tokens // max_num_ips, dim=-1)
    dummy = dummy.repeat_interleave(num_dummy_tokens, dim=-1)
    m = torch.cat([dummy, ip], dim=-1)[:, None].expand(batch, heads, sequence_length, -1)
    return m.to(dtype)


def masked_ip_cross_attention(x: Tensor, enc: Tensor, bbox: Tensor, aspect_ratio: float,
                              wq: Tensor, wk: Tensor, wv: Tensor, wk_ip: Tensor, wv_ip: Tensor,
                              wo: Tensor, bo: Tensor, heads: int, scale: float,
                              num_ip_tokens: int, num_dummy_tokens: int, q: Callable = _id) -> Tensor:
    """MaskedIPAttnProcessor2_0 for BasicTransformerBlock.attn2.

    reference :207-261: q = to_q(x); split enc into text / ip at L-(num_ip+num_dummy); text SDPA;
    ip SDPA with the region mask; `text + scale*ip`; to_out[0].
    """
    b, n, c = x.shape
    end_pos = enc.shape[1] - (num_ip_tokens + num_dummy_tokens)
    txt, ip = enc[:, :end_pos], enc[:, end_pos:]
    query = q(x @ wq.t())
    key, value = q(txt @ wk.t()), q(txt @ wv.t())
    qh = _heads(query, heads)
    t_out = sdpa(qh, _heads(key, heads), _heads(value, heads))
    t_out = q(t_out.transpose(1, 2).reshape(b, n, c))
    mask = ip_region_mask(bbox, n, heads, aspect_ratio, num_ip_tokens, num_dummy_tokens, dtype=x.dtype)
    ip_key, ip_value = q(ip @ wk_ip.t()), q(ip @ wv_ip.t())
    i_out = sdpa(qh, _heads(ip_key, heads), _heads(ip_value, heads), mask)
    i_out = q(i_out.transpose(1, 2).reshape(b, n, c))
    h = q(t_out + q(scale * i_out))
    return q(h @ wo.t() + bo)
