"""Host mirror of reference src/models/attention_processor.py — same class names, constructor arguments,
attributes (`.scale`, `.to_k_ip`, `.to_v_ip`, `num_ip_tokens`, `num_dummy_tokens`) and call protocol
`proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, bbox=None,
aspect_ratio=None)`.  The arithmetic is the HIP kernels (ops.*): there is no torch fallback.

Inside `UNetMangaModel.forward` the engine does not call these objects per layer (it replays a launch
plan); they exist so the processor can also be used stand-alone, exactly like the reference's, and they are
what `unet.attn_processors` returns (`set_ip_scale` writes `.scale` on them;
reference src/pipelines/pipeline_diffsensei.py:172-178).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops

Tensor = torch.Tensor
LP = 96


def mask_grid_size(sequence_length: int, aspect_ratio: float) -> Tuple[int, int]:
    """(height, width) of the token grid the reference infers from N and H/W.

    Control-path restatement of reference src/models/attention_processor.py:131-139 (runs once per level on
    the host; the per-token region test itself runs inside the attention kernel)."""
    width = int((sequence_length / aspect_ratio) ** 0.5)
    width = max(width, 1)
    height = sequence_length // width
    while width * height != sequence_length:
        if width * height < sequence_length:
            width += 1
        else:
            width -= 1
        height = sequence_length // width
    return height, width


class _Linear:
    """Weight holder with the nn.Linear attribute surface the reference touches (`.weight`, `.bias`)."""

    def __init__(self, out_features: int, in_features: int, bias: bool, device=None, dtype=torch.float16):
        self.in_features, self.out_features = in_features, out_features
        self.weight = torch.zeros(out_features, in_features, device=device, dtype=dtype)
        self.bias = torch.zeros(out_features, device=device, dtype=dtype) if bias else None

    def __call__(self, x: Tensor, residual: Optional[Tensor] = None) -> Tensor:
        shp = x.shape
        y = ops.gemm(x.reshape(-1, shp[-1]).contiguous(), self.weight, self.bias,
                     None if residual is None else residual.reshape(-1, self.out_features))
        return y.reshape(*shp[:-1], self.out_features)


class AttentionWeights:
    """Stand-in for diffusers `Attention` holding to_q/to_k/to_v/to_out and `heads` (what the processors read)."""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int], heads: int, device=None):
        kv = cross_attention_dim or query_dim
        self.to_q = _Linear(query_dim, query_dim, False, device)
        self.to_k = _Linear(query_dim, kv, False, device)
        self.to_v = _Linear(query_dim, kv, False, device)
        self.to_out = [_Linear(query_dim, query_dim, True, device), None]
        self.heads = heads
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = False
        self.residual_connection = False
        self.rescale_output_factor = 1.0


def _vt(x: Tensor, w: Tensor) -> Tensor:
    """V^T panels [B, C, N] = W @ x[b]^T (values stored key-contiguous for the attention kernels)."""
    return ops.gemm_batched_nt(w, x.contiguous())


class AttnProcessor2_0:
    """Self-attention processor (reference src/models/attention_processor.py:7-96)."""

    def __call__(self, attn, hidden_states: Tensor, encoder_hidden_states=None, attention_mask=None, temb=None,
                 bbox=None, dialog_bbox=None, aspect_ratio=None, *args, **kwargs) -> Tensor:
        if attention_mask is not None or encoder_hidden_states is not None:
            raise NotImplementedError("AttnProcessor2_0 (HIP): only the unmasked self-attention form is on the hot path")
        if attn.spatial_norm is not None or attn.group_norm is not None:
            raise NotImplementedError("spatial_norm/group_norm attention variants are not used by the SDXL UNet")
        x = hidden_states
        q, k = attn.to_q(x), attn.to_k(x)
        vt = _vt(x, attn.to_v.weight)
        o = ops.self_attention(q.contiguous(), k.contiguous(), vt, attn.heads)
        return attn.to_out[0](o)


class MaskedIPAttnProcessor2_0:
    """Region-masked IP-Adapter cross-attention processor (reference src/models/attention_processor.py:99-273)."""

    def __init__(self, hidden_size, cross_attention_dim=None, scale=1.0, num_ip_tokens=4, num_dummy_tokens=4,
                 device=None):
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        self.scale = scale
        self.num_ip_tokens = num_ip_tokens
        self.num_dummy_tokens = num_dummy_tokens
        self.to_k_ip = _Linear(hidden_size, cross_attention_dim or hidden_size, False, device)
        self.to_v_ip = _Linear(hidden_size, cross_attention_dim or hidden_size, False, device)

    def state_dict(self):
        return {"to_k_ip.weight": self.to_k_ip.weight, "to_v_ip.weight": self.to_v_ip.weight}

    def load_state_dict(self, sd):
        self.to_k_ip.weight = sd["to_k_ip.weight"].to(self.to_k_ip.weight)
        self.to_v_ip.weight = sd["to_v_ip.weight"].to(self.to_v_ip.weight)

    def __call__(self, attn, hidden_states: Tensor, encoder_hidden_states: Tensor = None, attention_mask=None,
                 temb=None, bbox: Tensor = None, aspect_ratio: float = None, *args, **kwargs) -> Tensor:
        if attention_mask is not None:
            raise NotImplementedError("MaskedIPAttnProcessor2_0 (HIP): attention_mask is always None on the hot path")
        if encoder_hidden_states is None or bbox is None or aspect_ratio is None:
            raise ValueError("encoder_hidden_states, bbox and aspect_ratio are required")
        x = hidden_states
        b, n, c = x.shape
        n_ctx = self.num_ip_tokens + self.num_dummy_tokens
        end_pos = encoder_hidden_states.shape[1] - n_ctx
        enc = encoder_hidden_states.contiguous()
        txt = ops.pad_rows(enc, 0, end_pos, LP)
        ip = ops.pad_rows(enc, end_pos, n_ctx, LP)
        q = attn.to_q(x)
        kt, ki = attn.to_k(txt), self.to_k_ip(ip)
        vtt, vti = _vt(txt, attn.to_v.weight), _vt(ip, self.to_v_ip.weight)
        max_ips = bbox.shape[1]
        o = ops.masked_ip_attention(q.contiguous(), kt.contiguous(), vtt, ki.contiguous(), vti,
                                    bbox.to(device=x.device, dtype=torch.float32).contiguous(), attn.heads,
                                    mask_grid_size(n, aspect_ratio), float(self.scale), Lt=end_pos, Li=n_ctx,
                                    n_dummy=self.num_dummy_tokens, tok_per_ip=self.num_ip_tokens // max_ips)
        return attn.to_out[0](o)
