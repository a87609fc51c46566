"""Generate tests/golden/*.npz by EXECUTING the reference's importable modules (CPU, fp32).

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python -m oracle.make_golden

Imports reference src/models/attention_processor.py and src/models/resampler.py unmodified (they depend
only on torch), drives them with seeded inputs through a duck-typed `attn` stub that has exactly the
attributes the processors touch (to_q/to_k/to_v/to_out/heads/spatial_norm/group_norm/norm_cross/
residual_connection/rescale_output_factor), and stores inputs, weights and outputs.  The oracle
(oracle/attention_ref.py, oracle/resampler_ref.py) and the HIP path are both tested against these files.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("DIFFSENSEI_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class _AttnStub(nn.Module):
    """Stand-in for diffusers.models.attention_processor.Attention (only what the processors read)."""

    def __init__(self, dim, cross_dim, heads):
        super().__init__()
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(cross_dim, dim, bias=False)
        self.to_v = nn.Linear(cross_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim, bias=True), nn.Dropout(0.0)])
        self.heads = heads
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = False
        self.residual_connection = False
        self.rescale_output_factor = 1.0


MASK_CASES = [
    # name, H, W, bboxes [B,4,4]
    ("square32", 32, 32, [[[0.05, 0.10, 0.50, 0.95], [0.50, 0.10, 0.95, 0.95], [0, 0, 0, 0], [0, 0, 0, 0]],
                          [[0, 0, 0, 0]] * 4]),
    ("odd33_exact_half", 33, 33, [[[0.0, 0.0, 0.5, 0.5], [0.5, 0.5, 1.0, 1.0], [0.25, 0.25, 0.75, 0.75],
                                   [0.03125, 0.0625, 0.96875, 0.9375]]]),
    ("wide24x40", 24, 40, [[[0.1, 0.2, 0.4, 0.9], [0.35, 0.0, 0.8, 0.55], [0.9, 0.9, 1.0, 1.0], [0, 0, 0, 0]]]),
    ("tall48x32", 48, 32, [[[0.0, 0.0, 1.0, 1.0], [0.2, 0.3, 0.21, 0.31], [0.6, 0.1, 0.3, 0.9], [0.5, 0.5, 0.5, 0.5]]]),
    ("lvl16", 16, 16, [[[0.05, 0.10, 0.50, 0.95], [0.50, 0.10, 0.95, 0.95], [0, 0, 0, 0], [0, 0, 0, 0]]]),
    ("tiny8x12", 8, 12, [[[0.0, 0.0, 0.45, 1.0], [0.46, 0.0, 1.0, 1.0], [0.2, 0.2, 0.8, 0.8], [0.0, 0.5, 1.0, 0.5]]]),
]


def main():
    sys.path.insert(0, REF)
    from src.models.attention_processor import AttnProcessor2_0, MaskedIPAttnProcessor2_0
    from src.models.resampler import Resampler
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)

    # ---- 1. region masks (reference prepare_attention_mask_ip), stored as uint8 "masked" flags
    proc = MaskedIPAttnProcessor2_0(hidden_size=64, cross_attention_dim=32, num_ip_tokens=64, num_dummy_tokens=16)
    out = {}
    for name, h, w, boxes in MASK_CASES:
        bbox = torch.tensor(boxes, dtype=torch.float32)
        hs = torch.zeros(bbox.shape[0], h * w, 64)
        m = proc.prepare_attention_mask_ip(bbox, hs, 1, h / w)      # [B,1,N,80]
        out[name + "_bbox"] = bbox.numpy()
        out[name + "_hw"] = np.array([h, w], dtype=np.int32)
        out[name + "_masked"] = (m[:, 0] < -1.0).numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "ip_region_masks.npz"), **out)

    # ---- 2. masked IP cross-attention processor end to end (small dims, SDXL token counts 77 + 80)
    dim, xdim, heads, hh, ww, b = 128, 64, 2, 16, 16, 2
    attn = _AttnStub(dim, xdim, heads)
    proc = MaskedIPAttnProcessor2_0(hidden_size=dim, cross_attention_dim=xdim, scale=0.6, num_ip_tokens=64,
                                    num_dummy_tokens=16)
    x = torch.randn(b, hh * ww, dim)
    enc = torch.randn(b, 77 + 80, xdim)
    bbox = torch.tensor([[[0.05, 0.10, 0.50, 0.95], [0.50, 0.10, 0.95, 0.95], [0, 0, 0, 0], [0, 0, 0, 0]],
                         [[0, 0, 0, 0]] * 4], dtype=torch.float32)
    with torch.no_grad():
        y = proc(attn, x, encoder_hidden_states=enc, bbox=bbox, aspect_ratio=hh / ww)
    np.savez_compressed(
        os.path.join(OUT, "masked_ip_attn.npz"),
        x=x.numpy(), enc=enc.numpy(), bbox=bbox.numpy(), hw=np.array([hh, ww], np.int32), heads=np.int32(heads),
        scale=np.float32(0.6), wq=attn.to_q.weight.detach().numpy(), wk=attn.to_k.weight.detach().numpy(),
        wv=attn.to_v.weight.detach().numpy(), wk_ip=proc.to_k_ip.weight.detach().numpy(),
        wv_ip=proc.to_v_ip.weight.detach().numpy(), wo=attn.to_out[0].weight.detach().numpy(),
        bo=attn.to_out[0].bias.detach().numpy(), y=y.numpy())

    # ---- 3. self-attention processor
    attn1 = _AttnStub(dim, dim, heads)
    sp = AttnProcessor2_0()
    with torch.no_grad():
        y1 = sp(attn1, x)
    np.savez_compressed(
        os.path.join(OUT, "self_attn.npz"), x=x.numpy(), heads=np.int32(heads),
        wq=attn1.to_q.weight.detach().numpy(), wk=attn1.to_k.weight.detach().numpy(),
        wv=attn1.to_v.weight.detach().numpy(), wo=attn1.to_out[0].weight.detach().numpy(),
        bo=attn1.to_out[0].bias.detach().numpy(), y=y1.numpy())

    # ---- 4. Resampler (small dims; structure identical to configs/model/diffsensei.yaml)
    rs = Resampler(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, num_dummy_tokens=16, embedding_dim=96,
                   magi_embedding_dim=64, output_dim=256, ff_mult=4).eval()
    xi = torch.randn(1, 4, 17, 96)
    mg = torch.randn(1, 4, 64)
    with torch.no_grad():
        yo = rs(xi, mg)
        yz = rs(torch.zeros_like(xi), torch.zeros_like(mg))
    d = {"in_x": xi.numpy(), "in_magi": mg.numpy(), "out": yo.numpy(), "out_zero": yz.numpy()}
    for k, v in rs.state_dict().items():
        d["sd." + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "resampler.npz"), **d)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
