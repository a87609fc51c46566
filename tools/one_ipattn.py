#!/usr/bin/env python
"""Run the fused text + masked-IP cross-attention on ONE shape a few times (target for rocprofv3 --pmc passes).
    python tools/one_ipattn.py B heads H W [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import ops
B, h, H, W = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g, device="cuda").half()
N, C = H * W, h * 64
q, kt, ki, vtt, vti = R(B, N, C), R(B, 96, C), R(B, 96, C), R(B, C, 96), R(B, C, 96)
bbox = torch.zeros(B, 4, 4, device="cuda")
bbox[B // 2:, 0] = torch.tensor([0.05, 0.1, 0.5, 0.95]); bbox[B // 2:, 1] = torch.tensor([0.5, 0.1, 0.95, 0.95])
ops.masked_ip_attention(q, kt, vtt, ki, vti, bbox, h, (H, W), 0.6)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(reps):
    ops.masked_ip_attention(q, kt, vtt, ki, vti, bbox, h, (H, W), 0.6)
ev[1].record(); torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) / reps * 1e3
print(f"ip_attn B={B} h={h} N={N}: {us:.1f} us, Q+O {2 * B * N * C * 2 / us / 1e6:.2f} TB/s")
