#!/usr/bin/env python
"""ip_attn_kernel: register-staged 4-wave kernel (variant 1) vs the 8-wave LDS-DMA ring kernel (variant 2), interleaved rounds,
20 launches per number, at the UNet's shapes (heads x 64 channels, N tokens)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g, device="cuda").half()
for (B, h, hw) in [(64, 20, (32, 32)), (64, 10, (64, 64)), (32, 20, (32, 32)), (16, 20, (32, 32)), (8, 20, (32, 32)), (8, 20, (64, 64)), (2, 20, (32, 32))]:
    N, C = hw[0] * hw[1], h * 64
    q, kt, ki, vtt, vti = R(B, N, C), R(B, 96, C), R(B, 96, C), R(B, C, 96), R(B, C, 96)
    bbox = torch.zeros(B, 4, 4, device="cuda"); bbox[B // 2:, 0] = torch.tensor([0.05, 0.1, 0.5, 0.95]); bbox[B // 2:, 1] = torch.tensor([0.5, 0.1, 0.95, 0.95])
    rows, outs = {1: [], 2: [], 0: []}, {}
    for rnd in range(3):
        for v in (1, 2, 0):
            lib.ds_set_option(b"ip_attn_variant", v)
            outs[v] = ops.masked_ip_attention(q, kt, vtt, ki, vti, bbox, h, hw, 0.6)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(20):
                ops.masked_ip_attention(q, kt, vtt, ki, vti, bbox, h, hw, 0.6)
            ev[1].record(); torch.cuda.synchronize()
            rows[v].append(ev[0].elapsed_time(ev[1]) * 50)
    lib.ds_set_option(b"ip_attn_variant", 0)
    gb = 2.0 * 2 * B * N * C / 1e9
    print(f"B={B:3d} heads={h:2d} N={N:5d} | register-staged {min(rows[1]):7.1f} us {gb / min(rows[1]) * 1e3:5.2f} TB/s | ring {min(rows[2]):7.1f} us "
          f"{gb / min(rows[2]) * 1e3:5.2f} TB/s | auto {min(rows[0]):7.1f} us | bit-identical {torch.equal(outs[1], outs[2])}", flush=True)
