#!/usr/bin/env python
"""GroupNorm (three launches) at the UNet's shapes, batch 2 and 64: average of 20 back-to-back calls, HIP events.
Run twice with DIFFSENSEI_LIB pointing at two builds for an A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for B in (2, 8, 64):
    for (HW, C) in [(128 * 128, 320), (64 * 64, 640), (32 * 32, 1280), (32 * 32, 2560), (64 * 64, 1920)]:
        x = torch.randn(B, HW, C, generator=g, device="cuda").half()
        ga, be = torch.ones(C, device="cuda").half(), torch.zeros(C, device="cuda").half()
        for _ in range(3):
            y = ops.groupnorm(x, ga, be, 32, 1e-5, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = ops.groupnorm(x, ga, be, 32, 1e-5, True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        ref = torch.nn.functional.silu(torch.nn.functional.group_norm(x.float().transpose(1, 2), 32, eps=1e-5)).transpose(1, 2)
        err = ((y.float() - ref).norm() / ref.norm()).item()
        print(f"B={B:2d} HW={HW:5d} C={C:4d}  {us:7.1f} us  {2 * x.numel() * 2 / us / 1e6:6.2f} TB/s (read+write once)  rel-L2 vs torch fp32 {err:.2e}", flush=True)
