#!/usr/bin/env python
"""GroupNorm(+SiLU) at the UNet's shapes and batch 64: round-3 geometry (gn_variant 1) vs round-4 (0), interleaved, 10 launches per
number; effective bandwidth = (read x twice + write y) / time; both must match the fp32 reference at the usual tolerance."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from diffsensei_amd import _lib, ops
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device="cuda").manual_seed(0)
for HW, C1, C2 in [(16384, 320, 0), (4096, 640, 0), (1024, 1280, 0), (1024, 1280, 1280), (4096, 1280, 640), (4096, 640, 640),
                   (16384, 640, 320), (16384, 320, 320)]:
    C = C1 + C2
    x1 = torch.randn(B, HW, C1, generator=g, device="cuda").half()
    x2 = torch.randn(B, HW, C2, generator=g, device="cuda").half() if C2 else None
    gam, bet = torch.randn(C, generator=g, device="cuda").half(), torch.randn(C, generator=g, device="cuda").half()
    rows, outs = {0: [], 1: [], 2: []}, {}
    for rnd in range(3):
        for v in (1, 0, 2):
            lib.ds_set_option(b"gn_variant", v)
            outs[v] = ops.groupnorm(x1, gam, bet, 32, 1e-5, True, x2=x2)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                ops.groupnorm(x1, gam, bet, 32, 1e-5, True, x2=x2)
            ev[1].record(); torch.cuda.synchronize()
            rows[v].append(ev[0].elapsed_time(ev[1]) * 100)
    lib.ds_set_option(b"gn_variant", 0)
    gb = 3.0 * 2 * B * HW * C / 1e9
    d = (outs[0].float() - outs[1].float()).abs().max().item()
    xc = x1 if x2 is None else torch.cat([x1, x2], -1)
    ref = F.silu(F.group_norm(xc[:2].float().transpose(1, 2), 32, gam.float(), bet.float(), 1e-5)).transpose(1, 2)
    e = (outs[0][:2].float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"B={B} HW={HW:5d} C={C1}+{C2:<4d} | round 3 {min(rows[1]):7.1f} us {gb / min(rows[1]) * 1e3:5.2f} TB/s | round 4 {min(rows[0]):7.1f} us "
          f"{gb / min(rows[0]) * 1e3:5.2f} TB/s | 8 in flight {min(rows[2]):7.1f} us {gb / min(rows[2]) * 1e3:5.2f} TB/s | max |new - old| {d:.2e}, new vs fp32 {e:.2e}", flush=True)
