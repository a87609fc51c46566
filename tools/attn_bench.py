#!/usr/bin/env python
"""Self-attention kernels at the UNet's shapes: interleaved rounds of the software-pipelined kernel (attn_variant 3) and the
automatic choice among self_attn_kernel<1|2> (0); min and median of ROUNDS x 10 launches, HIP events."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
ROUNDS = int(os.environ.get("ROUNDS", "5"))
VARS = [int(v) for v in os.environ.get("VARS", "3,0").split(",")]
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g, device="cuda").half()
for (B, h, N) in [(32, 20, 1024), (32, 10, 4096), (8, 20, 1024), (8, 10, 4096), (2, 20, 1024), (2, 10, 4096), (2, 10, 16384)]:
    C = h * 64
    q, k, vt = R(B, N, C), R(B, N, C), R(B, h, 64, N)
    outs, t = {}, {v: [] for v in VARS}
    for rnd in range(ROUNDS):
        for var in VARS:
            lib.ds_set_option(b"attn_variant", var)
            outs[var] = ops.self_attention(q, k, vt, h)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                ops.self_attention(q, k, vt, h)
            ev[1].record()
            torch.cuda.synchronize()
            t[var].append(ev[0].elapsed_time(ev[1]) * 100)
    lib.ds_set_option(b"attn_variant", 0)
    fl = 4.0 * B * h * N * N * 64
    row = "  ".join(f"var{v}: min {min(t[v]):7.1f} us ({fl / min(t[v]) / 1e6:6.1f} TF) med {statistics.median(t[v]):7.1f}" for v in VARS)
    d = (outs[VARS[0]].float() - outs[VARS[-1]].float()).abs().max().item()
    print(f"B={B:2d} h={h:2d} N={N:5d}  {row}  maxdiff {d:.3g}", flush=True)
