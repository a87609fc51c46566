#!/usr/bin/env python
"""GEMM ablation on the GPU: per (variant, debug) time of one shape.  debug: 1 = no MFMA/ds_read, 2 = no tile loads,
3 = neither (barriers + epilogue only).  Results of debug != 0 runs are garbage by construction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
shapes = [(8192, 10240, 1280), (8192, 1280, 5120), (32768, 640, 2560)]
if len(sys.argv) > 3:
    shapes = [tuple(int(v) for v in sys.argv[1:4])]
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in shapes:
    x = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).half()
    y = torch.empty(M, N, dtype=torch.float16, device="cuda")
    for variant in (2, 5, 6):
        row = []
        for dbg in (0, 1, 2, 3):
            lib.ds_set_option(b"gemm_variant", variant)
            lib.ds_set_option(b"gemm_debug", dbg)
            ops.gemm(x, w, out=y)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                ops.gemm(x, w, out=y)
            ev[1].record()
            torch.cuda.synchronize()
            us = ev[0].elapsed_time(ev[1]) * 100
            row.append(f"dbg{dbg}: {us:7.1f} us ({2.0 * M * N * K / us / 1e6:6.1f} TF-eq)")
        print(f"M={M} N={N} K={K} v{variant}  " + "  ".join(row), flush=True)
lib.ds_set_option(b"gemm_debug", 0)
lib.ds_set_option(b"gemm_variant", 0)
