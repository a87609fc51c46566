#!/usr/bin/env python
"""Cost of the erf-GELU in the GEGLU epilogue of the ping-pong GEMM: same launch with / without the erf (debug 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in [(32768, 10240, 1280), (131072, 5120, 640), (32768, 1280, 5120), (32768, 2560, 1280)]:
    x = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).half()
    b = (torch.randn(N, generator=g, device="cuda") * 0.5).half()
    y = torch.empty(M, N // 2, dtype=torch.float16, device="cuda")
    for rnd in range(3):
        row = []
        for variant, dbg in ((3, 0), (3, 128), (3, 0), (3, 128)):
            lib.ds_set_option(b"gemm_variant", variant)
            lib.ds_set_option(b"gemm_debug", dbg)
            ops.gemm(x, w, b, geglu=True, out=y)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(20):
                ops.gemm(x, w, b, geglu=True, out=y)
            ev[1].record()
            torch.cuda.synchronize()
            us = ev[0].elapsed_time(ev[1]) * 50
            row.append(f"v{variant}/dbg{dbg}: {us:7.1f} us ({2.0 * M * N * K / us / 1e6:6.1f} TF)")
        print(f"M={M} N={N} K={K}  " + "  ".join(row), flush=True)
lib.ds_set_option(b"gemm_debug", 0)
lib.ds_set_option(b"gemm_variant", 0)
