#!/usr/bin/env python
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in [(8192, 10240, 1280), (8192, 1280, 5120), (32768, 640, 2560), (8192, 1280, 1280), (32768, 5120, 640), (8192, 1280, 11520)]:
    x = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).half()
    y = torch.empty(M, N, dtype=torch.float16, device="cuda")
    ref = None
    row = []
    for rnd in range(2):
        for dbg in (0, 8):
            lib.ds_set_option(b"gemm_variant", 2)
            lib.ds_set_option(b"gemm_debug", dbg)
            ops.gemm(x, w, out=y)
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            diff = (y.float() - ref.float()).abs().max().item()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                ops.gemm(x, w, out=y)
            ev[1].record()
            torch.cuda.synchronize()
            us = ev[0].elapsed_time(ev[1]) * 100
            row.append(f"dbg{dbg}: {us:7.1f} us ({2.0 * M * N * K / us / 1e6:6.1f} TF) d={diff:.2g}")
    print(f"M={M} N={N} K={K}  " + "  ".join(row), flush=True)
lib.ds_set_option(b"gemm_debug", 0)
