#!/usr/bin/env python
"""Ablation of the 256x256 ping-pong GEMM (variant 3) next to the 128x128 one-buffer kernel (variant 8):
debug 1 = no MFMA, 2 = no tile loads, 3 = neither.  Also cube shapes for comparison with published kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
shapes = [(16384, 10240, 1280), (8192, 8192, 8192)]
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3,8").split(",")]
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in shapes:
    for data in ("randn",):
        x = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
        w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).half()
        if data == "zeros":
            x.zero_(); w.zero_()
        y = torch.empty(M, N, dtype=torch.float16, device="cuda")
        for variant in variants:
            row = []
            for dbg in (0, 8, 16, 24, 18):
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
                extra = ""
                if dbg & 16:
                    pr = y.view(torch.int64).flatten()[:2].tolist()
                    extra = f" {pr[0] / max(pr[1], 1) * 0.1:5.2f} GHz"
                row.append(f"dbg{dbg}: {us:7.1f} us ({2.0 * M * N * K / us / 1e6:6.1f} TF-eq){extra}")
            print(f"M={M} N={N} K={K} {data} v{variant}  " + "  ".join(row), flush=True)
lib.ds_set_option(b"gemm_debug", 0)
lib.ds_set_option(b"gemm_variant", 0)
