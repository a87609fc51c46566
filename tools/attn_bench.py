#!/usr/bin/env python
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g, device="cuda").half()
for (B, h, N) in [(8, 20, 1024), (8, 10, 4096), (2, 20, 1024), (2, 10, 4096)]:
    C = h * 64
    q, k, vt = R(B, N, C), R(B, N, C), R(B, h, 64, N)
    outs, row = {}, []
    for rnd in range(2):
        for var in (1, 0):
            lib.ds_set_option(b"attn_variant", var)
            o = ops.self_attention(q, k, vt, h)
            torch.cuda.synchronize()
            outs[var] = o
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                ops.self_attention(q, k, vt, h)
            ev[1].record()
            torch.cuda.synchronize()
            us = ev[0].elapsed_time(ev[1]) * 100
            row.append(f"var{var}: {us:7.1f} us ({4.0 * B * h * N * N * 64 / us / 1e6:6.1f} TF)")
    d = (outs[0].float() - outs[1].float()).abs().max().item()
    print(f"B={B} h={h} N={N}  " + "  ".join(row) + f"  maxdiff {d:.3g}", flush=True)
lib.ds_set_option(b"attn_variant", 0)
# 64-row GEMM tiles on the N=1280 shapes
for (M, N, K) in [(8192, 1280, 1280), (8192, 1280, 5120), (8192, 2560, 1280), (32768, 640, 640), (32768, 640, 2560)]:
    x = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).half()
    b = R(N); r = R(M, N)
    y = torch.empty(M, N, dtype=torch.float16, device="cuda")
    row = []
    for rnd in range(2):
        for var in (2, 7):
            lib.ds_set_option(b"gemm_variant", var)
            ops.gemm(x, w, b, residual=r, out=y)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                ops.gemm(x, w, b, residual=r, out=y)
            ev[1].record()
            torch.cuda.synchronize()
            us = ev[0].elapsed_time(ev[1]) * 100
            row.append(f"v{var}: {us:6.1f} us ({2.0 * M * N * K / us / 1e6:6.1f} TF)")
    print(f"M={M} N={N} K={K}  " + "  ".join(row), flush=True)
lib.ds_set_option(b"gemm_variant", 0)
