#!/usr/bin/env python
"""Interleaved A/B on the shapes that are NOT on the ping-pong kernel's main road at UNet batch 32 / 64: the level-1
K = N = 640 projections (automatic: 128 x 128 one-buffer kernel) and the batched V^T projections (automatic since round 3:
ping-pong with the batch folded into the tile walk).  Variants: 0 automatic, 3 ping-pong forced, 8 one-buffer 128 x 128,
2 two-buffer 128 x 128.  min of 3 rounds x 20 launches, HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.5).half()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


VARS = (0, 3, 8, 2)
for B in (32, 64):
    for name, M, N, K, mode in [("proj_L1 +res", B * 4096, 640, 640, "res"), ("to_q_L1", B * 4096, 640, 640, None),
                                ("ff2_L1 +res", B * 4096, 640, 2560, "res"), ("proj_L2 +res", B * 1024, 1280, 1280, "res")]:
        x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
        res = R(M, N) if mode == "res" else None
        y = ops.gemm(x, w, b, residual=res)
        run = lambda: ops.gemm(x, w, b, residual=res, out=y)
        t = {v: [] for v in VARS}
        for rnd in range(3):
            for v in (VARS if rnd % 2 == 0 else VARS[::-1]):
                lib.ds_set_option(b"gemm_variant", v)
                t[v].append(timed(run))
        lib.ds_set_option(b"gemm_variant", 0)
        fl = 2.0 * M * N * K
        print(f"B={B:2d} {name:13s} M={M:6d} N={N:4d} K={K:4d} | " + " | ".join(f"v{v}: {min(t[v]):7.1f} us {fl / min(t[v]) / 1e6:5.0f} TF" for v in VARS), flush=True)
    for name, C, N in [("V^T L2", 1280, 1024), ("V^T L1", 640, 4096)]:
        x, wv = R(B, N, C), R(C, C) * (C ** -0.5) * 2
        out = ops.gemm_batched_nt(wv, x)
        run = lambda: ops.gemm_batched_nt(wv, x, out=out)
        t = {v: [] for v in VARS}
        for rnd in range(3):
            for v in (VARS if rnd % 2 == 0 else VARS[::-1]):
                lib.ds_set_option(b"gemm_variant", v)
                t[v].append(timed(run))
        lib.ds_set_option(b"gemm_variant", 0)
        fl = 2.0 * B * C * N * C
        print(f"B={B:2d} {name:13s} [{C} x {C}] x [{B} x {N} x {C}]^T | " + " | ".join(f"v{v}: {min(t[v]):7.1f} us {fl / min(t[v]) / 1e6:5.0f} TF" for v in VARS), flush=True)
