#!/usr/bin/env python
"""Interleaved A/B (3 rounds, alternating order) of the automatic GEMM dispatch vs gemm_pp_kernel forced, on the level-2
transformer shapes at UNet batches 6..16 - the data behind the `single` rule in gemm.hip choose()."""
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


for B in (6, 8, 10, 12, 16):
    for name, M, N, K, mode in [("ff2_L2", B * 1024, 1280, 5120, "res"), ("proj_L2", B * 1024, 1280, 1280, "res"),
                                ("qk_L2", B * 1024, 2560, 1280, None), ("qk_L1", B * 4096, 1280, 640, None),
                                ("ff2_L1", B * 4096, 640, 2560, "res"), ("proj_L1", B * 4096, 640, 640, "res")]:
        x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
        res = R(M, N) if mode == "res" else None
        y = ops.gemm(x, w, b, residual=res)
        run = lambda: ops.gemm(x, w, b, residual=res, out=y)
        t = {0: [], 3: [], 8: []}
        for rnd in range(3):
            for v in ((0, 3, 8) if rnd % 2 == 0 else (8, 3, 0)):
                lib.ds_set_option(b"gemm_variant", v)
                t[v].append(timed(run))
        lib.ds_set_option(b"gemm_variant", 0)
        T = ((M + 255) // 256) * ((N + 255) // 256)
        f = lambda v: "/".join(f"{u:6.1f}" for u in t[v])
        print(f"B={B:2d} {name:8s} tiles {T:4d} | auto {f(0)} | pp {f(3)} | glds128 {f(8)} us", flush=True)
