#!/usr/bin/env python
"""Experiment: gemm_pp_kernel with the persistent grid shrunk so every round of tiles is full ("gemm_pp_even" 1:
ceil(T / rounds) blocks) vs one block per CU with a partial last round.  On a power-limited part fewer active CUs may
clock higher.  Interleaved, 20 launches per number."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.5).half()


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


for name, M, N, K, mode in [("ff2_L2", 32768, 1280, 5120, "res"), ("proj_L2", 32768, 1280, 1280, "res"),
                            ("qk_L2", 32768, 2560, 1280, None), ("ff1_L2", 32768, 10240, 1280, "geglu"),
                            ("ff2_L1", 131072, 640, 2560, "res"), ("qk_L1", 131072, 1280, 640, None),
                            ("ff1_L1", 131072, 5120, 640, "geglu")]:
    x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
    res = R(M, N) if mode == "res" else None
    y = ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"))
    run = lambda: ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"), out=y)
    row = []
    for rnd in range(2):
        for ev in (0, 1):
            lib.ds_set_option(b"gemm_pp_even", ev)
            t = timed(run)
            row.append(f"even={ev}: {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF")
    lib.ds_set_option(b"gemm_pp_even", 0)
    T = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"{name:8s} M={M:6d} N={N:5d} K={K:4d} tiles {T:5d} ({T / 256:.2f} rounds) | " + " | ".join(row), flush=True)
