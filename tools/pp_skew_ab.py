#!/usr/bin/env python
"""Experiment: start skew of half the XCDs in gemm_pp_kernel (knob gemm_pp_skew, 1/1000 of an estimated tile time) on the UNet's
shapes at batch 32: interleaved rounds, min of 3 x 20 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
from diffsensei_amd.engine import pack_geglu
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


SK = [int(v) for v in os.environ.get("SKEWS", "0,300,500,700").split(",")]
for name, M, N, K, mode in [("ff1_L2 geglu", 32768, 10240, 1280, "geglu"), ("qk_L2", 32768, 2560, 1280, None),
                            ("out_L2 +res", 32768, 1280, 1280, "res"), ("ff2_L2 +res", 32768, 1280, 5120, "res"),
                            ("ff1_L1 geglu", 131072, 5120, 640, "geglu"), ("qk_L1", 131072, 1280, 640, None),
                            ("ff2_L1 +res", 131072, 640, 2560, "res"), ("out_L2 b64", 65536, 1280, 1280, "res")]:
    x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
    res = R(M, N) if mode == "res" else None
    if mode == "geglu":
        w, b = pack_geglu(w, b)
    y = ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"))
    ref = y.clone()
    run = lambda: ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"), out=y)
    t = {s: [] for s in SK}
    same = True
    for rnd in range(3):
        for s in (SK if rnd % 2 == 0 else SK[::-1]):
            lib.ds_set_option(b"gemm_pp_skew", s)
            t[s].append(timed(run))
            same = same and torch.equal(y, ref)
    lib.ds_set_option(b"gemm_pp_skew", 0)
    fl = 2.0 * M * N * K
    print(f"{name:13s} M={M:6d} N={N:5d} K={K:4d} | " + " | ".join(f"skew {s:4d}: {min(t[s]):7.1f} us {fl / min(t[s]) / 1e6:5.0f} TF" for s in SK) + f" | identical {same}", flush=True)
