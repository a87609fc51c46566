#!/usr/bin/env python
"""gemm_pp_kernel: how much of the gap to the matrix pipe is the L2-miss latency of the operand stream?

Ablation build (DIFFSENSEI_LIB=.../libdiffsensei_hip_ablation.so), gemm_debug 32: every block works on tile (0,0), so every
operand load hits L2 (results are garbage by construction) while the instruction stream, the LDS traffic and the C stores are
unchanged.  gemm_debug 0 next to it, interleaved rounds, UNet batch-64 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
from diffsensei_amd.engine import pack_geglu

lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.5).half()


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


SHAPES = [("ff1_L2 geglu", 65536, 10240, 1280, "geglu"), ("qk_L2", 65536, 2560, 1280, None),
          ("ff2_L2 +res", 65536, 1280, 5120, "res"), ("out_L2 +res", 65536, 1280, 1280, "res"),
          ("ff1_L1 geglu", 262144, 5120, 640, "geglu"), ("qk_L1", 262144, 1280, 640, None), ("cube 8192", 8192, 8192, 8192, None)]
dbgs = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,32").split(",")]
for name, M, N, K, mode in SHAPES:
    x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
    res = R(M, N) if mode == "res" else None
    if mode == "geglu":
        w, b = pack_geglu(w, b)
    y = torch.empty(M, N // 2 if mode == "geglu" else N, dtype=torch.float16, device="cuda")
    lib.ds_set_option(b"gemm_variant", 3)
    run = lambda: ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"), out=y)
    rows = {d: [] for d in dbgs}
    for rnd in range(3):
        for d in dbgs:
            lib.ds_set_option(b"gemm_debug", d)
            rows[d].append(timed(run))
    lib.ds_set_option(b"gemm_debug", 0)
    lib.ds_set_option(b"gemm_variant", 0)
    flop = 2.0 * M * N * K
    print(f"{name:14s} M={M:6d} N={N:5d} K={K:4d} | " + " | ".join(
        f"dbg {d:3d}: {min(rows[d]):7.1f} us {flop / min(rows[d]) / 1e6:5.0f} TF" for d in dbgs), flush=True)
    del x, w, b, res, y
