#!/usr/bin/env python
"""gemm_row640_kernel (128 rows x all 640 columns per block) vs the automatic choice without it (gemm_row_variant 1: the 128 x 128
one-buffer kernel for K = 640, the 256 x 256 kernel for K = 2560) at the 640-channel level's shapes; interleaved rounds, 10 launches
per number; TB/s = algorithmic bytes (A + W + C [+ residual]) / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
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


for name, M, K, res in [("proj_L1 +res", 262144, 640, True), ("to_q_L1", 262144, 640, False), ("ff2_L1 +res", 262144, 2560, True),
                        ("proj_L1 +res B=32", 131072, 640, True), ("proj_L1 +res B=16", 65536, 640, True), ("proj_L1 +res B=8", 32768, 640, True),
                        ("ff2_L1 +res B=16", 65536, 2560, True)]:
    N = 640
    x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
    r = R(M, N) if res else None
    y = torch.empty(M, N, dtype=torch.float16, device="cuda")
    rows, outs = {1: [], 2: []}, {}
    for rnd in range(3):
        for v in (1, 2):
            lib.ds_set_option(b"gemm_row_variant", v)
            outs[v] = ops.gemm(x, w, b, residual=r).clone()
            rows[v].append(timed(lambda: ops.gemm(x, w, b, residual=r, out=y)))
    lib.ds_set_option(b"gemm_row_variant", 0)
    fl = 2.0 * M * N * K
    by = 2.0 * (M * K + N * K + M * N * (2 if res else 1))
    print(f"{name:20s} M={M:6d} K={K:4d} | without {min(rows[1]):7.1f} us {fl / min(rows[1]) / 1e6:5.0f} TF {by / min(rows[1]) / 1e6:5.2f} TB/s | "
          f"row kernel {min(rows[2]):7.1f} us {fl / min(rows[2]) / 1e6:5.0f} TF {by / min(rows[2]) / 1e6:5.2f} TB/s | bit-identical {torch.equal(outs[1], outs[2])}", flush=True)
    del x, w, r, y, outs
