#!/usr/bin/env python
"""Level-1 (640-channel) GEMMs at UNet batch 64: kernel families side by side (gemm_variant 0 auto, 3 ping-pong 256x256, 2 two-buffer
128x128, 8 one-buffer 128x128), interleaved rounds, 10 launches per number."""
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


for name, M, N, K, res in [("ff2_L1 +res", 262144, 640, 2560, True), ("proj_L1 +res", 262144, 640, 640, True), ("to_q_L1", 262144, 640, 640, False),
                           ("shortcut_L0 960->320", 1048576, 320, 960, False)]:
    x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
    r = R(M, N) if res else None
    y = torch.empty(M, N, dtype=torch.float16, device="cuda")
    rows = {}
    for rnd in range(2):
        for v in (0, 3, 2, 8):
            lib.ds_set_option(b"gemm_variant", v)
            rows.setdefault(v, []).append(timed(lambda: ops.gemm(x, w, b, residual=r, out=y)))
    lib.ds_set_option(b"gemm_variant", 0)
    fl = 2.0 * M * N * K
    print(f"{name:22s} M={M} N={N} K={K} | " + " | ".join(f"v{v}: {min(t):7.1f} us {fl / min(t) / 1e6:5.0f} TF" for v, t in rows.items()), flush=True)
    del x, w, r, y
