#!/usr/bin/env python
"""First contact for the experimental 4-wave GEMM (csrc/experimental/gemm_w4.hip; library built with
`python -m diffsensei_amd.build --experimental`): bit-equality with the register-staged kernel (same MFMA order per output),
fp32 reference, and interleaved timing against the ping-pong kernel and F.linear on the UNet's level-2 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from diffsensei_amd import _lib, ops
from diffsensei_amd.engine import pack_geglu

lib = _lib.load()
assert lib.ds_set_option(b"gemm_variant", 12) == 0, "library was not built with --experimental"
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


bad = 0
for name, M, N, K, mode in [("small", 512, 512, 128, "res"), ("3 tiles/block", 4096, 12288, 256, None),
                            ("qk_L2", 32768, 2560, 1280, None), ("out_L2 +res", 32768, 1280, 1280, "res"),
                            ("ff2_L2 +res", 32768, 1280, 5120, "res"), ("ff1_L2 geglu", 32768, 10240, 1280, "geglu")]:
    x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
    res = R(M, N) if mode == "res" else None
    if mode == "geglu":
        w, b = pack_geglu(w, b)
    run = lambda out=None: ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"), out=out)
    lib.ds_set_option(b"gemm_variant", 1)
    ref = run().clone()
    y = torch.empty_like(ref)
    lib.ds_set_option(b"gemm_variant", 12)
    n = 0
    for _ in range(10):
        y.zero_(); run(y); n += int(not torch.equal(y, ref))
    bad += n
    t = {}
    for rnd in range(2):
        for v in (12, 3):
            lib.ds_set_option(b"gemm_variant", v)
            t.setdefault(v, []).append(timed(lambda: run(y)))
    lib.ds_set_option(b"gemm_variant", 0)
    flop = 2.0 * M * N * K
    tl = timed(lambda: F.linear(x, w, b)) if mode != "geglu" else float("nan")
    print(f"{name:14s} M={M:6d} N={N:5d} K={K:4d} | w4 {min(t[12]):7.1f} us {flop / min(t[12]) / 1e6:5.0f} TF | pp {min(t[3]):7.1f} us "
          f"{flop / min(t[3]) / 1e6:5.0f} TF | F.linear {tl:7.1f} us | launches differing from the register-staged kernel: {n}/10 "
          f"(max |diff| {(y.float() - ref.float()).abs().max().item():.3g})", flush=True)
print("TOTAL MISMATCHES", bad)
