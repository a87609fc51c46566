#!/usr/bin/env python
"""gemm_pp_kernel tile hand-over A/B (same binary, interleaved, 20 launches per number, UNet batch 32 shapes):

    gemm_debug 0    C stores of a tile keep draining under the next tile's first k-tiles (counted waits + EX_TAIL)
    gemm_debug 256  s_waitcnt vmcnt(0) before the next tile's first k-tile (the round-1 behaviour)
    gemm_debug 512  as 0, C stores carry the non-temporal hint
    gemm_debug 768  as 256, non-temporal

and F.linear (hipBLASLt) on the same box as the yardstick.  Every setting must produce the same bits: the outputs of 0 / 512 /
768 are compared with 256's on every repetition of a 30-launch stress loop (a counted wait that under-waits shows up as a
rare wrong tile, not as a crash).
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from diffsensei_amd import _lib, ops
from diffsensei_amd.engine import pack_geglu

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


SHAPES = [("ff1_L2 geglu", 32768, 10240, 1280, "geglu"), ("qk_L2", 32768, 2560, 1280, None),
          ("out_L2 +res", 32768, 1280, 1280, "res"), ("ff2_L2 +res", 32768, 1280, 5120, "res"),
          ("ff1_L1 geglu", 131072, 5120, 640, "geglu"), ("qk_L1", 131072, 1280, 640, None),
          ("ff2_L1 +res", 131072, 640, 2560, "res"), ("ragged M +res", 32768 - 48, 1280, 1280, "res"),
          ("ragged N", 16384, 2560 - 64, 1280, None)]
bad = 0
for name, M, N, K, mode in SHAPES:
    x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
    res = R(M, N) if mode == "res" else None
    if mode == "geglu":
        w, b = pack_geglu(w, b)
    lib.ds_set_option(b"gemm_variant", 3)      # force the ping-pong kernel (the ragged shapes would not pick it)
    run = lambda out=None: ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"), out=out)
    lib.ds_set_option(b"gemm_debug", 256)
    ref = run().clone()
    y = torch.empty_like(ref)
    mism = {}
    for dbg in (0, 512, 768):
        lib.ds_set_option(b"gemm_debug", dbg)
        n = 0
        for _ in range(30):
            y.zero_()
            run(y)
            n += int(not torch.equal(y, ref))
        mism[dbg] = n
        bad += n
    row = []
    for rnd in range(2):
        for dbg in (0, 256, 512, 768):
            lib.ds_set_option(b"gemm_debug", dbg)
            row.append((dbg, timed(lambda: run(y))))
    lib.ds_set_option(b"gemm_debug", 0)
    lib.ds_set_option(b"gemm_variant", 0)
    flop = 2.0 * M * N * K
    best = {d: min(t for dd, t in row if dd == d) for d in (0, 256, 512, 768)}
    wl = w if mode != "geglu" else None
    tl = ""
    if wl is not None:
        t = timed(lambda: F.linear(x, w, b))
        tl = f" | F.linear {t:7.1f} us {flop / t / 1e6:5.0f} TF"
    fp = (x.float() @ w.float().t() + b.float()) if mode != "geglu" else None
    err = ""
    if fp is not None:
        if res is not None:
            fp = fp + res.float()
        err = f" | rel-L2 vs fp32 {((ref.float() - fp).norm() / fp.norm()).item():.1e}"
    print(f"{name:14s} M={M:6d} N={N:5d} K={K:4d} | " +
          " | ".join(f"dbg {d:3d}: {best[d]:7.1f} us {flop / best[d] / 1e6:5.0f} TF" for d in (0, 256, 512, 768)) +
          f"{tl} | mismatching launches vs dbg 256: {mism}{err}", flush=True)
    del x, w, b, res, ref, y, fp
    torch.cuda.empty_cache()
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
