#!/usr/bin/env python
"""gemm_pp_kernel, round 4 A/B runs in one process (interleaved rounds, HIP events, 10 launches per number):
  (1) first MFMA of a tile with the constant 0 as C operand (gemm_debug 0) vs 128 v_mov zeroing the accumulators
      (gemm_debug 1024, the pre-round-4 loop top) - outputs must be bit-identical;
  (2) the K = N = 640 projections at UNet batch 64 (M = 262144): automatic dispatch (128 x 128 one-buffer kernel) vs the
      256 x 256 ping-pong kernel forced (gemm_variant 3) - the dispatch rule that excludes them dates from M = 65536."""
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


print("(1) zero C operand vs zeroed accumulators")
bad = 0
for name, M, N, K, mode in [("ff1_L2 geglu", 65536, 10240, 1280, "geglu"), ("qk_L2", 65536, 2560, 1280, None),
                            ("out_L2 +res", 65536, 1280, 1280, "res"), ("ff2_L2 +res", 65536, 1280, 5120, "res"),
                            ("qk_L1", 262144, 1280, 640, None), ("ragged +res", 32768 - 48, 1280, 1280, "res")]:
    x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
    res = R(M, N) if mode == "res" else None
    if mode == "geglu":
        w, b = pack_geglu(w, b)
    lib.ds_set_option(b"gemm_variant", 3)
    outs, rows = {}, {0: [], 1024: []}
    for rnd in range(3):
        for d in (0, 1024):
            lib.ds_set_option(b"gemm_debug", d)
            y = ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"))
            outs[d] = y
            rows[d].append(timed(lambda: ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"), out=y)))
    lib.ds_set_option(b"gemm_debug", 0)
    lib.ds_set_option(b"gemm_variant", 0)
    same = torch.equal(outs[0], outs[1024])
    bad += int(not same)
    fl = 2.0 * M * N * K
    print(f"  {name:14s} M={M:6d} N={N:5d} K={K:4d} | zero-C {min(rows[0]):7.1f} us {fl / min(rows[0]) / 1e6:5.0f} TF | "
          f"v_mov {min(rows[1024]):7.1f} us {fl / min(rows[1024]) / 1e6:5.0f} TF | bit-identical {same}", flush=True)
    del x, w, b, res, outs
print("  MISMATCHES", bad)

print("(2) K = N = 640 projections at M = 262144: auto (128 x 128) vs ping-pong forced")
for name, M, N, K, mode in [("proj_L1 +res", 262144, 640, 640, "res"), ("to_q_L1", 262144, 640, 640, None)]:
    x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
    res = R(M, N) if mode == "res" else None
    y = torch.empty(M, N, dtype=torch.float16, device="cuda")
    rows, outs = {0: [], 3: []}, {}
    for rnd in range(3):
        for v in (0, 3):
            lib.ds_set_option(b"gemm_variant", v)
            outs[v] = ops.gemm(x, w, b if mode else None, residual=res).clone()
            rows[v].append(timed(lambda: ops.gemm(x, w, b if mode else None, residual=res, out=y)))
    lib.ds_set_option(b"gemm_variant", 0)
    fl = 2.0 * M * N * K
    byts = 2.0 * (M * K + N * K + M * N * (2 if mode == "res" else 1))
    print(f"  {name:14s} | auto {min(rows[0]):7.1f} us {fl / min(rows[0]) / 1e6:5.0f} TF {byts / min(rows[0]) / 1e6:5.2f} TB/s | "
          f"ping-pong {min(rows[3]):7.1f} us {fl / min(rows[3]) / 1e6:5.0f} TF {byts / min(rows[3]) / 1e6:5.2f} TB/s | "
          f"bit-identical {torch.equal(outs[0], outs[3])}", flush=True)
