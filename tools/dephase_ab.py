#!/usr/bin/env python
"""De-phasing the two co-resident blocks of a CU (conv_halo256_kernel: gemm_debug bits 12-13 = delay units of s_sleep 127 for
blocks 256..511; self_attn_sp_kernel: attn_variant 5..7 = 1..3 units of s_sleep 40): us per launch per setting, interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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


shapes = [("L0_320", B, 128, 128, 320, 320, False), ("L1_640", B, 64, 64, 640, 640, False), ("L2_1280", B, 32, 32, 1280, 1280, False),
          ("up0_2560", B, 32, 32, 2560, 1280, False), ("up1_1920", B, 64, 64, 1920, 640, False), ("up2_960", B, 128, 128, 960, 320, False),
          ("upsample_L2", B, 32, 32, 1280, 1280, True)]
for name, Bc, H, W, Cin, Cout, up in shapes:
    x, w, b = R(Bc, H, W, Cin), R(Cout, 3, 3, Cin) * ((9 * Cin) ** -0.5) * 2, R(Cout)
    rb = R(Bc, Cout)
    best, ref = {}, None
    for rnd in range(3):
        for units in (0, 1, 2, 3):
            lib.ds_set_option(b"gemm_debug", units << 12)
            y = ops.conv3x3(x, w, b, rowbias=rb, upsample=up)
            ref = y if ref is None else ref
            assert torch.equal(y, ref)
            best[units] = min(best.get(units, 1e30), timed(lambda: ops.conv3x3(x, w, b, rowbias=rb, upsample=up)))
    lib.ds_set_option(b"gemm_debug", 0)
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    fl = 2.0 * Bc * Ho * Wo * Cout * 9 * Cin
    print(f"conv {name:12s} B={Bc} " + "  ".join(f"delay {u}: {best[u]:8.1f} us {fl / best[u] / 1e6:6.0f} TF" for u in best), flush=True)
    del x, w, b, rb, y, ref
    torch.cuda.empty_cache()
for (Bq, h, N) in [(B, 20, 1024), (B, 10, 4096), (8, 20, 1024), (2, 10, 16384)]:
    C = h * 64
    q, k, vt = R(Bq, N, C) * 4.0, R(Bq, N, C) * 2.0, R(Bq, h, 64, N)
    best, ref = {}, None
    for rnd in range(3):
        for var in (3, 5, 6, 7):
            lib.ds_set_option(b"attn_variant", var)
            y = ops.self_attention(q, k, vt, h)
            ref = y if ref is None else ref
            assert torch.equal(y, ref)
            best[var] = min(best.get(var, 1e30), timed(lambda: ops.self_attention(q, k, vt, h)))
    lib.ds_set_option(b"attn_variant", 0)
    fl = 4.0 * Bq * h * N * N * 64
    print(f"attn B={Bq} h={h} N={N} " + "  ".join(f"v{v}: {best[v]:8.1f} us {fl / best[v] / 1e6:6.0f} TF" for v in best), flush=True)
# ---- gemm_pp_kernel: staggered block start (gemm_debug 16384) on the batch-64 shapes
from diffsensei_amd.engine import pack_geglu
for name, M, N, K, mode in [("ff1 geglu", 65536, 10240, 1280, "geglu"), ("qk", 65536, 2560, 1280, None), ("out +res", 65536, 1280, 1280, "res"),
                            ("ff2 +res", 65536, 1280, 5120, "res"), ("ff1_L1 geglu", 262144, 5120, 640, "geglu")]:
    x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
    res = R(M, N) if mode == "res" else None
    if mode == "geglu":
        w, b = pack_geglu(w, b)
    y = torch.empty((M, N // 2 if mode == "geglu" else N), dtype=torch.float16, device="cuda")
    run = lambda: ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"), out=y)
    best = {}
    for rnd in range(3):
        for dbg in (0, 16384):
            lib.ds_set_option(b"gemm_debug", dbg)
            best[dbg] = min(best.get(dbg, 1e30), timed(run))
    lib.ds_set_option(b"gemm_debug", 0)
    fl = 2.0 * M * N * K
    print(f"gemm_pp {name:14s} " + "  ".join(f"dbg {d}: {best[d]:8.1f} us {fl / best[d] / 1e6:6.0f} TF" for d in best), flush=True)
    del x, w, b, res, y
    torch.cuda.empty_cache()
