#!/usr/bin/env python
"""What the tile boundary of gemm_pp_kernel costs in total - the upper bound of what ANY scheme that overlaps the epilogue with the
neighbouring tiles' MFMAs (phase-shifted wave groups, two blocks per CU, epilogue pieces under the k-loop) could recover.
Ablation build (python -m diffsensei_amd.build --ablation): gemm_debug 128 = NO epilogue (accumulators only marked used),
1 = no MFMA (loads + epilogue only), 129 = neither.  UNet batch-64 shapes, random operands, interleaved rounds.
    DIFFSENSEI_LIB=diffsensei_amd/lib/libdiffsensei_hip_ablation.so python tools/pp_boundary_ablation.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
from diffsensei_amd.engine import pack_geglu
lib = _lib.load()
shapes = [("GEGLU up-projection", 65536, 10240, 1280, "geglu"), ("out-projection +residual", 65536, 1280, 1280, "res"),
          ("FF down-projection +residual", 65536, 1280, 5120, "res"), ("q|k projection", 65536, 2560, 1280, None),
          ("640-level FF down +residual", 262144, 640, 2560, "res")]
g = torch.Generator(device="cuda").manual_seed(0)
lib.ds_set_option(b"gemm_variant", 3)
print("# gemm_pp_kernel, us per launch (TFLOP/s): full | no epilogue (dbg 128) | no MFMA (dbg 1); boundary = full - no epilogue")
for name, M, N, K, mode in shapes:
    x = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).half()
    b = (torch.randn(N, generator=g, device="cuda") * 0.1).half()
    res = (torch.randn(M, N, generator=g, device="cuda")).half() if mode == "res" else None
    if mode == "geglu":
        w, b = pack_geglu(w, b)
    y = torch.empty(M, N // 2 if mode == "geglu" else N, dtype=torch.float16, device="cuda")
    t = {0: [], 128: [], 1: []}
    for rnd in range(3):
        for dbg in (0, 128, 1):
            lib.ds_set_option(b"gemm_debug", dbg)
            ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"), out=y)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"), out=y)
            ev[1].record()
            torch.cuda.synchronize()
            t[dbg].append(ev[0].elapsed_time(ev[1]) * 100)
    lib.ds_set_option(b"gemm_debug", 0)
    full, noepi, nomma = (min(t[k]) for k in (0, 128, 1))
    tf = lambda us: 2.0 * M * N * K / us / 1e6
    tiles = (M // 256) * ((N + 255) // 256)
    rounds = -(-tiles // 256)
    print(f"{name:32s} M={M} N={N} K={K}: {full:8.1f} us ({tf(full):6.1f}) | {noepi:8.1f} us ({tf(noepi):6.1f}) | {nomma:8.1f} us | "
          f"boundary {full - noepi:7.1f} us = {100 * (full - noepi) / full:4.1f} % = {(full - noepi) / rounds:5.2f} us per tile round ({rounds} rounds)", flush=True)
lib.ds_set_option(b"gemm_variant", 0)
