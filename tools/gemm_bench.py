#!/usr/bin/env python
"""GPU micro-benchmark of the GEMM / implicit-conv kernel variants on the UNet's real shapes (num_samples=4 -> batch 8).

    python tools/gemm_bench.py [--variants 1,2,3] [--reps 10] [--batch 8]

Interleaved rounds in ONE process (A/B noise is correlated), random fp16 data, HIP-event timing on the launch
stream.  Every variant's output is compared with variant 1 (the register-staged kernel the parity suite pinned).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffsensei_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="1,2,3")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    lib = _lib.load()
    variants = [int(v) for v in args.variants.split(",")]
    B = args.batch
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    R = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).half()
    gemms = [("ff1_geglu_L2", B * 1024, 10240, 1280, "geglu"), ("ff2_L2", B * 1024, 1280, 5120, "res"),
             ("proj_L2", B * 1024, 1280, 1280, "res"), ("qk_L2", B * 1024, 2560, 1280, None),
             ("ff1_geglu_L1", B * 4096, 5120, 640, "geglu"), ("ff2_L1", B * 4096, 640, 2560, "res"),
             ("proj_L1", B * 4096, 640, 640, "res"), ("qk_L1", B * 4096, 1280, 640, None)]
    convs = [("conv_L0_320", B, 128, 128, 320, 320), ("conv_L1_640", B, 64, 64, 640, 640),
             ("conv_L2_1280", B, 32, 32, 1280, 1280), ("conv_up0_2560", B, 32, 32, 2560, 1280),
             ("conv_up1_1920", B, 64, 64, 1920, 640), ("conv_up2_960", B, 128, 128, 960, 320)]
    results = {}
    for name, M, N, K, mode in gemms:
        x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
        res = R(M, N) if mode == "res" else None
        outs = {}
        for v in variants:
            lib.ds_set_option(b"gemm_variant", v)
            y = ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"))
            torch.cuda.synchronize()
            outs[v] = y
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(args.reps):
                ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"), out=y)
            ev[1].record()
            torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / args.reps
            results[(name, v)] = (2.0 * M * N * K / (ms * 1e-3) / 1e12, ms)
        base = outs[variants[0]].float()
        diffs = {v: (outs[v].float() - base).abs().max().item() for v in variants}
        print(f"{name:16s} M={M:6d} N={N:5d} K={K:5d} " +
              " ".join(f"v{v}: {results[(name, v)][0]:7.1f} TF ({results[(name, v)][1] * 1e3:7.1f} us)" for v in variants) +
              f"  maxdiff {max(diffs.values()):.3g}", flush=True)
    for name, Bc, H, W, Cin, Cout in convs:
        x, w, b = R(Bc, H, W, Cin), R(Cout, 3, 3, Cin) * ((9 * Cin) ** -0.5) * 2, R(Cout)
        rb, res = R(Bc, Cout), R(Bc, H, W, Cout)
        outs = {}
        for v in variants:
            lib.ds_set_option(b"gemm_variant", v)
            y = ops.conv3x3(x, w, b, rowbias=rb, residual=res)
            torch.cuda.synchronize()
            outs[v] = y
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(args.reps):
                ops.conv3x3(x, w, b, rowbias=rb, residual=res)
            ev[1].record()
            torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / args.reps
            results[(name, v)] = (2.0 * Bc * H * W * Cout * 9 * Cin / (ms * 1e-3) / 1e12, ms)
        base = outs[variants[0]].float()
        diffs = {v: (outs[v].float() - base).abs().max().item() for v in variants}
        print(f"{name:16s} {Bc}x{H}x{W} {Cin:4d}->{Cout:4d}       " +
              " ".join(f"v{v}: {results[(name, v)][0]:7.1f} TF ({results[(name, v)][1] * 1e3:7.1f} us)" for v in variants) +
              f"  maxdiff {max(diffs.values()):.3g}", flush=True)
    lib.ds_set_option(b"gemm_variant", 0)


if __name__ == "__main__":
    main()
