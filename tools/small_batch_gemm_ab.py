#!/usr/bin/env python
"""A/B of the two round-2 GEMM changes on the UNet's transformer shapes at num_samples 1 / 4 / 16 (UNet batch 2 / 8 / 32):

  ring     gemm_glds_kernel<64,false,3|4> (ring of LDS buffers) vs the one-buffer kernel      - option "gemm_ring"
  split-K  gemm_pp_kernel with its partial last round cut into S k-slices, S = 1 (off), auto, and forced values, vs the
           automatic dispatch (128x128 kernels where gemm_pp was not chosen)                    - option "gemm_split_k"

Interleaved in ONE process, random fp16 data, HIP-event timing over `--reps` launches.  Output: one line per shape.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffsensei_amd import _lib, ops  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batches", default="2,8,32")
    ap.add_argument("--splits", default="2,3,4,5,8")
    args = ap.parse_args()
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    R = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).half()
    splits = [int(v) for v in args.splits.split(",")]
    opt = lambda k, v: lib.ds_set_option(k, v)
    for B in [int(v) for v in args.batches.split(",")]:
        shapes = [("qk_L2", B * 1024, 2560, 1280, None), ("proj_L2", B * 1024, 1280, 1280, "res"),
                  ("ff1_L2", B * 1024, 10240, 1280, "geglu"), ("ff2_L2", B * 1024, 1280, 5120, "res"),
                  ("qk_L1", B * 4096, 1280, 640, None), ("proj_L1", B * 4096, 640, 640, "res"),
                  ("ff1_L1", B * 4096, 5120, 640, "geglu"), ("ff2_L1", B * 4096, 640, 2560, "res")]
        for name, M, N, K, mode in shapes:
            x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
            res = R(M, N) if mode == "res" else None
            y = ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"))
            run = lambda: ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"), out=y)
            fl = 2.0 * M * N * K
            cols = []
            opt(b"gemm_variant", 0); opt(b"gemm_split_k", 0); opt(b"gemm_ring", 0)
            t_auto = timed(run, args.reps)
            cols.append(f"auto {t_auto:7.1f}us {fl / t_auto / 1e6:6.0f}TF")
            opt(b"gemm_ring", 1)
            t = timed(run, args.reps)
            cols.append(f"noring {t:7.1f}")
            opt(b"gemm_ring", 0)
            opt(b"gemm_variant", 3)
            for s in [1, 0] + splits:
                opt(b"gemm_split_k", s)
                try:
                    t = timed(run, args.reps)
                    cols.append(f"pp/S{'auto' if s == 0 else s} {t:7.1f}")
                except Exception as e:   # shape not supported by the forced family
                    cols.append(f"pp/S{s} n/a")
                    break
            opt(b"gemm_variant", 0); opt(b"gemm_split_k", 0)
            print(f"B={B:2d} {name:8s} M={M:6d} N={N:5d} K={K:4d} | " + " | ".join(cols), flush=True)


if __name__ == "__main__":
    main()
