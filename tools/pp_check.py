#!/usr/bin/env python
"""GPU check of the 256x256 ping-pong GEMM (gemm_variant 3) against the register-staged kernel (variant 1) and torch.

    python tools/pp_check.py            # correctness sweep, then timings on the UNet shapes (batch 16)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffsensei_amd import _lib, ops  # noqa: E402


def main():
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    R = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).half()
    bad = 0
    for (M, N, K) in [(256, 256, 128), (256, 256, 192), (512, 256, 64 * 6), (256, 512, 64 * 8), (1024, 768, 640),
                      (2048, 1280, 1280), (272, 640, 256), (1040, 384, 384), (16, 128, 128), (4096 + 48, 1280 + 128, 640),
                      (70000 // 16 * 16, 640, 640)]:
        for mode in (None, "bias", "res", "geglu", "gelu"):
            x, w = R(M, K), R(N, K) * (K ** -0.5) * 2
            b = R(N) if mode else None
            res = R(M, N) if mode == "res" else None
            outs = []
            for v in (1, 3):
                lib.ds_set_option(b"gemm_variant", v)
                outs.append(ops.gemm(x, w, b, residual=res, geglu=(mode == "geglu"), act=("gelu" if mode == "gelu" else None)))
            torch.cuda.synchronize()
            d = (outs[0].float() - outs[1].float()).abs().max().item()
            ref = x.float() @ w.float().t()
            print(f"M={M} N={N} K={K} {mode}: maxdiff vs v1 {d:.3g}", flush=True)
            bad += d > (1e-3 if mode == "gelu" else 0)
    # repeatability / race screen: same launch many times must be bit-identical
    M, N, K = 4096, 2560, 1280
    x, w, b = R(M, K), R(N, K) * (K ** -0.5) * 2, R(N)
    lib.ds_set_option(b"gemm_variant", 1)
    base = ops.gemm(x, w, b)
    lib.ds_set_option(b"gemm_variant", 3)
    nbad = 0
    for i in range(50):
        y = ops.gemm(x, w, b)
        nbad += int(not torch.equal(y, base))
    print(f"race screen: {nbad}/50 launches differ from v1", flush=True)
    bad += nbad
    lib.ds_set_option(b"gemm_variant", 0)
    print("PP_CHECK", "FAIL" if bad else "OK")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
