#!/usr/bin/env python
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.5).half()
B = 8
for name, H, W, Cin, Cout in [("L0_320", 128, 128, 320, 320), ("L1_640", 64, 64, 640, 640), ("L2_1280", 32, 32, 1280, 1280),
                              ("up0_2560", 32, 32, 2560, 1280), ("up1_1920", 64, 64, 1920, 640), ("up2_960", 128, 128, 960, 320)]:
    x, w, b = R(B, H, W, Cin), R(Cout, 3, 3, Cin) * ((9 * Cin) ** -0.5) * 2, R(Cout)
    row = []
    for rnd in range(2):
        for (var, dbg) in ((1, 0), (2, 0), (2, 8)):
            lib.ds_set_option(b"gemm_variant", var)
            lib.ds_set_option(b"gemm_debug", dbg)
            ops.conv3x3(x, w, b)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                ops.conv3x3(x, w, b)
            ev[1].record()
            torch.cuda.synchronize()
            us = ev[0].elapsed_time(ev[1]) * 100
            row.append(f"v{var}d{dbg}: {2.0 * B * H * W * Cout * 9 * Cin / us / 1e6:6.1f}")
    print(f"{name:10s} " + "  ".join(row), flush=True)
lib.ds_set_option(b"gemm_debug", 0); lib.ds_set_option(b"gemm_variant", 0)
