#!/usr/bin/env python
"""A/B of the two halo-conv block shapes (8x16 vs 16x16 output pixels) on the UNet / VAE conv shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.5).half()
shapes = [("L0_320", B, 128, 128, 320, 320, False), ("L1_640", B, 64, 64, 640, 640, False), ("L2_1280", B, 32, 32, 1280, 1280, False),
          ("up0_2560", B, 32, 32, 2560, 1280, False), ("up1_1920", B, 64, 64, 1920, 640, False), ("up2_960", B, 128, 128, 960, 320, False),
          ("upsample_L2", B, 32, 32, 1280, 1280, True), ("upsample_L1", B, 64, 64, 640, 640, True)]
for name, Bc, H, W, Cin, Cout, up in shapes:
    x, w, b = R(Bc, H, W, Cin), R(Cout, 3, 3, Cin) * ((9 * Cin) ** -0.5) * 2, R(Cout)
    rb = R(Bc, Cout)
    outs, row = {}, []
    for rnd in range(2):
        for v in (1, 2):
            lib.ds_set_option(b"conv_halo_variant", v)
            y = ops.conv3x3(x, w, b, rowbias=rb, upsample=up)
            torch.cuda.synchronize()
            outs[v] = y
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                ops.conv3x3(x, w, b, rowbias=rb, upsample=up)
            ev[1].record()
            torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / 10
            Ho, Wo = (2 * H, 2 * W) if up else (H, W)
            row.append(f"v{v}: {2.0 * Bc * Ho * Wo * Cout * 9 * Cin / ms / 1e9:7.1f} TF ({ms * 1e3:7.1f} us)")
    d = (outs[1].float() - outs[2].float()).abs().max().item()
    print(f"{name:12s} B={Bc} " + "  ".join(row) + f"  maxdiff {d:.3g}", flush=True)
lib.ds_set_option(b"conv_halo_variant", 0)
