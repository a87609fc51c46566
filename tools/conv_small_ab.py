#!/usr/bin/env python
"""conv_halo_kernel (the 8 x 16-pixel block kernel small batches run): tail tiles (Cout = 320: every third channel tile has 64
valid columns) as full 128-column tiles (gemm_debug 4096) vs split four ways over one 64-column strip (default); interleaved,
10 launches per number, UNet batch-2 / batch-8 shapes; outputs must be bit-identical.  Run it under another library build
(DIFFSENSEI_LIB=...) to compare builds on one box.
    python tools/conv_small_ab.py [batch 2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = torch.Generator(device="cuda").manual_seed(0)
print(f"library {os.environ.get('DIFFSENSEI_LIB', 'default')}")
for name, H, W, Cin, Cout in [("L0 320->320", 128, 128, 320, 320), ("L0 960->320", 128, 128, 960, 320), ("L0 640->320", 128, 128, 640, 320),
                              ("L1 640->640 (no tail)", 64, 64, 640, 640), ("L1 1920->640 (no tail)", 64, 64, 1920, 640),
                              ("L2 1280->1280 (no tail)", 32, 32, 1280, 1280)]:
    x = torch.randn(B, H, W, Cin, generator=g, device="cuda").half()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g, device="cuda") * (9 * Cin) ** -0.5).half()
    b = torch.randn(Cout, generator=g, device="cuda").half()
    rows, outs = {0: [], 4096: []}, {}
    for rnd in range(3):
        for d in (4096, 0):
            lib.ds_set_option(b"gemm_debug", d)
            outs[d] = ops.conv3x3(x, w, b)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                ops.conv3x3(x, w, b)
            ev[1].record(); torch.cuda.synchronize()
            rows[d].append(ev[0].elapsed_time(ev[1]) * 100)
    lib.ds_set_option(b"gemm_debug", 0)
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print(f"{name:24s} B={B} | full tail tiles {min(rows[4096]):8.1f} us {fl / min(rows[4096]) / 1e6:5.0f} TF | split tail {min(rows[0]):8.1f} us "
          f"{fl / min(rows[0]) / 1e6:5.0f} TF | bit-identical {torch.equal(outs[0], outs[4096])}", flush=True)
    del x, w, outs
