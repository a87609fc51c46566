#!/usr/bin/env python
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd import _lib, ops
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g, device="cuda").half()
for (B, h, hw) in [(8, 20, (32, 32)), (8, 10, (64, 64)), (32, 20, (32, 32)), (32, 10, (64, 64))]:
    N, C = hw[0] * hw[1], h * 64
    q, kt, ki, vtt, vti = R(B, N, C), R(B, 96, C), R(B, 96, C), R(B, C, 96), R(B, C, 96)
    bbox = torch.zeros(B, 4, 4, device="cuda"); bbox[B // 2:, 0] = torch.tensor([0.05, 0.1, 0.5, 0.95]); bbox[B // 2:, 1] = torch.tensor([0.5, 0.1, 0.95, 0.95])
    row, ref = [], None
    for mb in (1000000, 2048, 1024, 512, 256, 128):
        lib.ds_set_option(b"ip_attn_min_blocks", mb)
        o = ops.masked_ip_attention(q, kt, vtt, ki, vti, bbox, h, hw, 0.6)
        torch.cuda.synchronize()
        if ref is None: ref = o.clone()
        d = (o.float() - ref.float()).abs().max().item()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(20):
            ops.masked_ip_attention(q, kt, vtt, ki, vti, bbox, h, hw, 0.6)
        ev[1].record(); torch.cuda.synchronize()
        row.append(f"min_blocks {mb}: {ev[0].elapsed_time(ev[1]) * 50:6.1f} us (d={d:.2g})")
    print(f"B={B} h={h} N={N}  " + "  ".join(row), flush=True)
lib.ds_set_option(b"ip_attn_min_blocks", 1024)
