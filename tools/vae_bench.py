#!/usr/bin/env python
"""GPU timing of the VAE decoder at SDXL shapes in both storage modes: per-image ms and algorithmic TFLOP/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsensei_amd.vae import VaeConfig, VaeDecoderEngine

cfg = VaeConfig()
for precision, (B, hw) in [(pr, sh) for pr in ("fp16-scaled", "bf16") for sh in [(1, 128), (4, 128), (4, 64)]]:
    eng = VaeDecoderEngine.init_random(cfg, 0, "cuda", precision=precision)
    lat = torch.randn(B, 4, hw, hw, device="cuda") * 0.9
    img = eng.decode(lat, return_dict=False, scaling_factor=cfg.scaling_factor, denormalize=True)[0]
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(3):
        eng.decode(lat, return_dict=False, scaling_factor=cfg.scaling_factor, denormalize=True)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 3
    fl = eng.decode_flops(hw, hw) * B
    print(f"{precision:11s} B={B} latent {hw}x{hw} -> {8*hw}x{8*hw}: {ms:8.2f} ms  ({ms / B:7.2f} ms/image, {fl / ms / 1e9:7.1f} TFLOP/s algorithmic, "
          f"{fl / B / 1e12:.2f} TFLOP/image)  finite={bool(torch.isfinite(img).all())}", flush=True)
