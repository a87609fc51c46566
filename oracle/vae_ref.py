"""TEST INFRASTRUCTURE ONLY (oracle): fp32 CPU restatement of the SDXL VAE decode used by the reference pipeline.

Follows reference src/pipelines/pipeline_diffsensei.py:339-367 (`latents / scaling_factor` -> `vae.decode`), whose
`AutoencoderKL` lives in the third-party dependency diffusers (absent from /root/reference; the reference pins
diffusers 0.30.x in requirements).  Restated from the published module structure:

    AutoencoderKL.decode(z)      = Decoder(post_quant_conv(z))                       post_quant_conv: Conv2d(4, 4, 1)
    Decoder                      = conv_in(4 -> C3) -> UNetMidBlock2D -> 4 x UpDecoderBlock2D -> GroupNorm(32, eps 1e-6)
                                   -> SiLU -> conv_out(C0 -> 3)          block_out_channels (C0..C3) = (128,256,512,512)
    UNetMidBlock2D               = ResnetBlock2D, Attention(heads = 1, dim_head = C3, group_norm 32 / 1e-6,
                                   residual_connection), ResnetBlock2D
    UpDecoderBlock2D i           = (layers_per_block + 1) x ResnetBlock2D [+ Upsample2D: nearest x2, Conv2d 3x3]
                                   channels reversed(C): 512, 512, 256, 128; no upsample in the last block
    ResnetBlock2D (temb = None)  = x' + conv2(silu(norm2(conv1(silu(norm1(x)))))),  x' = conv_shortcut(x) (1x1) if Cin != Cout

PARITY UNPINNED: diffusers is not importable here, so no golden vector of the real module exists; the restatement is
checked only for self-consistency (tests/test_oracle_unet.py style) and is the fp32 yard-stick of the bf16 HIP path.
State-dict keys are diffusers' (`decoder.up_blocks.0.resnets.1.conv1.weight`, `post_quant_conv.weight`, ...);
`diffsensei_amd.vae.vae_param_shapes` lists them.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _resnet(sd, p, x, groups, eps):
    h = F.silu(F.group_norm(x, groups, sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], eps))
    h = F.conv2d(h, sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], eps))
    h = F.conv2d(h, sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], padding=1)
    if f"{p}.conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.conv_shortcut.weight"], sd[f"{p}.conv_shortcut.bias"])
    return x + h


def _attention(sd, p, x, groups, eps):
    B, C, H, W = x.shape
    h = F.group_norm(x, groups, sd[f"{p}.group_norm.weight"], sd[f"{p}.group_norm.bias"], eps)
    t = h.view(B, C, H * W).transpose(1, 2)                       # [B, N, C]
    q = F.linear(t, sd[f"{p}.to_q.weight"], sd[f"{p}.to_q.bias"])
    k = F.linear(t, sd[f"{p}.to_k.weight"], sd[f"{p}.to_k.bias"])
    v = F.linear(t, sd[f"{p}.to_v.weight"], sd[f"{p}.to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), dim=-1) @ v   # one head of dim C
    o = F.linear(a, sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(B, C, H, W)


def vae_decode(sd: Dict[str, Tensor], z: Tensor, layers_per_block: int = 2, groups: int = 32, eps: float = 1e-6,
               n_up: int = 4, taps: Optional[dict] = None) -> Tensor:
    """`vae.decode(z)[0]`: z [B,4,h,w] (already divided by scaling_factor) -> image [B,3,8h,8w], fp32.
    `taps`: filled with max |activation| of the residual stream after each block (tests that push the decoder into the
    magnitude range where the real SDXL VAE overflows fp16)."""
    sd = {k: v.float() for k, v in sd.items()}
    tap = (lambda name, t: taps.__setitem__(name, float(t.abs().max()))) if taps is not None else (lambda name, t: None)
    x = F.conv2d(z.float(), sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = F.conv2d(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    x = _resnet(sd, "decoder.mid_block.resnets.0", x, groups, eps)
    x = _attention(sd, "decoder.mid_block.attentions.0", x, groups, eps)
    x = _resnet(sd, "decoder.mid_block.resnets.1", x, groups, eps)
    tap("mid_block", x)
    for i in range(n_up):
        for j in range(layers_per_block + 1):
            x = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, groups, eps)
        if f"decoder.up_blocks.{i}.upsamplers.0.conv.weight" in sd:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                         sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
        tap(f"up_blocks.{i}", x)
    x = F.silu(F.group_norm(x, groups, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], eps))
    return F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
