"""VAE decode on the HIP kernels: host mirror of `AutoencoderKL.decode` as the reference pipeline uses it
(reference src/pipelines/pipeline_diffsensei.py:339-367 -> diffusers AutoencoderKL [3P]).

    engine = VaeDecoderEngine.from_state_dict(sd)          # diffusers key names (vae.state_dict())
    image = engine.decode(latents / scaling_factor)[0]     # drop-in for vae.decode(z, return_dict=False)
    image = engine.decode(latents, scaling_factor=sf)[0]   # same, with the division folded into the first kernel

Precision: bf16 storage, fp32 accumulation, fp32 GroupNorm statistics, fp32 softmax, fp32 image out.  The reference
upcasts the VAE to fp32 for this step because fp16 overflows inside the decoder (:340-344); bf16 has fp32's range.
Layout: NHWC between kernels.  Kernels: `conv_halo_kernel<bf16>` (every 3x3 conv, upsample fused), `gemm_pp_kernel<bf16>`
(1x1 shortcuts, attention projections), `gn_*<bf16>`, `wide_attn_kernel` (the 1-head, dim-512 mid-block attention),
`vae_conv_in_kernel` (post_quant_conv + conv_in), `vae_conv_out_kernel`.  No torch arithmetic on the data path: the
only host-side math is the one-off weight re-layout (and folding the V bias through to_out, see `_pack`).

Shape rules of the kernels: any latent height x width (conv patches at ragged edges are masked; the mid-block attention
pads its token matrices to a multiple of 16 rows and masks the padding keys), channel widths multiples of 128, mid-block
width exactly 512 (SDXL / SD VAE).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import ops

Tensor = torch.Tensor
BF = torch.bfloat16


@dataclass
class VaeConfig:
    """The decoder-relevant fields of diffusers' AutoencoderKL config (SDXL values)."""
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    latent_channels: int = 4
    out_channels: int = 3
    norm_num_groups: int = 32
    scaling_factor: float = 0.13025
    force_upcast: bool = True          # what makes the reference decode in fp32; here: bf16 storage, fp32 math
    latents_mean: Optional[Sequence[float]] = None
    latents_std: Optional[Sequence[float]] = None
    eps: float = 1e-6


def vae_param_shapes(cfg: VaeConfig = VaeConfig()) -> Dict[str, tuple]:
    """Decoder-side parameter names and shapes of diffusers' AutoencoderKL."""
    C = list(cfg.block_out_channels)
    lc = cfg.latent_channels
    sh: Dict[str, tuple] = {"post_quant_conv.weight": (lc, lc, 1, 1), "post_quant_conv.bias": (lc,)}

    def resnet(prefix, cin, cout):
        sh[f"{prefix}.norm1.weight"] = (cin,)
        sh[f"{prefix}.norm1.bias"] = (cin,)
        sh[f"{prefix}.conv1.weight"] = (cout, cin, 3, 3)
        sh[f"{prefix}.conv1.bias"] = (cout,)
        sh[f"{prefix}.norm2.weight"] = (cout,)
        sh[f"{prefix}.norm2.bias"] = (cout,)
        sh[f"{prefix}.conv2.weight"] = (cout, cout, 3, 3)
        sh[f"{prefix}.conv2.bias"] = (cout,)
        if cin != cout:
            sh[f"{prefix}.conv_shortcut.weight"] = (cout, cin, 1, 1)
            sh[f"{prefix}.conv_shortcut.bias"] = (cout,)

    top = C[-1]
    sh["decoder.conv_in.weight"] = (top, lc, 3, 3)
    sh["decoder.conv_in.bias"] = (top,)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    sh[f"{a}.group_norm.weight"] = (top,)
    sh[f"{a}.group_norm.bias"] = (top,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        sh[f"{a}.{n}.weight"] = (top, top)
        sh[f"{a}.{n}.bias"] = (top,)
    resnet("decoder.mid_block.resnets.1", top, top)
    rev = C[::-1]
    prev = rev[0]
    for i, out in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else out, out)
        if i != len(rev) - 1:
            sh[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (out, out, 3, 3)
            sh[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (out,)
        prev = out
    sh["decoder.conv_norm_out.weight"] = (C[0],)
    sh["decoder.conv_norm_out.bias"] = (C[0],)
    sh["decoder.conv_out.weight"] = (cfg.out_channels, C[0], 3, 3)
    sh["decoder.conv_out.bias"] = (cfg.out_channels,)
    return sh


def random_state_dict(cfg: VaeConfig = VaeConfig(), seed: int = 0) -> Dict[str, Tensor]:
    """Seeded weights at the true shapes (no checkpoint is reachable offline): fan-in scaled, norm gains near 1."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, s in vae_param_shapes(cfg).items():
        if k.endswith("weight") and len(s) == 1:
            sd[k] = 1.0 + 0.1 * torch.randn(s, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.05 * torch.randn(s, generator=g)
        else:
            sd[k] = torch.randn(s, generator=g) / math.sqrt(math.prod(s[1:]))
    return sd


_OLD_ATTN_NAMES = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}  # pre-0.18 checkpoints


@dataclass
class DecoderOutput:
    sample: Tensor


class VaeDecoderEngine:
    """Weights re-laid-out for the kernels + the launch sequence of one decode."""

    dtype = torch.bfloat16  # so `pipe.vae.dtype == torch.float16 and force_upcast` (reference :340) is False

    def __init__(self, cfg: VaeConfig, state_dict: Dict[str, Tensor], device="cuda"):
        self.config = cfg
        self.device = torch.device(device)
        C = cfg.block_out_channels
        if C[-1] != 512:
            raise ValueError(f"mid-block width must be 512 (the dim-512 attention kernel), got {C[-1]}")
        if any(c % 128 for c in C):
            raise ValueError(f"block_out_channels must be multiples of 128, got {C}")
        if cfg.latents_mean is not None or cfg.latents_std is not None:
            raise NotImplementedError("latents_mean / latents_std (not part of the SDXL VAE config)")
        self.w: Dict[str, Tensor] = {}
        self._pack({k: v.detach() for k, v in state_dict.items()})

    # ---- construction
    @classmethod
    def from_state_dict(cls, sd: Dict[str, Tensor], cfg: Optional[VaeConfig] = None, device="cuda"):
        return cls(cfg or VaeConfig(), sd, device)

    @classmethod
    def from_diffusers(cls, vae, device="cuda"):
        """`vae`: a diffusers AutoencoderKL (only `.config` and `.state_dict()` are touched)."""
        c = vae.config
        cfg = VaeConfig(tuple(c.block_out_channels), c.layers_per_block, c.latent_channels, c.out_channels,
                        c.norm_num_groups, float(c.scaling_factor), bool(getattr(c, "force_upcast", True)),
                        getattr(c, "latents_mean", None), getattr(c, "latents_std", None))
        return cls(cfg, vae.state_dict(), device)

    @classmethod
    def init_random(cls, cfg: Optional[VaeConfig] = None, seed: int = 0, device="cuda"):
        cfg = cfg or VaeConfig()
        return cls(cfg, random_state_dict(cfg, seed), device)

    def _pack(self, sd: Dict[str, Tensor]) -> None:
        dev = self.device

        def get(name):
            if name in sd:
                return sd[name].float()
            for new, old in _OLD_ATTN_NAMES.items():  # old attention naming
                if f".{new}." in name and name.replace(f".{new}.", f".{old}.") in sd:
                    return sd[name.replace(f".{new}.", f".{old}.")].float()
            raise KeyError(f"VAE state dict has no '{name}'")

        for name, shape in vae_param_shapes(self.config).items():
            t = get(name)
            if name.startswith("post_quant_conv"):
                self.w[name] = t.reshape(shape[0], -1).contiguous().to(dev) if name.endswith("weight") else t.contiguous().to(dev)
            elif len(shape) == 4 and shape[2] == 3:      # 3x3 conv: [Cout,Cin,3,3] -> [Cout,3,3,Cin]
                self.w[name] = t.permute(0, 2, 3, 1).contiguous().to(dev, BF)
            elif len(shape) == 4:                        # 1x1 shortcut -> linear [Cout,Cin]
                self.w[name] = t.reshape(shape[0], shape[1]).contiguous().to(dev, BF)
            elif len(shape) == 2 and t.dim() == 4:       # old checkpoints store attention linears as 1x1 convs
                self.w[name] = t.reshape(shape).contiguous().to(dev, BF)
            else:
                self.w[name] = t.contiguous().to(dev, BF)
        # V is produced transposed ([B, C, N], keys contiguous) by a GEMM whose bias runs along the other axis, so its
        # bias is carried through the attention instead: softmax rows sum to 1, hence attn(V + 1 b^T) = attn(V) + b and
        # to_out(o + b_v) = W_o o + (W_o b_v + b_o).
        a = "decoder.mid_block.attentions.0"
        wo, bo, bv = get(f"{a}.to_out.0.weight").reshape(512, 512), get(f"{a}.to_out.0.bias"), get(f"{a}.to_v.bias")
        self.w[f"{a}.to_out.0.bias+v"] = (bo + wo @ bv).contiguous().to(dev, BF)

    # ---- building blocks ([B,H,W,C] bf16 NHWC)
    def _gn(self, x: Tensor, name: str, silu: bool) -> Tensor:
        B, H, W, C = x.shape
        y = ops.groupnorm_bf16(x.view(B, H * W, C), self.w[f"{name}.weight"], self.w[f"{name}.bias"],
                               self.config.norm_num_groups, self.config.eps, silu)
        return y.view(B, H, W, C)

    def _resnet(self, x: Tensor, p: str) -> Tensor:
        B, H, W, Cin = x.shape
        h = self._gn(x, f"{p}.norm1", True)
        h = ops.conv3x3_bf16(h, self.w[f"{p}.conv1.weight"], self.w[f"{p}.conv1.bias"])
        h = self._gn(h, f"{p}.norm2", True)
        if f"{p}.conv_shortcut.weight" in self.w:
            ws = self.w[f"{p}.conv_shortcut.weight"]
            x = ops.gemm_bf16(x.view(B * H * W, Cin), ws, self.w[f"{p}.conv_shortcut.bias"]).view(B, H, W, ws.shape[0])
        return ops.conv3x3_bf16(h, self.w[f"{p}.conv2.weight"], self.w[f"{p}.conv2.bias"], residual=x)

    def _attention(self, x: Tensor, p: str) -> Tensor:
        B, H, W, C = x.shape
        N = H * W
        h = self._gn(x, f"{p}.group_norm", False).view(B, N, C)
        # The bf16 GEMMs take row counts that are multiples of 16.  Other latent sizes (the reference accepts every image
        # side that is a multiple of 8): the token matrices get zero rows up to the next multiple of 16 - pure copies, the
        # statistics above were taken on the real tokens - and the attention kernel masks the padding keys.
        Np = (N + 15) // 16 * 16
        res = x.view(B, N, C)
        if Np != N:
            hp = torch.zeros((B, Np, C), dtype=h.dtype, device=h.device)
            hp[:, :N] = h
            rp = torch.zeros((B, Np, C), dtype=x.dtype, device=x.device)
            rp[:, :N] = res
            h, res = hp, rp
        h2 = h.view(B * Np, C)
        q = ops.gemm_bf16(h2, self.w[f"{p}.to_q.weight"], self.w[f"{p}.to_q.bias"]).view(B, Np, C)
        k = ops.gemm_bf16(h2, self.w[f"{p}.to_k.weight"], self.w[f"{p}.to_k.bias"]).view(B, Np, C)
        vt = ops.gemm_batched_nt_bf16(self.w[f"{p}.to_v.weight"], h)                         # [B, C, Np], bias deferred
        o = ops.wide_attention_bf16(q, k, vt, 1.0 / math.sqrt(C), n_valid=N)
        out = ops.gemm_bf16(o.view(B * Np, C), self.w[f"{p}.to_out.0.weight"], self.w[f"{p}.to_out.0.bias+v"],
                            residual=res.reshape(B * Np, C)).view(B, Np, C)
        if Np != N:
            out = out[:, :N].contiguous()
        return out.view(B, H, W, C)

    # ---- the decode (mirrors AutoencoderKL.decode / Decoder.forward)
    def decode(self, z: Tensor, return_dict: bool = True, generator=None, scaling_factor: float = 1.0,
               denormalize: bool = False):
        """`vae.decode(z)`; `scaling_factor` folds the pipeline's `latents / scaling_factor` (:359) into the first kernel,
        `denormalize` the image processor's `(x / 2 + 0.5).clamp(0, 1)` (:367) into the last one."""
        if z.dim() != 4 or z.shape[1] != self.config.latent_channels:
            raise ValueError(f"expected latents [B,{self.config.latent_channels},h,w], got {tuple(z.shape)}")
        B, _, h, w = z.shape
        # the conv kernel addresses its input with 32-bit element offsets: the widest full-resolution activation
        # ([chunk, 8h, 8w, C1]) bounds how many images go through one launch sequence (7 at 1024^2 -> chunks of 4)
        per_image = (8 * h) * (8 * w) * max(self.config.block_out_channels[1], self.config.block_out_channels[0])
        chunk = max(1, min(B, (2 ** 31 - 1) // per_image))
        chunk = 1 << (chunk.bit_length() - 1)
        if B > chunk:
            parts = [self.decode(z[i:i + chunk], False, generator, scaling_factor, denormalize)[0] for i in range(0, B, chunk)]
            img = torch.cat(parts)
            return DecoderOutput(img) if return_dict else (img,)
        lat = z.to(self.device, torch.float32).contiguous()
        x = ops.vae_conv_in(lat, self.w["post_quant_conv.weight"], self.w["post_quant_conv.bias"],
                            self.w["decoder.conv_in.weight"], self.w["decoder.conv_in.bias"], scaling_factor)
        x = self._resnet(x, "decoder.mid_block.resnets.0")
        x = self._attention(x, "decoder.mid_block.attentions.0")
        x = self._resnet(x, "decoder.mid_block.resnets.1")
        n_up = len(self.config.block_out_channels)
        for i in range(n_up):
            for j in range(self.config.layers_per_block + 1):
                x = self._resnet(x, f"decoder.up_blocks.{i}.resnets.{j}")
            up = f"decoder.up_blocks.{i}.upsamplers.0.conv"
            if f"{up}.weight" in self.w:
                x = ops.conv3x3_bf16(x, self.w[f"{up}.weight"], self.w[f"{up}.bias"], upsample=True)
        x = self._gn(x, "decoder.conv_norm_out", True)
        img = ops.vae_conv_out(x, self.w["decoder.conv_out.weight"], self.w["decoder.conv_out.bias"], denormalize)
        return DecoderOutput(img) if return_dict else (img,)

    # ---- plumbing the reference pipeline touches
    def to(self, *a, **k):
        return self

    def tensors(self):
        """Every weight tensor (for the one-off RCCL broadcast)."""
        return list(self.w.values())

    def decode_flops(self, h: int, w: int) -> float:
        """Algorithmic flops of one image decode (convs + linears + attention)."""
        C = self.config.block_out_channels
        fl = 0.0
        for name, t in self.w.items():
            if t.dim() == 4 and t.dtype == BF:  # 3x3 convs
                lvl = _level_of(name, len(C))
                hw = (h << lvl) * (w << lvl) * (4 if "upsamplers" in name else 1)
                fl += 2.0 * hw * t.shape[0] * t.shape[1] * t.shape[2] * t.shape[3]
            elif t.dim() == 2 and t.dtype == BF and "bias" not in name:
                lvl = _level_of(name, len(C))
                fl += 2.0 * (h << lvl) * (w << lvl) * t.shape[0] * t.shape[1]
        fl += 4.0 * (h * w) ** 2 * C[-1]
        return fl


def _level_of(name: str, n_up: int) -> int:
    """Resolution level (0 = latent size) a decoder parameter is applied at."""
    if ".up_blocks." in name:
        return min(int(name.split(".up_blocks.")[1].split(".")[0]), n_up - 1)
    if "conv_norm_out" in name or "conv_out" in name:
        return n_up - 1
    return 0
