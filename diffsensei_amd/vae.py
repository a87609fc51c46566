"""VAE decode on the HIP kernels: host mirror of `AutoencoderKL.decode` as the reference pipeline uses it
(reference src/pipelines/pipeline_diffsensei.py:339-367 -> diffusers AutoencoderKL [3P]).

    engine = VaeDecoderEngine.from_state_dict(sd)          # diffusers key names (vae.state_dict())
    image = engine.decode(latents / scaling_factor)[0]     # drop-in for vae.decode(z, return_dict=False)
    image = engine.decode(latents, scaling_factor=sf)[0]   # same, with the division folded into the first kernel

Precision (`precision=`; default from the config's `force_upcast`, env DIFFSENSEI_VAE_PRECISION overrides):
  "fp16-scaled"  (force_upcast = true, i.e. wherever the reference upcasts the VAE to fp32 because "it overflows in
                 float16", :339-344)  fp16 operands, fp32 accumulation / statistics / softmax, and every stored conv output and
                 the residual stream multiplied by S = 2^-6: range 4.2e6 instead of 65504 (the real SDXL decoder reaches
                 1e4-1e5 in its up blocks), fp16's 11-bit mantissa instead of bf16's 8.  Nothing is rescaled at run time: conv
                 INPUTS are GroupNorm+SiLU outputs (O(1)), emitted pre-multiplied by S (`out_scale` of the norm kernel), so the
                 accumulators already carry the factor; biases (and conv_in / to_out weights, whose inputs are unscaled) are
                 multiplied by S once at load time (exact: a power of two); GroupNorm of a scaled tensor is exact with
                 eps * S^2.  Measured against the fp32 oracle: image rel-L2 ~5e-4, uint8 image within 1 LSB everywhere
                 (tests/test_gpu_vae.py), at the bf16 path's speed.
  "bf16"         bf16 storage (fp32's range, 8-bit mantissa): image rel-L2 ~4e-3 - 4 % of the uint8 bytes differ from the fp32
                 decode by more than 1 LSB (profiles/r03_*); kept for A/B.
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
import os
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import ops

Tensor = torch.Tensor
BF = torch.bfloat16
STORE_SCALE = 2.0 ** -6     # "fp16-scaled": stored activations = true value * 2^-6 (see the module docstring)


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


def fold_latents_affine(w: Tensor, b: Tensor, mean: Sequence[float], std: Sequence[float]) -> Tuple[Tensor, Tensor]:
    """reference pipeline_diffsensei.py:348-357: with `latents_mean` / `latents_std` in the VAE config the pipeline decodes
    `latents * std / scaling_factor + mean` (per latent channel) instead of `latents / scaling_factor`.  The first decoder op is
    post_quant_conv, a 1x1 conv on exactly that tensor, so the affine map folds into its weights once at load time:
    W'[o,c] = W[o,c] * std[c],  b'[o] = b[o] + sum_c W[o,c] * mean[c]  - and `decode(z, scaling_factor=sf)` is unchanged."""
    w2 = w.reshape(w.shape[0], -1).float()
    m = torch.tensor(list(mean), dtype=torch.float32, device=w.device)
    sd = torch.tensor(list(std), dtype=torch.float32, device=w.device)
    return (w2 * sd[None, :]).contiguous(), (b.float() + w2 @ m).contiguous()


_OLD_ATTN_NAMES = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}  # pre-0.18 checkpoints


@dataclass
class DecoderOutput:
    sample: Tensor


class VaeDecoderEngine:
    """Weights re-laid-out for the kernels + the launch sequence of one decode."""

    dtype = torch.bfloat16  # so `pipe.vae.dtype == torch.float16 and force_upcast` (reference :340) is False

    def __init__(self, cfg: VaeConfig, state_dict: Dict[str, Tensor], device="cuda", precision: Optional[str] = None):
        self.config = cfg
        self.device = torch.device(device)
        # force_upcast (SDXL: true) is what makes the reference decode in fp32; without it the reference decodes in the VAE's
        # own fp16 - the scaled-fp16 path is at least as precise as either request, so it is the default for both
        precision = precision or os.environ.get("DIFFSENSEI_VAE_PRECISION") or "fp16-scaled"
        if precision not in ("fp16-scaled", "bf16"):
            raise ValueError(f"VAE precision {precision!r}: 'fp16-scaled' or 'bf16'")
        self.precision = precision
        self.dt = BF if precision == "bf16" else torch.float16
        self.S = 1.0 if precision == "bf16" else STORE_SCALE
        C = cfg.block_out_channels
        if C[-1] != 512:
            raise ValueError(f"mid-block width must be 512 (the dim-512 attention kernel), got {C[-1]}")
        if any(c % 128 for c in C):
            raise ValueError(f"block_out_channels must be multiples of 128, got {C}")
        if (cfg.latents_mean is None) != (cfg.latents_std is None):
            raise ValueError("latents_mean and latents_std come as a pair (reference pipeline_diffsensei.py:348-357)")
        self.w: Dict[str, Tensor] = {}
        self._pack({k: v.detach() for k, v in state_dict.items()})

    # ---- construction
    @classmethod
    def from_state_dict(cls, sd: Dict[str, Tensor], cfg: Optional[VaeConfig] = None, device="cuda",
                        precision: Optional[str] = None):
        return cls(cfg or VaeConfig(), sd, device, precision)

    @classmethod
    def from_diffusers(cls, vae, device="cuda", precision: Optional[str] = None):
        """`vae`: a diffusers AutoencoderKL (only `.config` and `.state_dict()` are touched)."""
        c = vae.config
        cfg = VaeConfig(tuple(c.block_out_channels), c.layers_per_block, c.latent_channels, c.out_channels,
                        c.norm_num_groups, float(c.scaling_factor), bool(getattr(c, "force_upcast", True)),
                        getattr(c, "latents_mean", None), getattr(c, "latents_std", None))
        return cls(cfg, vae.state_dict(), device, precision)

    @classmethod
    def init_random(cls, cfg: Optional[VaeConfig] = None, seed: int = 0, device="cuda", precision: Optional[str] = None):
        cfg = cfg or VaeConfig()
        return cls(cfg, random_state_dict(cfg, seed), device, precision)

    def _pack(self, sd: Dict[str, Tensor]) -> None:
        dev = self.device

        def get(name):
            if name in sd:
                return sd[name].float()
            for new, old in _OLD_ATTN_NAMES.items():  # old attention naming
                if f".{new}." in name and name.replace(f".{new}.", f".{old}.") in sd:
                    return sd[name.replace(f".{new}.", f".{old}.")].float()
            raise KeyError(f"VAE state dict has no '{name}'")

        DT, S = self.dt, self.S
        a = "decoder.mid_block.attentions.0"
        for name, shape in vae_param_shapes(self.config).items():
            t = get(name)
            # scaled-fp16 mode: biases of everything that writes a stored (scaled) tensor carry S; so do the weights of the
            # two ops whose input is NOT scaled (conv_in reads the fp32 latents, to_out the attention output); conv_out and
            # to_q / to_k produce unscaled values from unscaled inputs
            unscaled_out = name.startswith("decoder.conv_out") or f"{a}.to_q" in name or f"{a}.to_k" in name or ".norm" in name \
                or "group_norm" in name or "conv_norm_out" in name
            if name.endswith("bias") and not unscaled_out and not name.startswith("post_quant_conv"):
                t = t * S
            if name in ("decoder.conv_in.weight", f"{a}.to_out.0.weight"):
                t = t * S
            if name.startswith("post_quant_conv"):
                self.w[name] = t.reshape(shape[0], -1).contiguous().to(dev) if name.endswith("weight") else t.contiguous().to(dev)
            elif len(shape) == 4 and shape[2] == 3:      # 3x3 conv: [Cout,Cin,3,3] -> [Cout,3,3,Cin]
                self.w[name] = t.permute(0, 2, 3, 1).contiguous().to(dev, DT)
            elif len(shape) == 4:                        # 1x1 shortcut -> linear [Cout,Cin]
                self.w[name] = t.reshape(shape[0], shape[1]).contiguous().to(dev, DT)
            elif len(shape) == 2 and t.dim() == 4:       # old checkpoints store attention linears as 1x1 convs
                self.w[name] = t.reshape(shape).contiguous().to(dev, DT)
            else:
                self.w[name] = t.contiguous().to(dev, DT)
        if self.config.latents_mean is not None:
            w2, b2 = fold_latents_affine(get("post_quant_conv.weight"), get("post_quant_conv.bias"),
                                         self.config.latents_mean, self.config.latents_std)
            self.w["post_quant_conv.weight+ms"], self.w["post_quant_conv.bias+ms"] = w2.to(dev), b2.to(dev)
        # V is produced transposed ([B, C, N], keys contiguous) by a GEMM whose bias runs along the other axis, so its
        # bias is carried through the attention instead: softmax rows sum to 1, hence attn(V + 1 b^T) = attn(V) + b and
        # to_out(o + b_v) = W_o o + (W_o b_v + b_o).
        wo, bo, bv = get(f"{a}.to_out.0.weight").reshape(512, 512), get(f"{a}.to_out.0.bias"), get(f"{a}.to_v.bias")
        self.w[f"{a}.to_out.0.bias+v"] = ((bo + wo @ bv) * S).contiguous().to(dev, DT)

    # ---- building blocks ([B,H,W,C] NHWC in self.dt; "stored" tensors carry the factor self.S, see the module docstring)
    def _gn(self, x: Tensor, name: str, silu: bool, out_scale: float = 1.0) -> Tensor:
        """x: a stored tensor (true * S).  -> act(GroupNorm(true)) * out_scale."""
        B, H, W, C = x.shape
        g, bt = self.w[f"{name}.weight"], self.w[f"{name}.bias"]
        if self.dt == BF:
            y = ops.groupnorm_bf16(x.view(B, H * W, C), g, bt, self.config.norm_num_groups, self.config.eps, silu)
        else:
            y = ops.groupnorm_scaled(x.view(B, H * W, C), g, bt, self.config.norm_num_groups,
                                     self.config.eps * self.S * self.S, silu, out_scale)
        return y.view(B, H, W, C)

    def _conv(self, x: Tensor, name: str, upsample: bool = False, residual: Optional[Tensor] = None) -> Tensor:
        w, b = self.w[f"{name}.weight"], self.w[f"{name}.bias"]
        if self.dt == BF:
            return ops.conv3x3_bf16(x, w, b, upsample=upsample, residual=residual)
        return ops.conv3x3(x, w, b, upsample=upsample, residual=residual)

    def _linear(self, x: Tensor, w: Tensor, bias: Optional[Tensor], residual: Optional[Tensor] = None) -> Tensor:
        return ops.gemm_bf16(x, w, bias, residual=residual) if self.dt == BF else ops.gemm(x, w, bias, residual=residual)

    def _resnet(self, x: Tensor, p: str) -> Tensor:
        B, H, W, Cin = x.shape
        S = self.S
        h = self._gn(x, f"{p}.norm1", True, S)           # conv inputs carry S, so the accumulators do
        h = self._conv(h, f"{p}.conv1")
        h = self._gn(h, f"{p}.norm2", True, S)
        if f"{p}.conv_shortcut.weight" in self.w:         # 1x1 on the stored stream itself (already * S)
            ws = self.w[f"{p}.conv_shortcut.weight"]
            x = self._linear(x.view(B * H * W, Cin), ws, self.w[f"{p}.conv_shortcut.bias"]).view(B, H, W, ws.shape[0])
        return self._conv(h, f"{p}.conv2", residual=x)

    def _attention(self, x: Tensor, p: str) -> Tensor:
        B, H, W, C = x.shape
        N = H * W
        h = self._gn(x, f"{p}.group_norm", False).view(B, N, C)          # unscaled: q, k, v are true values
        # The MFMA GEMMs take row counts that are multiples of 16.  Other latent sizes (the reference accepts every image
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
        q = self._linear(h2, self.w[f"{p}.to_q.weight"], self.w[f"{p}.to_q.bias"]).view(B, Np, C)
        k = self._linear(h2, self.w[f"{p}.to_k.weight"], self.w[f"{p}.to_k.bias"]).view(B, Np, C)
        scale = 1.0 / math.sqrt(C)
        if self.dt == BF:
            vt = ops.gemm_batched_nt_bf16(self.w[f"{p}.to_v.weight"], h)                     # [B, C, Np], bias deferred
            o = ops.wide_attention_bf16(q, k, vt, scale, n_valid=N)
        else:
            vt = ops.gemm_batched_nt(self.w[f"{p}.to_v.weight"], h)
            o = ops.wide_attention_f16(q, k, vt, scale, n_valid=N)
        # to_out's weights and folded bias carry S: its output joins the stored stream
        out = self._linear(o.view(B * Np, C), self.w[f"{p}.to_out.0.weight"], self.w[f"{p}.to_out.0.bias+v"],
                           residual=res.reshape(B * Np, C)).view(B, Np, C)
        if Np != N:
            out = out[:, :N].contiguous()
        return out.view(B, H, W, C)

    # ---- the decode (mirrors AutoencoderKL.decode / Decoder.forward)
    def decode(self, z: Tensor, return_dict: bool = True, generator=None, scaling_factor: float = 1.0,
               denormalize: bool = False, latents_affine: bool = False):
        """`vae.decode(z)`; `scaling_factor` folds the pipeline's `latents / scaling_factor` (:359) into the first kernel,
        `denormalize` the image processor's `(x / 2 + 0.5).clamp(0, 1)` (:367) into the last one; `latents_affine` (the
        pipeline sets it when the config has latents_mean / latents_std) decodes `z * std / scaling_factor + mean` (:348-357)
        through the folded post_quant_conv weights."""
        if z.dim() != 4 or z.shape[1] != self.config.latent_channels:
            raise ValueError(f"expected latents [B,{self.config.latent_channels},h,w], got {tuple(z.shape)}")
        B, _, h, w = z.shape
        chunk = self.decode_chunk(h, w, B)
        if B > chunk:
            parts = [self.decode(z[i:i + chunk], False, generator, scaling_factor, denormalize, latents_affine)[0]
                     for i in range(0, B, chunk)]
            img = torch.cat(parts)
            return DecoderOutput(img) if return_dict else (img,)
        lat = z.to(self.device, torch.float32).contiguous()
        ms = "+ms" if latents_affine and self.config.latents_mean is not None else ""
        x = ops.vae_conv_in(lat, self.w["post_quant_conv.weight" + ms], self.w["post_quant_conv.bias" + ms],
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
                x = self._conv(x, up, upsample=True)            # on the stored stream (already * S)
        x = self._gn(x, "decoder.conv_norm_out", True)
        img = ops.vae_conv_out(x, self.w["decoder.conv_out.weight"], self.w["decoder.conv_out.bias"], denormalize)
        return DecoderOutput(img) if return_dict else (img,)

    def decode_chunk(self, h: int, w: int, B: int) -> int:
        """Images that go through one launch sequence: the conv kernel addresses its input with 32-bit element offsets, so the
        widest full-resolution activation ([chunk, 8h, 8w, C1]) bounds it (7 at 1024^2 -> chunks of 4)."""
        per_image = (8 * h) * (8 * w) * max(self.config.block_out_channels[1], self.config.block_out_channels[0])
        chunk = max(1, min(B, (2 ** 31 - 1) // per_image))
        return 1 << (chunk.bit_length() - 1)

    # ---- plumbing the reference pipeline touches
    def to(self, *a, **k):
        return self

    def weights_changed(self) -> None:
        """Called after `tensors()` were rewritten in place / re-homed (the RCCL start-up broadcast).  The kernels read the
        listed tensors themselves at every call - nothing derived from them is cached here - so there is nothing to drop; the
        hook exists so that `distributed.broadcast_pipeline` can require it of every engine."""

    def tensors(self):
        """Every weight tensor (for the one-off RCCL broadcast)."""
        return list(self.w.values())

    def decode_flops(self, h: int, w: int) -> float:
        """Algorithmic flops of one image decode (convs + linears + attention)."""
        C = self.config.block_out_channels
        fl = 0.0
        for name, t in self.w.items():
            if t.dim() == 4 and t.dtype == self.dt:  # 3x3 convs
                lvl = _level_of(name, len(C))
                hw = (h << lvl) * (w << lvl) * (4 if "upsamplers" in name else 1)
                fl += 2.0 * hw * t.shape[0] * t.shape[1] * t.shape[2] * t.shape[3]
            elif t.dim() == 2 and t.dtype == self.dt and "bias" not in name:
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
