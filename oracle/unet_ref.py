"""Oracle (test infrastructure only): fp32 CPU restatement of `UNetMangaModel.forward`.

Follows reference src/models/unet.py:185-338 step by step.  The diffusers blocks it delegates to are
NOT importable here (diffusers absent) and are restated from their published semantics [3P]:
  ResnetBlock2D        GN(32, eps 1e-5) -> SiLU -> conv3x3 -> + time_emb_proj(SiLU(emb)) -> GN -> SiLU -> conv3x3
                       -> + (x or conv1x1(x))            (output_scale_factor 1)
  Transformer2DModel   GN(32, eps 1e-6) -> linear proj_in -> N x BasicTransformerBlock -> linear proj_out -> + residual
  BasicTransformerBlock LN -> attn1 -> +x ; LN -> attn2 -> +x ; LN -> GEGLU FF -> +x   (LN eps 1e-5)
  GEGLU                h, gate = proj(x).chunk(2); h * gelu(gate)  (erf GELU)
  Downsample2D         conv3x3 stride 2 pad 1 ; Upsample2D: nearest x2 then conv3x3 pad 1
  Timesteps            sinusoid, flip_sin_to_cos=True, freq_shift=0 ; TimestepEmbedding: linear, SiLU, linear
  text_time            add_emb = MLP(cat[text_embeds, sinusoid256(time_ids).flatten])
Parity of these against diffusers itself is UNPINNED (see oracle/__init__.py).  The block ORDER, channel widths and
skip pairing come from oracle/unet_topology_ref.py (a channel-flow simulation of the diffusers forward from the plain
config dict), NOT from the product's `unet_config.build_topology`; tests/test_oracle_unet.py checks that the two agree
and pins both to public SDXL-base anchors (1680 tensors, 2 567 463 684 parameters, known up-block conv shapes).

`q` is an optional rounding hook (identity = pure fp32; `lambda t: t.half().float()` emulates the
reference's fp16 storage between ops, which is what the fp16 HIP path is compared against).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

from oracle.attention_ref import masked_ip_cross_attention, self_attention
from oracle.unet_topology_ref import config_dict, unet_program

Tensor = torch.Tensor
_id = lambda t: t


def timestep_sinusoid(t: Tensor, dim: int, flip_sin_to_cos: bool = True, freq_shift: float = 0.0,
                      max_period: int = 10000) -> Tensor:
    """diffusers get_timestep_embedding [3P]."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def encode_dialog_bbox(sample: Tensor, dialog_bbox: Tensor, embedding: Tensor) -> Tensor:
    """reference src/models/unet.py:88-114 — int() truncation, clamp, ASSIGN (later boxes overwrite), then add."""
    batch, channel, height, width = sample.shape
    masked = torch.zeros_like(sample)
    for i in range(batch):
        for j in range(dialog_bbox.shape[1]):
            x1 = int(dialog_bbox[i, j, 0] * width)
            y1 = int(dialog_bbox[i, j, 1] * height)
            x2 = int(dialog_bbox[i, j, 2] * width)
            y2 = int(dialog_bbox[i, j, 3] * height)
            x1, x2 = max(0, x1), min(width, x2)
            y1, y2 = max(0, y1), min(height, y2)
            masked[i, :, y1:y2, x1:x2] = embedding.view(channel, 1, 1).to(sample.dtype)
    return sample + masked


class UNetOracle:
    def __init__(self, cfg, sd: Dict[str, Tensor], q: Callable = _id):
        self.cfg = cfg          # scalar hyper-parameters only (eps, group count, token counts); no topology table
        self.sd = {k: v.detach().float().cpu() for k, v in sd.items()}
        self.q = q
        self.program = unet_program(config_dict(cfg))
        self.ip_scale = 1.0

    # ---- leaf ops
    def _gn(self, x, name, eps, silu):
        y = F.group_norm(x, self.cfg.norm_num_groups, self.sd[name + ".weight"], self.sd[name + ".bias"], eps)
        if silu:
            y = F.silu(y)
        return self.q(y)

    def _conv(self, x, name, stride=1):
        return self.q(F.conv2d(x, self.sd[name + ".weight"], self.sd[name + ".bias"], stride=stride, padding=1))

    def _lin(self, x, name, bias=True):
        y = x @ self.sd[name + ".weight"].t()
        if bias:
            y = y + self.sd[name + ".bias"]
        return self.q(y)

    def _ln(self, x, name):
        c = x.shape[-1]
        return self.q(F.layer_norm(x, (c,), self.sd[name + ".weight"], self.sd[name + ".bias"], 1e-5))

    # ---- blocks
    def resnet(self, x, emb_act, prefix, cin, cout):
        q, sd = self.q, self.sd
        assert x.shape[1] == cin, (prefix, x.shape, cin)
        h = self._gn(x, prefix + ".norm1", self.cfg.norm_eps, True)
        h = self._conv(h, prefix + ".conv1")
        t = self._lin(emb_act, prefix + ".time_emb_proj")
        h = q(h + t[:, :, None, None])
        h = self._gn(h, prefix + ".norm2", self.cfg.norm_eps, True)
        h = self._conv(h, prefix + ".conv2")
        if cin != cout:             # diffusers: use_in_shortcut = in_channels != out_channels -> 1x1 conv_shortcut
            w = sd[prefix + ".conv_shortcut.weight"]
            x = q(F.conv2d(x, w, sd[prefix + ".conv_shortcut.bias"]))
        return q(x + h)

    def transformer(self, x, enc, prefix, depth, heads, bbox, aspect_ratio):
        q, sd, cfg = self.q, self.sd, self.cfg
        b, c, hh, ww = x.shape
        res = x
        h = self._gn(x, prefix + ".norm", 1e-6, False)
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        h = self._lin(h, prefix + ".proj_in")
        for k in range(depth):
            t = f"{prefix}.transformer_blocks.{k}"
            n = self._ln(h, t + ".norm1")
            o = self_attention(n, sd[t + ".attn1.to_q.weight"], sd[t + ".attn1.to_k.weight"],
                               sd[t + ".attn1.to_v.weight"], sd[t + ".attn1.to_out.0.weight"],
                               sd[t + ".attn1.to_out.0.bias"], heads, q)
            h = q(o + h)
            n = self._ln(h, t + ".norm2")
            o = masked_ip_cross_attention(
                n, enc, bbox, aspect_ratio,
                sd[t + ".attn2.to_q.weight"], sd[t + ".attn2.to_k.weight"], sd[t + ".attn2.to_v.weight"],
                sd[t + ".attn2.processor.to_k_ip.weight"], sd[t + ".attn2.processor.to_v_ip.weight"],
                sd[t + ".attn2.to_out.0.weight"], sd[t + ".attn2.to_out.0.bias"], heads, self.ip_scale,
                cfg.max_num_ips * cfg.num_vision_tokens, cfg.num_vision_tokens, q)
            h = q(o + h)
            n = self._ln(h, t + ".norm3")
            p = self._lin(n, t + ".ff.net.0.proj")
            hid, gate = p.chunk(2, dim=-1)
            g = q(hid * q(F.gelu(gate)))
            o = self._lin(g, t + ".ff.net.2")
            h = q(o + h)
        h = self._lin(h, prefix + ".proj_out")
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
        return q(h + res)

    # ---- forward
    def embeddings(self, timestep, text_embeds, time_ids, batch):
        """reference src/models/unet.py:190-199 (get_time_embed / time_embedding / get_aug_embed [3P])."""
        cfg, q = self.cfg, self.q
        t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(batch)
        t_emb = q(timestep_sinusoid(t, cfg.block_out_channels[0], cfg.flip_sin_to_cos, cfg.freq_shift))
        emb = self._lin(q(F.silu(self._lin(t_emb, "time_embedding.linear_1"))), "time_embedding.linear_2")
        tid = q(timestep_sinusoid(time_ids.float().flatten(), cfg.addition_time_embed_dim, cfg.flip_sin_to_cos,
                                  cfg.freq_shift)).reshape(batch, -1)
        add = torch.cat([text_embeds.float(), tid], dim=-1)
        aug = self._lin(q(F.silu(self._lin(add, "add_embedding.linear_1"))), "add_embedding.linear_2")
        return q(emb + aug)

    def forward(self, sample: Tensor, timestep, encoder_hidden_states: Tensor, text_embeds: Tensor,
                time_ids: Tensor, bbox: Tensor, aspect_ratio: float, dialog_bbox: Optional[Tensor] = None) -> Tensor:
        q, cfg = self.q, self.cfg
        sample = q(sample.float())
        enc = q(encoder_hidden_states.float())
        emb = self.embeddings(timestep, text_embeds, time_ids, sample.shape[0])
        emb_act = q(F.silu(emb))
        x = self._conv(sample, "conv_in")
        if dialog_bbox is not None:
            # the box tensor keeps its own dtype: the reference holds it in the UNet's dtype (fp16, prepare_dialog_bbox) and
            # `int(coord * width)` is evaluated in that dtype - 0.65 * 20 is 13 in fp16 and 12.998 in fp32
            x = q(encode_dialog_bbox(x, dialog_bbox, self.sd["dialog_bbox_embedding"]))
        skips = []
        up_factor = 2 ** sum(1 for st in self.program if st[0] == "upsample")
        forward_upsample_size = any(d % up_factor != 0 for d in sample.shape[-2:])
        for st in self.program:      # reference src/models/unet.py:244-332, flattened by oracle/unet_topology_ref.py
            op = st[0]
            if op == "push":
                skips.append(x)
            elif op == "pop_cat":
                x = torch.cat([x, skips.pop()], dim=1)
            elif op == "resnet":
                x = self.resnet(x, emb_act, st[1], st[2], st[3])
            elif op == "attn":
                assert x.shape[1] == st[2], (st, x.shape)
                x = self.transformer(x, enc, st[1], st[3], st[4], bbox, aspect_ratio)
            elif op == "downsample":
                x = self._conv(x, st[1], stride=2)
            elif op == "upsample":
                # diffusers: when a latent side is not a multiple of 2^(number of upsamplers) the forward passes
                # `upsample_size = down_block_res_samples[-1].shape[2:]` and Upsample2D resizes to it [3P]
                if forward_upsample_size:
                    x = F.interpolate(x, size=tuple(skips[-1].shape[-2:]), mode="nearest")
                else:
                    x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                x = self._conv(x, st[1])
            else:
                raise ValueError(st)
        assert not skips
        x = self._gn(x, "conv_norm_out", cfg.norm_eps, True)
        return self._conv(x, "conv_out")
