"""Host mirror of reference src/models/unet.py `UNetMangaModel` — same public surface
(`from_config`, `set_manga_modules`, `load_state_dict` with diffusers key names, `config.*`,
`attn_processors`, `forward(...)` returning an object with `.sample`), executed by the gfx950 launch plan
(`engine.UNetEngine`).  There is no PyTorch execution path: without the HIP library `forward` raises.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, fields
from typing import Any, Dict, Optional, Tuple, Union

import torch

from . import ops
from .attention_processor import AttnProcessor2_0, MaskedIPAttnProcessor2_0
from .engine import PackedUNet, UNetEngine
from .unet_config import (UNetMangaConfig, attn_processor_names, param_shapes, random_state_dict,
                          sdxl_config)

Tensor = torch.Tensor


@dataclass
class UNet2DConditionOutput:
    sample: Tensor = None


def dialog_pixel_boxes(dialog_bbox: Tensor, height: int, width: int) -> Tensor:
    """Pixel boxes of `encode_dialog_bbox` (reference src/models/unet.py:101-108): `int(bbox * size)` evaluated in
    the tensor's own dtype (fp16 in the reference, `prepare_dialog_bbox` casts to unet.dtype), truncation toward
    zero, then clamped.  Control-path arithmetic on <= B*8 boxes, done on the host; returns int32 [B,nd,4]."""
    db = dialog_bbox.detach().to("cpu")
    size = torch.tensor([width, height, width, height], dtype=db.dtype)
    px = (db * size).to(torch.int32)          # product rounded in db.dtype, then truncated like int()
    x1 = px[..., 0].clamp(min=0)
    y1 = px[..., 1].clamp(min=0)
    x2 = px[..., 2].clamp(max=width)
    y2 = px[..., 3].clamp(max=height)
    return torch.stack([x1, y1, x2, y2], dim=-1).contiguous()


class UNetMangaModel:
    """SDXL UNet + DiffSensei manga modules on MI355X."""

    def __init__(self, config: Optional[UNetMangaConfig] = None, device: Union[str, torch.device] = "cuda",
                 dtype: torch.dtype = torch.float16):
        self.config = config or sdxl_config()
        self.device = torch.device(device)
        self.dtype = dtype
        if dtype != torch.float16:
            raise ValueError("the MI355X engine computes in fp16 (the reference's inference dtype)")
        self._sd: Dict[str, Tensor] = {}
        self._packed: Optional[PackedUNet] = None
        self._engines: Dict[Tuple, UNetEngine] = {}
        self._attn_processors: Dict[str, Any] = {n: AttnProcessor2_0() for n in attn_processor_names(self.config)}
        self._manga = False
        # "fp16": the reference's arithmetic - the only value (the e4m3 variant of rounds 2-5 was retired in round 6)
        self.attention_dtype = "fp16"

    # ---- construction (reference scripts/demo/gradio_wo_mllm.py:161-169)
    @classmethod
    def from_config(cls, config, subfolder: Optional[str] = None, torch_dtype: torch.dtype = torch.float16,
                    device: Union[str, torch.device] = "cuda", **kwargs) -> "UNetMangaModel":
        if isinstance(config, UNetMangaConfig):
            cfg = config
        else:
            if isinstance(config, (str, os.PathLike)):
                path = os.fspath(config)
                if subfolder:
                    path = os.path.join(path, subfolder)
                if os.path.isdir(path):
                    path = os.path.join(path, "config.json")
                with open(path) as fh:
                    config = json.load(fh)
            known = {f.name for f in fields(UNetMangaConfig)}
            picked = {}
            for k, v in dict(config).items():
                if k in known:
                    picked[k] = tuple(v) if isinstance(v, list) else v
            n = len(picked.get("block_out_channels", (320, 640, 1280)))
            for k in ("transformer_layers_per_block", "attention_head_dim"):
                if k in picked and isinstance(picked[k], int):
                    picked[k] = (picked[k],) * n
            cfg = UNetMangaConfig(**picked)
        return cls(cfg, device=device, dtype=torch_dtype or torch.float16)

    def set_manga_modules(self, max_num_ips=4, num_vision_tokens=16, max_num_dialogs=8):
        """reference src/models/unet.py:44-86: register config keys, install the processors (IP K/V initialised from
        the text K/V), create `dialog_bbox_embedding`."""
        cfg = self.config
        cfg.max_num_ips, cfg.max_num_dialogs, cfg.num_vision_tokens = max_num_ips, max_num_dialogs, num_vision_tokens
        procs = {}
        for name in attn_processor_names(cfg):
            if name.endswith("attn1.processor"):
                procs[name] = AttnProcessor2_0()
                continue
            if name.startswith("mid_block"):
                hidden = cfg.block_out_channels[-1]
            elif name.startswith("up_blocks"):
                hidden = list(reversed(cfg.block_out_channels))[int(name[len("up_blocks.")])]
            else:
                hidden = cfg.block_out_channels[int(name[len("down_blocks.")])]
            layer = name.split(".processor")[0]
            proc = MaskedIPAttnProcessor2_0(hidden_size=hidden, cross_attention_dim=cfg.cross_attention_dim,
                                            num_ip_tokens=max_num_ips * num_vision_tokens,
                                            num_dummy_tokens=num_vision_tokens, device=self.device)
            if layer + ".to_k.weight" in self._sd:
                self._sd[name + ".to_k_ip.weight"] = self._sd[layer + ".to_k.weight"].clone()
                self._sd[name + ".to_v_ip.weight"] = self._sd[layer + ".to_v.weight"].clone()
            procs[name] = proc
        self._attn_processors = procs
        g = torch.Generator(device="cpu")
        g.manual_seed(torch.initial_seed() % (2 ** 31))
        self._sd["dialog_bbox_embedding"] = torch.randn(cfg.block_out_channels[0], generator=g).to(self.device, self.dtype)
        self._manga = True
        self._invalidate()

    # ---- weights
    def _invalidate(self):
        self._packed = None
        self._engines.clear()

    def state_dict(self) -> Dict[str, Tensor]:
        return dict(self._sd)

    def tensors(self):
        """Frozen weights in state-dict layout (the multi-GPU weight broadcast list).  They are re-packed for the kernels
        lazily, so a broadcast into these tensors must be followed by `weights_changed()`."""
        return list(self._sd.values())

    def weights_changed(self):
        """Drop the packed copies / launch plans derived from the state dict (after an in-place update of `tensors()`)."""
        self._invalidate()

    def load_state_dict(self, sd: Dict[str, Tensor], strict: bool = True):
        shapes = param_shapes(self.config)
        missing = [k for k in shapes if k not in sd]
        unexpected = [k for k in sd if k not in shapes]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]} (+{max(0, len(missing) - 5)}), "
                               f"unexpected {unexpected[:5]} (+{max(0, len(unexpected) - 5)})")
        for k, shp in shapes.items():
            if k in sd:
                if tuple(sd[k].shape) != tuple(shp):
                    raise RuntimeError(f"load_state_dict: {k} has shape {tuple(sd[k].shape)}, expected {tuple(shp)}")
                self._sd[k] = sd[k].detach().to(self.device, self.dtype).contiguous()
        self._invalidate()
        return missing, unexpected

    def init_random(self, seed: int = 0) -> "UNetMangaModel":
        """Seeded synthetic weights (see unet_config.random_state_dict), generated directly on the device."""
        self._sd = random_state_dict(self.config, seed, self.device, self.dtype)
        self._manga = True
        self._invalidate()
        return self

    def to(self, device=None, dtype=None, **kwargs):
        if dtype is not None and dtype != torch.float16:
            raise ValueError("the MI355X engine computes in fp16")
        if device is not None and torch.device(device) != self.device:
            self.device = torch.device(device)
            self._sd = {k: v.to(self.device) for k, v in self._sd.items()}
            self._invalidate()
        return self

    def eval(self):
        return self

    @property
    def attn_processors(self) -> Dict[str, Any]:
        return self._attn_processors

    def set_attn_processor(self, procs: Dict[str, Any]):
        """diffusers protocol `unet.set_attn_processor({name: processor})` (reference src/models/unet.py:84).  The launch
        plan implements exactly the reference's two processors, so anything else is refused instead of being silently
        ignored: every key must be a known processor slot, attn1 slots take `AttnProcessor2_0`, attn2 slots take
        `MaskedIPAttnProcessor2_0` (whose `.scale` the plan reads) or, before `set_manga_modules`, `AttnProcessor2_0`."""
        known = attn_processor_names(self.config)
        if not isinstance(procs, dict):
            procs = {n: procs for n in known}
        unknown = [n for n in procs if n not in known]
        if unknown:
            raise ValueError(f"set_attn_processor: unknown processor slots {unknown[:3]} (+{max(0, len(unknown) - 3)})")
        missing = [n for n in known if n not in procs]
        if missing:
            raise ValueError(f"set_attn_processor: a processor is needed for every attention layer; missing {missing[:3]} "
                             f"(+{max(0, len(missing) - 3)})")
        for n, pr in procs.items():
            ok = isinstance(pr, (AttnProcessor2_0, MaskedIPAttnProcessor2_0)) if n.endswith("attn2.processor") \
                else isinstance(pr, AttnProcessor2_0)
            if not ok:
                raise ValueError(f"set_attn_processor: {type(pr).__name__} at {n} is not executed by the MI355X launch plan "
                                 f"(attn1: AttnProcessor2_0, attn2: MaskedIPAttnProcessor2_0)")
        self._attn_processors = dict(procs)

    # ---- execution
    def packed(self) -> PackedUNet:
        if self._packed is None:
            shapes = param_shapes(self.config)
            missing = [k for k in shapes if k not in self._sd]
            if missing:
                raise RuntimeError(f"UNetMangaModel has no weights for {missing[:4]} ... call load_state_dict / "
                                   f"set_manga_modules first")
            if self.device.type != "cuda":
                raise RuntimeError("UNetMangaModel runs on an MI355X only (device must be cuda/hip); no CPU fallback")
            self._packed = PackedUNet(self.config, self._sd, self.device)
        return self._packed

    # Launch plans are cached per (batch, latent size): each owns its activation buffers, K/V panels and a captured
    # hipGraph (GBs at batch 32).  A serving queue with many bucket fill levels must not accumulate them without bound,
    # so the cache is LRU with a byte budget (DIFFSENSEI_ENGINE_CACHE_GB, default 64 of the 288 GB); the engine in use
    # is never evicted.
    engine_cache_bytes = int(float(os.environ.get("DIFFSENSEI_ENGINE_CACHE_GB", "64")) * (1 << 30))

    def engine(self, batch: int, height: int, width: int, aspect_ratio: Optional[float] = None) -> UNetEngine:
        key = (batch, height, width, None if aspect_ratio is None else round(float(aspect_ratio), 6), self.attention_dtype)
        eng = self._engines.pop(key, None)
        if eng is None:
            eng = UNetEngine(self.packed(), batch, height, width, aspect_ratio, attention=self.attention_dtype)
        self._engines[key] = eng                      # dict order = recency (most recent last)
        total = sum(e.nbytes() for e in self._engines.values())
        for k in list(self._engines):
            if total <= self.engine_cache_bytes or k == key:
                break
            total -= self._engines.pop(k).nbytes()
        return eng

    def ip_scale(self) -> float:
        for p in self._attn_processors.values():
            if hasattr(p, "scale"):
                return float(p.scale)
        return 1.0

    def forward(self, sample: Tensor, timestep, encoder_hidden_states: Tensor, timestep_cond=None,
                attention_mask=None, cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                added_cond_kwargs: Optional[Dict[str, Tensor]] = None, down_block_additional_residuals=None,
                mid_block_additional_residual=None, down_intrablock_additional_residuals=None,
                encoder_attention_mask=None, return_dict: bool = True, dialog_bbox: Tensor = None):
        """Same signature as reference src/models/unet.py:116-132.  sample: [B,4,H,W]."""
        if any(a is not None for a in (timestep_cond, attention_mask, down_block_additional_residuals,
                                       mid_block_additional_residual, down_intrablock_additional_residuals,
                                       encoder_attention_mask)):
            raise NotImplementedError("ControlNet/T2I-adapter residuals and attention masks are outside the DiffSensei "
                                      "sampling path")
        if not cross_attention_kwargs or "bbox" not in cross_attention_kwargs:
            raise ValueError("cross_attention_kwargs={'bbox':..., 'aspect_ratio':...} is required")
        if not added_cond_kwargs or "text_embeds" not in added_cond_kwargs or "time_ids" not in added_cond_kwargs:
            raise ValueError("added_cond_kwargs={'text_embeds':..., 'time_ids':...} is required")
        B, Cin, H, W = sample.shape
        ar = cross_attention_kwargs.get("aspect_ratio", H / W)
        eng = self.engine(B, H, W, ar)
        boxes = None if dialog_bbox is None else dialog_pixel_boxes(dialog_bbox, H, W)
        eng.set_request(encoder_hidden_states, added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"],
                        cross_attention_kwargs["bbox"], boxes, self.ip_scale())
        t = float(timestep) if not torch.is_tensor(timestep) else float(timestep.reshape(-1)[0])
        eng.table[0, 0] = t
        eng.ctr.zero_()
        x = sample.to(self.device, torch.float16).reshape(B, Cin, H * W).contiguous()
        eng.x_in.copy_(ops.nchw_to_nhwc(x))
        eng.forward_plan.run()
        out = ops.nhwc_to_nchw(eng.eps).reshape(B, self.config.out_channels, H, W)
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out)

    __call__ = forward
