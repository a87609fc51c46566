"""Checkpoint directory reader for `DiffSenseiPipeline.from_pretrained` — the construction call of the reference's
recipe (scripts/demo/gradio_wo_mllm.py:189-194, scripts/demo/gradio.py:232-237):

    pipeline = DiffSenseiPipeline.from_pretrained(os.path.join(ckpt, "image_generator"), unet=unet,
                                                  image_encoder=image_encoder, torch_dtype=torch.float16)

The reference inherits it from diffusers' `DiffusionPipeline`; diffusers is not a dependency here, so the diffusers
directory layout [3P] is read directly:

    model_index.json                      component table {name: [library, class]} + pipeline flags
    scheduler/scheduler_config.json       `_class_name` EulerDiscreteScheduler | DDIMScheduler + its constructor arguments
    vae/config.json + weights             AutoencoderKL (only the decoder + post_quant_conv are used on this path)
    text_encoder/, text_encoder_2/        transformers CLIPTextModel / CLIPTextModelWithProjection (config.json + weights)
    tokenizer/, tokenizer_2/              transformers CLIPTokenizer files (loaded with transformers, host only)
    unet/ (optional here)                 config.json + weights, when `unet=` is not passed
    image_encoder/ (optional here)        when `image_encoder=` is not passed and the directory exists

Weights: `*.safetensors` (preferred; `diffusion_pytorch_model[.fp16].safetensors`, `model[.fp16].safetensors`), sharded
safetensors through their `*.index.json`, or `*.bin` pickles (`torch.load(weights_only=True)`).  Every component ends up
as a HIP engine (`vae.VaeDecoderEngine`, `encoders.ClipTextEngine`, ...); nothing here computes.
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, Optional

import torch

Tensor = torch.Tensor

_WEIGHT_NAMES = ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors", "model.safetensors",
                 "model.fp16.safetensors", "diffusion_pytorch_model.bin", "pytorch_model.bin")
_INDEX_NAMES = ("diffusion_pytorch_model.safetensors.index.json", "model.safetensors.index.json")


def read_json(path: str) -> dict:
    with open(path) as fh:
        return json.load(fh)


def load_weights(folder: str) -> Dict[str, Tensor]:
    """State dict of one component directory (first match of the diffusers / transformers file names)."""
    for name in _INDEX_NAMES:                      # sharded safetensors
        idx = os.path.join(folder, name)
        if os.path.exists(idx):
            from safetensors.torch import load_file
            sd: Dict[str, Tensor] = {}
            for shard in sorted(set(read_json(idx)["weight_map"].values())):
                sd.update(load_file(os.path.join(folder, shard)))
            return sd
    for name in _WEIGHT_NAMES:
        p = os.path.join(folder, name)
        if os.path.exists(p):
            if p.endswith(".safetensors"):
                from safetensors.torch import load_file
                return load_file(p)
            return torch.load(p, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no weight file ({', '.join(_WEIGHT_NAMES)}) in {folder}")


class _Loaded:
    """What the engines' `from_transformers` constructors touch of a transformers model: `.config`, `.state_dict()`."""

    def __init__(self, config, sd: Dict[str, Tensor]):
        self.config, self._sd = config, sd

    def state_dict(self):
        return self._sd


def load_scheduler(folder: str):
    from .schedulers import DDIMScheduler, EulerDiscreteScheduler
    cfg = read_json(os.path.join(folder, "scheduler_config.json"))
    classes = {"EulerDiscreteScheduler": EulerDiscreteScheduler, "DDIMScheduler": DDIMScheduler}
    name = cfg.get("_class_name", "EulerDiscreteScheduler")
    if name not in classes:
        raise NotImplementedError(f"scheduler {name}: the MI355X sampler kernel implements {sorted(classes)}")
    kw = {k: v for k, v in cfg.items() if not k.startswith("_")}
    return classes[name](**kw)


def load_vae(folder: str, device):
    from .vae import VaeConfig, VaeDecoderEngine
    c = read_json(os.path.join(folder, "config.json"))
    cfg = VaeConfig(tuple(c.get("block_out_channels", (128, 256, 512, 512))), int(c.get("layers_per_block", 2)),
                    int(c.get("latent_channels", 4)), int(c.get("out_channels", 3)), int(c.get("norm_num_groups", 32)),
                    float(c.get("scaling_factor", 0.13025)), bool(c.get("force_upcast", True)), c.get("latents_mean"),
                    c.get("latents_std"))
    return VaeDecoderEngine.from_state_dict(load_weights(folder), cfg, device)


def load_text_encoder(folder: str, device):
    from transformers import CLIPTextConfig
    from .encoders import ClipTextEngine
    cfg = CLIPTextConfig.from_json_file(os.path.join(folder, "config.json"))
    return ClipTextEngine.from_transformers(_Loaded(cfg, load_weights(folder)), device)


def load_clip_vision(folder: str, device):
    from transformers import CLIPVisionConfig
    from .encoders import ClipVisionEngine
    raw = read_json(os.path.join(folder, "config.json"))
    cfg = CLIPVisionConfig(**raw["vision_config"]) if "vision_config" in raw else CLIPVisionConfig.from_json_file(
        os.path.join(folder, "config.json"))
    return ClipVisionEngine.from_transformers(_Loaded(cfg, load_weights(folder)), device)


def load_tokenizer(folder: str):
    from transformers import CLIPTokenizer
    return CLIPTokenizer.from_pretrained(folder)


def load_unet(path: str, device, set_manga: Optional[dict] = None):
    """`UNetMangaModel.from_config(path, subfolder="unet")` + weights, the two calls of gradio_wo_mllm.py:162-169."""
    from .unet import UNetMangaModel
    unet = UNetMangaModel.from_config(path, subfolder="unet", device=device)
    if set_manga is not None:
        unet.set_manga_modules(**set_manga)
    unet.load_state_dict(load_weights(os.path.join(path, "unet")))
    return unet


def load_pipeline_components(path: str, device, have: Dict[str, Any]) -> Dict[str, Any]:
    """Everything `DiffSenseiPipeline.__init__` takes, from a diffusers-layout directory; entries of `have` that are not
    None (the keyword arguments of `from_pretrained`, e.g. `unet=`, `image_encoder=`) are used as given."""
    path = os.fspath(path)
    index_file = os.path.join(path, "model_index.json")
    if not os.path.exists(index_file):
        raise FileNotFoundError(f"{index_file} not found: `from_pretrained` needs a diffusers pipeline directory")
    index = read_json(index_file)
    sub = lambda n: os.path.join(path, n)
    listed = lambda n: isinstance(index.get(n), (list, tuple)) and index[n][0] is not None and os.path.isdir(sub(n))
    out: Dict[str, Any] = dict(have)
    if out.get("scheduler") is None:
        out["scheduler"] = load_scheduler(sub("scheduler"))
    if out.get("vae") is None:
        out["vae"] = load_vae(sub("vae"), device)
    for n in ("text_encoder", "text_encoder_2"):
        if out.get(n) is None:
            out[n] = load_text_encoder(sub(n), device)
    for n in ("tokenizer", "tokenizer_2"):
        if out.get(n) is None:
            out[n] = load_tokenizer(sub(n))
    if out.get("unet") is None:
        out["unet"] = load_unet(path, device)
    if out.get("image_encoder") is None and listed("image_encoder"):
        out["image_encoder"] = load_clip_vision(sub("image_encoder"), device)
    if out.get("feature_extractor") is None and listed("feature_extractor"):
        from transformers import CLIPImageProcessor
        out["feature_extractor"] = CLIPImageProcessor.from_pretrained(sub("feature_extractor"))
    out.setdefault("force_zeros_for_empty_prompt", bool(index.get("force_zeros_for_empty_prompt", True)))
    return out
