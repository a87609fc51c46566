"""Host mirror of reference src/pipelines/pipeline_diffsensei.py `DiffSenseiPipeline`.

Same constructor, `register_manga_modules`, `check_inputs`, `prepare_ip_image_embeds`, `prepare_dialog_bbox`,
`set_ip_scale` and `__call__` signature (reference :43-57, :73-79, :81-102, :104-154, :156-170, :172-178, :181-203;
extra TRAILING keyword arguments only).  Returns an object with `.images`.

What runs where
  * denoising loop (UNet + CFG + scheduler step): C++ launch plan over the HIP kernels, one `step_plan` run (or
    hipGraph replay) per step, zero host arithmetic and zero host<->device syncs inside the loop;
  * character encoders (CLIP-H, Magi ViT-MAE) + Resampler: HIP engines (`encoders.py`, `resampler.py`);
  * the two SDXL CLIP text encoders (`encode_prompt`): HIP engine (`encoders.ClipTextEngine`),
    transformers models passed to the constructor are re-laid-out for it; tokenisation stays on the host;
  * the VAE decode + denormalisation (reference :339-367): HIP engine `vae.VaeDecoderEngine`
    (bf16 storage / fp32 math where the reference upcasts the VAE to fp32); a diffusers `AutoencoderKL` passed to the
    constructor is re-laid-out for it, any other object with the `decode` protocol is used as is;
    `output_type` in {"pil", "np", "pt", "latent"}.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, List, Optional, Sequence, Tuple, Union

import torch

from .encoders import ClipTextEngine, ClipVisionEngine, ViTMAEEngine
from .unet import UNetMangaModel, dialog_pixel_boxes
from .vae import VaeDecoderEngine

Tensor = torch.Tensor


@dataclass
class StableDiffusionXLPipelineOutput:
    images: Any


def _black_image():
    from PIL import Image
    return Image.new("RGB", (224, 224), (0, 0, 0))


class DiffSenseiPipeline:
    def __init__(self, vae, text_encoder, text_encoder_2, tokenizer, tokenizer_2, scheduler, unet: UNetMangaModel,
                 image_encoder, feature_extractor=None, force_zeros_for_empty_prompt: bool = True):
        self.scheduler, self.unet = scheduler, unet          # first: the engine conversions below read unet.device
        self.vae = self._as_vae_engine(vae)
        self.tokenizer, self.tokenizer_2 = tokenizer, tokenizer_2
        self.text_encoder = self._as_text_engine(text_encoder)
        self.text_encoder_2 = self._as_text_engine(text_encoder_2)
        self.image_encoder = self._as_clip_engine(image_encoder)
        self.feature_extractor = feature_extractor
        self.force_zeros_for_empty_prompt = force_zeros_for_empty_prompt
        self.vae_scale_factor = 8
        self.default_sample_size = unet.config.sample_size
        self.progress_bar_config = {"disable": True}
        self.magi_image_encoder = None
        self.image_proj_model = None
        self._clip_proc = None
        # character references are resized / cropped / normalised by csrc/preprocess.hip (bytes identical to Pillow; e2e
        # character tokens vs the host processors rel-L2 6.9e-4, profiles/r02_device_preprocess_e2e.log).  False, or a
        # reference that is not a PIL image, goes through the transformers processors on the host like the reference.
        # DIFFSENSEI_DEVICE_PREPROCESS=0 restores the reference's host path process-wide.
        self.device_preprocess = os.environ.get("DIFFSENSEI_DEVICE_PREPROCESS", "1") != "0"
        self._device_pre = None
        self._magi_proc = None
        self._guidance_scale = 1.0
        self._interrupt = False
        self._stream = None
        self.use_graph = os.environ.get("DIFFSENSEI_GRAPH", "1") != "0"
        self.last_run_info = {}

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, unet: Optional[UNetMangaModel] = None, image_encoder=None,
                        torch_dtype: Optional[torch.dtype] = torch.float16, device: Optional[Union[str, torch.device]] = None,
                        **components) -> "DiffSenseiPipeline":
        """The reference's construction call (scripts/demo/gradio_wo_mllm.py:189-194, gradio.py:232-237; inherited there
        from diffusers' DiffusionPipeline [3P]): read a diffusers-layout directory - `model_index.json`,
        `scheduler/scheduler_config.json`, `vae/`, `text_encoder/`, `text_encoder_2/`, `tokenizer/`, `tokenizer_2/`
        (safetensors or .bin) - without diffusers, and build the HIP engines.  Components passed as keyword arguments
        (`unet=`, `image_encoder=`, also `vae=`, `scheduler=`, ...) are used as given, exactly like the reference does for
        its UNetMangaModel and CLIP image encoder; a missing `unet=` is loaded from `unet/`."""
        if torch_dtype not in (None, torch.float16):
            raise ValueError("the MI355X engines compute in fp16 (the reference's inference dtype): torch_dtype=torch.float16")
        if not os.path.isdir(os.fspath(pretrained_model_name_or_path)):
            raise FileNotFoundError(f"{pretrained_model_name_or_path}: a local checkpoint directory is required "
                                    f"(there is no hub download in this framework)")
        from .loading import load_pipeline_components
        dev = torch.device(device) if device is not None else (unet.device if unet is not None else torch.device("cuda"))
        names = ("vae", "text_encoder", "text_encoder_2", "tokenizer", "tokenizer_2", "scheduler", "feature_extractor")
        unknown = set(components) - set(names) - {"force_zeros_for_empty_prompt"}
        if unknown:
            raise TypeError(f"from_pretrained: unexpected components {sorted(unknown)}")
        have = {n: components.get(n) for n in names}
        have.update(unet=unet, image_encoder=image_encoder)
        if "force_zeros_for_empty_prompt" in components:
            have["force_zeros_for_empty_prompt"] = components["force_zeros_for_empty_prompt"]
        c = load_pipeline_components(pretrained_model_name_or_path, dev, have)
        return cls(vae=c["vae"], text_encoder=c["text_encoder"], text_encoder_2=c["text_encoder_2"], tokenizer=c["tokenizer"],
                   tokenizer_2=c["tokenizer_2"], scheduler=c["scheduler"], unet=c["unet"], image_encoder=c["image_encoder"],
                   feature_extractor=c.get("feature_extractor"),
                   force_zeros_for_empty_prompt=c["force_zeros_for_empty_prompt"])

    # ---- plumbing
    @property
    def _execution_device(self):
        return self.unet.device

    @property
    def device(self):
        return self.unet.device

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def interrupt(self):
        return self._interrupt

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1

    def to(self, device=None, dtype=None, **kw):
        self.unet.to(device=device, dtype=dtype)
        if self.image_proj_model is not None:
            self.image_proj_model.to(device=device, dtype=dtype)
        return self

    def _as_clip_engine(self, m):
        if m is None or isinstance(m, ClipVisionEngine):
            return m
        return ClipVisionEngine.from_transformers(m, self.unet.device)

    def _as_vae_engine(self, m):
        """A diffusers AutoencoderKL is re-laid-out for the HIP decoder; anything else with `.decode` is used as is."""
        if m is None or isinstance(m, VaeDecoderEngine):
            return m
        if hasattr(m, "state_dict") and hasattr(getattr(m, "config", None), "block_out_channels"):
            return VaeDecoderEngine.from_diffusers(m, self.unet.device)
        return m

    def _as_text_engine(self, m):
        if m is None or isinstance(m, ClipTextEngine):
            return m
        return ClipTextEngine.from_transformers(m, self.unet.device)

    def _as_magi_engine(self, m):
        if m is None or isinstance(m, ViTMAEEngine):
            return m
        return ViTMAEEngine.from_transformers(m, self.unet.device)

    def tensors(self) -> List[Tensor]:
        """Every frozen weight tensor of the pipeline's engines (UNet state dict, text encoders, CLIP-H, Magi, Resampler,
        VAE decoder): the list `distributed.broadcast_pipeline` sends from rank 0 over RCCL.  Components that are not HIP
        engines (a user-supplied VAE object with its own `.decode`) are skipped."""
        out: List[Tensor] = []
        for m in (self.unet, self.text_encoder, self.text_encoder_2, self.image_encoder, self.magi_image_encoder,
                  self.image_proj_model, self.vae):
            if m is not None and hasattr(m, "tensors"):
                out += list(m.tensors())
        return out

    def register_manga_modules(self, magi_image_encoder, image_proj_model):
        """reference :73-79"""
        self.magi_image_encoder = self._as_magi_engine(magi_image_encoder)
        self.image_proj_model = image_proj_model

    # ---- reference :81-102 (same checks, same messages)
    def check_inputs(self, prompt, prompt_2, ip_images, ip_image_embeds, ip_bbox):
        if prompt is None:
            raise ValueError(f"`prompt` has to be of type `str` but is {type(prompt)}")
        elif prompt is not None and not isinstance(prompt, str):
            raise ValueError(f"`prompt` has to be of type `str` but is {type(prompt)}")
        elif prompt_2 is not None and not isinstance(prompt_2, str):
            raise ValueError(f"`prompt_2` has to be of type `str` but is {type(prompt_2)}")
        if len(ip_images) > 0 and ip_image_embeds is not None:
            raise ValueError(f"`ip_images` and `ip_image_embeds` can not be input together!")
        num_ips = len(ip_image_embeds) if ip_image_embeds is not None else len(ip_images)
        if num_ips != len(ip_bbox):
            raise ValueError(f"`ip_images` must have the same length as `ip_bbox`. But they are in length {num_ips} "
                             f"and {len(ip_bbox)}!")

    def _processors(self):
        if self._clip_proc is None:
            from transformers import CLIPImageProcessor, ViTImageProcessor
            self._clip_proc, self._magi_proc = CLIPImageProcessor(), ViTImageProcessor()
        return self._clip_proc, self._magi_proc

    def _encode_refs(self, ip_images, zero_padded: bool):
        """CLIP penultimate states [1,4,257,1280] and Magi CLS embeddings [1,4,768] of the (black-padded) references."""
        max_num_ips = self.unet.config.max_num_ips
        ip_images = list(ip_images)[:max_num_ips]
        num_ips = len(ip_images)
        while len(ip_images) < max_num_ips:
            ip_images.append(_black_image())
        if self.device_preprocess and all(hasattr(im, "convert") and hasattr(im, "size") for im in ip_images):
            # Pillow's resize + crop + normalise on the device (preprocess.py): only the RGB bytes go up
            if self._device_pre is None:
                from .preprocess import DevicePreprocessor
                self._device_pre = DevicePreprocessor(self._execution_device)
            clip_px, magi_px = self._device_pre.clip(ip_images), self._device_pre.vit(ip_images)
        else:
            clip_proc, magi_proc = self._processors()
            clip_px = clip_proc(images=ip_images, return_tensors="pt").pixel_values
            magi_px = magi_proc(images=ip_images, return_tensors="pt").pixel_values
        clip_embeds = self.image_encoder.penultimate_hidden(clip_px).unsqueeze(0)
        magi_embeds = self.magi_image_encoder.cls_embedding(magi_px).unsqueeze(0)
        if zero_padded:
            clip_embeds[0, num_ips:] = 0
            magi_embeds[0, num_ips:] = 0
        return clip_embeds, magi_embeds

    def encode_ip_tokens(self, ip_images) -> Tensor:
        """Resampler tokens [1, num_dummy + max_num_ips*num_vision_tokens, dim] of the references exactly as the MLLM
        pre-pass computes them (reference scripts/demo/gradio.py:85-98: black padding, padded slots NOT zeroed)."""
        clip_embeds, magi_embeds = self._encode_refs(ip_images, zero_padded=False)
        x = self.image_proj_model(clip_embeds, magi_embeds)
        return x.view(1, -1, x.shape[-1])

    # ---- reference :104-154
    def prepare_ip_image_embeds(self, ip_images, ip_image_embeds, ip_bbox, num_samples):
        cfg = self.unet.config
        dev = self._execution_device
        max_num_ips = cfg.max_num_ips
        if ip_image_embeds is not None:
            ip_image_embeds = ip_image_embeds[:max_num_ips]
        ip_bbox = [list(b) for b in ip_bbox][:max_num_ips]
        while len(ip_bbox) < max_num_ips:
            ip_bbox.append([0.0, 0.0, 0.0, 0.0])
        clip_embeds, magi_embeds = self._encode_refs(ip_images, zero_padded=True)   # [1,4,257,1280], [1,4,768]
        image_embeds = self.image_proj_model(clip_embeds, magi_embeds)
        negative_image_embeds = self.image_proj_model(torch.zeros_like(clip_embeds), torch.zeros_like(magi_embeds))
        bbox = torch.tensor(ip_bbox, dtype=torch.float32).unsqueeze(0).to(dev)
        negative_bbox = torch.zeros_like(bbox)
        nv = cfg.num_vision_tokens
        image_embeds = image_embeds.view(1, nv + max_num_ips * nv, image_embeds.shape[-1])
        if ip_image_embeds is not None:
            n_e, _, dim = ip_image_embeds.shape
            image_embeds[0, nv:(1 + n_e) * nv, :] = ip_image_embeds.to(image_embeds).view(1, -1, dim)
        negative_image_embeds = negative_image_embeds.view(1, nv + max_num_ips * nv, image_embeds.shape[-1])
        image_embeds = image_embeds.repeat(num_samples, 1, 1).to(torch.float16)
        negative_image_embeds = negative_image_embeds.repeat(num_samples, 1, 1).to(torch.float16)
        return (negative_image_embeds, image_embeds, negative_bbox.repeat(num_samples, 1, 1),
                bbox.repeat(num_samples, 1, 1))

    # ---- reference :156-170
    def prepare_dialog_bbox(self, dialog_bbox, num_samples):
        max_num_dialogs = self.unet.config.max_num_dialogs
        dialog_bbox = [list(b) for b in dialog_bbox][:max_num_dialogs]
        while len(dialog_bbox) < max_num_dialogs:
            dialog_bbox.append([0.0, 0.0, 0.0, 0.0])
        db = torch.tensor(dialog_bbox, dtype=torch.float32).unsqueeze(0).to(dtype=self.unet.dtype)
        db = db.repeat(num_samples, 1, 1)
        return torch.zeros_like(db), db

    # ---- reference :172-178
    def set_ip_scale(self, scale):
        for attn_processor in self.unet.attn_processors.values():
            if hasattr(attn_processor, "scale"):
                attn_processor.scale = scale

    # ---- text encoding (transformers modules, not on this path; diffusers SDXL `encode_prompt` semantics [3P])
    def encode_prompt(self, prompt, prompt_2, device, num_images_per_prompt, do_cfg, negative_prompt, negative_prompt_2):
        if self.text_encoder is None or self.tokenizer is None:
            raise ValueError("no text encoders registered: pass prompt_embeds / negative_prompt_embeds / "
                             "pooled_prompt_embeds / negative_pooled_prompt_embeds")
        prompts = [prompt, prompt_2 or prompt]
        negs = [negative_prompt or "", negative_prompt_2 or negative_prompt or ""]
        toks, encs = [self.tokenizer, self.tokenizer_2], [self.text_encoder, self.text_encoder_2]

        def enc(texts):
            embs, pooled = [], None
            for text, tok, te in zip(texts, toks, encs):
                ids = tok(text, padding="max_length", max_length=tok.model_max_length, truncation=True,
                          return_tensors="pt").input_ids
                hidden, pooled = te.encode(ids)     # HIP text-encoder engine: (hidden_states[-2], out[0])
                embs.append(hidden)
            return torch.cat(embs, dim=-1), pooled

        with torch.no_grad():
            pe, pp = enc(prompts)
            if do_cfg and negative_prompt is None and self.force_zeros_for_empty_prompt:
                ne, npool = torch.zeros_like(pe), torch.zeros_like(pp)
            else:
                ne, npool = enc(negs)
        rep = lambda t: t.repeat_interleave(num_images_per_prompt, dim=0).to(device, torch.float16)
        return rep(pe), rep(ne), rep(pp), rep(npool)

    def prepare_latents(self, batch_size, num_channels, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels, int(height) // self.vae_scale_factor, int(width) // self.vae_scale_factor)
        if latents is None:
            gdev = generator.device if generator is not None and not isinstance(generator, list) else torch.device(device)
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            latents = latents.to(device=device, dtype=dtype)
        return latents * self.scheduler.init_noise_sigma

    # ---- reference :180-372
    @torch.no_grad()
    def __call__(self, prompt: str, prompt_2: str = None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 40, guidance_scale: float = 5.0,
                 negative_prompt: Optional[Union[str, List[str]]] = None,
                 negative_prompt_2: Optional[Union[str, List[str]]] = None, num_samples: Optional[int] = 1,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
                 original_size: Optional[Tuple[int, int]] = None, crops_coords_top_left: Tuple[int, int] = (0, 0),
                 target_size: Optional[Tuple[int, int]] = None, min_size_step: Optional[int] = 8,
                 ip_images=[], ip_image_embeds: Optional[Tensor] = None, ip_bbox: Optional[List[List[float]]] = [],
                 ip_scale: Optional[int] = 1.0, dialog_bbox: Optional[List[List[float]]] = [],
                 # ---- trailing extensions (not in the reference signature)
                 latents: Optional[Tensor] = None, prompt_embeds: Optional[Tensor] = None,
                 negative_prompt_embeds: Optional[Tensor] = None, pooled_prompt_embeds: Optional[Tensor] = None,
                 negative_pooled_prompt_embeds: Optional[Tensor] = None, output_type: str = "pil",
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs: Sequence[str] = ("latents",)):
        """`callback_on_step_end(pipe, step_index, timestep, {"latents": device tensor}) -> dict | None` is diffusers'
        SDXL-pipeline hook [3P]; together with `pipe._interrupt = True` it is the reference's early exit: the loop
        `continue`s over the remaining steps (reference :314-315) and the call still decodes and post-processes.  Latents
        may be edited in place or returned (`{"latents": new}`), as in diffusers; `latents` is the only tensor the launch
        plan can hand out per step, so other `callback_on_step_end_tensor_inputs` are refused like diffusers refuses unknown
        names."""
        bad = [k for k in callback_on_step_end_tensor_inputs if k != "latents"]
        if bad:
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in ['latents'], but found {bad}")
        self._interrupt = False                                   # reference :226
        cond = self._conditioning(prompt, prompt_2, height, width, num_inference_steps, guidance_scale, negative_prompt,
                                  negative_prompt_2, num_samples, generator, original_size, crops_coords_top_left,
                                  target_size, ip_images, ip_image_embeds, ip_bbox, ip_scale, dialog_bbox, latents,
                                  prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds,
                                  negative_pooled_prompt_embeds)
        out_latents = self._denoise([cond], num_inference_steps, guidance_scale, ip_scale, callback_on_step_end)
        return StableDiffusionXLPipelineOutput(images=self._postprocess(out_latents, output_type))

    # ---- one request's conditioning tensors (reference :205-309), `num_samples` rows each, conditional and negative
    def _conditioning(self, prompt, prompt_2, height, width, num_inference_steps, guidance_scale, negative_prompt,
                      negative_prompt_2, num_samples, generator, original_size, crops_coords_top_left, target_size,
                      ip_images, ip_image_embeds, ip_bbox, ip_scale, dialog_bbox, latents, prompt_embeds,
                      negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds):
        height = height or self.default_sample_size * self.vae_scale_factor
        width = width or self.default_sample_size * self.vae_scale_factor
        original_size = original_size or (height, width)
        target_size = target_size or (height, width)
        self.check_inputs(prompt, prompt_2, ip_images, ip_image_embeds, ip_bbox)
        if height % self.vae_scale_factor or width % self.vae_scale_factor:
            raise ValueError(f"`height` and `width` have to be divisible by {self.vae_scale_factor} but are {height} and {width}.")
        self._guidance_scale = guidance_scale
        device = self._execution_device
        self.set_ip_scale(ip_scale)
        do_cfg = self.do_classifier_free_guidance

        if prompt_embeds is None:
            prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = \
                self.encode_prompt(prompt, prompt_2, device, num_samples, do_cfg, negative_prompt, negative_prompt_2)
        else:
            f = lambda t: None if t is None else t.to(device, torch.float16)
            prompt_embeds, pooled_prompt_embeds = f(prompt_embeds), f(pooled_prompt_embeds)
            negative_prompt_embeds = f(negative_prompt_embeds) if negative_prompt_embeds is not None \
                else torch.zeros_like(prompt_embeds)
            negative_pooled_prompt_embeds = f(negative_pooled_prompt_embeds) if negative_pooled_prompt_embeds is not None \
                else torch.zeros_like(pooled_prompt_embeds)
            if prompt_embeds.shape[0] == 1 and num_samples > 1:
                prompt_embeds = prompt_embeds.repeat(num_samples, 1, 1)
                negative_prompt_embeds = negative_prompt_embeds.repeat(num_samples, 1, 1)
                pooled_prompt_embeds = pooled_prompt_embeds.repeat(num_samples, 1)
                negative_pooled_prompt_embeds = negative_pooled_prompt_embeds.repeat(num_samples, 1)

        self.scheduler.set_timesteps(num_inference_steps, device=device)
        lat = self.prepare_latents(num_samples, self.unet.config.in_channels, height, width, torch.float16, device,
                                   generator, latents)
        neg_img, img, neg_bbox, bbox = self.prepare_ip_image_embeds(ip_images, ip_image_embeds, list(ip_bbox), num_samples)
        add_time_ids = torch.tensor([list(original_size) + list(crops_coords_top_left) + list(target_size)],
                                    dtype=torch.float16, device=device).repeat(num_samples, 1)
        neg_dialog, dialog = self.prepare_dialog_bbox(list(dialog_bbox), num_samples)
        to = lambda t: t.to(device)
        return {"n": num_samples, "lat": lat, "time_ids": add_time_ids,
                "pos": (to(prompt_embeds), to(pooled_prompt_embeds), to(img), to(bbox), to(dialog)),
                "neg": (to(negative_prompt_embeds), to(negative_pooled_prompt_embeds), to(neg_img), to(neg_bbox),
                        to(neg_dialog))}

    # ---- the denoising loop over one UNet batch assembled from >= 1 requests of the same shape (reference :310-337)
    def _denoise(self, conds, num_inference_steps, guidance_scale, ip_scale, callback_on_step_end=None) -> Tensor:
        device = self._execution_device
        self._guidance_scale = guidance_scale
        do_cfg = self.do_classifier_free_guidance
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        lat = torch.cat([c["lat"] for c in conds], dim=0)
        num_samples = lat.shape[0]
        H, W = lat.shape[-2], lat.shape[-1]
        aspect_ratio = H / W
        cat = lambda side, i: torch.cat([c[side][i] for c in conds], dim=0)
        prompt_embeds, add_text_embeds, img, bbox, dialog = (cat("pos", i) for i in range(5))
        add_time_ids = torch.cat([c["time_ids"] for c in conds], dim=0)
        if do_cfg:  # CFG batch layout: all negative rows, then all conditional rows (reference :300-309)
            prompt_embeds = torch.cat([cat("neg", 0), prompt_embeds], dim=0)
            add_text_embeds = torch.cat([cat("neg", 1), add_text_embeds], dim=0)
            add_time_ids = torch.cat([add_time_ids, add_time_ids], dim=0)
            img = torch.cat([cat("neg", 2), img], dim=0)
            bbox = torch.cat([cat("neg", 3), bbox], dim=0)
            dialog = torch.cat([cat("neg", 4), dialog], dim=0)
        else:
            # Without CFG the reference still passes bbox = cat([negative_ip_bbox, ip_bbox]) (reference :270-273) while the
            # UNet batch is only the `num_samples` conditional rows, so its mask builder indexes the FIRST num_samples rows:
            # the all-zero negative boxes (attention_processor.py:141-163 loops `for i in range(batch)`).  Mirrored here so
            # guidance_scale <= 1 gives the reference's images; `dialog_bbox` is not concatenated there and stays positive.
            bbox = cat("neg", 3)
        enc = torch.cat([prompt_embeds, img], dim=1)

        # one plan replay per step
        B = enc.shape[0]
        eng = self.unet.engine(B, H, W, aspect_ratio)
        eng.build_sampler(num_samples, self.scheduler.kind, do_cfg)
        eng.set_request(enc, add_text_embeds, add_time_ids, bbox, dialog_pixel_boxes(dialog, H, W), float(ip_scale))
        eng.load_schedule(torch.from_numpy(self.scheduler.coef_table(float(guidance_scale))))
        eng.latents.copy_(lat)
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        st = self._stream
        st.wait_stream(torch.cuda.current_stream(device))
        graph = False
        with torch.cuda.stream(st):
            eng.prep_plan.run(st.cuda_stream)
            n0 = 0
            if self.use_graph:
                if not eng.step_plan.captured:
                    eng.step_plan.run(st.cuda_stream)          # first step eager (also warms lazy kernel state)
                    n0 = 1
                    eng.step_plan.capture(st.cuda_stream)
                graph = True
            timesteps = self.scheduler.timesteps

            def hook(i):
                # diffusers' contract [3P]: `latents = callback_outputs.pop("latents", latents)` - a callback may return a
                # dict with replaced latents instead of editing `eng.latents` in place; both forms are honoured
                ret = callback_on_step_end(self, i, timesteps[i], {"latents": eng.latents})
                if isinstance(ret, dict):
                    new = ret.get("latents")
                    if isinstance(new, Tensor) and new.data_ptr() != eng.latents.data_ptr():
                        eng.latents.copy_(new.to(eng.latents.device, eng.latents.dtype).reshape(eng.latents.shape))

            if n0 and callback_on_step_end is not None:
                hook(0)
            for i in range(n0, num_inference_steps):
                if self._interrupt:                               # reference :314-315 `if self.interrupt: continue`
                    continue
                if graph:
                    eng.step_plan.replay(st.cuda_stream)
                else:
                    eng.step_plan.run(st.cuda_stream)
                if callback_on_step_end is not None:              # launched, not synchronised: the hook sees device tensors
                    hook(i)
        torch.cuda.current_stream(device).wait_stream(st)
        self.last_run_info = {"graph": graph, "ops_per_step": eng.step_plan.n, "batch": B, "latent_hw": (H, W)}
        return eng.latents.clone()

    # ---- reference :339-367: VAE decode + image_processor.postprocess
    def _postprocess(self, out_latents: Tensor, output_type: str):
        if output_type == "latent" or self.vae is None:
            if output_type != "latent" and self.vae is None:
                raise ValueError("no VAE registered: call with output_type='latent'")
            return out_latents
        scaling = getattr(getattr(self.vae, "config", None), "scaling_factor", 0.13025)
        if output_type == "pil" and isinstance(self.vae, VaeDecoderEngine) and out_latents.is_cuda \
                and (out_latents.shape[2] * out_latents.shape[3] * 64) % 4 == 0 and out_latents.shape[0] > 1:
            return self._decode_to_pil_pipelined(out_latents, scaling)
        if isinstance(self.vae, VaeDecoderEngine):  # incl. postprocess' denormalize, all on the HIP kernels
            image = self.vae.decode(out_latents, return_dict=False, scaling_factor=scaling, denormalize=True,
                                    latents_affine=True)[0]      # latents_mean / latents_std (:348-357) folded at load time
        else:
            vc = getattr(self.vae, "config", None)
            lm, ls = getattr(vc, "latents_mean", None), getattr(vc, "latents_std", None)
            z = out_latents.float()
            if lm is not None and ls is not None:                # reference :348-357, a user-supplied decoder object
                view = lambda v: torch.tensor(list(v), dtype=z.dtype, device=z.device).view(1, -1, 1, 1)
                z = z * view(ls) / scaling + view(lm)
            else:
                z = z / scaling
            image = self.vae.decode(z, return_dict=False)[0]
            image = (image / 2 + 0.5).clamp(0, 1)
        if output_type == "pt":
            return image
        from PIL import Image
        if output_type == "pil" and image.is_cuda and image.dtype == torch.float32 and image.shape[1] == 3 \
                and (image.shape[2] * image.shape[3]) % 4 == 0:
            # (x*255).round() -> uint8 NHWC on the device: 3 B/pixel cross PCIe instead of 12, no host arithmetic
            from . import ops
            u8 = ops.image_to_u8(image.contiguous())
            host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
            host.copy_(u8, non_blocking=True)
            torch.cuda.current_stream(image.device).synchronize()
            return [Image.fromarray(im) for im in host.numpy()]
        image = image.permute(0, 2, 3, 1).float().cpu().numpy()
        if output_type == "np":
            return image
        return [Image.fromarray((im * 255).round().astype("uint8")) for im in image]

    def _decode_to_pil_pipelined(self, out_latents: Tensor, scaling: float):
        """`vae.decode` + `image_processor.postprocess(output_type="pil")` (reference :359-367) for several images, the same
        kernels and bytes as the one-shot path above, but chunk by chunk: while the decoder works on chunk i + 1 the host wraps the
        uint8 pixels of chunk i (already copied to pinned memory) into PIL images - the ~1.5 ms per 1024 x 1024 image that
        `Image.fromarray` costs no longer sits behind the whole decode with the GPU idle."""
        from PIL import Image
        from . import ops
        B, _, h, w = out_latents.shape
        chunk = self.vae.decode_chunk(h, w, B)
        stream = torch.cuda.current_stream(out_latents.device)
        pending, images = [], []

        def drain(n_keep):
            while len(pending) > n_keep:
                ev, host = pending.pop(0)
                ev.synchronize()
                images.extend(Image.fromarray(im) for im in host.numpy())

        for i in range(0, B, chunk):
            img = self.vae.decode(out_latents[i:i + chunk], return_dict=False, scaling_factor=scaling, denormalize=True,
                                  latents_affine=True)[0]
            u8 = ops.image_to_u8(img.contiguous())
            host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
            host.copy_(u8, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
            pending.append((ev, host))
            drain(1)          # wrap the PREVIOUS chunk while this one is still on the GPU
        drain(0)
        return images

    # ---- several requests of one shape in ONE UNet batch (serving front-end)
    @torch.no_grad()
    def generate_batch(self, requests: List[dict], output_type: str = "pil") -> List[Any]:
        """Each request: the keyword arguments of `__call__` (without `output_type`).  All must share height, width,
        num_inference_steps, guidance_scale and ip_scale (`serving.bucket_key`); prompts, character references, boxes,
        seeds and `num_samples` are per request.  Returns one `.images`-like object per request, in order."""
        if not requests:
            return []
        self._interrupt = False          # like `__call__` (reference :226): an earlier interrupted call must not leak into this one
        key = lambda r: (r.get("height"), r.get("width"), r.get("num_inference_steps", 40), r.get("guidance_scale", 5.0),
                         r.get("ip_scale", 1.0))
        if any(key(r) != key(requests[0]) for r in requests):
            raise ValueError("generate_batch: requests must share height/width/steps/guidance_scale/ip_scale")
        names = ("prompt", "prompt_2", "height", "width", "num_inference_steps", "guidance_scale", "negative_prompt",
                 "negative_prompt_2", "num_samples", "generator", "original_size", "crops_coords_top_left", "target_size",
                 "ip_images", "ip_image_embeds", "ip_bbox", "ip_scale", "dialog_bbox", "latents", "prompt_embeds",
                 "negative_prompt_embeds", "pooled_prompt_embeds", "negative_pooled_prompt_embeds")
        defaults = dict(prompt_2=None, height=None, width=None, num_inference_steps=40, guidance_scale=5.0,
                        negative_prompt=None, negative_prompt_2=None, num_samples=1, generator=None, original_size=None,
                        crops_coords_top_left=(0, 0), target_size=None, ip_images=[], ip_image_embeds=None, ip_bbox=[],
                        ip_scale=1.0, dialog_bbox=[], latents=None, prompt_embeds=None, negative_prompt_embeds=None,
                        pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None)
        conds = []
        for r in requests:
            unknown = set(r) - set(names)
            if unknown:
                raise TypeError(f"generate_batch: unknown request fields {sorted(unknown)}")
            conds.append(self._conditioning(*[r[n] if n in r else defaults[n] for n in names]))
        r0 = requests[0]
        out = self._denoise(conds, r0.get("num_inference_steps", 40), r0.get("guidance_scale", 5.0), r0.get("ip_scale", 1.0))
        images = self._postprocess(out, output_type)
        res, off = [], 0
        for c in conds:
            res.append(images[off:off + c["n"]])
            off += c["n"]
        return res
