"""Oracle (test infrastructure only): the denoising loop of `DiffSenseiPipeline.__call__`
(reference src/pipelines/pipeline_diffsensei.py:310-337) on the CPU, fp32, over `UNetOracle`.

Also the `cpu_baseline` workload of bench.py: BASELINE.json configs[0] — 512x512, 20-step Euler, text-only
(`ip_images=[]`: the IP branch still runs on zeroed embeddings, reference :119-135), batch 1.
"""
from __future__ import annotations

import time
from typing import Callable, Optional

import torch

from oracle.scheduler_ref import DDIMOracle, EulerDiscreteOracle, cfg_combine
from oracle.unet_ref import UNetOracle

_id = lambda t: t
_h = lambda t: t.half().float()


def sample_loop(unet: UNetOracle, scheduler, latents: torch.Tensor, enc: torch.Tensor, text_embeds: torch.Tensor,
                time_ids: torch.Tensor, bbox: torch.Tensor, dialog_bbox: Optional[torch.Tensor], guidance_scale: float,
                num_steps: int, ip_scale: float, q: Callable = _id, max_steps: Optional[int] = None) -> torch.Tensor:
    """latents: [ns,4,H,W] already multiplied by init_noise_sigma; enc/text_embeds/time_ids/bbox/dialog_bbox are the
    CFG-concatenated ([neg..., pos...]) tensors of reference :294-303.  guidance_scale <= 1 (`do_classifier_free_guidance`
    False, reference :315, :333): the conditional rows only - the caller passes `num_samples` rows, and for `bbox` the
    FIRST `num_samples` rows of cat([negative_ip_bbox, ip_bbox]) like the reference's mask builder ends up using."""
    scheduler.set_timesteps(num_steps)
    unet.ip_scale = ip_scale
    x = q(latents.float())
    ar = latents.shape[-2] / latents.shape[-1]
    n = num_steps if max_steps is None else min(num_steps, max_steps)
    for i in range(n):
        t = float(scheduler.timesteps[i])
        do_cfg = guidance_scale > 1
        xin = torch.cat([x] * 2) if do_cfg else x                  # :315
        xin = q(scheduler.scale_model_input(xin, i))               # :317
        eps = unet.forward(xin, t, enc, text_embeds, time_ids, bbox, ar, dialog_bbox)   # :322-329
        if do_cfg:
            u, c = eps.chunk(2)
            e = q(u + q(guidance_scale * q(c - u)))                # :333-334 (fp16 tensor ops in the reference)
        else:
            e = eps
        x = q(scheduler.step(e, i, x))                             # :337
    return x


def time_cpu_baseline(cfg, sd, height: int = 512, width: int = 512, steps: int = 20, budget_s: float = 25.0,
                      threads: Optional[int] = None, keep_state: bool = False, min_steps: int = 1):
    """Time the oracle on a BOUNDED sample of configs[0]: as many of the 20 Euler steps as fit `budget_s`
    (at least `min_steps`), extrapolated to the full 20-step panel.  Returns dict(value=panels/s, ...).
    `keep_state`: also return, under "_state", the seeded inputs and the latents after the measured steps, so that the
    caller can run the product on the SAME inputs and compare (bench.py's `parity` object)."""
    if threads:
        torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    H, W = height // 8, width // 8
    ns = 1
    n_txt, n_ip, xdim = cfg.num_text_tokens, cfg.num_ip_tokens, cfg.cross_attention_dim
    enc = torch.randn(2 * ns, n_txt + n_ip, xdim, generator=g)
    pooled = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim
    text_embeds = torch.randn(2 * ns, pooled, generator=g)
    time_ids = torch.tensor([[height, width, 0, 0, height, width]] * (2 * ns), dtype=torch.float32)
    bbox = torch.zeros(2 * ns, cfg.max_num_ips, 4)
    unet = UNetOracle(cfg, sd)
    sch = EulerDiscreteOracle().set_timesteps(steps)
    lat = torch.randn(ns, 4, H, W, generator=g) * sch.init_noise_sigma
    done, t0 = 0, time.perf_counter()
    x = lat
    sch.set_timesteps(steps)
    unet.ip_scale = 0.6
    with torch.no_grad():
        for i in range(steps):
            xin = sch.scale_model_input(torch.cat([x] * 2), i)
            eps = unet.forward(xin, float(sch.timesteps[i]), enc, text_embeds, time_ids, bbox, H / W, None)
            x = sch.step(cfg_combine(eps, 7.5), i, x)
            done += 1
            if done >= min_steps and time.perf_counter() - t0 > budget_s:
                break
    dt = time.perf_counter() - t0
    per_step = dt / done
    out = {"value": 1.0 / (per_step * steps), "unit": "panels/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{done} of {steps} Euler steps of a {height}x{width} text-only panel (CFG batch 2), fp32 torch "
                     f"oracle, {dt:.1f} s measured, extrapolated to {steps} steps; VAE decode not included"}
    if keep_state:
        out["_state"] = {"enc": enc, "text_embeds": text_embeds, "time_ids": time_ids, "bbox": bbox, "latents0": lat,
                         "latents": x, "steps_done": done, "steps": steps, "ip_scale": 0.6, "guidance_scale": 7.5,
                         "height": height, "width": width}
    return out


# ------------------------------------------------------------------------------------------------------------------
# The whole `DiffSenseiPipeline.__call__` (reference src/pipelines/pipeline_diffsensei.py:180-372) on the CPU in fp32:
# tokenise -> both SDXL text encoders (:237-245; diffusers' SDXL `encode_prompt` [3P]: hidden_states[-2] of both,
# concatenated; pooled = text_encoder_2's projected output; the empty negative prompt forced to zeros) ->
# prepare_ip_image_embeds (:104-154: black padding, HF image processors, CLIP-H hidden_states[-2], ViT-MAE CLS, padded
# slots zeroed, Resampler on the embeddings and on zeros) -> prepare_dialog_bbox (:156-170) -> CFG concatenation
# (:294-309) -> the Euler loop (:310-337; `max_steps` = the reference's `interrupt` flag raised after that many steps:
# the remaining iterations `continue`) -> latents / scaling_factor -> VAE decode in fp32 (:339-361) ->
# image_processor.postprocess(output_type="pil") [3P]: (x / 2 + 0.5).clamp(0, 1), NHWC, (x * 255).round().astype(uint8).
# The encoder modules are the installed `transformers` classes themselves (third-party code on the reference's path).
def call_oracle(mods: dict, unet: UNetOracle, prompt: str, negative_prompt: Optional[str], height: int, width: int,
                num_inference_steps: int, guidance_scale: float, latents: torch.Tensor, ip_images=(), ip_bbox=(),
                ip_scale: float = 1.0, dialog_bbox=(), num_samples: int = 1, max_steps: Optional[int] = None,
                timings: Optional[dict] = None) -> dict:
    """mods: {"tokenizer", "tokenizer_2", "text_encoder", "text_encoder_2" (transformers CLIPTextModel /
    CLIPTextModelWithProjection, fp32, CPU), "image_encoder" (CLIPVisionModel), "magi" (ViTMAEModel), "resampler_sd",
    "resampler_heads", "resampler_dim_head", "vae_sd", "vae_cfg" (layers_per_block, norm_num_groups, eps,
    scaling_factor)}.  `latents`: [num_samples,4,H/8,W/8] unit-variance noise (what `prepare_latents` draws).
    Returns {"latents", "image" (fp32 [ns,3,H,W] in [-1,1]), "u8" (uint8 [ns,H,W,3]), "steps_done"}."""
    import numpy as np
    from PIL import Image
    from oracle.resampler_ref import resampler_forward
    from oracle.vae_ref import vae_decode
    tm = timings if timings is not None else {}
    clk = time.perf_counter
    cfg = unet.cfg
    ns = num_samples
    do_cfg = guidance_scale > 1
    with torch.no_grad():
        t0 = clk()
        # ---- encode_prompt
        def enc(text):
            hs, pooled = [], None
            for tok, te in ((mods["tokenizer"], mods["text_encoder"]), (mods["tokenizer_2"], mods["text_encoder_2"])):
                ids = tok(text, padding="max_length", max_length=tok.model_max_length, truncation=True,
                          return_tensors="pt").input_ids
                out = te(ids, output_hidden_states=True)
                pooled = out[0]
                hs.append(out.hidden_states[-2])
            return torch.cat(hs, dim=-1), pooled
        pe, pp = enc(prompt)
        if do_cfg and negative_prompt is None:
            ne, npool = torch.zeros_like(pe), torch.zeros_like(pp)        # force_zeros_for_empty_prompt
        else:
            ne, npool = enc(negative_prompt or "")
        tm["text_encoders_s"] = clk() - t0
        # ---- prepare_ip_image_embeds
        t0 = clk()
        n_max, nv = cfg.max_num_ips, cfg.num_vision_tokens
        imgs = list(ip_images)[:n_max]
        n_ip = len(imgs)
        imgs += [Image.new("RGB", (224, 224), (0, 0, 0))] * (n_max - n_ip)
        boxes = [list(b) for b in ip_bbox][:n_max]
        boxes += [[0.0] * 4] * (n_max - len(boxes))
        from transformers import CLIPImageProcessor, ViTImageProcessor
        clip_px = CLIPImageProcessor()(images=imgs, return_tensors="pt").pixel_values
        magi_px = ViTImageProcessor()(images=imgs, return_tensors="pt").pixel_values
        ce = mods["image_encoder"](clip_px, output_hidden_states=True).hidden_states[-2].unsqueeze(0)
        me = mods["magi"](magi_px).last_hidden_state[:, 0].unsqueeze(0)
        ce[0, n_ip:], me[0, n_ip:] = 0, 0
        rs = lambda a, b: resampler_forward(mods["resampler_sd"], a, b, mods["resampler_heads"], mods["resampler_dim_head"])
        img_e, neg_e = rs(ce, me), rs(torch.zeros_like(ce), torch.zeros_like(me))
        bbox_pos = torch.tensor(boxes, dtype=torch.float32).unsqueeze(0)
        tm["character_encoders_s"] = clk() - t0
        # ---- CFG batch ([negative rows, conditional rows]), dialog boxes in fp16 like the reference's `.to(unet.dtype)`
        d = [list(b) for b in dialog_bbox][:cfg.max_num_dialogs]
        d += [[0.0] * 4] * (cfg.max_num_dialogs - len(d))
        db_pos = torch.tensor(d, dtype=torch.float32).half().unsqueeze(0)
        rep = lambda t: t.repeat(ns, *([1] * (t.dim() - 1)))
        tid = torch.tensor([[height, width, 0, 0, height, width]], dtype=torch.float32)
        if do_cfg:
            enc_all = torch.cat([torch.cat([rep(ne), rep(pe)]), torch.cat([rep(neg_e), rep(img_e)])], dim=1)
            te_all = torch.cat([rep(npool), rep(pp)])
            tid_all = rep(tid).repeat(2, 1)
            bbox_all = torch.cat([torch.zeros_like(rep(bbox_pos)), rep(bbox_pos)])
            db_all = torch.cat([torch.zeros_like(rep(db_pos)), rep(db_pos)])
        else:
            enc_all, te_all, tid_all = torch.cat([rep(pe), rep(img_e)], dim=1), rep(pp), rep(tid)
            bbox_all, db_all = torch.zeros_like(rep(bbox_pos)), rep(db_pos)
        # ---- denoising loop
        t0 = clk()
        sch = EulerDiscreteOracle().set_timesteps(num_inference_steps)
        n = num_inference_steps if max_steps is None else min(num_inference_steps, max_steps)
        x = sample_loop(unet, EulerDiscreteOracle(), latents.float() * sch.init_noise_sigma, enc_all, te_all, tid_all,
                        bbox_all, db_all, guidance_scale, num_inference_steps, ip_scale, max_steps=n)
        tm["denoise_s"], tm["steps_done"] = clk() - t0, n
        # ---- decode + postprocess
        t0 = clk()
        vc = mods["vae_cfg"]
        image = vae_decode(mods["vae_sd"], x / vc["scaling_factor"], vc["layers_per_block"], vc["norm_num_groups"], vc["eps"])
        pix = (image / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
        u8 = (pix * 255).round().astype("uint8")
        tm["vae_postprocess_s"] = clk() - t0
    return {"latents": x, "image": image, "u8": np.ascontiguousarray(u8), "steps_done": n,
            "conditioning": {"enc": enc_all, "text_embeds": te_all, "time_ids": tid_all, "bbox": bbox_all, "dialog": db_all}}
