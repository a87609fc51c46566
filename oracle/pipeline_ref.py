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
