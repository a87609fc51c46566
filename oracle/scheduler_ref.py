"""Oracle (test infrastructure only): scheduler arithmetic the reference pipeline calls.

reference call sites: src/pipelines/pipeline_diffsensei.py:248-249 (set_timesteps), :317 (scale_model_input),
:333-334 (classifier-free guidance), :337 (step).  The schedulers themselves are diffusers classes [3P, not
vendored, unpinned]; restated here from their published algorithm with the SDXL-base scheduler config
(scaled_linear betas 0.00085->0.012, 1000 train steps, "leading" spacing, steps_offset 1, epsilon prediction).
numpy float64/float32 exactly where diffusers uses numpy, torch fp32 where it uses torch.
"""
from __future__ import annotations

import numpy as np
import torch


def _alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012) -> torch.Tensor:
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class EulerDiscreteOracle:
    """diffusers EulerDiscreteScheduler, s_churn=0 (deterministic Euler), final sigma 0."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        self.T = num_train_timesteps
        self.steps_offset = steps_offset
        self.alphas_cumprod = _alphas_cumprod(num_train_timesteps, beta_start, beta_end)

    def set_timesteps(self, n: int):
        step_ratio = self.T // n
        ts = (np.arange(0, n) * step_ratio).round()[::-1].copy().astype(np.float32)
        ts += self.steps_offset
        ac = self.alphas_cumprod.numpy()
        sig = np.array(((1 - ac) / ac) ** 0.5)
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps = ts
        self.init_noise_sigma = float((self.sigmas.max() ** 2 + 1) ** 0.5)  # "leading" spacing branch
        return self

    def scale_model_input(self, x: torch.Tensor, i: int) -> torch.Tensor:
        s = float(self.sigmas[i])
        return x / ((s ** 2 + 1) ** 0.5)

    def step(self, eps: torch.Tensor, i: int, x: torch.Tensor) -> torch.Tensor:
        s = torch.tensor(self.sigmas[i], dtype=torch.float32)
        s_next = torch.tensor(self.sigmas[i + 1], dtype=torch.float32)
        x32 = x.float()
        pred_x0 = x32 - s * eps.float()
        derivative = (x32 - pred_x0) / s
        return x32 + derivative * (s_next - s)


class DDIMOracle:
    """diffusers DDIMScheduler, eta=0, clip_sample False, set_alpha_to_one False."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        self.T = num_train_timesteps
        self.steps_offset = steps_offset
        self.alphas_cumprod = _alphas_cumprod(num_train_timesteps, beta_start, beta_end)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n: int):
        self.n = n
        step_ratio = self.T // n
        self.timesteps = (np.arange(0, n) * step_ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        return self

    def scale_model_input(self, x, i):
        return x

    def step(self, eps: torch.Tensor, i: int, x: torch.Tensor) -> torch.Tensor:
        t = int(self.timesteps[i])
        prev_t = t - self.T // self.n
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        x32, e32 = x.float(), eps.float()
        pred_x0 = (x32 - (1 - a_t) ** 0.5 * e32) / a_t ** 0.5
        return a_prev ** 0.5 * pred_x0 + (1 - a_prev) ** 0.5 * e32


def cfg_combine(noise_pred: torch.Tensor, guidance_scale: float) -> torch.Tensor:
    """reference src/pipelines/pipeline_diffsensei.py:333-334."""
    u, c = noise_pred.chunk(2)
    return u + guidance_scale * (c - u)
