"""Scheduler objects with the diffusers protocol the reference pipeline drives
(`set_timesteps(n, device)`, `.timesteps`, `.init_noise_sigma`, `scale_model_input(x, t)`,
`step(eps, t, x, return_dict=False)[0]`; reference src/pipelines/pipeline_diffsensei.py:248-249, :317, :337).

Host side = the schedule tables only (a few hundred scalars computed once per `set_timesteps`, in numpy exactly
where diffusers uses numpy).  All per-element arithmetic (CFG combine, the update, the next step's input scaling)
runs in ONE HIP kernel (`ds_cfg_sampler_step_f16`) reading a per-step scalar table the engine indexes with a
device-side step counter — the reference issues 5+ elementwise launches per step here.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import ops

KIND_EULER, KIND_DDIM = 0, 1


def _alphas_cumprod(T: int, beta_start: float, beta_end: float) -> torch.Tensor:
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=torch.float32) ** 2  # scaled_linear
    return torch.cumprod(1.0 - betas, dim=0)


class _SchedulerBase:
    kind = -1
    order = 1
    _FIXED = {"trained_betas": (None,), "rescale_betas_zero_snr": (False,), "use_karras_sigmas": (False,),
              "use_exponential_sigmas": (False,), "use_beta_sigmas": (False,), "interpolation_type": ("linear",),
              "final_sigmas_type": ("zero",), "timestep_type": ("discrete",), "sigma_min": (None,), "sigma_max": (None,),
              "clip_sample": (False,), "set_alpha_to_one": (False,), "thresholding": (False,)}

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 steps_offset=1, timestep_spacing="leading", prediction_type="epsilon", **unused):
        if beta_schedule != "scaled_linear" or timestep_spacing != "leading" or prediction_type != "epsilon":
            raise NotImplementedError("only the SDXL scheduler configuration (scaled_linear / leading / epsilon)")
        # scheduler_config.json keys of diffusers' Euler / DDIM classes [3P] that CHANGE the sigma schedule or the update
        # rule: only the value the device kernel implements is accepted - anything else would sample on a different
        # schedule than the reference's scheduler without a word.  Keys that do not touch the arithmetic are ignored.
        for key, ok in self._FIXED.items():
            if key in unused and unused[key] not in ok:
                raise NotImplementedError(f"scheduler config {key}={unused[key]!r}: the MI355X sampler kernel implements "
                                          f"{key} in {ok} only")
        self.T = num_train_timesteps
        self.steps_offset = steps_offset
        self.alphas_cumprod = _alphas_cumprod(num_train_timesteps, beta_start, beta_end)
        self.timesteps = None
        self.num_inference_steps = None
        self._step_index = 0
        self._dev_table = None

    # -- table for the engine: rows [n_steps, 8] = {t, c_in_div, k0..k3, c_in_div_next, guidance}
    def coef_table(self, guidance_scale: float) -> np.ndarray:
        raise NotImplementedError

    def _table_on(self, device, guidance: float) -> torch.Tensor:
        return torch.from_numpy(self.coef_table(guidance)).to(device)

    def _index_of(self, t) -> int:
        tv = float(t)
        idx = np.nonzero(np.isclose(np.asarray(self.timesteps_np, dtype=np.float64), tv))[0]
        return int(idx[0]) if len(idx) else self._step_index

    # -- stand-alone protocol (one kernel launch each; the pipeline's fused loop does not go through these)
    def scale_model_input(self, sample: torch.Tensor, timestep) -> torch.Tensor:
        i = self._index_of(timestep)
        div = float(self.coef_table(1.0)[i, 1])
        if div == 1.0:
            return sample
        ns, c, h, w = sample.shape
        table = torch.tensor([[0, div, 0, 0, 0, 0, 1, 1]], dtype=torch.float32, device=sample.device)
        tmp = torch.empty((ns, h * w, c), dtype=torch.float16, device=sample.device)
        ops.prepare_model_input(sample.contiguous(), tmp, table, do_cfg=False)
        return ops.nhwc_to_nchw(tmp).reshape(ns, c, h, w)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True, **kw):
        """model_output: the (already CFG-combined) noise prediction, NCHW like `sample`."""
        i = self._index_of(timestep)
        row = self.coef_table(1.0)[i:i + 1].copy()
        table = torch.from_numpy(row).to(sample.device)
        ns, c, h, w = sample.shape
        eps = ops.nchw_to_nhwc(model_output.to(torch.float16).reshape(ns, c, h * w).contiguous())
        lat = sample.to(torch.float16).contiguous().clone()
        scratch = torch.empty((ns, h * w, c), dtype=torch.float16, device=sample.device)
        ops.cfg_sampler_step(eps, lat, scratch, table, self.kind, do_cfg=False)
        self._step_index = i + 1
        return (lat,) if not return_dict else {"prev_sample": lat}


class EulerDiscreteScheduler(_SchedulerBase):
    """diffusers EulerDiscreteScheduler [3P] (deterministic: s_churn = 0, final sigma 0, linear interpolation)."""
    kind = KIND_EULER

    def set_timesteps(self, num_inference_steps: int, device=None):
        n = num_inference_steps
        step_ratio = self.T // n
        ts = (np.arange(0, n) * step_ratio).round()[::-1].copy().astype(np.float32) + self.steps_offset
        ac = self.alphas_cumprod.numpy()
        sig = np.array(((1 - ac) / ac) ** 0.5)
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps_np = ts
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)
        self.num_inference_steps = n
        self.init_noise_sigma = float((self.sigmas.max() ** 2 + 1) ** 0.5)
        self._step_index = 0

    def coef_table(self, guidance_scale: float) -> np.ndarray:
        n = self.num_inference_steps
        tab = np.zeros((n, 8), dtype=np.float32)
        s = self.sigmas.astype(np.float32)
        div = ((s ** 2 + 1) ** 0.5).astype(np.float32)   # fp32 like the 0-dim sigma tensor arithmetic in diffusers
        tab[:, 0] = self.timesteps_np
        tab[:, 1] = div[:n]
        tab[:, 2] = s[:n]
        tab[:, 3] = s[1:n + 1]
        tab[:, 6] = div[1:n + 1]
        tab[:, 7] = guidance_scale
        return tab


class DDIMScheduler(_SchedulerBase):
    """diffusers DDIMScheduler [3P], eta = 0, clip_sample False, set_alpha_to_one False."""
    kind = KIND_DDIM
    init_noise_sigma = 1.0

    def set_timesteps(self, num_inference_steps: int, device=None):
        n = num_inference_steps
        step_ratio = self.T // n
        ts = (np.arange(0, n) * step_ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps_np = ts
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)
        self.num_inference_steps = n
        self._step_index = 0

    def coef_table(self, guidance_scale: float) -> np.ndarray:
        n = self.num_inference_steps
        tab = np.zeros((n, 8), dtype=np.float32)
        ac = self.alphas_cumprod.numpy().astype(np.float32)
        for i, t in enumerate(self.timesteps_np):
            prev = int(t) - self.T // n
            a_t = ac[int(t)]
            a_p = ac[prev] if prev >= 0 else ac[0]
            tab[i] = [float(t), 1.0, a_t ** 0.5, (1 - a_t) ** 0.5, a_p ** 0.5, (1 - a_p) ** 0.5, 1.0, guidance_scale]
        return tab
