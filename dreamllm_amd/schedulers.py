"""DDPM / DDIM scheduler arithmetic -- restated from diffusers==0.24 `DDPMScheduler` / `DDIMScheduler` as the reference
uses them (omni/models/dreamllm/modeling_plugins.py:379-381,534-536,551,787-788,812,833; `_compute_snr` :468-491).
SD-2.1 scheduler_config: scaled_linear betas 0.00085 -> 0.012 over 1000 steps, steps_offset 1, clip_sample False,
set_alpha_to_one False, "leading" timestep spacing.  Host-side tables + tiny latent updates; the
denoising loop's per-step update runs as the fused HIP kernel `dllm_cfg_ddim_step` (see `DDIMScheduler.step_cfg_fused_`).
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch

from .utils import randn_tensor

SD21_SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                  prediction_type="epsilon", steps_offset=1, clip_sample=False, set_alpha_to_one=False,
                  timestep_spacing="leading", variance_type="fixed_small")


def _load_cfg(name_or_path):
    d = dict(SD21_SCHED)
    if isinstance(name_or_path, dict):
        d.update(name_or_path.get("scheduler", {}))
    elif isinstance(name_or_path, str):
        fp = os.path.join(name_or_path, "scheduler", "scheduler_config.json")
        if os.path.isfile(fp):
            with open(fp) as f:
                raw = json.load(f)
            d.update({k: raw[k] for k in d if k in raw})
    return d


class _SchedulerBase:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, **cfg):
        d = dict(SD21_SCHED, **cfg)
        self.config = SimpleNamespace(**d)
        T = d["num_train_timesteps"]
        if d["beta_schedule"] == "scaled_linear":
            betas = torch.linspace(d["beta_start"] ** 0.5, d["beta_end"] ** 0.5, T, dtype=torch.float32) ** 2
        elif d["beta_schedule"] == "linear":
            betas = torch.linspace(d["beta_start"], d["beta_end"], T, dtype=torch.float32)
        else:
            raise ValueError(f"unsupported beta_schedule {d['beta_schedule']}")
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if d["set_alpha_to_one"] else self.alphas_cumprod[0]
        self.timesteps = torch.arange(T - 1, -1, -1)
        self.num_inference_steps = None

    @classmethod
    def from_name_or_path(cls, name_or_path):
        return cls(**_load_cfg(name_or_path))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        """'leading' spacing: t_i = i * (T // N) + steps_offset, descending."""
        T = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        ratio = T // num_inference_steps
        ts = (torch.arange(0, num_inference_steps) * ratio).round().flip(0).long() + self.config.steps_offset
        self.timesteps = ts.to(device) if device is not None else ts

    def _ab(self, x, timesteps):
        ac = self.alphas_cumprod.to(device=x.device)
        a = ac[timesteps].float() ** 0.5
        b = (1 - ac[timesteps].float()) ** 0.5
        while a.dim() < x.dim():
            a, b = a.unsqueeze(-1), b.unsqueeze(-1)
        return a, b

    def add_noise(self, original_samples, noise, timesteps):
        """x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) eps   (modeling_plugins.py:534-536)."""
        a, b = self._ab(original_samples, timesteps)
        return (a * original_samples.float() + b * noise.float()).to(original_samples.dtype)

    def get_velocity(self, sample, noise, timesteps):
        """v = sqrt(abar_t) eps - sqrt(1 - abar_t) x_0   (modeling_plugins.py:551)."""
        a, b = self._ab(sample, timesteps)
        return (a * noise.float() - b * sample.float()).to(sample.dtype)

    def _x0_eps(self, model_output, t, sample):
        a_t = self.alphas_cumprod[t].item()
        if self.config.prediction_type == "epsilon":
            eps = model_output
            x0 = (sample - (1 - a_t) ** 0.5 * eps) / a_t**0.5
        elif self.config.prediction_type == "v_prediction":
            x0 = a_t**0.5 * sample - (1 - a_t) ** 0.5 * model_output
            eps = a_t**0.5 * model_output + (1 - a_t) ** 0.5 * sample
        else:
            raise ValueError(self.config.prediction_type)
        return x0, eps, a_t


class DDPMScheduler(_SchedulerBase):
    """Ancestral sampling step, variance_type fixed_small (the reference head's own scheduler, modeling_plugins.py:833)."""

    def step(self, model_output, t, sample, eta=0.0, generator=None):
        t = int(t)
        T = self.config.num_train_timesteps
        prev_t = t - T // (self.num_inference_steps or T)
        x0, _, a_t = self._x0_eps(model_output.float(), t, sample.float())
        a_prev = self.alphas_cumprod[prev_t].item() if prev_t >= 0 else 1.0
        cur_alpha = a_t / a_prev
        cur_beta = 1 - cur_alpha
        mean = (a_prev**0.5 * cur_beta) / (1 - a_t) * x0 + cur_alpha**0.5 * (1 - a_prev) / (1 - a_t) * sample.float()
        if t > 0:
            var = max((1 - a_prev) / (1 - a_t) * cur_beta, 1e-20)
            noise = randn_tensor(sample.shape, generator=generator, device=sample.device, dtype=torch.float32)
            mean = mean + var**0.5 * noise
        return mean.to(sample.dtype)


class DDIMScheduler(_SchedulerBase):
    """Deterministic DDIM (eta = 0) as installed by the benchmark (BASELINE config 3: "50 DDIM steps")."""

    def alphas_for(self, t):
        t = int(t)
        T = self.config.num_train_timesteps
        prev_t = t - T // self.num_inference_steps
        a_t = self.alphas_cumprod[t].item()
        a_prev = self.alphas_cumprod[prev_t].item() if prev_t >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_prev

    def step(self, model_output, t, sample, eta=0.0, generator=None):
        x0, eps, a_t = self._x0_eps(model_output.float(), int(t), sample.float())
        _, a_prev = self.alphas_for(t)
        var = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
        std = eta * var**0.5
        prev = a_prev**0.5 * x0 + max(1 - a_prev - std**2, 0.0) ** 0.5 * eps
        if eta > 0:
            prev = prev + std * randn_tensor(sample.shape, generator=generator, device=sample.device, dtype=torch.float32)
        return prev.to(sample.dtype)

    def step_cfg_fused_(self, noise_pred_nhwc, t, latents_nhwc, next_in, guidance_scale):
        """CFG combine + eta=0 update + next UNet input in ONE kernel (in place on the fp32 NHWC latents)."""
        from . import ops
        a_t, a_prev = self.alphas_for(t)
        return ops.cfg_ddim_step_(noise_pred_nhwc, latents_nhwc, next_in, guidance_scale, a_t, a_prev,
                                  self.config.prediction_type == "v_prediction")
