"""TEST INFRASTRUCTURE (oracle) -- duck-typed stand-ins for the three diffusers==0.24 objects the reference's
`StableDiffusionHead` / `StableDiffusionXLHead` hold (`AutoencoderKL`, `UNet2DConditionModel`, `DDPMScheduler`;
omni/models/dreamllm/modeling_plugins.py:375-381), plus `VaeImageProcessor.postprocess` (:847) and a recorder for the
random draws of the wrapper.

diffusers is neither vendored in /root/reference nor installable here, so the reference head cannot be *constructed*; but
its `forward` (:493-577), `pipeline` (:671-850), `_compute_snr` (:468-491), `_rescale_noise_cfg` (:658-669) and the SDXL
overrides (omni/models/dreamllm_sdxl/modeling_plugins.py:151-236,239-445) only *call* those objects.  `oracle/make_golden_sdhead.py`
builds the real reference classes with `__new__`, attaches these adapters (which delegate the arithmetic to the restated
`oracle/{unet,vae,sched}_ref.py`), and EXECUTES the reference wrappers.  That pins the wrapper logic by execution; the
third-party UNet/VAE/scheduler arithmetic underneath stays a restatement (PARITY UNPINNED against diffusers itself).

The adapters expose exactly the attribute / call surface the reference touches, with diffusers' signatures (the wrapper
inspects `scheduler.step`'s signature for `eta` / `generator`, :625-641).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
from torch import nn

from . import sched_ref, unet_ref, vae_ref


class _SdModule(nn.Module):
    """nn.Module over a plain {name: tensor} dict with dotted diffusers names; `.to()` / `.float()` map over the dict."""

    def __init__(self, sd):
        super().__init__()
        self._sd = dict(sd)
        self._anchor = nn.Parameter(torch.zeros(1), requires_grad=False)  # gives get_model_device/dtype something to find

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self._sd = {k_: fn(v) for k_, v in self._sd.items()}
        return self

    @property
    def dtype(self):
        return next(iter(self._sd.values())).dtype


class _LatentDist:
    def __init__(self, moments):
        self.moments = moments

    def sample(self, generator=None):
        """diffusers DiagonalGaussianDistribution.sample: mean + std * randn(mean.shape) in the moments' dtype."""
        mean, logvar = torch.chunk(self.moments, 2, dim=1)
        logvar = torch.clamp(logvar, -30.0, 20.0)
        std = torch.exp(0.5 * logvar)
        noise = torch.randn(mean.shape, generator=generator, device=mean.device, dtype=mean.dtype)
        return mean + std * noise


class DuckVAE(_SdModule):
    """AutoencoderKL surface used by the reference: .config.{scaling_factor,block_out_channels}, .encode(x).latent_dist.sample(),
    .decode(z, return_dict=False)[0]."""

    def __init__(self, cfg, sd):
        super().__init__(sd)
        self._cfg = dict(cfg)
        self.config = SimpleNamespace(**cfg)

    def encode(self, images):
        return SimpleNamespace(latent_dist=_LatentDist(vae_ref.encode_moments(images, self._sd, self._cfg)))

    def decode(self, z, return_dict=True):
        img = vae_ref.decode(z, self._sd, self._cfg)
        return (img,) if not return_dict else SimpleNamespace(sample=img)


class DuckUNet(_SdModule):
    """UNet2DConditionModel surface: .config.{in_channels,sample_size,cross_attention_dim}; call with diffusers' kwargs."""

    def __init__(self, cfg, sd):
        super().__init__(sd)
        self._cfg = dict(cfg)
        self.config = SimpleNamespace(**cfg)
        self.calls = []  # (timestep, batch) of every call: lets the golden script assert the loop structure

    def forward(self, sample, timestep, encoder_hidden_states=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                return_dict=True):
        t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
        self.calls.append((t.detach().reshape(-1).tolist(), sample.shape[0]))
        out = unet_ref.unet_forward(sample, t, encoder_hidden_states, self._sd, self._cfg, added_cond_kwargs=added_cond_kwargs)
        return (out,) if not return_dict else SimpleNamespace(sample=out)


class _DuckSchedulerBase:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, prediction_type="epsilon", num_train_timesteps=1000, steps_offset=1):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, prediction_type=prediction_type,
                                      steps_offset=steps_offset)
        self._ac = sched_ref.alphas_cumprod(num_train_timesteps)
        self.alphas_cumprod = torch.tensor(self._ac, dtype=torch.float32)  # read by `_compute_snr` (:473)
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)
        self.num_inference_steps = None

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        self.timesteps = torch.tensor(sched_ref.leading_timesteps(num_inference_steps, self.config.num_train_timesteps,
                                                                  self.config.steps_offset), dtype=torch.int64)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        return torch.stack([sched_ref.add_noise(original_samples[i], noise[i], int(timesteps[i]), self._ac)
                            for i in range(original_samples.shape[0])]).to(original_samples.dtype)

    def get_velocity(self, sample, noise, timesteps):
        return torch.stack([sched_ref.velocity(sample[i], noise[i], int(timesteps[i]), self._ac)
                            for i in range(sample.shape[0])]).to(sample.dtype)

    def _x0_eps(self, out, t, x):
        a = self._ac[t]
        if self.config.prediction_type == "epsilon":
            return (x - math.sqrt(1 - a) * out) / math.sqrt(a), out
        if self.config.prediction_type == "v_prediction":
            return math.sqrt(a) * x - math.sqrt(1 - a) * out, math.sqrt(a) * out + math.sqrt(1 - a) * x
        raise ValueError(self.config.prediction_type)


class DuckDDPMScheduler(_DuckSchedulerBase):
    """diffusers DDPMScheduler.step (variance_type fixed_small, clip_sample False): NO `eta` parameter, has `generator`."""

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        t = int(timestep)
        T = self.config.num_train_timesteps
        prev = t - T // (self.num_inference_steps or T)
        a_t = self._ac[t]
        a_p = self._ac[prev] if prev >= 0 else 1.0
        x0, _ = self._x0_eps(model_output, t, sample)
        cur_alpha = a_t / a_p
        cur_beta = 1 - cur_alpha
        mean = math.sqrt(a_p) * cur_beta / (1 - a_t) * x0 + math.sqrt(cur_alpha) * (1 - a_p) / (1 - a_t) * sample
        if t > 0:
            var = max((1 - a_p) / (1 - a_t) * cur_beta, 1e-20)
            noise = torch.randn(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            mean = mean + math.sqrt(var) * noise
        return (mean,) if not return_dict else SimpleNamespace(prev_sample=mean)


class DuckDDIMScheduler(_DuckSchedulerBase):
    """diffusers DDIMScheduler.step signature (eta, use_clipped_model_output, generator, variance_noise)."""

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None, variance_noise=None,
             return_dict=True):
        t = int(timestep)
        T = self.config.num_train_timesteps
        prev = t - T // self.num_inference_steps
        a_t = self._ac[t]
        a_p = self._ac[prev] if prev >= 0 else self._ac[0]  # set_alpha_to_one False
        x0, eps = self._x0_eps(model_output, t, sample)
        var = (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)
        std = eta * math.sqrt(var)
        out = math.sqrt(a_p) * x0 + math.sqrt(max(1 - a_p - std**2, 0.0)) * eps
        if eta > 0:
            out = out + std * torch.randn(model_output.shape, generator=generator, device=model_output.device,
                                          dtype=model_output.dtype)
        return (out,) if not return_dict else SimpleNamespace(prev_sample=out)


class DuckImageProcessor:
    """VaeImageProcessor.postprocess for output_type in {"latent", "pt"} (do_normalize default True)."""

    def postprocess(self, image, output_type="pil", do_denormalize=None):
        if output_type == "latent":
            return image
        image = torch.stack([(im / 2 + 0.5).clamp(0, 1) if (do_denormalize is None or do_denormalize[i]) else im
                             for i, im in enumerate(image)])
        if output_type == "pt":
            return image
        raise NotImplementedError(output_type)


class DrawLog:
    """Context manager that records (or replays) every `torch.randn / randn_like / randint / bernoulli` result produced inside
    it, in call order.  Recording runs pin WHICH random numbers the reference wrapper consumed and in which order; the HIP
    head replays them (`dreamllm_amd.utils.replay_draws`) so that the two sides see identical noise."""

    KINDS = ("randn", "randn_like", "randint", "bernoulli")

    def __init__(self, replay=None):
        self.draws = []
        self._replay = list(replay) if replay is not None else None
        self._orig = {}

    def _wrap(self, kind):
        orig = self._orig[kind]

        def fn(*a, **k):
            out = orig(*a, **k)
            if self._replay is not None:
                rk, rt = self._replay.pop(0)
                assert rk == kind and tuple(rt.shape) == tuple(out.shape), (rk, kind, rt.shape, out.shape)
                out = rt.to(dtype=out.dtype, device=out.device)
            self.draws.append((kind, out.detach().clone()))
            return out

        return fn

    def __enter__(self):
        for kind in self.KINDS:
            self._orig[kind] = getattr(torch, kind)
            setattr(torch, kind, self._wrap(kind))
        return self

    def __exit__(self, *exc):
        for kind, fn in self._orig.items():
            setattr(torch, kind, fn)
        if exc[0] is None and self._replay is not None:
            assert not self._replay, f"{len(self._replay)} recorded draws were not consumed"
        return False
