"""TEST INFRASTRUCTURE (oracle) -- closed-form DDPM/DDIM arithmetic restated from diffusers==0.24, written
independently of dreamllm_amd/schedulers.py (plain loops over scalars) so the two can be checked against each other.
PARITY UNPINNED against diffusers itself (not installable here).  Call sites in the reference:
omni/models/dreamllm/modeling_plugins.py:534-536 (add_noise), :551 (get_velocity), :787-788 (set_timesteps), :833 (step)."""
import math

import torch


def alphas_cumprod(T=1000, b0=0.00085, b1=0.012):
    betas = [(math.sqrt(b0) + (math.sqrt(b1) - math.sqrt(b0)) * i / (T - 1)) ** 2 for i in range(T)]
    out, p = [], 1.0
    for b in betas:
        p *= 1.0 - b
        out.append(p)
    return out


def leading_timesteps(n, T=1000, offset=1):
    r = T // n
    return [i * r + offset for i in range(n)][::-1]


def add_noise(x0, eps, t, ac):
    return math.sqrt(ac[t]) * x0 + math.sqrt(1 - ac[t]) * eps


def velocity(x0, eps, t, ac):
    return math.sqrt(ac[t]) * eps - math.sqrt(1 - ac[t]) * x0


def ddim_step(eps, t, x, n, ac, T=1000):
    prev = t - T // n
    a_t, a_p = ac[t], (ac[prev] if prev >= 0 else ac[0])
    x0 = (x - math.sqrt(1 - a_t) * eps) / math.sqrt(a_t)
    return math.sqrt(a_p) * x0 + math.sqrt(1 - a_p) * eps


def cfg(eu, ec, s):
    return eu + s * (ec - eu)


def ddim_loop(unet_fn, latents, ctx_uncond, ctx_text, n_steps, guidance):
    """Reference denoising loop (modeling_plugins.py:809-839) with deterministic DDIM: returns final latents (fp32)."""
    ac = alphas_cumprod()
    x = latents.float()
    ctx = torch.cat([ctx_uncond, ctx_text])
    for t in leading_timesteps(n_steps):
        pred = unet_fn(torch.cat([x, x]), t, ctx).float()
        eu, ec = pred.chunk(2)
        x = ddim_step(cfg(eu, ec, guidance), t, x, n_steps, ac)
    return x
