"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of `diffusers.AutoencoderKL` encode/decode (diffusers==0.24) as the
reference calls it at omni/models/dreamllm/modeling_plugins.py:511-512 (encode -> latent_dist.sample * scaling_factor) and
:842 (decode).  PARITY UNPINNED (diffusers not installable here); diffusers key names.  NCHW, any float dtype."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _res(x, sd, p, g):
    h = F.silu(F.group_norm(x, g, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-6))
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, g, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-6))
    h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def _attn(x, sd, p, g):
    N, C, H, W = x.shape
    h = F.group_norm(x, g, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], 1e-6)
    h = h.view(N, C, H * W).transpose(1, 2)
    q = F.linear(h, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(h, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(h, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    w = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), dim=-1)
    o = F.linear(w @ v, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(N, C, H, W)


def _mid(x, sd, p, g):
    x = _res(x, sd, p + ".resnets.0", g)
    x = _attn(x, sd, p + ".attentions.0", g)
    return _res(x, sd, p + ".resnets.1", g)


def encode_moments(images, sd, cfg):
    g, boc, lpb = cfg["norm_num_groups"], cfg["block_out_channels"], cfg["layers_per_block"]
    x = F.conv2d(images, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for i in range(len(boc)):
        for j in range(lpb):
            x = _res(x, sd, f"encoder.down_blocks.{i}.resnets.{j}", g)
        if i < len(boc) - 1:
            x = F.pad(x, (0, 1, 0, 1))
            x = F.conv2d(x, sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"],
                         sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
    x = _mid(x, sd, "encoder.mid_block", g)
    x = F.silu(F.group_norm(x, g, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], 1e-6))
    x = F.conv2d(x, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return F.conv2d(x, sd["quant_conv.weight"], sd["quant_conv.bias"])


def sample_latents(moments, noise, scaling):
    mean, logvar = moments.float().chunk(2, dim=1)
    std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
    return (mean + std * noise) * scaling


def decode(z, sd, cfg):
    g, boc, lpb = cfg["norm_num_groups"], cfg["block_out_channels"], cfg["layers_per_block"]
    x = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = F.conv2d(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    x = _mid(x, sd, "decoder.mid_block", g)
    for i in range(len(boc)):
        for j in range(lpb + 1):
            x = _res(x, sd, f"decoder.up_blocks.{i}.resnets.{j}", g)
        if i < len(boc) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                         sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(F.group_norm(x, g, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], 1e-6))
    return F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def param_shapes(cfg):
    """name -> shape of every AutoencoderKL parameter (diffusers >= 0.18 key names) for the given config."""
    boc, lpb = tuple(cfg["block_out_channels"]), cfg["layers_per_block"]
    L, cin, cout = cfg.get("latent_channels", 4), cfg.get("in_channels", 3), cfg.get("out_channels", 3)
    sh = {}

    def conv(p, i, o, k):
        sh[p + ".weight"], sh[p + ".bias"] = (o, i, k, k), (o,)

    def lin(p, i, o):
        sh[p + ".weight"], sh[p + ".bias"] = (o, i), (o,)

    def norm(p, c):
        sh[p + ".weight"], sh[p + ".bias"] = (c,), (c,)

    def res(p, i, o):
        norm(p + ".norm1", i)
        conv(p + ".conv1", i, o, 3)
        norm(p + ".norm2", o)
        conv(p + ".conv2", o, o, 3)
        if i != o:
            conv(p + ".conv_shortcut", i, o, 1)

    def mid(p, c):
        res(p + ".resnets.0", c, c)
        norm(p + ".attentions.0.group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(p + ".attentions.0." + n, c, c)
        res(p + ".resnets.1", c, c)

    conv("encoder.conv_in", cin, boc[0], 3)
    c = boc[0]
    for i, o in enumerate(boc):
        for j in range(lpb):
            res(f"encoder.down_blocks.{i}.resnets.{j}", c if j == 0 else o, o)
        if i < len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", o, o, 3)
        c = o
    mid("encoder.mid_block", boc[-1])
    norm("encoder.conv_norm_out", boc[-1])
    conv("encoder.conv_out", boc[-1], 2 * L, 3)
    conv("quant_conv", 2 * L, 2 * L, 1)
    conv("post_quant_conv", L, L, 1)
    rev = boc[::-1]
    conv("decoder.conv_in", L, rev[0], 3)
    mid("decoder.mid_block", rev[0])
    c = rev[0]
    for i, o in enumerate(rev):
        for j in range(lpb + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", c if j == 0 else o, o)
        if i < len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", o, o, 3)
        c = o
    norm("decoder.conv_norm_out", boc[0])
    conv("decoder.conv_out", boc[0], cout, 3)
    return sh


def random_state_dict(cfg, seed=0, dtype=torch.float32, device="cpu"):
    """Seeded default-PyTorch-like init (uniform +-1/sqrt(fan_in)), norm weights around 1 -- reproducible on any host."""
    g = torch.Generator(device=device).manual_seed(seed)  # device="cuda": the full-size tests draw 0.9-2.6 G weights on the GPU
    sd = {}
    for k, s in param_shapes(cfg).items():
        if "norm" in k:
            t = (1.0 + 0.1 * torch.randn(s, generator=g, device=device)) if k.endswith("weight") else 0.1 * torch.randn(s, generator=g, device=device)
        elif k.endswith(".weight"):
            t = (torch.rand(s, generator=g, device=device) * 2 - 1) / math.sqrt(math.prod(s[1:]))
        else:
            t = (torch.rand(s, generator=g, device=device) * 2 - 1) * 0.05
        sd[k] = t.to(dtype)
    return sd
