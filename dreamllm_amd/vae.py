"""AutoencoderKL (SD VAE) encode/decode, forward only -- replaces `diffusers.AutoencoderKL` at
omni/models/dreamllm/modeling_plugins.py:375,511-512,842.  diffusers state_dict keys (`encoder.down_blocks.{i}.resnets.{j}.*`,
`encoder.mid_block.attentions.0.{group_norm,to_q,to_k,to_v,to_out.0}`, `quant_conv`, `post_quant_conv`, `decoder.*`).

SURVEY.md lists the VAE as "stock PyTorch for now / next row (f2)"; it already runs on this package's HIP kernels because
every piece exists: NHWC implicit-GEMM conv (incl. the encoder's asymmetric-pad stride-2 downsample and the decoder's
fused nearest-upsample conv), GroupNorm(+SiLU).  The single-head 512-wide mid-block attention (head_dim 512, outside the
flash kernel's 64/128) runs as GEMM -> row softmax (`dllm_softmax_rows`) -> GEMM (`ops.attention_wide_head`).  Frozen => no
backward.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import ops
from .unet import HipConv2d, HipGroupNorm, _Lin, _pad8
from .utils import logger

SD_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
              norm_num_groups=32, scaling_factor=0.18215, sample_size=512)
PRESET_SCALING = {"sdxl-base": 0.13025, "stabilityai/stable-diffusion-xl-base-1.0": 0.13025}


class _Cfg(SimpleNamespace):
    def to_dict(self):
        return dict(self.__dict__)


def load_vae_config(name_or_path_or_cfg):
    c = name_or_path_or_cfg
    d = dict(SD_VAE)
    if isinstance(c, dict):
        d.update(c.get("vae", {}))
    elif isinstance(c, str) and os.path.isfile(os.path.join(c, "vae", "config.json")):
        with open(os.path.join(c, "vae", "config.json")) as f:
            raw = json.load(f)
        d.update({k: raw[k] for k in d if k in raw})
    elif isinstance(c, str) and c in PRESET_SCALING:
        d["scaling_factor"] = PRESET_SCALING[c]
    d["block_out_channels"] = tuple(d["block_out_channels"])
    return _Cfg(**d)


class _Res(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = HipGroupNorm(groups, cin, 1e-6, act=True)
        self.conv1 = HipConv2d(cin, cout, 3)
        self.norm2 = HipGroupNorm(groups, cout, 1e-6, act=True)
        self.conv2 = HipConv2d(cout, cout, 3)
        self.conv_shortcut = HipConv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(self.norm1(x))
        sc = self.conv_shortcut(x) if self.conv_shortcut is not None else x
        return self.conv2(self.norm2(h), residual=sc)


class _Attn(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = HipGroupNorm(groups, c, 1e-6, act=False)
        self.to_q = _Lin(c, c)
        self.to_k = _Lin(c, c)
        self.to_v = _Lin(c, c)
        self.to_out = nn.ModuleList([_Lin(c, c), nn.Identity()])

    def forward(self, x):
        N, H, W, C = x.shape
        res = x.reshape(N, H * W, C)
        h = self.group_norm(x).reshape(N, H * W, C)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        o = ops.attention_wide_head(q, k, v)  # 1 head of width C = 512: two MFMA GEMMs around the row-softmax kernel
        return self.to_out[0](o, residual=res).reshape(N, H, W, C)


class _Sampler(nn.Module):
    def __init__(self, c, mode):
        super().__init__()
        self.conv = HipConv2d(c, c, 3, mode=mode)


class _Stage(nn.Module):
    def __init__(self, io, groups, sampler):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(i, o, groups) for i, o in io])
        if sampler == "down":
            self.downsamplers = nn.ModuleList([_Sampler(io[-1][1], "down_asym")])
        elif sampler == "up":
            self.upsamplers = nn.ModuleList([_Sampler(io[-1][1], "up")])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if hasattr(self, "downsamplers"):
            x = self.downsamplers[0].conv(x)
        if hasattr(self, "upsamplers"):
            x = self.upsamplers[0].conv(x)
        return x


class _Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([_Attn(c, groups)])
        self.resnets = nn.ModuleList([_Res(c, c, groups), _Res(c, c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = HipConv2d(cfg.in_channels, boc[0], 3)
        self.down_blocks = nn.ModuleList()
        c = boc[0]
        for i, o in enumerate(boc):
            io = [(c if j == 0 else o, o) for j in range(cfg.layers_per_block)]
            self.down_blocks.append(_Stage(io, g, "down" if i < len(boc) - 1 else None))
            c = o
        self.mid_block = _Mid(boc[-1], g)
        self.conv_norm_out = HipGroupNorm(g, boc[-1], 1e-6, act=True)
        self.conv_out = HipConv2d(boc[-1], 2 * cfg.latent_channels, 3)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(self.conv_norm_out(self.mid_block(x)))


class _Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        rev = boc[::-1]
        self.conv_in = HipConv2d(cfg.latent_channels, rev[0], 3)
        self.mid_block = _Mid(rev[0], g)
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, o in enumerate(rev):
            io = [(c if j == 0 else o, o) for j in range(cfg.layers_per_block + 1)]
            self.up_blocks.append(_Stage(io, g, "up" if i < len(rev) - 1 else None))
            c = o
        self.conv_norm_out = HipGroupNorm(g, boc[0], 1e-6, act=True)
        self.conv_out = HipConv2d(boc[0], cfg.out_channels, 3)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(self.conv_norm_out(x))


class DiagonalGaussian:
    """diffusers DiagonalGaussianDistribution on NCHW moments."""

    def __init__(self, moments):
        self.mean, logvar = torch.chunk(moments.float(), 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class AutoencoderKLLite(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.encoder = _Encoder(cfg)
        self.decoder = _Decoder(cfg)
        self.quant_conv = HipConv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
        self.post_quant_conv = HipConv2d(cfg.latent_channels, cfg.latent_channels, 1)

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype

    def load_pretrained(self, path, subfolder="vae"):
        if not isinstance(path, str):
            return False
        d = os.path.join(path, subfolder)
        for fn in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin"):
            fp = os.path.join(d, fn)
            if os.path.isfile(fp):
                if fn.endswith(".safetensors"):
                    from safetensors.torch import load_file
                    sd = load_file(fp)
                else:
                    sd = torch.load(fp, map_location="cpu")
                ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}  # pre-0.18 attention names
                sd = {".".join(ren.get(p, p) for p in k.split(".")): (v[..., 0, 0] if "attentions" in k and v.dim() == 4 else v)
                      for k, v in sd.items()}
                self.load_state_dict(sd, strict=True)
                logger.info(f"loaded VAE weights from {fp}")
                return True
        return False

    @staticmethod
    def _nhwc(x, cpad):
        x = x.permute(0, 2, 3, 1)
        if x.shape[-1] != cpad:
            x = torch.nn.functional.pad(x, (0, cpad - x.shape[-1]))
        return x.contiguous()

    @torch.no_grad()
    def encode(self, images):
        """images [N,3,H,W] in [-1,1] -> DiagonalGaussian over [N,4,H/8,W/8] (NCHW, fp32)."""
        x = self._nhwc(images.to(self.dtype), _pad8(self.config.in_channels))
        m = self.quant_conv(self.encoder(x))
        return DiagonalGaussian(m.permute(0, 3, 1, 2))

    @torch.no_grad()
    def decode(self, z):
        """z [N,4,h,w] -> images [N,3,8h,8w]."""
        x = self._nhwc(z.to(self.dtype), _pad8(self.config.latent_channels))
        x = self.post_quant_conv(x)
        x = torch.nn.functional.pad(x, (0, _pad8(x.shape[-1]) - x.shape[-1])) if x.shape[-1] % 8 else x
        return self.decoder(x).permute(0, 3, 1, 2)
