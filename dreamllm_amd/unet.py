"""Stable-Diffusion UNet (SD-2.1 and SDXL layouts) on HIP kernels -- replaces `diffusers.UNet2DConditionModel` at
omni/models/dreamllm/modeling_plugins.py:375-377,556,815-821 and omni/models/dreamllm_sdxl/modeling_plugins.py:215,406-413.

* Parameters keep diffusers' names and shapes (`down_blocks.{i}.resnets.{j}.conv1.weight` [CO,CI,3,3], ...), so real
  `unet/diffusion_pytorch_model` checkpoints load by key.
* Activations are NHWC bf16 end to end: a pixel's channels are contiguous, which is what the implicit-GEMM 3x3 conv
  (k = (tap, ci), ci contiguous), GroupNorm's 16-byte channel vectors and the attention token view [N, HW, C] all want --
  the NCHW<->[N,HW,C] permutes of Transformer2DModel disappear.
* Weights are frozen (every dreamllm recipe): each op implements forward + input gradient only; re-laid-out weight copies
  ([CO, 9*CI] forward, flipped [CI, 9*CO] for the input gradient) are cached per device.
* ResnetBlock2D: GroupNorm+SiLU (3 HBM-bound launches) -> conv1 with bias + per-image time-embedding in the epilogue
  -> GroupNorm+SiLU -> conv2 with bias + shortcut residual in the epilogue.  Upsample2D is fused into its conv's gather.
* Transformer: q/k/v as ONE GEMM for self-attention, k/v of the 64 dream tokens as one GEMM (cacheable across the
  denoising loop: the context does not change between steps), flash attention head_dim 64, to_out/ff.net.2 with bias +
  residual epilogues, GEGLU as one elementwise pass over the ff.net.0 GEMM output (training) or inside that GEMM's epilogue
  (no_grad: `ops.linear_geglu`, round 4).
"""
from __future__ import annotations

import json
import math
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import ops
from .utils import logger

SD21_BASE = dict(
    sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    attention_head_dim=(5, 10, 20, 20), transformer_layers_per_block=1, cross_attention_dim=1024, norm_num_groups=32,
    norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0, addition_embed_type=None, addition_time_embed_dim=None,
    projection_class_embeddings_input_dim=None,
)
SDXL_BASE = dict(
    SD21_BASE, sample_size=128, block_out_channels=(320, 640, 1280),
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"), attention_head_dim=(5, 10, 20),
    transformer_layers_per_block=(1, 2, 10), cross_attention_dim=2048, addition_embed_type="text_time",
    addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816,
)
PRESETS = {"sd21-base": SD21_BASE, "stabilityai/stable-diffusion-2-1-base": SD21_BASE, "sdxl-base": SDXL_BASE,
           "stabilityai/stable-diffusion-xl-base-1.0": SDXL_BASE}


class _Cfg(SimpleNamespace):
    def to_dict(self):
        return dict(self.__dict__)


def load_unet_config(name_or_path_or_cfg):
    c = name_or_path_or_cfg
    if isinstance(c, dict) and "unet" in c:
        c = c["unet"]
    if isinstance(c, dict):
        d = dict(SD21_BASE, **c)
    elif isinstance(c, str) and os.path.isfile(os.path.join(c, "unet", "config.json")):
        with open(os.path.join(c, "unet", "config.json")) as f:
            raw = json.load(f)
        d = dict(SD21_BASE, **{k: raw[k] for k in SD21_BASE if k in raw})
    elif isinstance(c, str):
        d = dict(PRESETS.get(c, SD21_BASE))
    else:
        raise TypeError(f"cannot derive a UNet config from {type(c)}")
    for k in ("block_out_channels", "down_block_types", "up_block_types"):
        d[k] = tuple(d[k])
    return _Cfg(**d)


def _tuple(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def _pad8(c):
    return (c + 7) // 8 * 8


class HipConv2d(nn.Module):
    """Conv2d parameters in diffusers layout; NHWC implicit-GEMM execution; frozen (input gradient only)."""

    def __init__(self, cin, cout, k, mode="same"):
        super().__init__()
        self.cin, self.cout, self.k, self.mode = cin, cout, k, mode
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        bound = 1.0 / math.sqrt(cin * k * k)
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)
        self._cache = None

    def _weights(self):
        w = self.weight
        key = (w.device, w.dtype, w._version, w.data_ptr())
        if self._cache is None or self._cache[0] != key:
            wd = w.detach()
            cin_p, cout_p = _pad8(self.cin), _pad8(self.cout)
            wf = wd.permute(0, 2, 3, 1)  # [CO, KH, KW, CI]
            if cin_p != self.cin:
                wf = torch.nn.functional.pad(wf, (0, cin_p - self.cin))
            wf = wf.reshape(self.cout, -1).contiguous()
            wb = wd.flip(2, 3).permute(1, 2, 3, 0)  # [CI, KH', KW', CO]
            if cout_p != self.cout:
                wb = torch.nn.functional.pad(wb, (0, cout_p - self.cout))
            wb = wb.reshape(self.cin, -1).contiguous()
            self._cache = (key, wf, wb)
        return self._cache[1], self._cache[2]

    def forward(self, x, residual=None, image_bias=None):
        """x [N,H,W,CI] (CI already padded to a multiple of 8 if needed) -> [N,OH,OW,CO]."""
        wf, wb = self._weights()
        return ops.ConvFn.apply(x, wf, wb, self.bias, residual, image_bias, self.cout, self.k, self.mode)


class HipGroupNorm(nn.GroupNorm):
    def __init__(self, groups, channels, eps, act=False):
        super().__init__(groups, channels, eps=eps, affine=True)
        self.act = act

    def forward(self, x):
        return ops.groupnorm(x, self.weight, self.bias, self.num_groups, self.eps, self.act)


class _Lin(nn.Linear):
    def forward(self, x, residual=None):
        return ops.linear(x, self.weight, self.bias, residual)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb, groups, eps):
        super().__init__()
        self.norm1 = HipGroupNorm(groups, cin, eps, act=True)
        self.conv1 = HipConv2d(cin, cout, 3)
        self.time_emb_proj = _Lin(temb, cout)
        self.norm2 = HipGroupNorm(groups, cout, eps, act=True)
        self.conv2 = HipConv2d(cout, cout, 3)
        self.conv_shortcut = HipConv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, emb_act, time_bias=None):
        """emb_act = SiLU(emb), shared by every ResBlock of the step.  `time_bias` [N, cout]: this block's `time_emb_proj(emb_act)`
        computed ahead of the denoising loop (`HipUNet2DConditionModel.precompute_time_bias`); it does not depend on the latents."""
        t = time_bias if time_bias is not None else self.time_emb_proj(emb_act)
        h = self.conv1(self.norm1(x), image_bias=t)
        sc = self.conv_shortcut(x) if self.conv_shortcut is not None else x
        return self.conv2(self.norm2(h), residual=sc)


class Attention(nn.Module):
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(ctx_dim, dim, bias=False)
        self.to_v = nn.Linear(ctx_dim, dim, bias=False)
        self.to_out = nn.ModuleList([_Lin(dim, dim), nn.Identity()])
        self.is_self = ctx_dim == dim
        self._cache = None

    def _fused(self):
        ws = (self.to_q.weight, self.to_k.weight, self.to_v.weight)
        key = tuple((w.device, w.dtype, w._version, w.data_ptr()) for w in ws)
        if self._cache is None or self._cache[0] != key:
            kv = torch.cat([ws[1], ws[2]], 0).detach().contiguous()
            qkv = torch.cat(ws, 0).detach().contiguous() if self.is_self else None
            self._cache = (key, kv, qkv)
        return self._cache[1], self._cache[2]

    def project_context(self, ctx):
        """K/V of the conditioning tokens: [N, L, 2, heads, 64] (constant over the denoising loop)."""
        kv, _ = self._fused()
        N, L, _ = ctx.shape
        return ops.linear(ctx, kv).view(N, L, 2, self.heads, -1)

    def forward(self, x, ctx=None, residual=None, kv_cache=None):
        N, S, C = x.shape
        hd = C // self.heads
        if ctx is None:
            _, qkv = self._fused()
            o = ops.packed_self_attn(ops.linear(x, qkv).view(N, S, 3, self.heads, hd), False, hd**-0.5)
        else:
            q = ops.linear(x, self.to_q.weight).view(N, S, self.heads, hd)
            kvt = kv_cache if kv_cache is not None else self.project_context(ctx)
            o = ops.packed_cross_attn(q, kvt, hd**-0.5)
        return self.to_out[0](o.reshape(N, S, C), residual=residual)


GEGLU_FUSED_MAX_ROWS = 1 << 30


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = _Lin(dim, inner * 2)

    def forward(self, x):
        # inference / denoising loop: projection + GEGLU in one launch.  In the loop's graph the fused form wins at every level but
        # one, where it ties (UNet batch 16: 64 x 64 level 286 us against 243 + ~120 us for GEMM + element-wise pass; 32 x 32 level
        # 213 against 172 + ~40 us; profiles/r04_unet_gemm_b16.log, r04_denoise_b8_launch_table.txt).
        # (forward-only: taken only when NOTHING it touches needs a gradient -- a trainable `proj` behind a frozen input would otherwise
        # silently receive none)
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or self.proj.weight.requires_grad or
                                                  (self.proj.bias is not None and self.proj.bias.requires_grad))
        if not needs_grad and x.numel() // x.shape[-1] <= GEGLU_FUSED_MAX_ROWS:
            y = ops.linear_geglu(x, self.proj.weight, self.proj.bias)
            if y is not None:
                return y
        return ops.geglu_packed(self.proj(x))  # hidden * gelu(gate); hidden, gate = chunk(2)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), _Lin(dim * 4, dim)])

    def forward(self, x, residual):
        return self.net[2](self.net[0](x), residual=residual)


class _LN(nn.LayerNorm):
    def forward(self, x):
        return ops.layernorm(x, self.weight, self.bias, self.eps)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, ctx_dim):
        super().__init__()
        self.norm1 = _LN(dim, eps=1e-5)
        self.attn1 = Attention(dim, dim, heads)
        self.norm2 = _LN(dim, eps=1e-5)
        self.attn2 = Attention(dim, ctx_dim, heads)
        self.attn2.is_self = False
        self.norm3 = _LN(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx, kv_cache=None):
        x = self.attn1(self.norm1(x), residual=x)
        x = self.attn2(self.norm2(x), ctx=ctx, residual=x, kv_cache=kv_cache)
        return self.ff(self.norm3(x), residual=x)


class Transformer2DModel(nn.Module):
    def __init__(self, dim, heads, depth, ctx_dim, groups):
        super().__init__()
        self.norm = HipGroupNorm(groups, dim, 1e-6, act=False)
        self.proj_in = _Lin(dim, dim)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, ctx_dim) for _ in range(depth)])
        self.proj_out = _Lin(dim, dim)

    def forward(self, x, ctx, kv_caches=None):
        N, H, W, C = x.shape
        res = x.reshape(N, H * W, C)
        h = self.proj_in(self.norm(x).reshape(N, H * W, C))
        for d, blk in enumerate(self.transformer_blocks):
            h = blk(h, ctx, None if kv_caches is None else kv_caches[d])
        return self.proj_out(h, residual=res).reshape(N, H, W, C)


class _Sampler(nn.Module):
    def __init__(self, c, mode):
        super().__init__()
        self.conv = HipConv2d(c, c, 3, mode=mode)

    def forward(self, x):
        return self.conv(x)


class _Block(nn.Module):
    """Down / mid / up block: resnets [+ attentions] [+ one down/up-sampler]."""

    def __init__(self, res_io, attn_dim, heads, depth, ctx_dim, temb, groups, eps, sampler=None):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(i, o, temb, groups, eps) for i, o in res_io])
        if attn_dim is not None:
            n_attn = 1 if sampler == "mid" else len(res_io)
            self.attentions = nn.ModuleList([Transformer2DModel(attn_dim, heads, depth, ctx_dim, groups)
                                             for _ in range(n_attn)])
        if sampler == "down":
            self.downsamplers = nn.ModuleList([_Sampler(res_io[-1][1], "down")])
        elif sampler == "up":
            self.upsamplers = nn.ModuleList([_Sampler(res_io[-1][1], "up")])


class _TimeEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = _Lin(cin, dim)
        self.linear_2 = _Lin(dim, dim)

    def forward(self, x):
        return self.linear_2(ops.silu(self.linear_1(x)))


def timestep_embedding(timesteps, dim, flip_sin_to_cos=True, freq_shift=0, max_period=10000):
    """diffusers get_timestep_embedding: a [N, dim] table computed on device from N scalars (host-trivial arithmetic)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / (half - freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class HipUNet2DConditionModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        boc = cfg.block_out_channels
        nb = len(boc)
        heads = _tuple(cfg.attention_head_dim, nb)
        depth = _tuple(cfg.transformer_layers_per_block, nb)
        lpb, cd, groups, eps = cfg.layers_per_block, cfg.cross_attention_dim, cfg.norm_num_groups, cfg.norm_eps
        temb = boc[0] * 4
        for i, h in enumerate(heads):
            if boc[i] // h != 64:
                raise ValueError("UNet attention head_dim must be 64")
        self.time_embedding = _TimeEmbedding(boc[0], temb)
        if cfg.addition_embed_type == "text_time":
            self.add_embedding = _TimeEmbedding(cfg.projection_class_embeddings_input_dim, temb)
        self.conv_in = HipConv2d(cfg.in_channels, boc[0], 3)
        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i, bt in enumerate(cfg.down_block_types):
            in_c, out_c = out_c, boc[i]
            io = [(in_c if j == 0 else out_c, out_c) for j in range(lpb)]
            self.down_blocks.append(_Block(io, out_c if bt.startswith("CrossAttn") else None, heads[i], depth[i], cd, temb,
                                           groups, eps, "down" if i < nb - 1 else None))
        self.mid_block = _Block([(boc[-1], boc[-1])] * 2, boc[-1], heads[-1], depth[-1], cd, temb, groups, eps, "mid")
        self.up_blocks = nn.ModuleList()
        rev, rheads, rdepth = boc[::-1], heads[::-1], depth[::-1]
        out_c = rev[0]
        for i, bt in enumerate(cfg.up_block_types):
            prev, out_c, in_c = out_c, rev[i], rev[min(i + 1, nb - 1)]
            io = [((prev if j == 0 else out_c) + (in_c if j == lpb else out_c), out_c) for j in range(lpb + 1)]
            self.up_blocks.append(_Block(io, out_c if bt.startswith("CrossAttn") else None, rheads[i], rdepth[i], cd, temb,
                                         groups, eps, "up" if i < nb - 1 else None))
        self.conv_norm_out = HipGroupNorm(groups, boc[0], eps, act=True)
        self.conv_out = HipConv2d(boc[0], cfg.out_channels, 3)

    @property
    def device(self):
        return self.conv_in.weight.device

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def load_pretrained(self, path, subfolder="unet"):
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
                self.load_state_dict(sd, strict=True)
                logger.info(f"loaded UNet weights from {fp}")
                return True
        return False

    # ---- helpers -----------------------------------------------------------------------------------------------
    def _attn_modules(self):
        for blk in list(self.down_blocks) + [self.mid_block] + list(self.up_blocks):
            if hasattr(blk, "attentions"):
                for t in blk.attentions:
                    yield t

    @torch.no_grad()
    def prepare_context(self, encoder_hidden_states):
        """Project the conditioning tokens through every cross-attention's to_k/to_v ONCE for a whole denoising loop."""
        ctx = encoder_hidden_states.to(self.dtype)
        return {id(t): [b.attn2.project_context(ctx) for b in t.transformer_blocks] for t in self._attn_modules()}

    def _resnets(self):
        """Every ResnetBlock2D in execution order (the layout of the precomputed time-bias buffer)."""
        out = []
        for blk in self.down_blocks:
            out += list(blk.resnets)
        out += list(self.mid_block.resnets)
        for blk in self.up_blocks:
            out += list(blk.resnets)
        return out

    def _embedding(self, timesteps, N, added_cond_kwargs=None):
        """emb = time_embedding(sinusoid(t)) [+ add_embedding(text_embeds | sinusoid(time_ids))] for N samples."""
        cfg = self.config
        dev = timesteps.device
        t_emb = timestep_embedding(timesteps, cfg.block_out_channels[0], cfg.flip_sin_to_cos, cfg.freq_shift).to(self.dtype)
        emb = self.time_embedding(t_emb)
        if cfg.addition_embed_type == "text_time":
            te = timestep_embedding(added_cond_kwargs["time_ids"].flatten().to(dev), cfg.addition_time_embed_dim,
                                    cfg.flip_sin_to_cos, cfg.freq_shift).reshape(N, -1).to(self.dtype)
            add = torch.cat([added_cond_kwargs["text_embeds"].to(self.dtype), te], dim=-1)
            emb = ops.add(emb, self.add_embedding(add))
        return emb

    def time_bias_layout(self, N):
        """[(offset, cout)] of every ResBlock's [N, cout] slice in the flat time-bias buffer, and its total length."""
        offs, off = [], 0
        for r in self._resnets():
            c = r.time_emb_proj.out_features
            offs.append((off, c))
            off += N * c
        return offs, off

    @torch.no_grad()
    def precompute_time_bias(self, timesteps, N, added_cond_kwargs=None):
        """The part of the UNet that depends on the TIMESTEP only -- sinusoid, time-embedding MLP, SiLU and every ResBlock's
        `time_emb_proj` (~38 tiny launches per denoising step at SD-2.1 dims) -- for ALL steps of a loop at once: one GEMM per
        ResBlock over M = len(timesteps) rows instead of one per step over M = N.  -> [S, total] bf16 table whose row s is the flat
        buffer `forward(time_bias=...)` reads (`time_bias_layout`): per block [N, cout], the N images sharing the step's timestep.
        Reference loop: modeling_plugins.py:809-821 (diffusers recomputes the embedding inside every UNet call)."""
        ts = torch.as_tensor(timesteps, device=self.device).reshape(-1)
        S = ts.numel()
        if self.config.addition_embed_type == "text_time":
            # the micro-conditioning rows differ per image: embed (step, image) pairs
            emb = self._embedding(ts.repeat_interleave(N), S * N,
                                  {k: v.repeat(S, *([1] * (v.dim() - 1))) for k, v in added_cond_kwargs.items()})   # [S*N, temb]
        else:
            emb = self._embedding(ts, S)                                                                                  # [S, temb]
        emb_act = ops.silu(emb)
        parts = []
        for r in self._resnets():
            tb = r.time_emb_proj(emb_act)                                   # [S or S*N, cout]
            tb = tb.view(S, N, -1) if tb.shape[0] == S * N else tb[:, None, :].expand(S, N, tb.shape[-1])
            parts.append(tb.reshape(S, -1))
        return torch.cat(parts, dim=1).contiguous()

    @staticmethod
    def to_nhwc(x, cpad=None):
        x = x.permute(0, 2, 3, 1)
        if cpad is not None and x.shape[-1] != cpad:
            x = torch.nn.functional.pad(x, (0, cpad - x.shape[-1]))
        return x.contiguous()

    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None, context_cache=None, return_dict=True,
                nhwc_io=False, time_bias=None, **unused):
        """sample: [N,4,H,W] (diffusers layout) or, with nhwc_io, [N,H,W,8] channel-padded NHWC; returns `.sample` in the
        same layout family ([N,4,H,W] or [N,H,W,4]).  `time_bias`: flat buffer of one row of `precompute_time_bias` (then
        `timestep` / `added_cond_kwargs` are not read: everything they feed was computed ahead of the loop)."""
        cfg = self.config
        N = sample.shape[0]
        dev = sample.device
        tb_of = None
        if time_bias is not None:
            offs, total = self.time_bias_layout(N)
            assert time_bias.numel() == total, "time_bias does not match this batch size"
            tb_of = {id(r): time_bias[o:o + N * c].view(N, c) for r, (o, c) in zip(self._resnets(), offs)}
            emb_act = None
        else:
            if not torch.is_tensor(timestep):
                timestep = torch.tensor([timestep], device=dev)
            timesteps = timestep.reshape(-1).to(dev).expand(N)
            emb_act = ops.silu(self._embedding(timesteps, N, added_cond_kwargs))
        ctx = encoder_hidden_states.to(self.dtype)
        x = sample if nhwc_io else self.to_nhwc(sample.to(self.dtype), _pad8(cfg.in_channels))
        kvc = (lambda t: context_cache[id(t)]) if context_cache is not None else (lambda t: None)
        tbo = (lambda r: tb_of[id(r)]) if tb_of is not None else (lambda r: None)

        x = self.conv_in(x)
        skips = [x]
        for blk in self.down_blocks:
            for j, r in enumerate(blk.resnets):
                x = r(x, emb_act, tbo(r))
                if hasattr(blk, "attentions"):
                    x = blk.attentions[j](x, ctx, kvc(blk.attentions[j]))
                skips.append(x)
            if hasattr(blk, "downsamplers"):
                x = blk.downsamplers[0](x)
                skips.append(x)
        m = self.mid_block
        x = m.resnets[0](x, emb_act, tbo(m.resnets[0]))
        x = m.attentions[0](x, ctx, kvc(m.attentions[0]))
        x = m.resnets[1](x, emb_act, tbo(m.resnets[1]))
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                x = torch.cat([x, skips.pop()], dim=-1)  # channel concat on NHWC: pure data movement
                x = r(x, emb_act, tbo(r))
                if hasattr(blk, "attentions"):
                    x = blk.attentions[j](x, ctx, kvc(blk.attentions[j]))
            if hasattr(blk, "upsamplers"):
                x = blk.upsamplers[0](x)
        x = self.conv_out(self.conv_norm_out(x))
        if not nhwc_io:
            x = x.permute(0, 3, 1, 2)
        if not return_dict:
            return (x,)
        return SimpleNamespace(sample=x)
