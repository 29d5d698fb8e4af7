"""TEST INFRASTRUCTURE (oracle) -- CPU PyTorch restatement of `diffusers.UNet2DConditionModel` (diffusers==0.24.0, the
pin in /root/reference/pyproject.toml:74) as called by the reference at
omni/models/dreamllm/modeling_plugins.py:556 (training) and :815-821 (denoise loop), and of the SDXL variant called at
omni/models/dreamllm_sdxl/modeling_plugins.py:215,406-413.

PARITY UNPINNED: diffusers is neither vendored in /root/reference nor installed here (no network), so this restatement
follows the published diffusers-0.24 algorithm and cannot be checked against the real
class in this environment.  State-dict keys are diffusers' own, so wherever diffusers IS available a one-line
`load_state_dict` cross-check is possible.  Only tests/, smoke() and bench.py's cpu_baseline may import this module.

Layout NCHW, any float dtype; weights from a dict with diffusers key names.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

SD21_BASE = dict(
    sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    attention_head_dim=(5, 10, 20, 20), transformer_layers_per_block=1, cross_attention_dim=1024, norm_num_groups=32,
    norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0, addition_embed_type=None,
)
SDXL_BASE = dict(
    sample_size=128, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    attention_head_dim=(5, 10, 20), transformer_layers_per_block=(1, 2, 10), cross_attention_dim=2048, norm_num_groups=32,
    norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0, addition_embed_type="text_time", addition_time_embed_dim=256,
    projection_class_embeddings_input_dim=2816,
)


def tiny_config(cross_dim=64, sdxl=False):
    """A structurally complete miniature (every block type, down/up-sampling, skip concatenations) for fast tests."""
    if sdxl:
        return dict(SDXL_BASE, sample_size=16, block_out_channels=(64, 128, 128), attention_head_dim=(1, 2, 2),
                    transformer_layers_per_block=(1, 1, 2), cross_attention_dim=cross_dim, layers_per_block=1,
                    addition_time_embed_dim=32, projection_class_embeddings_input_dim=6 * 32 + 40)
    return dict(SD21_BASE, sample_size=16, block_out_channels=(64, 128, 128, 128), attention_head_dim=(1, 2, 2, 2),
                cross_attention_dim=cross_dim, layers_per_block=1)


def _tuple(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def timestep_embedding(timesteps, dim, flip_sin_to_cos=True, freq_shift=0, max_period=10000):
    """diffusers.models.embeddings.get_timestep_embedding."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / (half - freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def _lin(x, sd, p, bias=True):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias") if bias else None)


def resnet(x, emb, sd, p, groups, eps):
    """ResnetBlock2D.forward (time_embedding_norm='default', output_scale_factor=1)."""
    h = F.silu(F.group_norm(x, groups, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps))
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    t = _lin(F.silu(emb), sd, p + ".time_emb_proj")
    h = h + t[:, :, None, None]
    h = F.silu(F.group_norm(h, groups, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps))
    h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def attention(x, ctx, sd, p, heads):
    """diffusers Attention (AttnProcessor2_0): q/k/v no bias, to_out.0 with bias, scale head_dim^-0.5, no mask."""
    B, S, C = x.shape
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    hd = C // heads
    q, k, v = (t.view(B, -1, heads, hd).transpose(1, 2) for t in (q, k, v))
    w = torch.softmax((q @ k.transpose(-1, -2)) * hd**-0.5, dim=-1)
    o = (w @ v).transpose(1, 2).reshape(B, S, C)
    return _lin(o, sd, p + ".to_out.0")


def basic_transformer_block(x, ctx, sd, p, heads):
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    x = x + attention(h, h, sd, p + ".attn1", heads)
    h = F.layer_norm(x, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    x = x + attention(h, ctx, sd, p + ".attn2", heads)
    h = F.layer_norm(x, (C,), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"], 1e-5)
    hg = _lin(h, sd, p + ".ff.net.0.proj")
    hidden, gate = hg.chunk(2, dim=-1)
    return x + _lin(hidden * F.gelu(gate), sd, p + ".ff.net.2")


def transformer2d(x, ctx, sd, p, heads, depth, groups):
    """Transformer2DModel.forward with use_linear_projection=True."""
    N, C, H, W = x.shape
    res = x
    h = F.group_norm(x, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(N, H * W, C)
    h = _lin(h, sd, p + ".proj_in")
    for d in range(depth):
        h = basic_transformer_block(h, ctx, sd, f"{p}.transformer_blocks.{d}", heads)
    h = _lin(h, sd, p + ".proj_out")
    h = h.reshape(N, H, W, C).permute(0, 3, 1, 2)
    return h + res


def unet_forward(sample, timesteps, encoder_hidden_states, sd, cfg, added_cond_kwargs=None):
    """UNet2DConditionModel.forward -> predicted noise [N, out_channels, H, W]."""
    boc = cfg["block_out_channels"]
    nb = len(boc)
    groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    heads = _tuple(cfg["attention_head_dim"], nb)
    depth = _tuple(cfg["transformer_layers_per_block"], nb)
    lpb = cfg["layers_per_block"]
    N = sample.shape[0]
    if not torch.is_tensor(timesteps):
        timesteps = torch.tensor([timesteps], device=sample.device)
    timesteps = timesteps.reshape(-1).expand(N)
    t_emb = timestep_embedding(timesteps, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"]).to(sample.dtype)
    emb = _lin(F.silu(_lin(t_emb, sd, "time_embedding.linear_1")), sd, "time_embedding.linear_2")
    if cfg.get("addition_embed_type") == "text_time":
        text_embeds, time_ids = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
        te = timestep_embedding(time_ids.flatten(), cfg["addition_time_embed_dim"], cfg["flip_sin_to_cos"], cfg["freq_shift"])
        te = te.reshape(N, -1).to(sample.dtype)
        add = torch.cat([text_embeds, te], dim=-1)
        emb = emb + _lin(F.silu(_lin(add, sd, "add_embedding.linear_1")), sd, "add_embedding.linear_2")

    x = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    skips = [x]
    for i, bt in enumerate(cfg["down_block_types"]):
        for j in range(lpb):
            x = resnet(x, emb, sd, f"down_blocks.{i}.resnets.{j}", groups, eps)
            if bt.startswith("CrossAttn"):
                x = transformer2d(x, encoder_hidden_states, sd, f"down_blocks.{i}.attentions.{j}", heads[i], depth[i], groups)
            skips.append(x)
        if i < nb - 1:
            x = F.conv2d(x, sd[f"down_blocks.{i}.downsamplers.0.conv.weight"], sd[f"down_blocks.{i}.downsamplers.0.conv.bias"],
                         stride=2, padding=1)
            skips.append(x)
    x = resnet(x, emb, sd, "mid_block.resnets.0", groups, eps)
    x = transformer2d(x, encoder_hidden_states, sd, "mid_block.attentions.0", heads[-1], depth[-1], groups)
    x = resnet(x, emb, sd, "mid_block.resnets.1", groups, eps)
    rheads, rdepth = heads[::-1], depth[::-1]
    for i, bt in enumerate(cfg["up_block_types"]):
        for j in range(lpb + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet(x, emb, sd, f"up_blocks.{i}.resnets.{j}", groups, eps)
            if bt.startswith("CrossAttn"):
                x = transformer2d(x, encoder_hidden_states, sd, f"up_blocks.{i}.attentions.{j}", rheads[i], rdepth[i], groups)
        if i < nb - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"up_blocks.{i}.upsamplers.0.conv.weight"], sd[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(F.group_norm(x, groups, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps))
    return F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


def param_shapes(cfg):
    """name -> shape of every UNet parameter (diffusers key names) for the given config."""
    boc = cfg["block_out_channels"]
    nb = len(boc)
    depth = _tuple(cfg["transformer_layers_per_block"], nb)
    lpb, cd = cfg["layers_per_block"], cfg["cross_attention_dim"]
    temb = boc[0] * 4
    sh = {}

    def lin(p, i, o, bias=True):
        sh[p + ".weight"] = (o, i)
        if bias:
            sh[p + ".bias"] = (o,)

    def conv(p, i, o, k):
        sh[p + ".weight"] = (o, i, k, k)
        sh[p + ".bias"] = (o,)

    def norm(p, c):
        sh[p + ".weight"] = (c,)
        sh[p + ".bias"] = (c,)

    def res(p, i, o):
        norm(p + ".norm1", i)
        conv(p + ".conv1", i, o, 3)
        lin(p + ".time_emb_proj", temb, o)
        norm(p + ".norm2", o)
        conv(p + ".conv2", o, o, 3)
        if i != o:
            conv(p + ".conv_shortcut", i, o, 1)

    def tfm(p, c, d):
        norm(p + ".norm", c)
        lin(p + ".proj_in", c, c)
        for k in range(d):
            q = f"{p}.transformer_blocks.{k}"
            norm(q + ".norm1", c)
            for a, kv in (("attn1", c), ("attn2", cd)):
                lin(f"{q}.{a}.to_q", c, c, False)
                lin(f"{q}.{a}.to_k", kv, c, False)
                lin(f"{q}.{a}.to_v", kv, c, False)
                lin(f"{q}.{a}.to_out.0", c, c)
            norm(q + ".norm2", c)
            norm(q + ".norm3", c)
            lin(q + ".ff.net.0.proj", c, 8 * c)
            lin(q + ".ff.net.2", 4 * c, c)
        lin(p + ".proj_out", c, c)

    lin("time_embedding.linear_1", boc[0], temb)
    lin("time_embedding.linear_2", temb, temb)
    if cfg.get("addition_embed_type") == "text_time":
        lin("add_embedding.linear_1", cfg["projection_class_embeddings_input_dim"], temb)
        lin("add_embedding.linear_2", temb, temb)
    conv("conv_in", cfg["in_channels"], boc[0], 3)
    out_c = boc[0]
    for i, bt in enumerate(cfg["down_block_types"]):
        in_c, out_c = out_c, boc[i]
        for j in range(lpb):
            res(f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c)
            if bt.startswith("CrossAttn"):
                tfm(f"down_blocks.{i}.attentions.{j}", out_c, depth[i])
        if i < nb - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3)
    res("mid_block.resnets.0", boc[-1], boc[-1])
    tfm("mid_block.attentions.0", boc[-1], depth[-1])
    res("mid_block.resnets.1", boc[-1], boc[-1])
    rev, rdepth = boc[::-1], depth[::-1]
    out_c = rev[0]
    for i, bt in enumerate(cfg["up_block_types"]):
        prev, out_c, in_c = out_c, rev[i], rev[min(i + 1, nb - 1)]
        for j in range(lpb + 1):
            skip = in_c if j == lpb else out_c
            rin = prev if j == 0 else out_c
            res(f"up_blocks.{i}.resnets.{j}", rin + skip, out_c)
            if bt.startswith("CrossAttn"):
                tfm(f"up_blocks.{i}.attentions.{j}", out_c, rdepth[i])
        if i < nb - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)
    norm("conv_norm_out", boc[0])
    conv("conv_out", boc[0], cfg["out_channels"], 3)
    return sh


def random_state_dict(cfg, seed=0, dtype=torch.float32, device="cpu"):
    """Default-PyTorch-like init (uniform +-1/sqrt(fan_in)); norm weights around 1."""
    g = torch.Generator(device=device).manual_seed(seed)  # device="cuda": the full-size tests draw 0.9-2.6 G weights on the GPU
    sd = {}
    for k, s in param_shapes(cfg).items():
        if ".norm" in k or k.startswith("conv_norm_out"):
            t = (1.0 + 0.1 * torch.randn(s, generator=g, device=device)) if k.endswith("weight") else 0.1 * torch.randn(s, generator=g, device=device)
        elif k.endswith(".weight"):
            fan_in = math.prod(s[1:])
            t = (torch.rand(s, generator=g, device=device) * 2 - 1) / math.sqrt(fan_in)
        else:
            t = (torch.rand(s, generator=g, device=device) * 2 - 1) * 0.05
        sd[k] = t.to(dtype)
    return sd


def unet_flops(cfg, ctx_len=64, latent=None):
    """Analytic forward FLOPs per sample (multiply-add = 2): convs + linears + attention matmuls."""
    latent = latent or cfg["sample_size"]
    boc = cfg["block_out_channels"]
    nb = len(boc)
    heads = _tuple(cfg["attention_head_dim"], nb)
    depth = _tuple(cfg["transformer_layers_per_block"], nb)
    lpb, cd = cfg["layers_per_block"], cfg["cross_attention_dim"]
    temb = boc[0] * 4
    fl = {"conv": 0.0, "tfm": 0.0}

    def res(i, o, hw):
        fl["conv"] += 2 * hw * 9 * i * o + 2 * hw * 9 * o * o + (2 * hw * i * o if i != o else 0) + 2 * temb * o

    def tfm(c, d, hw):
        per = 2 * hw * c * c * 4 + 4 * hw * hw * c            # self-attn q,k,v,o + QK^T, PV
        per += 2 * hw * c * c * 2 + 2 * ctx_len * cd * c * 2 + 4 * hw * ctx_len * c  # cross
        per += 2 * hw * c * 8 * c + 2 * hw * 4 * c * c        # GEGLU ff
        fl["tfm"] += d * per + 2 * 2 * hw * c * c             # proj_in/out

    hw = latent * latent
    fl["conv"] += 2 * hw * 9 * cfg["in_channels"] * boc[0]
    out_c = boc[0]
    for i, bt in enumerate(cfg["down_block_types"]):
        in_c, out_c = out_c, boc[i]
        for j in range(lpb):
            res(in_c if j == 0 else out_c, out_c, hw)
            if bt.startswith("CrossAttn"):
                tfm(out_c, depth[i], hw)
        if i < nb - 1:
            hw //= 4
            fl["conv"] += 2 * hw * 9 * out_c * out_c
    res(boc[-1], boc[-1], hw)
    tfm(boc[-1], depth[-1], hw)
    res(boc[-1], boc[-1], hw)
    rev, rdepth = boc[::-1], depth[::-1]
    out_c = rev[0]
    for i, bt in enumerate(cfg["up_block_types"]):
        prev, out_c, in_c = out_c, rev[i], rev[min(i + 1, nb - 1)]
        for j in range(lpb + 1):
            res((prev if j == 0 else out_c) + (in_c if j == lpb else out_c), out_c, hw)
            if bt.startswith("CrossAttn"):
                tfm(out_c, rdepth[i], hw)
        if i < nb - 1:
            hw *= 4
            fl["conv"] += 2 * hw * 9 * out_c * out_c
    fl["conv"] += 2 * hw * 9 * boc[0] * cfg["out_channels"]
    return fl
