"""`-m gpu` parity of the CLIP encoder, the SD UNet (SD-2.1 and SDXL structures), the VAE, the schedulers and the
StableDiffusionHead training / denoising paths against the CPU oracles (oracle/{clip,unet,vae,sched}_ref.py) on
identical seeded inputs and weights.  Tolerances follow tests/test_model_gpu.py: bf16 storage => compare against the
fp32 oracle with the oracle-in-bf16 error as the yard-stick where the graph is deep."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def bf16r(t):
    return t.to(BF).float()


def _ops():
    from dreamllm_amd import ops
    return ops


# ----------------------------------------------------------------------------- conv / groupnorm kernels
@pytest.mark.parametrize("N,H,W,CI,CO,K,mode", [
    (2, 16, 16, 64, 128, 3, "same"), (1, 12, 20, 320, 320, 3, "same"), (3, 8, 8, 128, 64, 1, "same"),
    (2, 16, 16, 64, 64, 3, "down"), (2, 8, 8, 128, 128, 3, "up"), (1, 16, 16, 64, 64, 3, "down_asym"),
    (2, 16, 16, 4, 64, 3, "same"), (2, 16, 16, 64, 4, 3, "same"),
    (2, 8, 8, 1280, 1280, 3, "same"), (2, 16, 16, 640, 640, 3, "down"),   # small grid, deep K: split-K path
    (4, 32, 32, 128, 256, 3, "same"), (3, 24, 40, 192, 320, 3, "same"),   # full 256-tiles: LDS-DMA gather + staged epilogue
])
@pytest.mark.parametrize("tile", [128, 256, 259])
def test_conv_fwd_dgrad(N, H, W, CI, CO, K, mode, tile):
    """Implicit-GEMM NHWC conv against F.conv2d (fp32, NCHW) incl. stride-2, fused nearest-upsample, asymmetric pad,
    channel-padded conv_in / 4-channel conv_out; input gradient through the flipped-weight conv."""
    from dreamllm_amd.unet import HipConv2d, _pad8
    from dreamllm_amd import _lib
    _lib.check("dllm_gemm_set_tile", tile)
    torch.manual_seed(N * H + CI + CO)
    conv = HipConv2d(CI, CO, K, mode=mode)
    conv.weight.data = bf16r(conv.weight.data)
    conv.bias.data = bf16r(conv.bias.data)
    x = bf16r(torch.randn(N, CI, H, W))
    xr = x.clone().requires_grad_(True)
    w, b = conv.weight.data.clone(), conv.bias.data.clone()
    if mode == "same":
        yr = F.conv2d(xr, w, b, padding=K // 2)
    elif mode == "down":
        yr = F.conv2d(xr, w, b, stride=2, padding=1)
    elif mode == "down_asym":
        yr = F.conv2d(F.pad(xr, (0, 1, 0, 1)), w, b, stride=2)
    else:
        yr = F.conv2d(F.interpolate(xr, scale_factor=2.0, mode="nearest"), w, b, padding=1)
    dy = bf16r(torch.randn_like(yr))
    yr.backward(dy)
    conv = conv.to(DEV, BF)
    xn = x.permute(0, 2, 3, 1)
    if CI % 8:
        xn = F.pad(xn, (0, _pad8(CI) - CI))
    xd = xn.contiguous().to(BF).to(DEV).requires_grad_(mode != "down_asym")
    y = conv(xd)
    assert y.shape == (N, yr.shape[2], yr.shape[3], CO)
    assert rel_l2(y.permute(0, 3, 1, 2), yr) < 4e-3
    if mode != "down_asym":
        y.backward(dy.permute(0, 2, 3, 1).contiguous().to(BF).to(DEV))
        assert rel_l2(xd.grad[..., :CI].permute(0, 3, 1, 2), xr.grad) < 4e-3
    _lib.check("dllm_gemm_set_tile", 0)


def test_conv_epilogue_image_bias_and_residual():
    from dreamllm_amd.unet import HipConv2d
    torch.manual_seed(3)
    N, H, W, C = 2, 8, 8, 64
    conv = HipConv2d(C, C, 3)
    conv.weight.data, conv.bias.data = bf16r(conv.weight.data), bf16r(conv.bias.data)
    x, res, tb = bf16r(torch.randn(N, C, H, W)), bf16r(torch.randn(N, C, H, W)), bf16r(torch.randn(N, C))
    ref = F.conv2d(x, conv.weight.data, conv.bias.data, padding=1) + tb[:, :, None, None] + res
    conv = conv.to(DEV, BF)
    y = conv(x.permute(0, 2, 3, 1).contiguous().to(BF).to(DEV), residual=res.permute(0, 2, 3, 1).contiguous().to(BF).to(DEV),
             image_bias=tb.to(BF).to(DEV))
    assert rel_l2(y.permute(0, 3, 1, 2), ref) < 4e-3


@pytest.mark.parametrize("N,HW,C,act", [(2, 64, 320, True), (3, 256, 64, False), (1, 1024, 1280, True), (2, 100, 960, True),
                                        (2, 4096, 128, True)])
def test_groupnorm_fwd_bwd(N, HW, C, act):
    ops = _ops()
    torch.manual_seed(C + HW)
    x = bf16r(torch.randn(N, HW, C) * 2 + 0.5)
    g, b = bf16r(1 + 0.1 * torch.randn(C)), bf16r(0.1 * torch.randn(C))
    dy = bf16r(torch.randn(N, HW, C))
    xr = x.clone().requires_grad_(True)
    yr = F.group_norm(xr.transpose(1, 2), 32, g, b, 1e-5).transpose(1, 2)
    if act:
        yr = F.silu(yr)
    yr.backward(dy)
    xd = x.to(BF).to(DEV).requires_grad_(True)
    y = ops.groupnorm(xd, g.to(BF).to(DEV), b.to(BF).to(DEV), 32, 1e-5, act)
    assert rel_l2(y, yr) < 4e-3
    y.backward(dy.to(BF).to(DEV))
    assert rel_l2(xd.grad, xr.grad) < 6e-3


def test_cfg_ddim_fused_kernel():
    ops = _ops()
    from oracle import sched_ref
    torch.manual_seed(0)
    B, P = 2, 64
    pred = bf16r(torch.randn(2 * B, P, 4))
    lat = torch.randn(B, P, 4)
    ac = sched_ref.alphas_cumprod()
    t, n = 981, 50
    eu, ec = pred.chunk(2)
    ref = sched_ref.ddim_step(sched_ref.cfg(eu, ec, 7.5), t, lat, n, ac)
    latd = lat.clone().to(DEV)
    nxt = torch.empty(2 * B, P, 8, dtype=BF, device=DEV)
    ops.cfg_ddim_step_(pred.to(BF).to(DEV), latd, nxt, 7.5, ac[t], ac[t - 20])
    assert rel_l2(latd, ref) < 1e-5
    assert torch.equal(nxt[:B, :, :4].float().cpu(), latd.to(BF).float().cpu()) and torch.count_nonzero(nxt[..., 4:]) == 0
    assert torch.equal(nxt[:B], nxt[B:])


# ----------------------------------------------------------------------------- CLIP
def test_clip_encoder_vs_transformers_and_oracle():
    """HIP tower == installed transformers.CLIPVisionModel (fp32) == oracle restatement, hidden_states[-2][:, 1:]."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from dreamllm_amd.clip_vit import HipCLIPVisionModel, load_clip_config
    from oracle import clip_ref
    torch.manual_seed(0)
    cfgd = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=4, num_attention_heads=2, image_size=56, patch_size=14)
    hf = CLIPVisionModel(CLIPVisionConfig(**cfgd)).eval()
    for p in hf.parameters():
        p.data = bf16r(p.data)
    px = bf16r(torch.randn(3, 3, 56, 56))
    with torch.no_grad():
        ref = hf(px, output_hidden_states=True).hidden_states
    sd = {("vision_model." + k if not k.startswith("vision_model.") else k): v for k, v in hf.state_dict().items()}
    ocfg = dict(cfgd, layer_norm_eps=1e-5)
    hs = clip_ref.clip_hidden_states(px, sd, ocfg)
    for a, b in zip(hs, ref):
        assert rel_l2(a, b) < 1e-5
    m = HipCLIPVisionModel(load_clip_config(cfgd))
    m.load_state_dict_compat(hf.state_dict())
    m = m.to(DEV, BF)
    out = m.encode(px.to(DEV), -2)
    hb = clip_ref.clip_hidden_states(px.to(BF), {k: v.to(BF) for k, v in sd.items()}, ocfg)
    e_ref = rel_l2(hb[-2], ref[-2])
    assert rel_l2(out, ref[-2]) <= 1.5 * e_ref + 2e-3, (rel_l2(out, ref[-2]), e_ref)
    assert rel_l2(m.encode(px.to(DEV), 0), ref[0]) < 6e-3


# ----------------------------------------------------------------------------- UNet
def _unet_pair(sdxl):
    from dreamllm_amd.unet import HipUNet2DConditionModel, load_unet_config
    from oracle import unet_ref
    cfg = unet_ref.tiny_config(cross_dim=64, sdxl=sdxl)
    sd = {k: bf16r(v) for k, v in unet_ref.random_state_dict(cfg, seed=1).items()}
    m = HipUNet2DConditionModel(load_unet_config(cfg))
    m.load_state_dict(sd, strict=True)
    return cfg, sd, m.to(DEV, BF).requires_grad_(False)


@pytest.mark.parametrize("sdxl", [False, True])
def test_unet_forward_and_context_gradient(sdxl):
    """Forward noise prediction and the gradient w.r.t. the conditioning tokens (the only gradient the frozen UNet passes
    on, modeling_plugins.py:405-407,556) against the fp32 oracle; yard-stick = oracle in bf16."""
    from oracle import unet_ref
    cfg, sd, m = _unet_pair(sdxl)
    torch.manual_seed(5)
    N = 2
    x = bf16r(torch.randn(N, 4, 16, 16))
    ctx = bf16r(torch.randn(N, 8, 64))
    t = torch.tensor([17, 801])
    added = dict(text_embeds=bf16r(torch.randn(N, 40)), time_ids=torch.tensor([[16., 16, 0, 0, 16, 16]] * N)) if sdxl else None
    cr = ctx.clone().requires_grad_(True)
    yr = unet_ref.unet_forward(x, t, cr, sd, cfg, added)
    dy = bf16r(torch.randn_like(yr))
    yr.backward(dy)
    with torch.no_grad():
        yb = unet_ref.unet_forward(x.to(BF), t, ctx.to(BF), {k: v.to(BF) for k, v in sd.items()}, cfg,
                                   None if added is None else {k: v.to(BF) for k, v in added.items()})
    e_ref = rel_l2(yb, yr)
    cd = ctx.to(BF).to(DEV).requires_grad_(True)
    addd = None if added is None else {k: v.to(DEV) for k, v in added.items()}
    y = m(x.to(BF).to(DEV), t.to(DEV), cd, added_cond_kwargs=addd).sample
    assert y.shape == yr.shape
    e = rel_l2(y, yr)
    assert e <= 1.5 * e_ref + 2e-3, (e, e_ref)
    y.backward(dy.to(BF).to(DEV))
    assert rel_l2(cd.grad, cr.grad) <= 4e-2


def test_unet_context_cache_matches():
    cfg, sd, m = _unet_pair(False)
    torch.manual_seed(6)
    x = torch.randn(2, 4, 16, 16, device=DEV, dtype=BF)
    ctx = torch.randn(2, 8, 64, device=DEV, dtype=BF)
    with torch.no_grad():
        a = m(x, 500, ctx).sample
        b = m(x, 500, ctx, context_cache=m.prepare_context(ctx)).sample
    assert torch.equal(a, b)


# ----------------------------------------------------------------------------- VAE
def _tiny_vae():
    from dreamllm_amd.vae import AutoencoderKLLite, load_vae_config
    cfgd = dict(block_out_channels=(32, 64, 64), layers_per_block=1)
    cfg = load_vae_config(dict(vae=cfgd))
    torch.manual_seed(2)
    v = AutoencoderKLLite(cfg)
    for n, p in v.named_parameters():
        if "norm" in n and n.endswith("weight"):
            p.data = 1 + 0.1 * torch.randn_like(p)
        p.data = bf16r(p.data)
    return cfg, v


def test_vae_encode_decode():
    from oracle import vae_ref
    cfg, v = _tiny_vae()
    sd = {k: t.clone() for k, t in v.state_dict().items()}
    od = dict(cfg.to_dict())
    img = bf16r(torch.rand(2, 3, 32, 32) * 2 - 1)
    mom = vae_ref.encode_moments(img, sd, od)
    z = bf16r(torch.randn(2, 4, 8, 8))
    dec = vae_ref.decode(z, sd, od)
    sdb = {k: t.to(BF) for k, t in sd.items()}
    e_enc = rel_l2(vae_ref.encode_moments(img.to(BF), sdb, od)[:, :4], mom[:, :4])
    e_dec = rel_l2(vae_ref.decode(z.to(BF), sdb, od), dec)
    v = v.to(DEV, BF)
    dist = v.encode(img.to(DEV))
    assert rel_l2(dist.mean, mom[:, :4]) <= 1.5 * e_enc + 2e-3
    assert rel_l2(v.decode(z.to(DEV)), dec) <= 1.5 * e_dec + 2e-3


# ----------------------------------------------------------------------------- StableDiffusionHead
def _tiny_head(embed=128):
    from dreamllm_amd.modeling_plugins import StableDiffusionHead
    from oracle import unet_ref
    ucfg = unet_ref.tiny_config(cross_dim=64)
    torch.manual_seed(4)
    head = StableDiffusionHead(dict(unet=ucfg, vae=dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1)),
                               embed_hidden_size=embed)
    usd = {k: bf16r(v) for k, v in unet_ref.random_state_dict(ucfg, seed=1).items()}
    head.unet.load_state_dict(usd)
    for p in head.parameters():
        p.data = bf16r(p.data)
    return head, ucfg, usd


def test_sd_head_training_loss_and_grad():
    """StableDiffusionHead.forward (modeling_plugins.py:493-577) with injected noise/timesteps: VAE-encode -> add_noise ->
    projector -> UNet -> MSE; loss and d(loss)/d(dream states) against the oracles."""
    from oracle import unet_ref, vae_ref, sched_ref
    head, ucfg, usd = _tiny_head()
    vsd = {k: v.clone() for k, v in head.vae.state_dict().items()}
    pw = head.projector.projector.weight.data.clone()
    N = 2
    img = bf16r(torch.rand(N, 3, 128, 128) * 2 - 1)
    enc = bf16r(torch.randn(N, 8, 128) * 0.5)
    noise = bf16r(torch.randn(N, 4, 16, 16))
    ts = torch.tensor([100, 700])
    head = head.to(DEV, BF)
    encd = enc.to(BF).to(DEV).requires_grad_(True)
    # the VAE sample uses torch.randn inside: fix it by seeding and mirroring on the oracle side through the mean (std ~ small)
    torch.manual_seed(9)
    loss = head(img.to(DEV), encd, None, None, noise=noise.to(DEV), timesteps=ts.to(DEV))
    loss.backward()
    # oracle: same latents (take them from the HIP VAE mean/std with the same RNG draw replaced by the mode for stability)
    vcfg = dict(head.vae.config.to_dict())
    mom = vae_ref.encode_moments(img, vsd, vcfg)
    torch.manual_seed(9)
    eps_v = torch.randn(mom[:, :4].shape, device=DEV).cpu()
    lat = vae_ref.sample_latents(mom, eps_v, vcfg["scaling_factor"])
    ac = sched_ref.alphas_cumprod()
    noisy = torch.stack([sched_ref.add_noise(lat[i], noise[i], int(ts[i]), ac) for i in range(N)])
    er = enc.clone().requires_grad_(True)
    pred = unet_ref.unet_forward(noisy, ts, F.linear(er, pw), usd, ucfg)
    lref = F.mse_loss(pred.float(), noise.float())
    lref.backward()
    assert abs(loss.item() - lref.item()) <= 2e-2 * abs(lref.item()), (loss.item(), lref.item())
    assert rel_l2(encd.grad, er.grad) <= 6e-2
    assert rel_l2(head.projector.projector.weight.grad, torch.autograd.grad(
        F.mse_loss(unet_ref.unet_forward(noisy, ts, F.linear(enc, pw.requires_grad_(True)), usd, ucfg).float(), noise.float()),
        pw)[0]) <= 6e-2


def test_sd_head_dummy_forward_and_config():
    head, _, _ = _tiny_head()
    head = head.to(DEV, BF)
    dq = torch.randn(1, 8, 128, device=DEV, dtype=BF, requires_grad=True)
    out = head(None, None, None, dq)
    out.backward()
    assert out.item() == 0.0 and head.projector.projector.weight.grad is not None
    assert set(head.config) >= {"diffusion_name_or_path", "freeze_unet", "snr_gamma"}
    assert head.plugin_type == "head"


def test_denoise_pipeline_ddim_vs_oracle_loop():
    """StableDiffusionHead.pipeline (modeling_plugins.py:671-850) with the deterministic DDIM scheduler, CFG 7.5:
    final latents after 1, 4 and 10 steps against the oracle loop over the oracle UNet."""
    from dreamllm_amd.schedulers import DDIMScheduler
    from oracle import unet_ref, sched_ref
    head, ucfg, usd = _tiny_head()
    pw = head.projector.projector.weight.data.clone()
    B = 2
    pe = bf16r(torch.randn(B, 8, 128) * 0.5)
    ne = bf16r(torch.randn(B, 8, 128) * 0.5)
    lat0 = torch.randn(B, 4, 16, 16, generator=torch.Generator().manual_seed(42))
    head = head.to(DEV, BF)
    unet_fn = lambda x, t, c: unet_ref.unet_forward(bf16r(x), torch.tensor([t]), c, usd, ucfg)
    for steps in (1, 4, 10):
        ref = sched_ref.ddim_loop(unet_fn, lat0, F.linear(ne, pw), F.linear(pe, pw), steps, 7.5)
        out = head.pipeline(num_inference_steps=steps, guidance_scale=7.5, latents=lat0.clone(), prompt_embeds=pe.to(DEV),
                            negative_prompt_embeds=ne.to(DEV), output_type="latent", scheduler=DDIMScheduler())
        e = rel_l2(out, ref)
        assert e <= 1e-2 * steps**0.5 + 5e-3, (steps, e)


# ----------------------------------------------------------------------------- SDXL head
def _tiny_xl_head(embed=128, gdim=40):
    from dreamllm_amd.modeling_plugins_sdxl import StableDiffusionXLHead
    from oracle import unet_ref
    ucfg = unet_ref.tiny_config(cross_dim=64, sdxl=True)
    torch.manual_seed(5)
    head = StableDiffusionXLHead(dict(unet=ucfg, vae=dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1)),
                                 embed_hidden_size=embed, global_condition_hidden_size=gdim)
    usd = {k: bf16r(v) for k, v in unet_ref.random_state_dict(ucfg, seed=2).items()}
    head.unet.load_state_dict(usd)
    for p in head.parameters():
        p.data = bf16r(p.data)
    return head, ucfg, usd


def test_sdxl_head_training_loss_and_grad():
    """StableDiffusionXLHead.forward (dreamllm_sdxl/modeling_plugins.py:151-236): mean-pooled global projector ->
    `text_embeds`, `add_time_ids` -> SDXL UNet; loss and gradients of the dream states / both projectors vs the oracle."""
    from oracle import unet_ref, vae_ref, sched_ref
    head, ucfg, usd = _tiny_xl_head()
    assert ucfg["projection_class_embeddings_input_dim"] == 40 + 6 * ucfg["addition_time_embed_dim"]
    vsd = {k: v.clone() for k, v in head.vae.state_dict().items()}
    pw = head.projector.projector.weight.data.clone()
    gw = head.global_projector.projector.weight.data.clone()
    N = 2
    img = bf16r(torch.rand(N, 3, 128, 128) * 2 - 1)
    enc = bf16r(torch.randn(N, 8, 128) * 0.5)
    noise = bf16r(torch.randn(N, 4, 16, 16))
    ts = torch.tensor([50, 800])
    tids = torch.tensor([[128., 128, 0, 0, 128, 128], [200., 160, 8, 16, 128, 128]])
    head = head.to(DEV, BF)
    encd = enc.to(BF).to(DEV).requires_grad_(True)
    torch.manual_seed(9)
    loss = head(img.to(DEV), encd, None, tids.to(DEV), None, noise=noise.to(DEV), timesteps=ts.to(DEV))
    loss.backward()
    vcfg = dict(head.vae.config.to_dict())
    mom = vae_ref.encode_moments(img, vsd, vcfg)
    torch.manual_seed(9)
    eps_v = torch.randn(mom[:, :4].shape, device=DEV).cpu()
    lat = vae_ref.sample_latents(mom, eps_v, vcfg["scaling_factor"])
    ac = sched_ref.alphas_cumprod()
    noisy = torch.stack([sched_ref.add_noise(lat[i], noise[i], int(ts[i]), ac) for i in range(N)])
    er = enc.clone().requires_grad_(True)
    pwr, gwr = pw.clone().requires_grad_(True), gw.clone().requires_grad_(True)
    added = dict(text_embeds=bf16r(F.linear(bf16r(er.mean(1)), gwr)), time_ids=tids)
    pred = unet_ref.unet_forward(noisy, ts, F.linear(er, pwr), usd, ucfg, added_cond_kwargs=added)
    lref = F.mse_loss(pred.float(), noise.float())
    lref.backward()
    assert abs(loss.item() - lref.item()) <= 2e-2 * abs(lref.item()), (loss.item(), lref.item())
    assert rel_l2(encd.grad, er.grad) <= 6e-2
    assert rel_l2(head.projector.projector.weight.grad, pwr.grad) <= 6e-2
    assert rel_l2(head.global_projector.projector.weight.grad, gwr.grad) <= 8e-2


def test_sdxl_head_dummy_forward_config_and_state_dict(tmp_path):
    head, _, _ = _tiny_xl_head()
    head = head.to(DEV, BF)
    dq = torch.randn(1, 8, 128, device=DEV, dtype=BF, requires_grad=True)
    out = head(None, None, None, None, dq)
    out.backward()
    assert out.item() == 0.0
    assert head.projector.projector.weight.grad is not None and head.global_projector.projector.weight.grad is not None
    assert set(head.config) >= {"diffusion_name_or_path", "global_condition_hidden_size", "freeze_unet", "snr_gamma"}
    keys = set(head.state_dict())
    assert "global_projector.projector.weight" in keys and "projector.projector.weight" in keys
    assert not any(k.endswith("projector.bias") for k in keys)
    head.save_model(str(tmp_path))
    assert (tmp_path / "stable_diffusion_xl_head.bin").is_file()
    w = head.global_projector.projector.weight.data.clone()
    head.global_projector.projector.weight.data.zero_()
    head.load_model(str(tmp_path))
    assert torch.equal(head.global_projector.projector.weight.data, w)
    assert head.fsdp_ignored_modules() == [head.vae, head.unet]


@pytest.mark.parametrize("use_graph", [True, False])
def test_sdxl_pipeline_ddim_vs_oracle_loop(use_graph):
    """StableDiffusionXLHead.pipeline (dreamllm_sdxl/modeling_plugins.py:239-445), deterministic DDIM, CFG 7.5, both the
    hipGraph + fused-update loop and the plain loop, against the oracle loop over the oracle SDXL UNet."""
    from dreamllm_amd.schedulers import DDIMScheduler
    from oracle import unet_ref, sched_ref
    head, ucfg, usd = _tiny_xl_head()
    pw = head.projector.projector.weight.data.clone()
    gw = head.global_projector.projector.weight.data.clone()
    B = 2
    pe = bf16r(torch.randn(B, 8, 128) * 0.5)
    ne = bf16r(torch.randn(B, 8, 128) * 0.5)
    lat0 = torch.randn(B, 4, 16, 16, generator=torch.Generator().manual_seed(42))
    head = head.to(DEV, BF)
    full = ucfg["sample_size"] * 8
    tids = torch.tensor([[float(full), full, 0, 0, full, full]] * (2 * B))
    gl = torch.cat([bf16r(F.linear(bf16r(ne.mean(1)), gw)), bf16r(F.linear(bf16r(pe.mean(1)), gw))])

    def unet_fn(x, t, c):  # the oracle loop calls it on the [uncond; cond] batch
        return unet_ref.unet_forward(bf16r(x), torch.tensor([t]), c, usd, ucfg,
                                     added_cond_kwargs=dict(text_embeds=gl, time_ids=tids))

    for steps in (1, 4):
        ref = sched_ref.ddim_loop(unet_fn, lat0, F.linear(ne, pw), F.linear(pe, pw), steps, 7.5)
        out = head.pipeline(num_inference_steps=steps, guidance_scale=7.5, latents=lat0.clone(), prompt_embeds=pe.to(DEV),
                            negative_prompt_embeds=ne.to(DEV), output_type="latent", scheduler=DDIMScheduler(),
                            use_graph=use_graph)
        e = rel_l2(out, ref)
        assert e <= 1e-2 * steps**0.5 + 5e-3, (steps, e)
