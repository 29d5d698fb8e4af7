"""TEST INFRASTRUCTURE -- generates tests/golden/sd_head.pt and tests/golden/sdxl_head.pt by EXECUTING the reference's
`StableDiffusionHead` (omni/models/dreamllm/modeling_plugins.py:335-850) and `StableDiffusionXLHead`
(omni/models/dreamllm_sdxl/modeling_plugins.py:48-445) wrapper code (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_sdhead

What is pinned BY EXECUTION: everything the reference's own files do -- `forward` (:493-577: latent scaling, noise offset,
input perturbation, timestep draw, add_noise call, CFG-drop mixing, projector, epsilon / v target, min-SNR weighting through
`_compute_snr` :468-491, the dummy branch), `pipeline` (:671-850: projector on both prompt halves, CFG concat order,
`scale_model_input`, UNet call convention, guidance, `_rescale_noise_cfg` :658-669, `scheduler.step` kwargs found by
signature inspection, VAE decode / `scaling_factor`, post-processing) and the SDXL overrides (:151-236 mean-pooled
`global_projector` -> `text_embeds`, `.float()` feeds, `add_time_ids`; :239-445 `original_size/crops/target_size`
micro-conditioning doubled for CFG).  What stays a RESTATEMENT: the diffusers==0.24 arithmetic underneath
(`oracle/{unet,vae,sched}_ref.py`, attached through `oracle/duck_diffusers.py`), because diffusers is neither vendored
nor installable here.  Status: "wrapper pinned by execution, third-party UNet/VAE/scheduler arithmetic restated".

Each case stores: the inputs, the random draws the wrapper consumed (in order), the fp32 result, and the result of the
SAME wrapper run with every module in bfloat16 (`*_bf16`: the `err_ref` yard-stick of the tolerance contract).  Weights
are NOT stored: they come from seeded generators (`unet_ref.random_state_dict`, `vae_ref.random_state_dict`, PROJ_SEED).
"""
from __future__ import annotations

import copy
import importlib
import sys

import torch

from . import duck_diffusers as dd
from . import ref_loader, unet_ref, vae_ref
from .make_golden import bf16r, rel, save

EMBED, NQ, GDIM = 128, 8, 40
VAE_CFG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(32, 64, 64, 64), layers_per_block=1,
               norm_num_groups=32, scaling_factor=0.18215, sample_size=128)
UNET_SEED, VAE_SEED, PROJ_SEED = 11, 12, 13
BF = torch.bfloat16


def load_reference_heads():
    """Import the REAL plugin modules (ref_loader normally replaces the base one by a 2-name stub)."""
    import transformers
    from transformers import CLIPVisionModel  # noqa: F401  (resolve the lazy attribute before torchvision gets stubbed)
    transformers.CLIPImageProcessor = ref_loader._Dummy  # the lazy import would probe the stubbed torchvision's version
    ref_loader.install()
    sys.modules.pop("omni.models.dreamllm.modeling_plugins", None)
    base = importlib.import_module("omni.models.dreamllm.modeling_plugins")
    xl = importlib.import_module("omni.models.dreamllm_sdxl.modeling_plugins")
    return base, xl


def projector_weights(cross_dim, xl):
    g = torch.Generator().manual_seed(PROJ_SEED)
    w = {"projector": bf16r(torch.randn(cross_dim, EMBED, generator=g) * 0.05)}
    if xl:
        w["global_projector"] = bf16r(torch.randn(GDIM, EMBED, generator=g) * 0.05)
    return w


def build_head(mods, xl, scheduler, **knobs):
    """The reference class, built without running its diffusers-dependent __init__ (`__new__` + the attributes __init__
    would set, modeling_plugins.py:357-407 / dreamllm_sdxl :89-113), with the real reference projector (`build_projector`)."""
    base, xlmod = mods
    cls = xlmod.StableDiffusionXLHead if xl else base.StableDiffusionHead
    ucfg = unet_ref.tiny_config(cross_dim=64, sdxl=xl)
    head = cls.__new__(cls)
    torch.nn.Module.__init__(head)
    head.save_model_name = "stable_diffusion_xl_head" if xl else "stable_diffusion_head"
    head.embed_hidden_size = EMBED
    for k, v in dict(drop_prob=None, noise_offset=0.0, input_perturbation=0.0, snr_gamma=None).items():
        setattr(head, k, knobs.get(k, v))
    vcfg = dict(VAE_CFG, scaling_factor=0.13025 if xl else 0.18215)
    head.vae = dd.DuckVAE(vcfg, {k: bf16r(v) for k, v in vae_ref.random_state_dict(vcfg, seed=VAE_SEED).items()})
    head.unet = dd.DuckUNet(ucfg, {k: bf16r(v) for k, v in unet_ref.random_state_dict(ucfg, seed=UNET_SEED).items()})
    head.noise_scheduler = scheduler
    pcfg = dict(projector="linear", freeze_projector=False, depth=1, save_model_name=head.save_model_name, model_name_or_path=None)
    build_projector = importlib.import_module("omni.models.projector.builder").build_projector
    pw = projector_weights(ucfg["cross_attention_dim"], xl)
    head.projector = build_projector(pcfg, in_hidden_size=EMBED, out_hidden_size=ucfg["cross_attention_dim"], bias=False)
    head.projector.projector.weight.data = pw["projector"].clone()
    if xl:
        head.global_condition_hidden_size = GDIM
        # the reference omits `bias` here and trips LinearProjector's own assert (dreamllm_sdxl/modeling_plugins.py:103,
        # projector/mlp_projector.py:18): bias=False is the only runnable reading
        head.global_projector = build_projector(pcfg, in_hidden_size=EMBED, out_hidden_size=GDIM, bias=False)
        head.global_projector.projector.weight.data = pw["global_projector"].clone()
    head.vae_scale_factor = 2 ** (len(vcfg["block_out_channels"]) - 1)
    head.image_processor = dd.DuckImageProcessor()
    head.set_progress_bar_config(disable=True)
    return head, ucfg, vcfg


def to_bf16(head, xl):
    h = copy.deepcopy(head)
    h.unet.to(BF)
    h.projector.to(BF)
    if xl:
        h.global_projector.to(BF)  # the reference's .to() keeps the VAE in fp32 (dreamllm_sdxl/modeling_plugins.py:140-149)
    else:
        h.vae.to(BF)
    return h


def run_forward(head, xl, inputs, replay=None, dtype=torch.float32):
    enc = inputs["enc"].to(dtype).clone().requires_grad_(True)
    u = inputs.get("u_enc")
    u = u.to(dtype).clone().requires_grad_(True) if u is not None else None
    images = inputs["images"] if (xl and dtype == BF) else inputs["images"].to(dtype)  # XL: VAE stays fp32
    head.zero_grad()
    # bf16 run of the CFG-drop case: `(1 - mask) * enc + mask * u` promotes to fp32 (the mask is an fp32 bernoulli draw,
    # :541-543) and the bf16 projector then rejects it unless autocast is on, as under the reference trainer's bf16 mode
    import contextlib
    # SDXL: the wrapper feeds `.float()` tensors to the (bf16) UNet (dreamllm_sdxl :212-215), which only runs under the
    # trainer's bf16 autocast -- so the SDXL yard-stick run is an autocast run too
    amp = torch.autocast("cpu", dtype=BF) if (dtype == BF and (u is not None or xl)) else contextlib.nullcontext()
    with dd.DrawLog(replay=replay) as log, amp:
        if xl:
            loss = head(images, enc, u, inputs["add_time_ids"], None)
        else:
            loss = head(images, enc, u, None)
    loss.backward()
    out = dict(loss=loss.detach().float(), grad_enc=enc.grad.detach().float(),
               grad_projector=head.projector.projector.weight.grad.detach().float())
    if u is not None and u.grad is not None:
        out["grad_u_enc"] = u.grad.detach().float()
    if xl:
        out["grad_global_projector"] = head.global_projector.projector.weight.grad.detach().float()
    return out, log.draws


def forward_case(mods, xl, name, seed, knobs, prediction_type="epsilon", with_u=False):
    head, ucfg, vcfg = build_head(mods, xl, dd.DuckDDPMScheduler(prediction_type=prediction_type), **knobs)
    g = torch.Generator().manual_seed(seed)
    N = 2
    inputs = dict(images=bf16r(torch.rand(N, 3, 128, 128, generator=g) * 2 - 1), enc=bf16r(torch.randn(N, NQ, EMBED, generator=g) * 0.5))
    if with_u:
        inputs["u_enc"] = bf16r(torch.randn(N, NQ, EMBED, generator=g) * 0.5)
    if xl:
        inputs["add_time_ids"] = torch.tensor([[128., 128, 0, 0, 128, 128], [200., 160, 8, 16, 128, 128]])
    torch.manual_seed(seed)
    out, draws = run_forward(head, xl, inputs)
    out_b, _ = run_forward(to_bf16(head, xl), xl, inputs, replay=draws, dtype=BF)
    case = dict(name=name, knobs=dict(knobs), prediction_type=prediction_type, inputs=inputs,
                draws=[(k, t.clone()) for k, t in draws], **out, **{k + "_bf16": v for k, v in out_b.items()})
    print(f"  {name}: loss {float(out['loss']):.6f}  (bf16 run {float(out_b['loss']):.6f}), draws {[k for k, _ in draws]}, "
          f"grad_enc err_ref {rel(out_b['grad_enc'], out['grad_enc']):.2e}")
    return case


def pipeline_case(mods, xl, name, seed, scheduler, steps, guidance_scale=7.5, guidance_rescale=0.0, output_type="latent",
                  eta=0.0, given_latents=True, extra=None):
    head, ucfg, vcfg = build_head(mods, xl, scheduler)
    g = torch.Generator().manual_seed(seed)
    B = 2
    pe = bf16r(torch.randn(B, NQ, EMBED, generator=g) * 0.5)
    ne = bf16r(torch.randn(B, NQ, EMBED, generator=g) * 0.5)
    lat0 = torch.randn(B, 4, 16, 16, generator=torch.Generator().manual_seed(42)) if given_latents else None
    kw = dict(num_inference_steps=steps, guidance_scale=guidance_scale, guidance_rescale=guidance_rescale, eta=eta,
              output_type=output_type, **(extra or {}))

    def run(h, dtype, replay):
        import contextlib
        gen = torch.Generator().manual_seed(seed + 1)
        amp = torch.autocast("cpu", dtype=BF) if (dtype == BF and xl) else contextlib.nullcontext()
        with dd.DrawLog(replay=replay) as log, amp:
            out = h.pipeline(latents=None if lat0 is None else lat0.clone().to(dtype), prompt_embeds=pe.to(dtype),
                             negative_prompt_embeds=ne.to(dtype) if guidance_scale > 1.0 else None, generator=gen, **kw)
        return out.float(), log.draws

    out, draws = run(head, torch.float32, None)
    # loop structure the HIP pipeline must reproduce: one UNet call per step at batch 2B (CFG) with the leading timesteps
    assert [c[1] for c in head.unet.calls] == [2 * B if guidance_scale > 1.0 else B] * steps
    out_b, _ = run(to_bf16(head, xl), BF, draws)
    print(f"  {name}: |out| {float(out.norm()):.4f}, err_ref (bf16 wrapper run) {rel(out_b, out):.2e}, {len(draws)} draws")
    return dict(name=name, steps=steps, kwargs=kw, scheduler=type(scheduler).__name__.replace("Duck", ""),
                prediction_type=scheduler.config.prediction_type, prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat0,
                gen_seed=seed + 1, draws=[(k, t.clone()) for k, t in draws], out=out, out_bf16=out_b,
                timesteps=[c[0][0] for c in head.unet.calls])


def compact(o):
    """fp32 tensors whose values are bf16-representable (inputs, bf16-run outputs) are stored as bf16: halves the fixture."""
    if torch.is_tensor(o):
        return o.to(BF) if (o.dtype == torch.float32 and torch.equal(o, bf16r(o))) else o
    if isinstance(o, dict):
        return {k: compact(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(compact(v) for v in o)
    return o


def main():
    mods = load_reference_heads()
    for xl in (False, True):
        tag = "sdxl_head" if xl else "sd_head"
        print(tag)
        fwd = [forward_case(mods, xl, "plain", 21, {})]
        fwd.append(forward_case(mods, xl, "offset_perturb_snr", 22, dict(noise_offset=0.1, input_perturbation=0.1, snr_gamma=5.0)))
        fwd.append(forward_case(mods, xl, "v_prediction_snr", 23, dict(snr_gamma=5.0), prediction_type="v_prediction"))
        if not xl:  # CFG-drop mixing exists only in the base head (:538-543); the SDXL forward ignores u_encoder_hidden_states
            for seed in range(24, 64):  # a seed whose bernoulli draw drops exactly one of the two samples
                case = forward_case(mods, xl, "cfg_drop", seed, dict(drop_prob=0.5), with_u=True)
                if float(case["draws"][-1][1].sum()) == 1.0:
                    break
            fwd.append(case)
        pipes = [
            pipeline_case(mods, xl, "ddim_1", 31, dd.DuckDDIMScheduler(), 1),
            pipeline_case(mods, xl, "ddim_10", 32, dd.DuckDDIMScheduler(), 10),
            pipeline_case(mods, xl, "ddim_50", 33, dd.DuckDDIMScheduler(), 50),
            pipeline_case(mods, xl, "ddim_10_rescale", 34, dd.DuckDDIMScheduler(), 10, guidance_rescale=0.7),
            pipeline_case(mods, xl, "ddim_4_nocfg", 35, dd.DuckDDIMScheduler(), 4, guidance_scale=1.0),
            pipeline_case(mods, xl, "ddpm_10", 36, dd.DuckDDPMScheduler(), 10),          # the head's own scheduler: ancestral noise
            pipeline_case(mods, xl, "ddpm_5_pt_randlat", 37, dd.DuckDDPMScheduler(), 5, output_type="pt", given_latents=False),
            pipeline_case(mods, xl, "ddim_4_vpred_eta", 38, dd.DuckDDIMScheduler(prediction_type="v_prediction"), 4, eta=0.5),
        ]
        if xl:
            pipes.append(pipeline_case(mods, xl, "ddim_4_microcond", 39, dd.DuckDDIMScheduler(), 4,
                                       extra=dict(original_size=[200, 160], crops_coords_top_left=[8, 16], target_size=[128, 128])))
        # dummy branch (:501-508): zero loss that still reaches the projector and the dream queries
        head, _, _ = build_head(mods, xl, dd.DuckDDPMScheduler())
        dq = bf16r(torch.randn(1, NQ, EMBED, generator=torch.Generator().manual_seed(5))).requires_grad_(True)
        dummy = None
        try:
            loss = head(None, None, None, None, dq) if xl else head(None, None, None, dq)
            loss.backward()
            dummy = dict(loss=float(loss), proj_grad_is_zero=bool((head.projector.projector.weight.grad == 0).all()),
                         dq_grad_is_zero=bool((dq.grad == 0).all()))
        except RuntimeError as e:  # SDXL: the projected dummy is fed to the global projector (:165): width mismatch
            dummy = dict(error=str(e).splitlines()[0])
        print("  dummy:", dummy)
        save(f"{tag}.pt", compact(dict(embed=EMBED, nq=NQ, gdim=GDIM, unet_cfg=unet_ref.tiny_config(cross_dim=64, sdxl=xl),
                               vae_cfg=dict(VAE_CFG, scaling_factor=0.13025 if xl else 0.18215),
                               seeds=dict(unet=UNET_SEED, vae=VAE_SEED, proj=PROJ_SEED), forward=fwd, pipeline=pipes, dummy=dummy)))


if __name__ == "__main__":
    main()
