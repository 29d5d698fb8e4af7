"""`StableDiffusionXLHead` -- the SDXL decoder head of DreamLLM-SDXL, behind the reference's plugin API.

Mirrors omni/models/dreamllm_sdxl/modeling_plugins.py:
  * `SDXLDataProcessor` (:14-45): resize / crop / flip + the 6 micro-conditioning numbers
    (original_size + crop_top_left + target_size) that become `add_time_ids`;
  * `StableDiffusionXLHead(StableDiffusionHead)` (:48-149): the SD head plus a `global_projector`
    (embed_hidden_size -> global_condition_hidden_size = 1280) fed with the MEAN over the dream queries; its output is
    the SDXL UNet's pooled `text_embeds`;
  * `forward` (:151-236): VAE-encode -> noise / timestep -> add_noise -> projector + global_projector -> SDXL UNet with
    `added_cond_kwargs = {time_ids, text_embeds}` -> MSE (optionally min-SNR weighted); dummy branch :159-166;
  * `pipeline` (:239-445): CFG denoising loop with the `added_cond_kwargs` batch-doubled like the prompt embeddings.

Execution: the UNet (`unet.HipUNet2DConditionModel`, SDXL-base layout: block_out (320,640,1280), transformer depth
(1,2,10), cross-attention dim 2048, `text_time` addition embedding 256/2816) and the VAE run on this package's HIP kernels
in bf16 with fp32 accumulation.  Two deliberate differences from the reference's dtype choreography, both on frozen
modules: the reference feeds the UNet `.float()` tensors at train time (:212-215) and keeps the VAE in fp32 (:140-149,
:433-434, an fp16-overflow work-around that bf16 does not need); here both compute in bf16.  State-dict keys are the
reference's: `vae.*`, `unet.*`, `projector.projector.weight`, `global_projector.projector.weight`; file name
`stable_diffusion_xl_head.bin`.
"""
from __future__ import annotations

import random
from typing import Any, Callable, Literal, Optional, Tuple

import torch

from . import ops
from .modeling_plugins import StableDiffusionHead
from .projector import build_projector
from .utils import logger


class SDXLDataProcessor:
    """Image processor for SDXL (dreamllm_sdxl/modeling_plugins.py:14-45) on tensors: `image` is a float [3,H,W] in
    [0,1] (or a PIL image, converted); returns (image in [-1,1] of size resolution x resolution, the 6 time ids)."""

    def __init__(self, resolution=1024, center_crop=False, random_flip=False):
        self.resolution = resolution
        self.center_crop = center_crop
        self.random_flip = random_flip

    def __call__(self, image):
        if not torch.is_tensor(image):  # PIL
            import numpy as np
            image = torch.from_numpy(np.asarray(image.convert("RGB"), dtype="float32") / 255.0).permute(2, 0, 1)
        res = self.resolution
        h0, w0 = int(image.shape[-2]), int(image.shape[-1])
        original_size = [h0, w0]
        # T.Resize(int): the SHORTER side becomes `resolution`, aspect ratio kept (bilinear)
        if h0 <= w0:
            nh, nw = res, max(res, int(res * w0 / h0))
        else:
            nh, nw = max(res, int(res * h0 / w0)), res
        image = torch.nn.functional.interpolate(image[None].float(), size=(nh, nw), mode="bilinear", align_corners=False,
                                                antialias=True)[0]
        if self.center_crop:
            y1 = max(0, int(round((nh - res) / 2.0)))
            x1 = max(0, int(round((nw - res) / 2.0)))
        else:
            y1 = random.randint(0, nh - res)
            x1 = random.randint(0, nw - res)
        image = image[:, y1:y1 + res, x1:x1 + res]
        if self.random_flip and random.random() < 0.5:
            x1 = nw - x1  # as the reference computes it (:39-42)
            image = image.flip(-1)
        crop_top_left = [y1, x1]
        return (image - 0.5) / 0.5, list(original_size + crop_top_left + [res, res])


class StableDiffusionXLHead(StableDiffusionHead):
    def __init__(self, diffusion_name_or_path, pretrained_model_name_or_path: str = None, projector_type="linear",
                 projector_depth: int = 1, projector_name_or_path: str = None, embed_hidden_size: int = 4096,
                 global_condition_hidden_size: int = 1280, drop_prob: float | None = None, noise_offset: float = 0.0,
                 input_perturbation: float = 0.0, snr_gamma: float | None = None, resolution: int = 1024,
                 center_crop: bool = True, random_flip: bool = True, freeze_vae: bool = True, freeze_unet: bool = True,
                 freeze_projector: bool = False, local_files_only: bool = True):
        super().__init__(diffusion_name_or_path=diffusion_name_or_path, pretrained_model_name_or_path=None,
                         projector_type=projector_type, projector_depth=projector_depth,
                         projector_name_or_path=projector_name_or_path, embed_hidden_size=embed_hidden_size,
                         drop_prob=drop_prob, noise_offset=noise_offset, input_perturbation=input_perturbation,
                         snr_gamma=snr_gamma, resolution=resolution, center_crop=center_crop, random_flip=random_flip,
                         freeze_vae=freeze_vae, freeze_unet=freeze_unet, freeze_projector=freeze_projector,
                         local_files_only=local_files_only)
        self.save_model_name = "stable_diffusion_xl_head"
        self.pretrained_model_name_or_path = pretrained_model_name_or_path
        self.global_condition_hidden_size = global_condition_hidden_size
        # a second, global projector, like the two CLIP encoders of unCLIP / SDXL (:93-103)
        projector_cfg = dict(projector=projector_type, freeze_projector=freeze_projector, depth=projector_depth,
                             save_model_name=self.save_model_name, model_name_or_path=None)
        # The reference calls build_projector without `bias` here (:103), which trips LinearProjector's own
        # `assert bias is not None` (projector/mlp_projector.py:18); bias=False is the convention of the head's other
        # projector (modeling_plugins.py:389-391) and keeps the state_dict to `global_projector.projector.weight`.
        self.global_projector = build_projector(projector_cfg, in_hidden_size=embed_hidden_size,
                                                out_hidden_size=self.global_condition_hidden_size, bias=False)
        if not self.global_projector.load_model(projector_name_or_path):
            self.global_projector.apply(self._init_weights)
        if pretrained_model_name_or_path is not None:
            self.load_model(pretrained_model_name_or_path)
        self.global_projector.requires_grad_(not freeze_projector)

    @property
    def processor(self):
        return SDXLDataProcessor(resolution=1024, center_crop=False, random_flip=False)

    @property
    def config(self):
        return dict(diffusion_name_or_path=self.diffusion_name_or_path,
                    pretrained_model_name_or_path=self.pretrained_model_name_or_path,
                    embed_hidden_size=self.embed_hidden_size,
                    global_condition_hidden_size=self.global_condition_hidden_size, drop_prob=self.drop_prob,
                    noise_offset=self.noise_offset, input_perturbation=self.input_perturbation, snr_gamma=self.snr_gamma,
                    freeze_vae=self.freeze_vae, freeze_unet=self.freeze_unet, freeze_projector=self.freeze_projector)

    def fsdp_ignored_modules(self) -> list:
        ignored = []
        if self.freeze_vae:
            ignored.append(self.vae)
        if self.freeze_unet:
            ignored.append(self.unet)
        if self.freeze_projector:
            ignored.append(self.projector)
            ignored.append(self.global_projector)
        return ignored

    def to(self, *args, **kwargs):
        """The reference (:140-149) moves only unet + projectors and keeps the VAE in fp32 (fp16 overflow guard).  Here the
        VAE runs on the bf16 HIP kernels, so it follows the module; the call signature and the warning are kept."""
        out = super().to(*args, **kwargs)
        dtype = kwargs.get("dtype", next((a for a in args if isinstance(a, torch.dtype)), None))
        if dtype is not None and dtype != torch.bfloat16:
            logger.warning("the HIP VAE/UNet compute in bfloat16; requested dtype: {}.".format(dtype))
        return out

    def _time_ids(self, add_time_ids, n, device):
        if add_time_ids is None:  # full-frame default, as `pipeline` builds it (:357-363)
            s = self.unet.config.sample_size * self.vae_scale_factor
            add_time_ids = torch.tensor([[s, s, 0, 0, s, s]], dtype=torch.float32).repeat(n, 1)
        if not torch.is_tensor(add_time_ids):
            add_time_ids = torch.tensor(add_time_ids, dtype=torch.float32)
        return add_time_ids.to(device=device, dtype=torch.float32).reshape(n, -1)

    def forward(self, images=None, encoder_hidden_states=None, u_encoder_hidden_states=None, add_time_ids=None,
                dream_embeddings=None, noise=None, timesteps=None):
        """dreamllm_sdxl/modeling_plugins.py:151-236.  `noise` / `timesteps` may be injected (tests, benchmarks)."""
        is_dummy = images is None
        if is_dummy:
            assert dream_embeddings is not None, "You must provide `dream_embeddings` when dummy forward."
            dummy = torch.zeros(1, dream_embeddings.shape[1], self.embed_hidden_size, device=self.device, dtype=self.dtype)
            dummy_local = self.projector(dummy)[-1]
            # the reference pushes the *projected* dummy through the global projector (:165), which only type-checks when
            # cross_attention_dim == embed_hidden_size; the zero tensor of the right width gives the same zero loss and
            # touches the same parameters
            dummy_global = self.global_projector(dummy.mean(1))[-1]
            return (0.0 * dummy_local).sum() + (0.0 * dummy_global).sum() + (0.0 * dream_embeddings).sum()

        latents, noise, timesteps, noisy_latents = self._noised_latents(images, encoder_hidden_states, noise, timesteps)
        bsz = latents.shape[0]

        global_states = encoder_hidden_states.mean(1)  # [N, D]: mean over the dream queries (:198); tiny, stays in torch
        global_states = self.global_projector(global_states)[-1]
        encoder_hidden_states = self.projector(encoder_hidden_states)[-1]

        if self.noise_scheduler.config.prediction_type == "epsilon":
            target = noise
        elif self.noise_scheduler.config.prediction_type == "v_prediction":
            target = self.noise_scheduler.get_velocity(latents, noise, timesteps)
        else:
            raise ValueError(f"Unknown prediction type {self.noise_scheduler.config.prediction_type}")

        added = {"time_ids": self._time_ids(add_time_ids, bsz, latents.device), "text_embeds": global_states}
        model_pred = self.unet(noisy_latents, timesteps, encoder_hidden_states, added_cond_kwargs=added).sample
        if self.snr_gamma is None:
            loss = ops.mse_loss(model_pred, target.float())
        else:
            snr = self._compute_snr(timesteps)
            if self.noise_scheduler.config.prediction_type == "v_prediction":
                snr = snr + 1
            w = torch.stack([snr, self.snr_gamma * torch.ones_like(snr)], dim=1).min(dim=1)[0] / snr
            loss = (ops.mse_loss_per_sample(model_pred, target.float()) * w).mean()
        return loss

    @torch.no_grad()
    def pipeline(self, height: int | None = None, width: int | None = None, num_inference_steps: int = 100,
                 guidance_scale: float = 7.5, num_images_per_prompt: int | None = 1, eta: float = 0.0, generator=None,
                 latents=None, prompt_embeds=None, negative_prompt_embeds=None,
                 output_type: Literal["latent", "pt", "np", "pil"] | None = "pil",
                 callback: Callable[[int, int, torch.FloatTensor], None] | None = None, callback_steps: int = 1,
                 cross_attention_kwargs: dict[str, Any] | None = None, guidance_rescale: float = 0.0,
                 original_size: Optional[Tuple[int, int]] = None, crops_coords_top_left: Tuple[int, int] = (0, 0),
                 target_size: Optional[Tuple[int, int]] = None, scheduler=None, use_graph: bool = True):
        """dreamllm_sdxl/modeling_plugins.py:239-445."""
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(height, width, callback_steps, prompt_embeds, negative_prompt_embeds)
        batch_size = prompt_embeds.shape[0]
        device = self.device
        do_cfg = guidance_scale > 1.0
        assert prompt_embeds is not None, "`prompt_embeds` must be provided by LLM."
        prompt_embeds = prompt_embeds.to(self.dtype)
        global_prompt = self.global_projector(prompt_embeds.mean(1))[-1]
        prompt_embeds = self.projector(prompt_embeds)[-1]
        add_text_embeds = global_prompt
        if original_size is None:
            full = self.unet.config.sample_size * (2 ** (len(self.vae.config.block_out_channels) - 1))
            original_size = [full, full]
        if target_size is None:
            target_size = original_size
        add_time_ids = torch.tensor([list(original_size) + list(crops_coords_top_left) + list(target_size)],
                                    dtype=torch.float32, device=device).repeat(batch_size, 1)
        if do_cfg:
            assert negative_prompt_embeds is not None, \
                "When using classifier free guidance, `negative_prompt_embeds` must be provided by LLM."
            negative_prompt_embeds = negative_prompt_embeds.to(self.dtype)
            global_negative = self.global_projector(negative_prompt_embeds.mean(1))[-1]
            negative_prompt_embeds = self.projector(negative_prompt_embeds)[-1]
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)
            add_text_embeds = torch.cat([global_negative, global_prompt], dim=0)
            add_time_ids = torch.cat([add_time_ids, add_time_ids], dim=0)
        sched = scheduler if scheduler is not None else self.noise_scheduler
        sched.set_timesteps(num_inference_steps, device=device)
        timesteps = sched.timesteps
        latents = self.prepare_latents(batch_size * num_images_per_prompt, self.unet.config.in_channels, height, width,
                                       torch.float32, device, generator, latents)
        added = {"text_embeds": add_text_embeds, "time_ids": add_time_ids}
        fused = (use_graph and do_cfg and guidance_rescale == 0.0 and eta == 0.0 and hasattr(sched, "step_cfg_fused_")
                 and callback is None and num_images_per_prompt == 1 and self.unet.config.in_channels == 4)
        if fused:
            latents = self._denoise_loop_graph(latents, prompt_embeds, timesteps.tolist(), sched, guidance_scale,
                                               added_cond_kwargs=added)
        else:
            ctx = self.unet.prepare_context(prompt_embeds)
            for i, t in enumerate(timesteps.tolist()):
                model_in = torch.cat([latents] * 2) if do_cfg else latents
                model_in = sched.scale_model_input(model_in, t)
                noise_pred = self.unet(model_in.to(self.dtype), t, encoder_hidden_states=prompt_embeds, context_cache=ctx,
                                       added_cond_kwargs=added, return_dict=False)[0].float()
                if do_cfg:
                    noise_pred_uncond, noise_pred_text = noise_pred.chunk(2)
                    noise_pred = noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond)
                    if guidance_rescale > 0.0:
                        noise_pred = self._rescale_noise_cfg(noise_pred, noise_pred_text, guidance_rescale=guidance_rescale)
                latents = sched.step(noise_pred, t, latents, eta=eta, generator=generator)
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, latents)
        if output_type == "latent":
            return latents
        image = self.vae.decode((latents / self.vae.config.scaling_factor).to(self.dtype))
        image = (image.float() / 2 + 0.5).clamp(0, 1)
        if output_type == "pt":
            return image
        image = image.permute(0, 2, 3, 1).cpu().numpy()
        if output_type == "np":
            return image
        import PIL.Image
        return [PIL.Image.fromarray((im * 255).round().astype("uint8")) for im in image]
