"""Omni plugin contract on HIP kernels -- API mirror of omni/models/dreamllm/modeling_plugins.py (drop-in boundary b1).

`PluginBase` / `MultimodalEmbedding` / `MultimodalHead` keep the reference's abstract surface (modeling_plugins.py:32-112):
`processor`, `config`, `save_model`, `load_model`, `forward`, `embed_len`/`embed_dim`, `pipeline`, `fsdp_ignored_modules`,
class attrs `initializer_range`, `plugin_type`.  The three concrete plugins keep their constructor kwargs, attribute
names, checkpoint file names and state_dict keys, so `projects/dreamllm/configs/common.py` switches to this implementation
by changing only `_class_=`.  Pretrained-weight *download* (HF hub) is host I/O outside the accelerated path: weights are
loaded from local checkpoints (`{save_model_name}.bin`, or the diffusers/transformers folder layout) when a path is
given and randomly initialised otherwise (the benchmark and tests run on random weights).
"""
from __future__ import annotations

import os
from abc import ABC, abstractmethod
from typing import Any, Callable, Literal

import torch
from torch import nn

from . import ops
from .projector import build_projector
from . import utils as _u
from .utils import FSDPMixin, check_path_and_file, get_model_device, get_model_dtype, logger, randn_tensor

PluginType = Literal["embedding", "head"]
PipelineImageType = Any


class PluginBase(ABC, nn.Module, FSDPMixin):
    """modeling_plugins.py:32-81."""

    initializer_range: float = 0.02
    plugin_type: PluginType | None = None

    def _init_weights(self, module):
        std = self.initializer_range
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()
        elif isinstance(module, nn.Parameter):
            module.data.normal_(mean=0.0, std=std)

    @property
    def device(self):
        return get_model_device(self)

    @property
    def dtype(self):
        return get_model_dtype(self)

    @property
    @abstractmethod
    def processor(self):
        pass

    @property
    @abstractmethod
    def config(self):
        pass

    @abstractmethod
    def save_model(self, output_dir: str):
        pass

    @abstractmethod
    def load_model(self, output_dir: str):
        pass

    @abstractmethod
    def forward(self):
        pass

    def _save(self, output_dir):
        torch.save(self.state_dict(), os.path.join(output_dir, f"{self.save_model_name}.bin"))

    def _load_bin(self, output_dir) -> bool:
        if check_path_and_file(output_dir, f"{self.save_model_name}.bin"):
            logger.info(f">>> loading `{type(self).__name__}`... from {output_dir}")
            sd = torch.load(os.path.join(output_dir, f"{self.save_model_name}.bin"), map_location="cpu")
            # strict, like the reference (modeling_plugins.py:157,296,446): a missing or renamed key must not pass silently.
            # The one tolerated difference: `...embeddings.position_ids`, a persistent buffer of CLIP towers saved by
            # transformers < 4.31 that newer towers (and this one) recompute.
            sd = {k: v for k, v in sd.items() if not k.endswith("embeddings.position_ids")}
            status = self.load_state_dict(sd, strict=True)
            logger.info(f"{status}")
            return True
        return False


class MultimodalEmbedding(PluginBase):
    """modeling_plugins.py:84-102."""

    initializer_range: float = 0.02
    plugin_type: PluginType | None = "embedding"

    @property
    @abstractmethod
    def embed_len(self):
        pass

    @property
    @abstractmethod
    def embed_dim(self):
        pass


class MultimodalHead(PluginBase):
    """modeling_plugins.py:105-112."""

    initializer_range: float = 0.02
    plugin_type: PluginType | None = "head"

    @abstractmethod
    @torch.no_grad()
    def pipeline(self):
        pass


# ------------------------------------------------------------------------------------------------ embedding modules
class DreamEmbedding(MultimodalEmbedding):
    """modeling_plugins.py:116-181: the learned dream queries."""

    def __init__(self, pretrained_model_name_or_path: str | None = None, num_dream_queries: int = 64,
                 embed_hidden_size: int = 4096, freeze_dream_queries: bool = False):
        super().__init__()
        self.save_model_name = "dream_embedding"
        self.pretrained_model_name_or_path = pretrained_model_name_or_path
        self.num_dream_queries = num_dream_queries
        self.embed_hidden_size = embed_hidden_size
        self.freeze_dream_queries = freeze_dream_queries
        self.dream_queries = nn.Parameter(torch.zeros(1, self.num_dream_queries, self.embed_hidden_size))
        self._init_weights(self.dream_queries)
        if pretrained_model_name_or_path is not None:
            self.load_model(pretrained_model_name_or_path)
        self.dream_queries.requires_grad_(not freeze_dream_queries)

    def fsdp_ignored_modules(self) -> list:
        return [self] if self.freeze_dream_queries else []

    @property
    def processor(self):
        return None

    @property
    def embed_len(self):
        return self.num_dream_queries

    @property
    def embed_dim(self):
        return self.embed_hidden_size

    @property
    def config(self):
        return dict(pretrained_model_name_or_path=self.pretrained_model_name_or_path,
                    num_dream_queries=self.num_dream_queries, embed_len=self.embed_len, embed_dim=self.embed_dim,
                    freeze_dream_queries=self.freeze_dream_queries)

    def save_model(self, output_dir: str):
        self._save(output_dir)

    def load_model(self, output_dir: str):
        self._load_bin(output_dir)
        if check_path_and_file(output_dir, "dream_queries.pt"):  # HACK: For compatibility (modeling_plugins.py:177-178)
            self.dream_queries = torch.load(os.path.join(output_dir, "dream_queries.pt"), map_location="cpu")

    def forward(self, batch_size: int = 1):
        return self.dream_queries.repeat(batch_size, 1, 1)


class CLIPVisionEmbedding(MultimodalEmbedding):
    """modeling_plugins.py:184-331: CLIP-ViT encoder (frozen) -> hidden_states[select_layer][:, 1:] -> projector.

    `clip_vision_model_name_or_path` may be a local HF folder (config.json + weights) or a dict / CLIPVisionConfig-like
    object describing the architecture (random init).  The encoder runs on the HIP kernels (`clip_vit.HipCLIPVisionModel`,
    HF state_dict keys) and skips the layers after `select_layer`, which the reference computes and discards."""

    def __init__(self, clip_vision_model_name_or_path, projector_type: str = "linear", projector_depth: int = 1,
                 projector_name_or_path: str = None, pretrained_model_name_or_path: str | None = None,
                 use_additional_post_layernorm: bool = False, select_layer: int = -2, embed_hidden_size: int = 4096,
                 freeze_clip_vision_model: bool = True, freeze_embedding_layers: bool = True, freeze_projector: bool = False,
                 local_files_only: bool = False):
        super().__init__()
        from .clip_vit import HipCLIPVisionModel, load_clip_config, CLIPImageProcessorLite

        self.save_model_name = "clip_vision_embedding"
        self.clip_vision_model_name_or_path = clip_vision_model_name_or_path
        self.projector_type = projector_type
        self.projector_depth = projector_depth
        self.projector_name_or_path = projector_name_or_path
        self.pretrained_model_name_or_path = pretrained_model_name_or_path
        self.use_additional_post_layernorm = use_additional_post_layernorm
        self.select_layer = select_layer
        self.embed_hidden_size = embed_hidden_size
        self.freeze_clip_vision_model = freeze_clip_vision_model
        self.freeze_embedding_layers = freeze_embedding_layers
        self.freeze_projector = freeze_projector
        if not freeze_clip_vision_model:
            raise NotImplementedError("the HIP CLIP encoder is forward-only; every dreamllm recipe freezes it "
                                      "(projects/dreamllm/configs/common.py:35)")
        clip_cfg = load_clip_config(clip_vision_model_name_or_path)
        self.clip_image_processor = CLIPImageProcessorLite(clip_cfg)
        self.clip_vision_model = HipCLIPVisionModel(clip_cfg)
        self.clip_vision_model.load_pretrained(clip_vision_model_name_or_path)
        projector_cfg = dict(projector=projector_type, freeze_projector=freeze_projector, depth=projector_depth,
                             save_model_name=self.save_model_name, model_name_or_path=None)
        self.projector = build_projector(projector_cfg, in_hidden_size=self.clip_vision_model.config.hidden_size,
                                         out_hidden_size=embed_hidden_size, bias=True)
        self.projector.apply(self._init_weights)
        if use_additional_post_layernorm:
            self.post_layernorm = ops.HipLayerNorm(embed_hidden_size, eps=self.clip_vision_model.config.layer_norm_eps)
        else:
            self.post_layernorm = nn.Identity()
        self.image_embed_len = (self.clip_vision_model.config.image_size // self.clip_vision_model.config.patch_size) ** 2
        if pretrained_model_name_or_path is not None:
            self.load_model(pretrained_model_name_or_path)
        if self.projector.load_model(projector_name_or_path):
            logger.info(f">>> loading `CLIPVisionEmbedding` projector from {projector_name_or_path}")
        self.clip_vision_model.requires_grad_(False)
        self.projector.requires_grad_(not freeze_projector)

    @property
    def processor(self):
        return self.clip_image_processor

    @property
    def embed_len(self):
        return self.image_embed_len

    @property
    def embed_dim(self):
        return self.embed_hidden_size

    @property
    def config(self) -> dict:
        return dict(clip_vision_model_name_or_path=self.clip_vision_model_name_or_path,
                    clip_vision_model_config=self.clip_vision_model.config.to_dict(),
                    pretrained_model_name_or_path=self.pretrained_model_name_or_path, select_layer=self.select_layer,
                    embed_len=self.embed_len, embed_dim=self.embed_dim,
                    freeze_clip_vision_model=self.freeze_clip_vision_model,
                    freeze_embedding_layers=self.freeze_embedding_layers, freeze_projector=self.freeze_projector)

    def fsdp_ignored_modules(self) -> list:
        ignored = [self.clip_vision_model]
        if self.freeze_projector:
            ignored.append(self.projector)
        return ignored

    def save_model(self, output_dir: str):
        self._save(output_dir)

    def load_model(self, output_dir: str):
        if self._load_bin(output_dir):
            return
        if check_path_and_file(output_dir, "clip_vision_model_projector.pt"):  # HACK: For compatibility
            self.projector.load_state_dict(torch.load(os.path.join(output_dir, "clip_vision_model_projector.pt"),
                                                      map_location="cpu"))
            return
        raise FileNotFoundError(f"{output_dir}: no {self.save_model_name}.bin (hub download is outside this path)")

    def forward(self, images: torch.FloatTensor | None = None):
        """modeling_plugins.py:314-331."""
        is_dummy = images is None
        if is_dummy:  # HACK: dummy forward so every trainable parameter gets a gradient (no find_unused_parameters)
            # the frozen encoder's output on a zero image is a constant: only the projector needs to see a tensor
            feats = torch.zeros(1, self.image_embed_len, self.clip_vision_model.config.hidden_size, device=self.device,
                                dtype=self.dtype)
        else:
            feats = self.clip_vision_model.encode(images.to(self.dtype), self.select_layer)[:, 1:]
        image_embeds = self.projector(feats)[-1]
        image_embeds = self.post_layernorm(image_embeds)
        if is_dummy:
            return (0.0 * image_embeds).sum()
        return image_embeds


# ------------------------------------------------------------------------------------------------ head modules
class StableDiffusionHead(MultimodalHead):
    """modeling_plugins.py:335-850: frozen VAE + frozen UNet + DDPM scheduler + trainable condition projector.

    `diffusion_name_or_path` is a local diffusers folder (unet/, vae/, scheduler/ sub-folders) or a preset name /
    config dict for random init ("sd21-base", "tiny").  The UNet runs on the HIP kernels in NHWC
    (`unet.HipUNet2DConditionModel`, diffusers state_dict keys); the VAE is the stock PyTorch-ROCm module (SURVEY.md
    §2.2: not on the accelerated path); the scheduler arithmetic is restated in `schedulers.py`."""

    def __init__(self, diffusion_name_or_path, projector_type="linear", projector_depth: int = 1,
                 projector_name_or_path: str = None, pretrained_model_name_or_path: str = None,
                 embed_hidden_size: int = 4096, drop_prob: float | None = None, noise_offset: float = 0.0,
                 input_perturbation: float = 0.0, snr_gamma: float | None = None, resolution: int = 512,
                 center_crop: bool = True, random_flip: bool = True, freeze_vae: bool = True, freeze_unet: bool = True,
                 freeze_projector: bool = False, local_files_only: bool = False):
        super().__init__()
        from .schedulers import DDPMScheduler
        from .unet import HipUNet2DConditionModel, load_unet_config
        from .vae import AutoencoderKLLite, load_vae_config

        self.save_model_name = "stable_diffusion_head"
        self.diffusion_name_or_path = diffusion_name_or_path
        self.projector_type = projector_type
        self.projector_depth = projector_depth
        self.projector_name_or_path = projector_name_or_path
        self.pretrained_model_name_or_path = pretrained_model_name_or_path
        self.embed_hidden_size = embed_hidden_size
        self.drop_prob = drop_prob
        self.noise_offset = noise_offset
        self.input_perturbation = input_perturbation
        self.snr_gamma = snr_gamma
        self.resolution = resolution
        self.center_crop = center_crop
        self.random_flip = random_flip
        self.freeze_vae = freeze_vae
        self.freeze_unet = freeze_unet
        self.freeze_projector = freeze_projector
        if not (freeze_vae and freeze_unet):
            raise NotImplementedError("the HIP UNet implements forward + input/condition gradients only (frozen weights), "
                                      "as in every dreamllm recipe (projects/dreamllm/configs/common.py:52-53)")
        self.vae = AutoencoderKLLite(load_vae_config(diffusion_name_or_path))
        self.unet = HipUNet2DConditionModel(load_unet_config(diffusion_name_or_path))
        self.noise_scheduler = DDPMScheduler.from_name_or_path(diffusion_name_or_path)
        self.vae.load_pretrained(diffusion_name_or_path, "vae")
        self.unet.load_pretrained(diffusion_name_or_path, "unet")
        projector_cfg = dict(projector=projector_type, freeze_projector=freeze_projector, depth=projector_depth,
                             save_model_name=self.save_model_name, model_name_or_path=None)
        self.projector = build_projector(projector_cfg, in_hidden_size=embed_hidden_size,
                                         out_hidden_size=self.unet.config.cross_attention_dim, bias=False)
        self.projector.apply(self._init_weights)
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)
        if pretrained_model_name_or_path is not None:
            self.load_model(pretrained_model_name_or_path)
        if self.projector.load_model(projector_name_or_path):
            logger.info(">>> loading `StableDiffusionHead` projector from {projector_name_or_path}")
        self.vae.requires_grad_(False)
        self.unet.requires_grad_(False)
        self.projector.requires_grad_(not freeze_projector)

    @property
    def processor(self):
        """Host-side preprocessing (modeling_plugins.py:410-420): resize/crop/flip/normalise to [-1, 1]."""
        res = self.resolution

        def _proc(img: torch.Tensor):  # img: float [3,H,W] in [0,1]
            img = torch.nn.functional.interpolate(img[None], size=(res, res), mode="bilinear", align_corners=False)[0]
            return (img - 0.5) / 0.5

        return _proc

    @property
    def config(self):
        return dict(diffusion_name_or_path=self.diffusion_name_or_path,
                    pretrained_model_name_or_path=self.pretrained_model_name_or_path,
                    embed_hidden_size=self.embed_hidden_size, drop_prob=self.drop_prob, noise_offset=self.noise_offset,
                    input_perturbation=self.input_perturbation, snr_gamma=self.snr_gamma, freeze_vae=self.freeze_vae,
                    freeze_unet=self.freeze_unet, freeze_projector=self.freeze_projector)

    def fsdp_ignored_modules(self) -> list:
        ignored = [self.vae, self.unet]
        if self.freeze_projector:
            ignored.append(self.projector)
        return ignored

    def save_model(self, output_dir: str):
        self._save(output_dir)

    def load_model(self, output_dir: str):
        if self._load_bin(output_dir):
            return
        if check_path_and_file(output_dir, "unet_projector.pt"):  # HACK: For compatibility
            self.projector.load_state_dict(torch.load(os.path.join(output_dir, "unet_projector.pt"), map_location="cpu"))
            return
        raise FileNotFoundError(f"{output_dir}: no {self.save_model_name}.bin (hub download is outside this path)")

    def _compute_snr(self, timesteps):
        """modeling_plugins.py:468-491."""
        ac = self.noise_scheduler.alphas_cumprod.to(device=timesteps.device)
        alpha = (ac**0.5)[timesteps].float()
        sigma = ((1.0 - ac) ** 0.5)[timesteps].float()
        return (alpha / sigma) ** 2

    def _noised_latents(self, images, encoder_hidden_states, noise=None, timesteps=None):
        """modeling_plugins.py:510-536: VAE-encode (sampled), scale, draw noise (+offset, +perturbation) and timesteps, add
        noise.  The random numbers are drawn in the reference's order: VAE sample, noise, offset, perturbation, timesteps."""
        with torch.no_grad():
            dist = self.vae.encode(images.to(self.dtype))
            latents = dist.sample(noise=_u.draws.randn(dist.mean.shape, device=dist.mean.device, dtype=dist.mean.dtype))
            latents = latents * self.vae.config.scaling_factor
        assert (
            encoder_hidden_states.shape[0] == latents.shape[0]
        ), f"encoder_hidden_states.shape[0]: {encoder_hidden_states.shape[0]} != latents.shape[0]: {latents.shape[0]}"
        bsz = latents.shape[0]
        if noise is None:
            noise = _u.draws.randn_like(latents)
        if self.noise_offset:
            noise = noise + self.noise_offset * _u.draws.randn((bsz, latents.shape[1], 1, 1), device=latents.device,
                                                              dtype=latents.dtype)
        new_noise = noise + self.input_perturbation * _u.draws.randn_like(noise) if self.input_perturbation else noise
        if timesteps is None:
            timesteps = _u.draws.randint(0, self.noise_scheduler.config.num_train_timesteps, (bsz,), device=latents.device)
        timesteps = timesteps.long()
        return latents, noise, timesteps, self.noise_scheduler.add_noise(latents, new_noise, timesteps)

    def forward(self, images=None, encoder_hidden_states=None, u_encoder_hidden_states=None, dream_embeddings=None,
                noise=None, timesteps=None):
        """modeling_plugins.py:493-577.  `noise` / `timesteps` may be injected (tests, reproducible benchmarks); when
        None they are sampled exactly where the reference samples them."""
        is_dummy = images is None
        if is_dummy:
            assert dream_embeddings is not None, "You must provide `dream_embeddings` when dummy forward."
            dummy = torch.zeros(1, dream_embeddings.shape[1], self.embed_hidden_size, device=self.device, dtype=self.dtype)
            dummy = self.projector(dummy)[-1]
            return (0.0 * dummy).sum() + (0.0 * dream_embeddings).sum()

        latents, noise, timesteps, noisy_latents = self._noised_latents(images, encoder_hidden_states, noise, timesteps)
        bsz = latents.shape[0]

        if u_encoder_hidden_states is not None and self.drop_prob is not None:
            mask = _u.draws.bernoulli(torch.zeros(bsz) + self.drop_prob).to(latents.device)[:, None, None]
            mask = mask.to(encoder_hidden_states.dtype)
            encoder_hidden_states = (1.0 - mask) * encoder_hidden_states + mask * u_encoder_hidden_states

        encoder_hidden_states = self.projector(encoder_hidden_states)[-1]
        if self.noise_scheduler.config.prediction_type == "epsilon":
            target = noise
        elif self.noise_scheduler.config.prediction_type == "v_prediction":
            target = self.noise_scheduler.get_velocity(latents, noise, timesteps)
        else:
            raise ValueError(f"Unknown prediction type {self.noise_scheduler.config.prediction_type}")

        model_pred = self.unet(noisy_latents, timesteps, encoder_hidden_states).sample
        if self.snr_gamma is None:
            loss = ops.mse_loss(model_pred, target.float())
        else:
            snr = self._compute_snr(timesteps)
            if self.noise_scheduler.config.prediction_type == "v_prediction":
                snr = snr + 1
            w = torch.stack([snr, self.snr_gamma * torch.ones_like(snr)], dim=1).min(dim=1)[0] / snr
            per = ops.mse_loss_per_sample(model_pred, target.float())
            loss = (per * w).mean()
        return loss

    # ---- inference pipeline ------------------------------------------------------------------------------------
    def check_inputs(self, height, width, callback_steps, prompt_embeds=None, negative_prompt_embeds=None):
        """modeling_plugins.py:579-606."""
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if (callback_steps is None) or (not isinstance(callback_steps, int) or callback_steps <= 0):
            raise ValueError(
                f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")
        if prompt_embeds is None:
            raise ValueError("Provide `prompt_embeds`. Cannot leave `prompt_embeds` undefined.")
        if prompt_embeds is not None and negative_prompt_embeds is not None:
            if prompt_embeds.shape != negative_prompt_embeds.shape:
                raise ValueError(
                    "`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, but"
                    f" got: `prompt_embeds` {prompt_embeds.shape} != `negative_prompt_embeds`"
                    f" {negative_prompt_embeds.shape}.")

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        """modeling_plugins.py:608-623."""
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(
                f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            latents = randn_tensor(shape, generator=generator, device=device, dtype=dtype)
        else:
            latents = latents.to(device)
        return latents * self.noise_scheduler.init_noise_sigma

    def _rescale_noise_cfg(self, noise_cfg, noise_pred_text, guidance_rescale=0.0):
        """modeling_plugins.py:658-669."""
        std_text = noise_pred_text.std(dim=list(range(1, noise_pred_text.ndim)), keepdim=True)
        std_cfg = noise_cfg.std(dim=list(range(1, noise_cfg.ndim)), keepdim=True)
        noise_pred_rescaled = noise_cfg * (std_text / std_cfg)
        return guidance_rescale * noise_pred_rescaled + (1 - guidance_rescale) * noise_cfg

    @torch.no_grad()
    def _denoise_loop_graph(self, latents, ctx_embeds, timesteps, sched, guidance_scale, added_cond_kwargs=None):
        """Deterministic-DDIM + CFG loop body (modeling_plugins.py:809-839) as: [fill t] -> hipGraph replay of one UNet
        forward on static NHWC buffers -> ONE fused kernel (CFG combine + DDIM update + next UNet input).  ~600 kernel
        launches per step collapse into one graph launch, which is what the loop is bound by at B_img = 1.  The graph
        (and its static buffers, including the cross-attention K/V of the conditioning tokens) is cached per shape."""
        B, C, H, W = latents.shape
        key = (B, H, W, tuple(ctx_embeds.shape), added_cond_kwargs is not None)
        cache = getattr(self, "_graph_cache", None)
        if cache is None:
            cache = self._graph_cache = {}
        dev = latents.device
        ent = cache.get(key)
        ctx_now = self.unet.prepare_context(ctx_embeds)
        # everything that depends on the timestep only (sinusoid, time-embedding MLP, every ResBlock's time_emb_proj: ~38 tiny
        # launches per step) is computed for ALL steps here, one GEMM per ResBlock; the captured forward reads one row of it
        addk = None if added_cond_kwargs is None else {k: v.to(dev) for k, v in added_cond_kwargs.items()}
        # ONE host conversion of the schedule (a device tensor would cost a sync per element); without micro-conditioning the table
        # depends on (schedule, batch) only and is kept across calls
        ts_host = tuple(float(t) for t in (timesteps.tolist() if torch.is_tensor(timesteps) else timesteps))
        # every parameter the table is computed from -- time-embedding MLP, each ResBlock's time_emb_proj weight AND bias -- with its
        # storage pointer, dtype and device beside identity / version: `module.to()` swaps `.data` without touching either
        def _sig(p):
            return (id(p), p._version, p.data_ptr(), p.dtype, str(p.device))
        w_sig = tuple(_sig(p) for p in self.unet.time_embedding.parameters()) + \
            tuple(_sig(p) for r in self.unet._resnets() for p in r.time_emb_proj.parameters())
        tb_key = (ts_host, 2 * B, w_sig)
        tb_cache = getattr(self, "_time_bias_cache", None)
        if addk is None and tb_cache is not None and tb_cache[0] == tb_key:
            tb_table = tb_cache[1]
        else:
            tb_table = self.unet.precompute_time_bias(list(ts_host), 2 * B, addk)
            self._time_bias_cache = (tb_key, tb_table) if addk is None else None
        if ent is None:
            x_in = torch.zeros(2 * B, H, W, 8, dtype=self.dtype, device=dev)
            tb_static = tb_table[0].clone()
            ctx_static = {k: [t.clone() for t in v] for k, v in ctx_now.items()}
            emb_static = ctx_embeds.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # warm-up outside capture (lazy caches, kernel attributes)
                for _ in range(2):
                    self.unet(x_in, None, emb_static, context_cache=ctx_static, nhwc_io=True, return_dict=False, time_bias=tb_static)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # capture ON the warm-up stream: the per-(device, stream) buffers of the operators (GroupNorm meeting slots, stream-K
            # workspace, split-K counters) were allocated by the warm-up calls above, outside the capture -- a different capture
            # stream would allocate new ones from the graph's private pool
            with torch.cuda.graph(graph, stream=side):
                pred = self.unet(x_in, None, emb_static, context_cache=ctx_static, nhwc_io=True, return_dict=False,
                                 time_bias=tb_static)[0]
            ent = cache[key] = dict(graph=graph, x_in=x_in, tb=tb_static, ctx=ctx_static, pred=pred, stream=side)  # (keeps the stream's handle alive)
        for k, v in ctx_now.items():
            for dst, src in zip(ent["ctx"][k], v):
                dst.copy_(src)
        lat = latents.permute(0, 2, 3, 1).contiguous().float()  # NHWC fp32 master copy
        x_in = ent["x_in"]
        x_in.zero_()
        x_in[:B, ..., :4] = lat.to(self.dtype)
        x_in[B:, ..., :4] = lat.to(self.dtype)
        for i, t in enumerate(ts_host):  # host floats: no per-step device read of the schedule
            ent["tb"].copy_(tb_table[i])
            ent["graph"].replay()
            sched.step_cfg_fused_(ent["pred"], t, lat, x_in, guidance_scale)
        return lat.permute(0, 3, 1, 2).contiguous()

    @torch.no_grad()
    def pipeline(self, height: int | None = None, width: int | None = None, num_inference_steps: int = 50,
                 guidance_scale: float = 7.5, num_images_per_prompt: int | None = 1, eta: float = 0.0,
                 generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None,
                 output_type: Literal["latent", "pt", "np", "pil"] | None = "pil",
                 callback: Callable[[int, int, torch.FloatTensor], None] | None = None, callback_steps: int = 1,
                 cross_attention_kwargs: dict[str, Any] | None = None, guidance_rescale: float = 0.0,
                 scheduler=None, use_graph: bool = True):
        """modeling_plugins.py:671-850.  `scheduler` optionally overrides the head's DDPM scheduler for the loop (the
        benchmark installs the deterministic DDIM eta=0 scheduler)."""
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(height, width, callback_steps, prompt_embeds, negative_prompt_embeds)
        batch_size = prompt_embeds.shape[0]
        device = self.device
        do_cfg = guidance_scale > 1.0
        assert prompt_embeds is not None, "`prompt_embeds` must be provided by LLM."
        prompt_embeds = self.projector(prompt_embeds.to(self.dtype))[-1]
        if do_cfg:
            assert negative_prompt_embeds is not None, \
                "When using classifier free guidance, `negative_prompt_embeds` must be provided by LLM."
            negative_prompt_embeds = self.projector(negative_prompt_embeds.to(self.dtype))[-1]
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])
        sched = scheduler if scheduler is not None else self.noise_scheduler
        sched.set_timesteps(num_inference_steps, device=device)
        timesteps = sched.timesteps
        latents = self.prepare_latents(batch_size * num_images_per_prompt, self.unet.config.in_channels, height, width,
                                       torch.float32, device, generator, latents)
        fused = (use_graph and do_cfg and guidance_rescale == 0.0 and eta == 0.0 and hasattr(sched, "step_cfg_fused_")
                 and callback is None and num_images_per_prompt == 1 and self.unet.config.in_channels == 4)
        if fused:
            latents = self._denoise_loop_graph(latents, prompt_embeds, timesteps.tolist(), sched, guidance_scale)
        else:
            ctx = self.unet.prepare_context(prompt_embeds)  # cross-attention K/V of the dream tokens: once, not per step
            for i, t in enumerate(timesteps.tolist()):
                model_in = torch.cat([latents] * 2) if do_cfg else latents
                model_in = sched.scale_model_input(model_in, t)
                noise_pred = self.unet(model_in.to(self.dtype), t, encoder_hidden_states=prompt_embeds, context_cache=ctx,
                                       return_dict=False)[0].float()
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
