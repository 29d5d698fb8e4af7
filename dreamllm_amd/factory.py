"""Builders for the benchmark / smoke / test models: DreamLLM (Vicuna-7B dims) + CLIP-ViT-L/14 + SD-2.1 head with RANDOM
weights created directly on the device (no checkpoints or network in this environment).  The plugin wiring
goes through the reference's own mechanism: `create_config_init_kwargs` -> `DreamLLMConfig.update_plugins` ->
`init_plugin_modules` (projects/dreamllm/configs/common.py:12-56, projects/dreamllm/train.py:99-139)."""
from __future__ import annotations

import contextlib

import torch

from .configuration_dreamllm import DreamLLMConfig, create_config_init_kwargs
from .modeling_dreamllm import DreamLLMForCausalMLM
from .modeling_plugins import CLIPVisionEmbedding, DreamEmbedding, StableDiffusionHead
from .tokenization_dreamllm import default_special_tokens2ids

VICUNA_7B = dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                 max_position_embeddings=2048, rms_norm_eps=1e-6)
TINY = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, max_position_embeddings=512,
            rms_norm_eps=1e-6)
# miniature plugin architectures for smoke() / `bench.py --model tiny` (every block type of the full-size graphs)
TINY_CLIP = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=56)
TINY_DIFFUSION = dict(
    unet=dict(sample_size=16, in_channels=4, out_channels=4, block_out_channels=(64, 128, 128, 128), layers_per_block=1,
              down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
              up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
              attention_head_dim=(1, 2, 2, 2), transformer_layers_per_block=1, cross_attention_dim=64, norm_num_groups=32,
              norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0, addition_embed_type=None),
    vae=dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1))


@contextlib.contextmanager
def _on(device, dtype):
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(device):
            yield
    finally:
        torch.set_default_dtype(old)


def plugin_configs(hidden_size, clip="openai/clip-vit-large-patch14", diffusion="sd21-base", num_dream_queries=64,
                   with_clip=True, with_sd=True, sdxl=False, global_condition_hidden_size=1280, freeze_clip_projector=False):
    """The three `ConfigAndInitKwargs` of projects/dreamllm/configs/common.py with this package's classes as `_class_`."""
    cfgs = [create_config_init_kwargs(dict(_class_=DreamEmbedding, _name_="dream_embedding", _plugin_type_="embedding",
                                           pretrained_model_name_or_path=None, num_dream_queries=num_dream_queries,
                                           embed_hidden_size=hidden_size, freeze_dream_queries=False))]
    if with_clip:
        cfgs.append(create_config_init_kwargs(dict(
            _class_=CLIPVisionEmbedding, _name_="clip_vision_embedding", _plugin_type_="embedding", projector_type="linear",
            projector_depth=1, clip_vision_model_name_or_path=clip, pretrained_model_name_or_path=None,
            embed_hidden_size=hidden_size, use_additional_post_layernorm=False, select_layer=-2,
            freeze_clip_vision_model=True, freeze_embedding_layers=True, freeze_projector=freeze_clip_projector,
            local_files_only=True)))
    if with_sd and not sdxl:
        cfgs.append(create_config_init_kwargs(dict(
            _class_=StableDiffusionHead, _name_="stable_diffusion_head", _plugin_type_="head", projector_type="linear",
            projector_depth=1, diffusion_name_or_path=diffusion, pretrained_model_name_or_path=None,
            embed_hidden_size=hidden_size, freeze_vae=True, freeze_unet=True, freeze_projector=False, local_files_only=True)))
    if with_sd and sdxl:  # projects/dreamllm_sdxl/configs/common.py:43-58: same plugin NAME, SDXL class
        from .modeling_plugins_sdxl import StableDiffusionXLHead
        cfgs.append(create_config_init_kwargs(dict(
            _class_=StableDiffusionXLHead, _name_="stable_diffusion_head", _plugin_type_="head", projector_type="linear",
            projector_depth=1, diffusion_name_or_path=diffusion, pretrained_model_name_or_path=None,
            embed_hidden_size=hidden_size, global_condition_hidden_size=global_condition_hidden_size, freeze_vae=True,
            freeze_unet=True, freeze_projector=False, local_files_only=True)))
    return cfgs


def build_dreamllm(llm=None, device="cuda", dtype=torch.bfloat16, seed=0, base_vocab=32000, clip="openai/clip-vit-large-patch14",
                   diffusion="sd21-base", with_clip=True, with_sd=True, num_dream_queries=64):
    """Random-init DreamLLM with plugins, built on `device` in `dtype` (weights N(0, 0.02) as `_init_weights`)."""
    llm = dict(VICUNA_7B if llm is None else llm)
    sp = default_special_tokens2ids(base_vocab)
    vocab = base_vocab + 1 + len(sp["additional_special_tokens"])  # [PAD] + 7 added tokens -> 32008
    cfg = DreamLLMConfig(vocab_size=vocab, pad_token_id=sp["[PAD]"], special_tokens2ids_dict=sp, loss_weight_lm=1.0,
                         loss_weight_vm=10.0, **llm)
    for pc in plugin_configs(llm["hidden_size"], clip, diffusion, num_dream_queries, with_clip, with_sd):
        cfg.update_plugins(pc)
    torch.manual_seed(seed)
    with _on(device, dtype):
        model = DreamLLMForCausalMLM(cfg)
        model.init_plugin_modules()
    return model.to(device=device, dtype=dtype)


def build_dreamllm_sdxl(llm=None, device="cuda", dtype=torch.bfloat16, seed=0, base_vocab=32000,
                        clip="openai/clip-vit-large-patch14", diffusion="sdxl-base", with_clip=True, num_dream_queries=196,
                        global_condition_hidden_size=1280, stage1=True):
    """Random-init DreamLLM-SDXL (projects/dreamllm_sdxl/configs/{common,stage1/base}.py): 196 dream queries, SDXL head.
    `stage1=True` applies the stage-I freeze policy of stage1/base.py:27-37,47-53: LLM, embeddings, lm_head, CLIP and its
    projector frozen; dream queries + the head's two projectors trainable; loss weights lm 0 / vm 1."""
    from .modeling_dreamllm_sdxl import DreamLLMSDXLConfig, DreamLLMSDXLForCausalMLM, default_special_tokens2ids as sdxl_ids
    llm = dict(VICUNA_7B if llm is None else llm)
    sp = sdxl_ids(base_vocab)
    vocab = base_vocab + 1 + len(sp["additional_special_tokens"])  # [PAD] + 8 added tokens -> 32009
    cfg = DreamLLMSDXLConfig(vocab_size=vocab, pad_token_id=sp["[PAD]"], special_tokens2ids_dict=sp,
                             loss_weight_lm=0.0 if stage1 else 1.0, loss_weight_vm=1.0 if stage1 else 10.0, **llm)
    for pc in plugin_configs(llm["hidden_size"], clip, diffusion, num_dream_queries, with_clip, True, sdxl=True,
                             global_condition_hidden_size=global_condition_hidden_size, freeze_clip_projector=stage1):
        cfg.update_plugins(pc)
    torch.manual_seed(seed)
    with _on(device, dtype):
        model = DreamLLMSDXLForCausalMLM(cfg)
        model.init_plugin_modules()
    model = model.to(device=device, dtype=dtype)
    if stage1:
        model.model.embed_tokens.requires_grad_(False)
        model.model.layers.requires_grad_(False)
        model.model.norm.requires_grad_(False)
        model.lm_head.requires_grad_(False)
    return model
