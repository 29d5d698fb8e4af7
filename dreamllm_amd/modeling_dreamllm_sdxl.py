"""DreamLLM-SDXL model classes -- omni/models/dreamllm_sdxl/{configuration_dreamllm_sdxl,modeling_dreamllm_sdxl,
tokenization_dreamllm}.py behind the same HIP decoder as `modeling_dreamllm`.

The reference's SDXL model file is its base model file with these differences (diff of the two files), all kept here:
  * class names `DreamLLMSDXL{Config,RMSNorm,Model,ForCausalMLM}` (modeling_dreamllm_sdxl.py:78,803,1208);
  * the tokenizer adds `<dream_patch>` to the additional special tokens (tokenization_dreamllm.py:73-90 of the sdxl
    package), and the unconditional prompt of the CFG-drop pass fills its dream slots with it (:1394);
  * `forward` takes `add_time_ids` and hands it to the head as 4th positional argument (:1357, :1440); the dummy call
    passes `None` in that position (:1444);
  * `rotary_emb.inv_freq` is registered non-persistent (:107) => not in the state_dict;
  * `loss` is divided by `loss_scale` twice (:1485-1487; a no-op for the shipped stage-I recipe, whose schedule gives
    loss_scale = 1 with loss_weight_lm = 0, loss_weight_vm = 1 -- projects/dreamllm_sdxl/configs/stage1/base.py:52-53).
Stage I of DreamLLM-SDXL trains only the dream queries and the two head projectors: the LLM is frozen, so the fused
decoder-layer Function runs forward + input-gradient only (its backward skips every weight-gradient GEMM whose parameter
does not require grad).
"""
from __future__ import annotations

import torch

from .configuration_dreamllm import DreamLLMConfig
from .modeling_dreamllm import DreamLLMForCausalMLM, DreamLLMModel, DreamLLMRMSNorm
from .tokenization_dreamllm import (DEFAULT_BOS_TOKEN, DEFAULT_DREAM_END_TOKEN, DEFAULT_DREAM_PATCH_TOKEN,
                                    DEFAULT_DREAM_START_TOKEN, DEFAULT_DREAM_TOKEN, DEFAULT_EOS_TOKEN, DEFAULT_IMAGE_END_TOKEN,
                                    DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_IMAGE_START_TOKEN, DEFAULT_IMAGE_TOKEN,
                                    DEFAULT_PAD_TOKEN, DEFAULT_UNK_TOKEN)

# tokenization_dreamllm.py:77-86 of the sdxl package: `<dream_patch>` sits between `<dream>` and `<dream_start>`
additional_special_tokens = [
    DEFAULT_IMAGE_TOKEN,
    DEFAULT_IMAGE_PATCH_TOKEN,
    DEFAULT_IMAGE_START_TOKEN,
    DEFAULT_IMAGE_END_TOKEN,
    DEFAULT_DREAM_TOKEN,
    DEFAULT_DREAM_PATCH_TOKEN,
    DEFAULT_DREAM_START_TOKEN,
    DEFAULT_DREAM_END_TOKEN,
]
special_tokens_dict = dict(bos_token=DEFAULT_BOS_TOKEN, eos_token=DEFAULT_EOS_TOKEN, unk_token=DEFAULT_UNK_TOKEN,
                           pad_token=DEFAULT_PAD_TOKEN, additional_special_tokens=additional_special_tokens)


def default_special_tokens2ids(base_vocab: int = 32000) -> dict:
    """[PAD] first, then the 8 additional special tokens in list order (vocabulary 32009)."""
    ids = {DEFAULT_BOS_TOKEN: 1, DEFAULT_EOS_TOKEN: 2, DEFAULT_UNK_TOKEN: 0, DEFAULT_PAD_TOKEN: base_vocab}
    ids["additional_special_tokens"] = {t: base_vocab + 1 + i for i, t in enumerate(additional_special_tokens)}
    return ids


class DreamLLMSDXLConfig(DreamLLMConfig):
    """configuration_dreamllm_sdxl.py:64 -- the base config under the SDXL name (`model_type` stays "dreamllm", :145)."""

    def __init__(self, *args, **kwargs):  # explicit: recent transformers synthesise an __init__ for bare subclasses
        super().__init__(*args, **kwargs)


class DreamLLMSDXLRMSNorm(DreamLLMRMSNorm):
    """modeling_dreamllm_sdxl.py:78-92."""


class DreamLLMSDXLModel(DreamLLMModel):
    """modeling_dreamllm_sdxl.py:803."""
    config_class = DreamLLMSDXLConfig

    def __init__(self, config):
        super().__init__(config)
        for layer in self.layers:  # modeling_dreamllm_sdxl.py:107: inv_freq is not persistent in this variant
            rope = layer.self_attn.rotary_emb
            inv = rope.inv_freq
            del rope._buffers["inv_freq"]
            rope.register_buffer("inv_freq", inv, persistent=False)


class DreamLLMSDXLForCausalMLM(DreamLLMForCausalMLM):
    """modeling_dreamllm_sdxl.py:1208-1509."""
    config_class = DreamLLMSDXLConfig
    _base_model_class = DreamLLMSDXLModel
    _dream_patch_token = DEFAULT_DREAM_PATCH_TOKEN
    _loss_scale_twice = True

    def _head_loss(self, head, images_dm, enc, u_enc):
        return head(images_dm, enc, u_enc, self._add_time_ids)

    def _head_dummy(self, head, images_dm):
        return head(images_dm, None, None, None, self.model.dream_embedding())

    def forward(self, input_ids=None, images=None, images_dm=None, add_time_ids=None, attention_mask=None, position_ids=None,
                past_key_values=None, inputs_embeds=None, labels=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, dream_index=None, image_index=None, seqlens=None, loss_index=None):
        """modeling_dreamllm_sdxl.py:1353-1509; `add_time_ids`: [N_dm, 6] = original_size + crop_top_left + target_size
        per dream image, as `SDXLDataProcessor` returns them."""
        self._add_time_ids = add_time_ids
        try:
            return super().forward(input_ids=input_ids, images=images, images_dm=images_dm, attention_mask=attention_mask,
                                   position_ids=position_ids, past_key_values=past_key_values, inputs_embeds=inputs_embeds,
                                   labels=labels, use_cache=use_cache, output_attentions=output_attentions,
                                   output_hidden_states=output_hidden_states, return_dict=return_dict, dream_index=dream_index,
                                   image_index=image_index, seqlens=seqlens, loss_index=loss_index)
        finally:
            self._add_time_ids = None
