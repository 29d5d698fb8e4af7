"""DreamLLM model configuration -- API mirror of omni/models/dreamllm/configuration_dreamllm.py (drop-in boundary b1).

Same constructor kwargs, same `special_tokens2ids_dict` / `plugins_init_kwargs` / `plugins_type` fields and the same
`update_plugins` contract ({_class_, _name_, _plugin_type_, **kwargs} -> {_target_: "pkg.Class", **kwargs}); omegaconf
is not required: plain dicts are stored (they serialise to the same JSON).
"""
from __future__ import annotations

from typing import Any

from transformers.configuration_utils import PretrainedConfig

from .utils import logger, target_to_string

CLASS_KEY = "_class_"
NAME_KEY = "_name_"
PLUGIN_TYPE_KEY = "_plugin_type_"
ConfigAndInitKwargs = dict


def create_config_init_kwargs(config_init_kwargs: dict) -> dict:
    """configuration_dreamllm.py:47-61 (validation identical; returns a plain dict instead of a DictConfig)."""
    config_class_ = config_init_kwargs.get(CLASS_KEY, None)
    config_name_ = config_init_kwargs.get(NAME_KEY, None)
    config_plugin_type_ = config_init_kwargs.get(PLUGIN_TYPE_KEY, None)
    if not isinstance(config_class_, type):
        raise ValueError(f"`config_init_kwargs` must have `_class_` field of type `{type}`, got `{type(config_class_)}`.")
    if not isinstance(config_name_, str):
        raise ValueError(f"`config_init_kwargs` must have `_name_` field of type `{str}`, got `{type(config_name_)}`.")
    if not isinstance(config_plugin_type_, str):
        raise ValueError(
            f"`config_init_kwargs` must have `_plugin_type_` field of type `{str}`, got `{type(config_plugin_type_)}`."
        )
    return dict(config_init_kwargs)


class DreamLLMConfig(PretrainedConfig):
    """configuration_dreamllm.py:64-278 (LLaMA hyper-parameters + plugin registry + loss weights)."""

    model_type = "dreamllm"
    keys_to_ignore_at_inference = ["past_key_values"]

    def __init__(
        self,
        vocab_size=32000,
        hidden_size=4096,
        intermediate_size=11008,
        num_hidden_layers=32,
        num_attention_heads=32,
        num_key_value_heads=None,
        hidden_act="silu",
        max_position_embeddings=2048,
        initializer_range=0.02,
        rms_norm_eps=1e-6,
        use_cache=True,
        pad_token_id=None,
        bos_token_id=1,
        eos_token_id=2,
        pretraining_tp=1,
        tie_word_embeddings=False,
        rope_theta=10000.0,
        rope_scaling=None,
        attention_bias=False,
        special_tokens2ids_dict=None,
        plugins_init_kwargs=None,
        plugins_type=None,
        loss_weight_lm=1.0,
        loss_weight_vm=10.0,
        loss_scale_schedule="none",
        log_attentions=False,
        log_hidden_states=False,
        **kwargs,
    ):
        self.vocab_size = vocab_size
        self.max_position_embeddings = max_position_embeddings
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        if num_key_value_heads is None:
            num_key_value_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads
        self.hidden_act = hidden_act
        self.initializer_range = initializer_range
        self.rms_norm_eps = rms_norm_eps
        self.pretraining_tp = pretraining_tp
        self.use_cache = use_cache
        self.rope_theta = rope_theta
        self.attention_bias = attention_bias
        super().__init__(
            pad_token_id=pad_token_id,
            bos_token_id=bos_token_id,
            eos_token_id=eos_token_id,
            tie_word_embeddings=tie_word_embeddings,
            **kwargs,
        )
        # newer transformers auto-populate rope parameters; the reference contract is None or {"type","factor"}
        self.rope_scaling = rope_scaling
        self._rope_scaling_validation()
        self.special_tokens2ids_dict: dict[str, Any] = special_tokens2ids_dict if special_tokens2ids_dict is not None else {}
        self.plugins_init_kwargs: dict[str, dict] = plugins_init_kwargs if plugins_init_kwargs is not None else {}
        self.plugins_type: dict[str, str] = plugins_type if plugins_type is not None else {}
        self.loss_weight_lm = loss_weight_lm
        self.loss_weight_vm = loss_weight_vm
        self.loss_scale_schedule = loss_scale_schedule
        self.log_attentions = log_attentions
        self.log_hidden_states = log_hidden_states

    def update_special_tokens2ids_dict(self, tokens_dict: dict, tokenizer):
        """configuration_dreamllm.py:225-235."""
        for key, token in tokens_dict.items():
            if isinstance(token, list):
                ids = tokenizer.convert_tokens_to_ids(token)
                if key not in self.special_tokens2ids_dict.keys():
                    self.special_tokens2ids_dict[key] = {}
                for _token, _id in zip(token, ids):
                    self.special_tokens2ids_dict[key][_token] = _id
            else:
                self.special_tokens2ids_dict[token] = tokenizer.convert_tokens_to_ids(token)

    def update_plugins(self, init_kwargs: dict):
        """configuration_dreamllm.py:237-255."""
        init_kwargs = dict(init_kwargs)
        cls = init_kwargs.pop(CLASS_KEY, None)
        name = init_kwargs.pop(NAME_KEY, None)
        plugin_type = init_kwargs.pop(PLUGIN_TYPE_KEY, None)
        assert (
            cls is not None and name is not None and plugin_type is not None
        ), f"`init_kwargs` must have `{CLASS_KEY}`, `{NAME_KEY}` and `{PLUGIN_TYPE_KEY}` fields"
        lazy_init = {"_target_": target_to_string(cls), **init_kwargs}
        if name not in self.plugins_init_kwargs.keys():
            self.plugins_init_kwargs[name] = lazy_init
        else:
            self.plugins_init_kwargs[name].update(lazy_init)
        self.plugins_type[name] = plugin_type
        return name

    def _rope_scaling_validation(self):
        """configuration_dreamllm.py:257-272."""
        if self.rope_scaling is None:
            return
        if not isinstance(self.rope_scaling, dict) or len(self.rope_scaling) != 2:
            raise ValueError(
                "`rope_scaling` must be a dictionary with with two fields, `type` and `factor`, " f"got {self.rope_scaling}"
            )
        rope_scaling_type = self.rope_scaling.get("type", None)
        rope_scaling_factor = self.rope_scaling.get("factor", None)
        if rope_scaling_type is None or rope_scaling_type not in ["linear", "dynamic"]:
            raise ValueError(f"`rope_scaling`'s type field must be one of ['linear', 'dynamic'], got {rope_scaling_type}")
        if rope_scaling_factor is None or not isinstance(rope_scaling_factor, float) or rope_scaling_factor <= 1.0:
            raise ValueError(f"`rope_scaling`'s factor field must be an float > 1, got {rope_scaling_factor}")

    def reset_plugins_init_kwargs(self, pretrained_plugin_model_name_or_path: str = None):
        """configuration_dreamllm.py:274-277."""
        for plugin_name in self.plugins_init_kwargs.keys():
            self.plugins_init_kwargs[plugin_name]["pretrained_model_name_or_path"] = pretrained_plugin_model_name_or_path
        logger.warning(f"reset all pretrained_model_name_or_path of plugin modules to {pretrained_plugin_model_name_or_path}")
