"""CLIP-ViT vision tower on HIP kernels (forward only, frozen) -- replaces `transformers.CLIPVisionModel` at
omni/models/dreamllm/modeling_plugins.py:214-219,321-323.

state_dict keys are the transformers-4.35 ones (`vision_model.embeddings.{class_embedding,patch_embedding.weight,
position_embedding.weight}`, `vision_model.pre_layrnorm.*`, `vision_model.encoder.layers.{i}.{self_attn.{q,k,v,out}_proj,
layer_norm1,mlp.{fc1,fc2},layer_norm2}.*`, `vision_model.post_layernorm.*`), so `clip_vision_embedding.bin` checkpoints of
the reference load unchanged; checkpoints written by newer transformers (no `vision_model.` prefix) are accepted too.

Execution: patch-embed conv 14x14/14 = one GEMM over an im2col view (K = 588 zero-padded to 592 so rows are 16-byte
aligned); per layer LN -> fused QKV GEMM (+bias) -> non-causal flash attention (head_dim 64, 257 tokens) -> out-proj GEMM
with bias + residual epilogue -> LN -> fc1 GEMM with bias + quick-GELU epilogue -> fc2 GEMM with bias + residual epilogue.
Layers after `select_layer` (the 24th for -2) and `post_layernorm` are never executed.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import ops
from .utils import logger

CLIP_VIT_L14 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=224,
                    patch_size=14, num_channels=3, layer_norm_eps=1e-5, hidden_act="quick_gelu")


class _Cfg(SimpleNamespace):
    def to_dict(self):
        return dict(self.__dict__)


def load_clip_config(name_or_path_or_cfg):
    """Local HF folder (config.json, possibly nested under "vision_config"), a dict, an object with attributes, or the
    preset name "openai/clip-vit-large-patch14" (architecture only; weights stay random unless a local folder is given)."""
    c = name_or_path_or_cfg
    if isinstance(c, dict):
        d = dict(CLIP_VIT_L14, **c)
    elif isinstance(c, str) and os.path.isfile(os.path.join(c, "config.json")):
        with open(os.path.join(c, "config.json")) as f:
            raw = json.load(f)
        raw = raw.get("vision_config", raw)
        d = dict(CLIP_VIT_L14, **{k: raw[k] for k in CLIP_VIT_L14 if k in raw})
    elif isinstance(c, str):
        d = dict(CLIP_VIT_L14)
    else:
        d = dict(CLIP_VIT_L14, **{k: getattr(c, k) for k in CLIP_VIT_L14 if hasattr(c, k)})
    if d["hidden_act"] != "quick_gelu":
        raise ValueError("only quick_gelu CLIP towers are supported (OpenAI CLIP)")
    if d["hidden_size"] // d["num_attention_heads"] != 64:
        raise ValueError("CLIP head_dim must be 64")
    return _Cfg(**d)


class CLIPImageProcessorLite:
    """Host-side preprocessing constants of CLIPImageProcessor (crop size, mean/std); resizing itself is data-pipeline work."""

    def __init__(self, cfg):
        self.crop_size = {"height": cfg.image_size, "width": cfg.image_size}
        self.image_mean = [0.48145466, 0.4578275, 0.40821073]
        self.image_std = [0.26862954, 0.26130258, 0.27577711]

    def __call__(self, images: torch.Tensor):
        """images float [N,3,H,W] in [0,1], already at crop size -> normalised."""
        mean = torch.tensor(self.image_mean, device=images.device).view(1, 3, 1, 1)
        std = torch.tensor(self.image_std, device=images.device).view(1, 3, 1, 1)
        return (images - mean) / std


class _Attn(nn.Module):
    def __init__(self, H):
        super().__init__()
        self.k_proj = nn.Linear(H, H)
        self.v_proj = nn.Linear(H, H)
        self.q_proj = nn.Linear(H, H)
        self.out_proj = nn.Linear(H, H)


class _MLP(nn.Module):
    def __init__(self, H, I):
        super().__init__()
        self.fc1 = nn.Linear(H, I)
        self.fc2 = nn.Linear(I, H)


class _Layer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        H = cfg.hidden_size
        self.self_attn = _Attn(H)
        self.layer_norm1 = nn.LayerNorm(H, eps=cfg.layer_norm_eps)
        self.mlp = _MLP(H, cfg.intermediate_size)
        self.layer_norm2 = nn.LayerNorm(H, eps=cfg.layer_norm_eps)


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        H = cfg.hidden_size
        self.class_embedding = nn.Parameter(torch.randn(H))
        self.patch_embedding = nn.Conv2d(cfg.num_channels, H, cfg.patch_size, cfg.patch_size, bias=False)
        self.num_positions = (cfg.image_size // cfg.patch_size) ** 2 + 1
        self.position_embedding = nn.Embedding(self.num_positions, H)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(cfg) for _ in range(cfg.num_hidden_layers)])


class _VisionTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Embeddings(cfg)
        self.pre_layrnorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)  # (sic) HF spelling
        self.encoder = _Encoder(cfg)
        self.post_layernorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class HipCLIPVisionModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.vision_model = _VisionTransformer(cfg)
        self._fused = None  # cache of fused QKV weights / padded patch weight (frozen tower)

    @property
    def device(self):
        return self.vision_model.pre_layrnorm.weight.device

    @property
    def dtype(self):
        return self.vision_model.pre_layrnorm.weight.dtype

    def load_pretrained(self, path):
        if not isinstance(path, str) or not os.path.isdir(path):
            return False
        for fn in ("pytorch_model.bin", "model.safetensors"):
            fp = os.path.join(path, fn)
            if os.path.isfile(fp):
                if fn.endswith(".safetensors"):
                    from safetensors.torch import load_file
                    sd = load_file(fp)
                else:
                    sd = torch.load(fp, map_location="cpu")
                self.load_state_dict_compat(sd)
                logger.info(f"loaded CLIP vision weights from {fp}")
                return True
        return False

    def load_state_dict_compat(self, sd):
        """Accept 4.35-style (`vision_model.*`) and newer (`embeddings.*`, `encoder.*`) key layouts; ignore text tower."""
        out = {}
        for k, v in sd.items():
            if k.startswith("text_model") or k.startswith("text_projection") or k.startswith("logit_scale") \
                    or k.startswith("visual_projection") or k.endswith("position_ids"):
                continue
            out[k if k.startswith("vision_model.") else "vision_model." + k] = v
        self._fused = None
        return self.load_state_dict(out, strict=True)

    def _load_from_state_dict(self, *args, **kwargs):
        # any state-dict load that reaches this module (also through a parent plugin's load_state_dict / load_model)
        # invalidates the fused QKV / padded patch weights
        self._fused = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _weights_key(self):
        """(data_ptr, version) of every tensor the fused cache was built from: in-place updates (`copy_`, optimizer steps,
        `load_state_dict`) bump the version, re-assignment / `.to()` change the pointer."""
        vm = self.vision_model
        ts = [vm.embeddings.patch_embedding.weight]
        for l in vm.encoder.layers:
            a = l.self_attn
            ts += [a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, a.q_proj.bias, a.k_proj.bias, a.v_proj.bias]
        return tuple((t.data_ptr(), t._version) for t in ts)

    def _prepare(self):
        key = (self.device, self.dtype, self._weights_key())
        if self._fused is not None and self._fused["dev"] == key:
            return self._fused
        vm = self.vision_model
        H = self.config.hidden_size
        w = vm.embeddings.patch_embedding.weight.detach().reshape(H, -1)
        K = w.shape[1]
        Kp = (K + 7) // 8 * 8
        wp = torch.zeros(H, Kp, dtype=w.dtype, device=w.device)
        wp[:, :K] = w
        layers = []
        for l in vm.encoder.layers:
            a = l.self_attn
            layers.append(dict(
                wqkv=torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0).detach().contiguous(),
                bqkv=torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0).detach().contiguous()))
        self._fused = dict(dev=key, patch_w=wp, K=K, Kp=Kp, layers=layers)
        return self._fused

    @torch.no_grad()
    def encode(self, pixel_values, select_layer=-2):
        """-> hidden_states[select_layer] of the HF model ([N, 1+patches, H]); runs only the layers that feed it."""
        cfg, vm = self.config, self.vision_model
        f = self._prepare()
        N = pixel_values.shape[0]
        P, G, H = cfg.patch_size, cfg.image_size // cfg.patch_size, cfg.hidden_size
        nl = cfg.num_hidden_layers
        n_run = select_layer if select_layer >= 0 else nl + 1 + select_layer
        if not 0 <= n_run <= nl:
            raise ValueError(f"select_layer {select_layer} out of range")
        # im2col is a pure re-layout for a stride == kernel conv: [N,3,G,P,G,P] -> [N*G*G, 3*P*P]
        cols = pixel_values.to(self.dtype).view(N, cfg.num_channels, G, P, G, P).permute(0, 2, 4, 1, 3, 5)
        patches = torch.zeros(N * G * G, f["Kp"], dtype=self.dtype, device=pixel_values.device)
        patches[:, : f["K"]] = cols.reshape(N * G * G, f["K"])
        emb = ops.linear_fwd(patches, f["patch_w"]).view(N, G * G, H)
        x = torch.cat([vm.embeddings.class_embedding.to(self.dtype).expand(N, 1, H), emb], dim=1)  # data movement only
        x = ops.add_bcast(x, vm.embeddings.position_embedding.weight)
        x, _, _ = ops.layernorm_fwd(x, vm.pre_layrnorm.weight, vm.pre_layrnorm.bias, cfg.layer_norm_eps, save_stats=False)
        S = x.shape[1]
        heads = cfg.num_attention_heads
        for i in range(n_run):
            l, lf = vm.encoder.layers[i], f["layers"][i]
            h, _, _ = ops.layernorm_fwd(x, l.layer_norm1.weight, l.layer_norm1.bias, cfg.layer_norm_eps, save_stats=False)
            qkv = ops.linear_fwd(h, lf["wqkv"], bias=lf["bqkv"]).view(N, S, 3, heads, 64)
            o, _ = ops.attn_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], False, 0.125, need_lse=False)
            x = ops.linear_fwd(o.view(N, S, H), l.self_attn.out_proj.weight, bias=l.self_attn.out_proj.bias, residual=x)
            h, _, _ = ops.layernorm_fwd(x, l.layer_norm2.weight, l.layer_norm2.bias, cfg.layer_norm_eps, save_stats=False)
            h = ops.linear_fwd(h, l.mlp.fc1.weight, bias=l.mlp.fc1.bias, epi="quick_gelu")
            x = ops.linear_fwd(h, l.mlp.fc2.weight, bias=l.mlp.fc2.bias, residual=x)
        return x

    def forward(self, pixel_values, output_hidden_states=False, select_layer=-2):
        hs = self.encode(pixel_values, select_layer)
        return SimpleNamespace(last_hidden_state=hs, hidden_states=None)
