"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the CLIP-ViT vision tower the reference calls through
`transformers.CLIPVisionModel` (omni/models/dreamllm/modeling_plugins.py:214-219,321-323; arithmetic lives in
transformers==4.35.2 `modeling_clip.py`, not vendored).  Weights use the 4.35 state_dict key names
(`vision_model.embeddings.*`, `vision_model.pre_layrnorm.*`, `vision_model.encoder.layers.{i}.*`).

Pinned by tests/test_oracle.py against the installed `transformers.CLIPVisionModel` (same architecture, version 5.x,
random weights): hidden_states[k] for every k.  Only tests/, smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def clip_hidden_states(pixel_values, sd, cfg, prefix="vision_model."):
    """-> list of hidden states as `output_hidden_states=True` returns them: [embeddings after pre-LN, layer1, ...]."""
    H, P, heads, eps = cfg["hidden_size"], cfg["patch_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
    N = pixel_values.shape[0]
    x = F.conv2d(pixel_values, sd[prefix + "embeddings.patch_embedding.weight"], stride=P)  # bias=False
    x = x.flatten(2).transpose(1, 2)
    cls = sd[prefix + "embeddings.class_embedding"].expand(N, 1, H)
    x = torch.cat([cls, x], dim=1) + sd[prefix + "embeddings.position_embedding.weight"][None]
    x = F.layer_norm(x, (H,), sd[prefix + "pre_layrnorm.weight"], sd[prefix + "pre_layrnorm.bias"], eps)
    hs = [x]
    hd = H // heads
    for i in range(cfg["num_hidden_layers"]):
        p = f"{prefix}encoder.layers.{i}."
        r = x
        h = F.layer_norm(x, (H,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], eps)
        S = h.shape[1]
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * hd**-0.5
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q, k, v = (t.view(N, S, heads, hd).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v
        a = a.transpose(1, 2).reshape(N, S, H)
        x = r + F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        r = x
        h = F.layer_norm(x, (H,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], eps)
        h = quick_gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        x = r + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        hs.append(x)
    return hs


def clip_vision_embedding(images, sd, cfg, proj_w, proj_b, select_layer=-2):
    """CLIPVisionEmbedding.forward, modeling_plugins.py:321-326 (linear projector)."""
    hs = clip_hidden_states(images, sd, cfg, prefix="clip_vision_model.vision_model.")
    feats = hs[select_layer][:, 1:]
    return F.linear(feats, proj_w, proj_b)
