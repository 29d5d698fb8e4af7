"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the reference's LLM hot path in plain PyTorch.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; nothing under dreamllm_amd/ does.
Every function cites the reference lines it restates (paths relative to /root/reference/omni/models/dreamllm/).
Pinned against the *imported reference classes* by oracle/make_golden.py (run in the authoring container), whose
outputs are committed under tests/golden/ and re-checked by tests/test_oracle.py.

All functions are dtype-agnostic: run them in fp32 for the oracle proper, or in bf16 to obtain the reference's own
bf16-vs-fp32 error as a yard-stick.  Weights are passed as a state_dict with the reference's key names.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def rmsnorm(x, weight, eps):
    """DreamLLMRMSNorm.forward, modeling_dreamllm.py:86-91."""
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return weight * h.to(dt)


def rope_tables(dim, max_pos, base=10000.0, dtype=torch.float32, device=None):
    """RotaryEmbedding.__init__/_set_cos_sin_cache, modeling_dreamllm.py:97-119 -> cos, sin [max_pos, dim]."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, device=device).float() / dim))
    t = torch.arange(max_pos, dtype=inv_freq.dtype, device=device)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    """modeling_dreamllm.py:176-180."""
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin, position_ids):
    """apply_rotary_pos_emb, modeling_dreamllm.py:184-209; q,k [B,H,S,D]."""
    cos = cos[position_ids].unsqueeze(1)
    sin = sin[position_ids].unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def causal_mask_4d(attention_mask, B, S, dtype, device=None):
    """HF _prepare_4d_causal_attention_mask as used at modeling_dreamllm.py:965-967 (no KV cache): additive mask."""
    minv = torch.finfo(dtype).min
    m = torch.full((S, S), minv, dtype=dtype, device=device)
    m = torch.triu(m, diagonal=1)[None, None].expand(B, 1, S, S).clone()
    if attention_mask is not None:
        pad = (attention_mask[:, None, None, :] == 0)
        m = m.masked_fill(pad, minv)
    return m


def attention(x, sd, prefix, n_heads, n_kv_heads, cos, sin, position_ids, mask4d):
    """DreamLLMAttention.forward (eager), modeling_dreamllm.py:309-400, pretraining_tp == 1, no cache."""
    B, S, H = x.shape
    hd = H // n_heads
    q = F.linear(x, sd[prefix + "q_proj.weight"]).view(B, S, n_heads, hd).transpose(1, 2)
    k = F.linear(x, sd[prefix + "k_proj.weight"]).view(B, S, n_kv_heads, hd).transpose(1, 2)
    v = F.linear(x, sd[prefix + "v_proj.weight"]).view(B, S, n_kv_heads, hd).transpose(1, 2)
    q, k = apply_rope(q, k, cos.to(x.dtype), sin.to(x.dtype), position_ids)
    rep = n_heads // n_kv_heads
    if rep > 1:  # repeat_kv, modeling_dreamllm.py:242-251
        k = k[:, :, None].expand(B, n_kv_heads, rep, S, hd).reshape(B, n_heads, S, hd)
        v = v[:, :, None].expand(B, n_kv_heads, rep, S, hd).reshape(B, n_heads, S, hd)
    w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(hd)
    if mask4d is not None:
        w = w + mask4d
        w = torch.max(w, torch.tensor(torch.finfo(w.dtype).min, dtype=w.dtype))
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(w, v).transpose(1, 2).contiguous().reshape(B, S, H)
    return F.linear(o, sd[prefix + "o_proj.weight"])


def mlp(x, sd, prefix):
    """DreamLLMMLP.forward, modeling_dreamllm.py:237."""
    return F.linear(F.silu(F.linear(x, sd[prefix + "gate_proj.weight"])) * F.linear(x, sd[prefix + "up_proj.weight"]),
                    sd[prefix + "down_proj.weight"])


def decoder_layer(x, sd, prefix, cfg, cos, sin, position_ids, mask4d):
    """DreamLLMDecoderLayer.forward, modeling_dreamllm.py:622-640."""
    r = x
    h = rmsnorm(x, sd[prefix + "input_layernorm.weight"], cfg["rms_norm_eps"])
    h = attention(h, sd, prefix + "self_attn.", cfg["num_attention_heads"], cfg["num_key_value_heads"], cos, sin,
                  position_ids, mask4d)
    x = r + h
    r = x
    h = rmsnorm(x, sd[prefix + "post_attention_layernorm.weight"], cfg["rms_norm_eps"])
    return r + mlp(h, sd, prefix + "mlp.")


def model_forward(inputs_embeds, sd, cfg, attention_mask=None, prefix="model."):
    """DreamLLMModel._forward, modeling_dreamllm.py:846-1043 (eager 4-D mask path, no cache) -> last_hidden_state."""
    B, S, H = inputs_embeds.shape
    hd = H // cfg["num_attention_heads"]
    dev = inputs_embeds.device  # the oracle also runs in fp32 ON the GPU for the full-size checks (tests/test_fullsize_models_gpu.py)
    cos, sin = rope_tables(hd, max(cfg["max_position_embeddings"], S), cfg.get("rope_theta", 10000.0), device=dev)
    position_ids = torch.arange(S, device=dev)[None]
    mask4d = causal_mask_4d(attention_mask, B, S, inputs_embeds.dtype, device=dev)
    x = inputs_embeds
    for i in range(cfg["num_hidden_layers"]):
        x = decoder_layer(x, sd, f"{prefix}layers.{i}.", cfg, cos, sin, position_ids, mask4d)
    return rmsnorm(x, sd[prefix + "norm.weight"], cfg["rms_norm_eps"])


def splice_inputs(input_ids, sd, cfg, dream_queries=None, image_features=None, prefix="model."):
    """DreamLLMModel.forward, modeling_dreamllm.py:1066-1141: embedding lookup, then overwrite the 64 rows after each
    <dream_start> with the dream queries and the 256 rows after each <im_start> with the projected CLIP features.
    dream_queries [n_q, H] (shared by every slot, `DreamEmbedding.forward` repeats it per sample);
    image_features [n_img, n_patch, H] consumed in order of appearance."""
    emb = F.embedding(input_ids, sd[prefix + "embed_tokens.weight"])
    out = emb.clone()
    ids = cfg["special_ids"]
    if dream_queries is not None:
        nq = dream_queries.shape[0]
        for b in range(input_ids.shape[0]):
            for p in torch.where(input_ids[b] == ids["dream_start"])[0].tolist():
                out[b, p + 1: p + 1 + nq] = dream_queries
    if image_features is not None:
        cur = 0
        for b in range(input_ids.shape[0]):
            for p in torch.where(input_ids[b] == ids["im_start"])[0].tolist():
                if cur >= image_features.shape[0]:
                    break
                n = image_features.shape[1]
                out[b, p + 1: p + 1 + n] = image_features[cur]
                cur += 1
    return out


def gather_dream_states(hidden, input_ids, cfg, nq, max_slots):
    """DreamLLMForCausalMLM.forward, modeling_dreamllm.py:1399-1418 -> [n_slots, nq, H]."""
    outs = []
    for b in range(input_ids.shape[0]):
        for p in torch.where(input_ids[b] == cfg["special_ids"]["dream_start"])[0].tolist():
            if len(outs) >= max_slots:
                break
            outs.append(hidden[b, p + 1: p + 1 + nq])
    return torch.stack(outs, 0)


def lm_loss(hidden, lm_head_weight, labels):
    """modeling_dreamllm.py:1452-1470: lm_head, .float(), shifted CE(reduction none), masked mean -> (loss, logits)."""
    logits = F.linear(hidden, lm_head_weight).float()
    V = logits.shape[-1]
    sl = logits[..., :-1, :].contiguous().view(-1, V)
    lb = labels[..., 1:].contiguous().view(-1)
    valid = lb != -100
    per = F.cross_entropy(sl, lb, reduction="none", ignore_index=-100)
    if valid.sum() > 0:
        loss = (per * valid).sum() / valid.sum()
    else:
        loss = per.mean()
    return loss, logits


def linear_projector(x, weight, bias=None):
    """LinearProjector.forward, omni/models/projector/mlp_projector.py:23-27 (returns the last list element)."""
    return F.linear(x, weight, bias)


def mlp_projector(x, weights, biases):
    """MLPProjector.forward, omni/models/projector/mlp_projector.py:40-50: Linear (GELU Linear)*."""
    h = F.linear(x, weights[0], biases[0])
    for w, b in zip(weights[1:], biases[1:]):
        h = F.linear(F.gelu(h), w, b)
    return h


def greedy_decode(input_ids, sd, cfg, steps):
    """Text-only greedy decode loop, omni/eval/language_eval/modeling_dreamllm.py:76-97 with temperature == 0 (argmax).
    Recomputes the full prefix each step (no KV cache): same tokens, used only at tiny sizes."""
    ids = input_ids.clone()
    for _ in range(steps):
        emb = F.embedding(ids, sd["model.embed_tokens.weight"])
        h = model_forward(emb, sd, cfg)
        logits = F.linear(h[:, -1], sd["lm_head.weight"]).float()
        ids = torch.cat([ids, logits.argmax(-1, keepdim=True)], dim=1)
    return ids


def random_state_dict(cfg, seed=0, n_added=0, added_boost=1.0):
    """Seeded DreamLLM decoder weights (bf16-representable fp32, reference key names, no plugin keys) -- reproducible on any
    host, so fixtures for configurations with a full-size vocabulary store the seed instead of 8 MB of embeddings.
    `added_boost` scales the lm_head rows of the last `n_added` (special) tokens: with a boost > 1 they win most argmaxes
    unless the decode loop slices the logits to the base vocabulary (omni/eval/language_eval/modeling_dreamllm.py:79,86)."""
    g = torch.Generator().manual_seed(seed)
    H, F_, V, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["num_hidden_layers"]
    nh, nkv = cfg["num_attention_heads"], cfg.get("num_key_value_heads") or cfg["num_attention_heads"]
    hd = H // nh
    r16 = lambda t: t.to(torch.bfloat16).to(torch.float32)
    w = lambda *s: r16(torch.randn(*s, generator=g) * 0.05)
    n = lambda s: r16(1.0 + 0.1 * torch.randn(s, generator=g))
    sd = {"model.embed_tokens.weight": w(V, H)}
    for i in range(L):
        p = f"model.layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = w(nh * hd, H)
        sd[p + "self_attn.k_proj.weight"] = w(nkv * hd, H)
        sd[p + "self_attn.v_proj.weight"] = w(nkv * hd, H)
        sd[p + "self_attn.o_proj.weight"] = w(H, nh * hd)
        sd[p + "mlp.gate_proj.weight"] = w(F_, H)
        sd[p + "mlp.up_proj.weight"] = w(F_, H)
        sd[p + "mlp.down_proj.weight"] = w(H, F_)
        sd[p + "input_layernorm.weight"] = n(H)
        sd[p + "post_attention_layernorm.weight"] = n(H)
    sd["model.norm.weight"] = n(H)
    head = w(V, H)
    if n_added:
        head[V - n_added:] = r16(head[V - n_added:] * added_boost)
    sd["lm_head.weight"] = head
    return sd
