"""DreamLLM decoder on hand-written gfx950 kernels -- API mirror of omni/models/dreamllm/modeling_dreamllm.py.

Same class names, constructor kwargs, forward signatures, output dataclasses and state_dict keys as the reference
(`model.layers.{i}.self_attn.{q,k,v,o}_proj.weight`, `...rotary_emb.inv_freq`, `model.embed_tokens.weight`,
`model.norm.weight`, `lm_head.weight`, plugins under `model.{dream,clip_vision}_embedding.*` and
`stable_diffusion_head.*`), so it drops in under omni/models/dreamllm.  What differs is *how* a layer executes:

* one `torch.autograd.Function` per decoder layer with a hand-written backward.  Only x, (q,k,v) after RoPE, the
  attention output + LSE, the post-attention residual stream and gate/up are kept (3.0 GB/layer at B=16,S=2048); the
  RMSNorm outputs and the SwiGLU product are recomputed in backward (HBM-bound, <1 % of the layer) instead of the
  reference's whole-layer gradient checkpointing (modeling_dreamllm.py:994-1003, +33 % FLOPs) -- 288 GB of HBM3E makes
  the 98 GB of activations resident.
* attention is always the causal flash kernel (the reference's DreamLLMFlashAttention2 semantics,
  modeling_dreamllm.py:403-583); a 2-D right-padding mask becomes per-sequence lengths.
* residual adds are GEMM epilogues; RoPE is applied in place on the QKV GEMM output ([B,S,H,D], no head transposes).
* the multimodal splice (modeling_dreamllm.py:1081-1141) is one index-scatter per modality driven by index tensors.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from dataclasses import dataclass, fields
from typing import Any

import torch
import torch.nn.functional as F  # noqa: F401  (kept for API parity of the module namespace)
from torch import nn
from transformers import PreTrainedModel
from transformers.utils import ModelOutput

from . import ops
from . import torch_ops  # noqa: F401  (registers torch.ops.dreamllm.*)
from .configuration_dreamllm import DreamLLMConfig
from .tokenization_dreamllm import (
    DEFAULT_BOS_TOKEN,
    DEFAULT_DREAM_END_TOKEN,
    DEFAULT_DREAM_START_TOKEN,
    DEFAULT_EOS_TOKEN,
    DEFAULT_IMAGE_PATCH_TOKEN,
    DEFAULT_IMAGE_START_TOKEN,
)
from .utils import FSDPMixin, deep_instantiate, logger

try:  # registration drives weight-decay grouping in the reference trainer (modeling_dreamllm.py:94)
    from transformers.pytorch_utils import ALL_LAYERNORM_LAYERS
except Exception:  # pragma: no cover
    ALL_LAYERNORM_LAYERS = []


class DreamLLMRMSNorm(nn.Module):
    """modeling_dreamllm.py:77-91."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states):
        if torch.compiler.is_compiling() and not torch.is_grad_enabled():   # torch.compile(model): registered op, no graph break
            return torch.ops.dreamllm.rmsnorm(hidden_states, self.weight, float(self.variance_epsilon))
        return ops.rmsnorm(hidden_states, self.weight, self.variance_epsilon)


ALL_LAYERNORM_LAYERS.append(DreamLLMRMSNorm)


class RotaryEmbedding(nn.Module):
    """modeling_dreamllm.py:97-128.  `inv_freq` stays a persistent buffer (it is in the reference's state_dict); the
    cos/sin cache is kept in fp32 as [max_pos, dim/2] (the two halves of the reference's table are identical)."""

    def __init__(self, dim, max_position_embeddings=2048, base=10000, device=None, scaling_factor=1.0, scaling_type=None):
        super().__init__()
        self.dim = dim
        self.max_position_embeddings = max_position_embeddings
        self.base = base
        self.scaling_factor = scaling_factor
        self.scaling_type = scaling_type
        inv_freq = 1.0 / (self.base ** (torch.arange(0, self.dim, 2).float().to(device) / self.dim))
        self.register_buffer("inv_freq", inv_freq)
        self._cache = {}  # device -> (seq_len, cos_half, sin_half); plain fp32 tensors, immune to model.to(bf16)

    def _build(self, seq_len, device):
        # inv_freq is recomputed in fp32 from (base, dim): the registered buffer may have been cast to bf16 by .to()
        base = self.base
        if self.scaling_type == "dynamic" and seq_len > self.max_position_embeddings:  # modeling_dreamllm.py:150-173
            base = self.base * ((self.scaling_factor * seq_len / self.max_position_embeddings) - (self.scaling_factor - 1)) ** (
                self.dim / (self.dim - 2))
        inv_freq = 1.0 / (base ** (torch.arange(0, self.dim, 2, device=device).float() / self.dim))
        t = torch.arange(seq_len, device=device, dtype=torch.float32)
        if self.scaling_type == "linear":  # modeling_dreamllm.py:131-147
            t = t / self.scaling_factor
        freqs = torch.einsum("i,j->ij", t, inv_freq)
        return freqs.cos().contiguous(), freqs.sin().contiguous()

    def tables(self, seq_len, device):
        """fp32 [>=seq_len, dim/2] cos and sin tables on `device`."""
        key = str(torch.device(device))
        ent = self._cache.get(key)
        need = max(seq_len, self.max_position_embeddings)
        if ent is None or ent[0] < need:
            c, s = self._build(need, device)
            ent = (need, c, s)
            self._cache[key] = ent
        return ent[1], ent[2]

    def forward(self, x, seq_len=None):
        """Reference-shaped output: (cos, sin) of shape [seq_len, dim] in x.dtype."""
        c, s = self.tables(seq_len, x.device)
        return torch.cat([c, c], -1)[:seq_len].to(x.dtype), torch.cat([s, s], -1)[:seq_len].to(x.dtype)


# ----------------------------------------------------------------------------------------- fused decoder layer
def _packed_view(*ws):
    """[sum(out_i), in] view over weights that are adjacent slices of one storage, in this order (see `pack_linear_weights`);
    None when they are not -- the caller then runs one GEMM per weight."""
    w0 = ws[0]
    if any(w.dtype != w0.dtype or w.device != w0.device or w.dim() != 2 or w.shape[1] != w0.shape[1] or not w.is_contiguous()
           for w in ws):
        return None
    es, ptr = w0.element_size(), w0.data_ptr()
    for w in ws:
        if w.data_ptr() != ptr:
            return None
        ptr += w.numel() * es
    rows = sum(w.shape[0] for w in ws)
    room = w0.untyped_storage().nbytes() - w0.storage_offset() * es
    if room < rows * w0.shape[1] * es:
        return None
    return torch.as_strided(w0.detach(), (rows, w0.shape[1]), (w0.shape[1], 1))


def pack_linear_weights(*linears):
    """Re-point the weights of `linears` (same in_features) at consecutive row blocks of ONE buffer, so that they can be applied
    as a single GEMM on the packed [sum(out), in] matrix while every `nn.Parameter` keeps its identity, shape and state_dict key
    (`q_proj.weight`, ...: modeling_dreamllm.py:273-275,219-220).  In-place updates (optimizers, `load_state_dict`, `copy_`)
    keep the packing; `module.to(dtype/device)` re-allocates the parameters and drops it (call again afterwards)."""
    ws = [l.weight for l in linears]
    if _packed_view(*[w.data for w in ws]) is not None:
        return
    buf = torch.cat([w.data for w in ws], dim=0).contiguous()
    r = 0
    for w in ws:
        n = w.shape[0]
        w.data = buf[r:r + n]
        r += n


def packed_parameter_groups(model):
    """The parameter groups every decoder layer of `model` applies as one packed GEMM: [[q, k, v], [gate, up]] per layer -- what
    `distributed.ShardedGradAdamW(atomic_groups=...)` must keep inside one flat bucket."""
    out = []
    for m in model.modules():
        if isinstance(m, DreamLLMDecoderLayer):
            a, f = m.self_attn, m.mlp
            out.append([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight])
            out.append([f.gate_proj.weight, f.up_proj.weight])
    return out


# 288 GB of HBM per MI355X: the normed inputs of the two GEMM groups and the SwiGLU product (1.26 GB per layer, 40 GB for the 7B
# model at B=16 x S=2048) are KEPT for the backward instead of recomputed (two RMSNorm passes and 0.7 GB of extra SwiGLU-backward
# traffic per layer); peak 161 -> 201 GB.  Set to False to trade the 40 GB back for ~0.7 % of the step (larger models / batches).
KEEP_LAYER_ACTIVATIONS = os.environ.get("DREAMLLM_KEEP_LAYER_ACTIVATIONS", "1") != "0"


class _DecoderLayerFn(torch.autograd.Function):
    """DreamLLMDecoderLayer.forward (modeling_dreamllm.py:622-640) with a hand-written backward.

    q/k/v and gate/up run as ONE GEMM each when their weights are packed (`pack_linear_weights`): forward y = h [Wq;Wk;Wv]^T
    straight into a packed [T, (H + 2 Hkv) D] buffer whose q/k/v column blocks are the strided views attention and RoPE take
    (one RoPE launch over the q and k heads); backward dh = [dq|dk|dv] [Wq;Wk;Wv] as one GEMM over K = 3H (no accumulating
    passes over dh), and one weight-gradient GEMM whose row blocks are the three gradients.  Same for [gate; up].  With
    unpacked weights the layer falls back to one GEMM per projection (identical results up to the GEMM's own rounding)."""

    @staticmethod
    def _run_forward(x, w_in, wq, wk, wv, wo, w_post, wg, wu, wd, cos, sin, pos, seqlens, seqstart, n_heads, n_kv, eps, need_bwd, ng,
                     keep, want_y=True, pack=None):
        """The layer's forward launches.  Returns (y, (k, v) views of the packed q|k|v buffer, the intermediates the backward reads).
        `want_y=False` (the recompute pass of `gradient_checkpointing`) stops in front of the down projection: nothing behind it is
        needed by the backward."""
        B, S, Hd = x.shape
        hd = Hd // n_heads
        T = B * S
        nq, nkv = n_heads * hd, n_kv * hd
        h, _, rstd1 = ops.rmsnorm_fwd(x, w_in, eps)
        wqkv = _packed_view(wq, wk, wv)
        roped = False
        fused = ops.linear_rope_qkv(h.view(T, Hd), wqkv, cos, sin, pos, n_heads + n_kv, hd, S) if wqkv is not None else None
        if fused is not None:          # q|k|v projection with the rotary embedding in its epilogue (round 6)
            qkv, roped = fused.view(B, S, nq + 2 * nkv), True
        elif wqkv is not None:
            qkv = ops.linear_fwd(h, wqkv).view(B, S, nq + 2 * nkv)
        else:
            qkv = torch.empty(B, S, nq + 2 * nkv, dtype=x.dtype, device=x.device)
            h2d, o2d = h.view(T, Hd), qkv.view(T, -1)
            ops.gemm(h2d, wq, T, nq, Hd, Hd, Hd, 0, 0, out=o2d[:, :nq])
            ops.gemm(h2d, wk, T, nkv, Hd, Hd, Hd, 0, 0, out=o2d[:, nq:nq + nkv])
            ops.gemm(h2d, wv, T, nkv, Hd, Hd, Hd, 0, 0, out=o2d[:, nq + nkv:])
        h_keep = h if (keep and (ng[2] or ng[3] or ng[4])) else None   # kept tensors only feed weight gradients (a frozen LLM keeps nothing extra)
        del h
        if not roped:
            qk = qkv[:, :, : nq + nkv].unflatten(-1, (n_heads + n_kv, hd))   # q heads then k heads: one RoPE launch
            ops.rope_(qk, cos, sin, pos)
        if pack is not None:
            # ragged batch on COMPACT rows (`_token_pack`): every GEMM / norm of the layer runs on the valid tokens only; attention alone
            # needs the padded [PB, PS] grid (one span per row), so q|k|v are scattered onto it and the output gathered back -- the
            # reference's unpad / pad round trip (modeling_dreamllm.py:523-545) the other way round
            vidx, tv, PB, PS = pack
            qkv_p = torch.empty(PB, PS, nq + 2 * nkv, dtype=x.dtype, device=x.device)
            ops.scatter_rows_(qkv_p.view(PB * PS, -1), vidx[:tv], qkv.view(T, -1)[:tv])
            qkv = qkv_p      # what the backward keeps
        q = qkv[:, :, :nq].unflatten(-1, (n_heads, hd))
        k = qkv[:, :, nq:nq + nkv].unflatten(-1, (n_kv, hd))
        v = qkv[:, :, nq + nkv:].unflatten(-1, (n_kv, hd))
        o, lse = ops.attn_fwd(q, k, v, True, 1.0 / math.sqrt(hd), seqlens, need_lse=need_bwd, seqstart=seqstart)
        o_c = o if pack is None else ops.gather_rows(o.view(-1, Hd), pack[0])   # (filler rows read a pad position: zeros)
        x2 = ops.linear_fwd(o_c.view(B, S, Hd), wo, residual=x)
        h2, _, rstd2 = ops.rmsnorm_fwd(x2, w_post, eps)
        F_ = wg.shape[0]
        wgu = _packed_view(wg, wu)
        y = act = None
        fused = ops.linear_swiglu_fwd(h2.view(T, Hd), wgu) if wgu is not None else None   # gate|up GEMM with the SwiGLU in its epilogue
        if fused is not None:
            gu, act = fused
        elif wgu is not None:
            gu = ops.linear_fwd(h2, wgu).view(T, 2 * F_)
        else:
            gu = torch.empty(T, 2 * F_, dtype=x.dtype, device=x.device)
            h22 = h2.view(T, Hd)
            ops.gemm(h22, wg, T, F_, Hd, Hd, Hd, 0, 0, out=gu[:, :F_])
            ops.gemm(h22, wu, T, F_, Hd, Hd, Hd, 0, 0, out=gu[:, F_:])
        h2_keep = h2 if (keep and (ng[7] or ng[8])) else None
        del h2
        if act is None and (want_y or (keep and ng[9])):
            act = ops.glu_fwd(gu[:, :F_], gu[:, F_:], 0)
        if want_y:
            y = ops.linear_fwd(act.view(B, S, F_), wd, residual=x2)
        return y, (k, v), (rstd1, qkv, o, lse, x2, rstd2, gu, h_keep, h2_keep, act if (keep and ng[9]) else None)

    @staticmethod
    def forward(ctx, x, w_in, wq, wk, wv, wo, w_post, wg, wu, wd, cos, sin, pos, seqlens, seqstart, n_heads, n_kv, eps, want_kv,
                recompute=False, pack=None):
        need_bwd = any(ctx.needs_input_grad[:10])
        ng = ctx.needs_input_grad
        # `recompute` = whole-layer activation recompute (the reference's gradient checkpointing, modeling_dreamllm.py:994-1003,
        # stage2/base.py:99): only the layer input is kept; the backward re-runs the forward launches (all but the down projection)
        # and then proceeds as usual -- same kernels on the same inputs, so the gradients are bit-identical to the keeping path.
        recompute = bool(recompute) and need_bwd
        keep = need_bwd and KEEP_LAYER_ACTIVATIONS and not recompute
        if pack is not None and want_kv:
            raise ValueError("packed ragged rows are a training path: no KV cache")
        y, (k, v), inter = _DecoderLayerFn._run_forward(x, w_in, wq, wk, wv, wo, w_post, wg, wu, wd, cos, sin, pos, seqlens, seqstart,
                                                        n_heads, n_kv, eps, need_bwd, ng, keep, pack=pack)
        ctx.pack = pack
        if need_bwd:
            if recompute:
                ctx.save_for_backward(x, w_in, wq, wk, wv, wo, w_post, wg, wu, wd, cos, sin, pos, seqlens, seqstart)
            else:
                ctx.save_for_backward(x, w_in, wq, wk, wv, wo, w_post, wg, wu, wd, cos, sin, pos, seqlens, seqstart, *inter)
            ctx.cfg = (n_heads, n_kv, eps)
            ctx.recompute = recompute
        if want_kv:
            k, v = k.contiguous(), v.contiguous()  # the cache must not pin the packed q/k/v buffer
            ctx.mark_non_differentiable(k, v)
            return y, k, v
        return y, None, None

    @staticmethod
    def backward(ctx, dy, _dk, _dv):
        n_heads, n_kv, eps = ctx.cfg
        if ctx.recompute:
            x, w_in, wq, wk, wv, wo, w_post, wg, wu, wd, cos, sin, pos, seqlens, seqstart = ctx.saved_tensors
            with torch.no_grad():   # intermediates of this layer only: freed again when this backward returns
                _, _, inter = _DecoderLayerFn._run_forward(x, w_in, wq, wk, wv, wo, w_post, wg, wu, wd, cos, sin, pos, seqlens, seqstart,
                                                           n_heads, n_kv, eps, True, ctx.needs_input_grad, KEEP_LAYER_ACTIVATIONS,
                                                           want_y=False, pack=ctx.pack)
            rstd1, qkv, o, lse, x2, rstd2, gu, h_keep, h2_keep, act_keep = inter
        else:
            (x, w_in, wq, wk, wv, wo, w_post, wg, wu, wd, cos, sin, pos, seqlens, seqstart, rstd1, qkv, o, lse, x2, rstd2,
             gu, h_keep, h2_keep, act_keep) = ctx.saved_tensors
        B, S, Hd = x.shape
        hd = Hd // n_heads
        T = B * S
        nq, nkv = n_heads * hd, n_kv * hd
        F_ = wg.shape[0]
        need = ctx.needs_input_grad
        dy = dy.contiguous()
        # ---- MLP
        g, u = gu[:, :F_], gu[:, F_:]
        act = act_keep
        # d(gate|up) straight out of the down projection's input-gradient GEMM (SwiGLU backward in its epilogue) whenever the product
        # itself need not be recomputed beside it
        dgu = ops.linear_dgrad_swiglu(dy.view(T, Hd), wd, gu) if (act_keep is not None or not need[9]) else None
        if dgu is None:
            d_act = ops.linear_dgrad(dy, wd)
            dgu = torch.empty_like(gu)
            if act is None and need[9]:
                act = torch.empty(T, F_, dtype=x.dtype, device=x.device)
            # dg, du (AND, when it was not kept, the recomputed product) in one pass
            ops.glu_bwd(d_act, g, u, 0, da=dgu[:, :F_], db=dgu[:, F_:], act_out=None if act_keep is not None else act)
            del d_act
        dwd = None
        if need[9]:
            dwd = ops.linear_wgrad(dy, act)
            del act
        wgu = _packed_view(wg, wu)
        dwg = dwu = None
        if need[7] or need[8]:
            h2 = h2_keep if h2_keep is not None else ops.rmsnorm_fwd(x2, w_post, eps)[0]  # kept, or recomputed
            if wgu is not None and need[7] and need[8]:
                dwgu = ops.linear_wgrad(dgu, h2)
                dwg, dwu = dwgu[:F_], dwgu[F_:]
            else:
                if need[7]:
                    dwg = ops.linear_wgrad(dgu[:, :F_], h2)
                if need[8]:
                    dwu = ops.linear_wgrad(dgu[:, F_:], h2)
            del h2
        if wgu is not None:
            dh2 = ops.gemm(dgu, wgu, T, Hd, 2 * F_, 2 * F_, Hd, 0, 1)
        else:
            dh2 = ops.gemm(dgu[:, :F_], wg, T, Hd, F_, 2 * F_, Hd, 0, 1)
            ops.gemm(dgu[:, F_:], wu, T, Hd, F_, 2 * F_, Hd, 0, 1, out=dh2, accumulate=True)
        del dgu
        dx2, dw_post = ops.rmsnorm_bwd(dh2, x2, w_post, rstd2, dh_in=dy, need_dw=need[6])
        del dh2
        # ---- attention
        pack = ctx.pack
        do = ops.linear_dgrad(dx2, wo)
        if pack is None:
            do = do.view(B, S, n_heads, hd)
            dwo = ops.linear_wgrad(dx2, o.view(B, S, Hd)) if need[5] else None
        else:   # compact rows: qkv / o were kept on the padded [PB, PS] grid the attention kernels ran on (see _run_forward)
            vidx, tv, PB, PS = pack
            dwo = ops.linear_wgrad(dx2, ops.gather_rows(o.view(-1, Hd), vidx)) if need[5] else None
            do_p = torch.zeros(PB * PS, Hd, dtype=x.dtype, device=x.device)
            ops.scatter_rows_(do_p, vidx[:tv], do.view(T, Hd)[:tv])
            do = do_p.view(PB, PS, n_heads, hd)
        q = qkv[:, :, :nq].unflatten(-1, (n_heads, hd))
        k = qkv[:, :, nq:nq + nkv].unflatten(-1, (n_kv, hd))
        v = qkv[:, :, nq + nkv:].unflatten(-1, (n_kv, hd))
        dqkv = torch.empty_like(qkv)
        dk_, dv_ = dqkv[:, :, nq:nq + nkv].unflatten(-1, (n_kv, hd)), dqkv[:, :, nq + nkv:].unflatten(-1, (n_kv, hd))
        ops.attn_bwd(do, q, k, v, o, lse, True, 1.0 / math.sqrt(hd), seqlens, dq=dqkv[:, :, :nq].unflatten(-1, (n_heads, hd)),
                     dk=dk_, dv=dv_, seqstart=seqstart)
        del do
        if pack is not None:   # back to compact rows (filler rows read a pad position: the backward kernels write zeros there)
            dqkv = ops.gather_rows(dqkv.view(-1, nq + 2 * nkv), pack[0]).view(B, S, nq + 2 * nkv)
        ops.rope_(dqkv[:, :, : nq + nkv].unflatten(-1, (n_heads + n_kv, hd)), cos, sin, pos, backward=True)
        d2 = dqkv.view(T, nq + 2 * nkv)
        wqkv = _packed_view(wq, wk, wv)
        dwq = dwk = dwv = None
        if need[2] or need[3] or need[4]:
            h = h_keep if h_keep is not None else ops.rmsnorm_fwd(x, w_in, eps)[0]  # kept, or recomputed
            if wqkv is not None and need[2] and need[3] and need[4]:
                dwqkv = ops.linear_wgrad(d2, h)
                dwq, dwk, dwv = dwqkv[:nq], dwqkv[nq:nq + nkv], dwqkv[nq + nkv:]
            else:
                if need[2]:
                    dwq = ops.linear_wgrad(d2[:, :nq], h)
                if need[3]:
                    dwk = ops.linear_wgrad(d2[:, nq:nq + nkv], h)
                if need[4]:
                    dwv = ops.linear_wgrad(d2[:, nq + nkv:], h)
            del h
        ld = nq + 2 * nkv
        if wqkv is not None:
            dh = ops.gemm(d2, wqkv, T, Hd, ld, ld, Hd, 0, 1)
        else:
            dh = ops.gemm(d2[:, :nq], wq, T, Hd, nq, ld, Hd, 0, 1)
            ops.gemm(d2[:, nq:nq + nkv], wk, T, Hd, nkv, ld, Hd, 0, 1, out=dh, accumulate=True)
            ops.gemm(d2[:, nq + nkv:], wv, T, Hd, nkv, ld, Hd, 0, 1, out=dh, accumulate=True)
        dx, dw_in = ops.rmsnorm_bwd(dh, x, w_in, rstd1, dh_in=dx2, need_dw=need[1])
        return (dx.view(x.shape), dw_in, dwq, dwk, dwv, dwo, dw_post, dwg, dwu, dwd) + (None,) * 11


class DreamLLMMLP(nn.Module):
    """modeling_dreamllm.py:212-239 (pretraining_tp > 1 is a numerics knob of HF checkpoints; results are identical up to
    summation order, so the single-GEMM path is always taken)."""

    def __init__(self, config: DreamLLMConfig):
        super().__init__()
        self.config = config
        self.hidden_size = config.hidden_size
        self.intermediate_size = config.intermediate_size
        self.gate_proj = nn.Linear(self.hidden_size, self.intermediate_size, bias=False)
        self.up_proj = nn.Linear(self.hidden_size, self.intermediate_size, bias=False)
        self.down_proj = nn.Linear(self.intermediate_size, self.hidden_size, bias=False)
        if config.hidden_act != "silu":
            raise ValueError("the HIP MLP kernel implements SwiGLU (hidden_act='silu') only")

    def forward(self, x):
        return ops.linear(ops.swiglu(ops.linear(x, self.gate_proj.weight), ops.linear(x, self.up_proj.weight)),
                          self.down_proj.weight)


class DreamLLMAttention(nn.Module):
    """modeling_dreamllm.py:254-400 parameters; executes as causal flash attention (the reference's
    DreamLLMFlashAttention2 semantics, modeling_dreamllm.py:403-583)."""

    def __init__(self, config: DreamLLMConfig):
        super().__init__()
        self.config = config
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.max_position_embeddings = config.max_position_embeddings
        self.rope_theta = config.rope_theta
        self.is_causal = True
        if (self.head_dim * self.num_heads) != self.hidden_size:
            raise ValueError(
                f"hidden_size must be divisible by num_heads (got `hidden_size`: {self.hidden_size} and `num_heads`: {self.num_heads})."
            )
        if self.head_dim not in (64, 128):
            raise ValueError("the HIP attention kernels support head_dim 64 and 128")
        if config.attention_bias:
            raise ValueError("attention_bias=True is not supported by the fused decoder layer")
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=False)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=False)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=False)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=False)
        self._init_rope()

    def _init_rope(self):
        """modeling_dreamllm.py:279-304."""
        rs = self.config.rope_scaling
        if rs is None:
            self.rotary_emb = RotaryEmbedding(self.head_dim, self.max_position_embeddings, base=self.rope_theta)
        else:
            if rs["type"] not in ("linear", "dynamic"):
                raise ValueError(f"Unknown RoPE scaling type {rs['type']}")
            self.rotary_emb = RotaryEmbedding(self.head_dim, self.max_position_embeddings, base=self.rope_theta,
                                              scaling_factor=rs["factor"], scaling_type=rs["type"])

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None, output_attentions=False,
                use_cache=False, **kwargs):
        """Stand-alone use (the decoder layer normally runs the fused function).  KV cache in the reference layout
        [B, Hkv, S, D]."""
        B, S, _ = hidden_states.shape
        q = ops.linear(hidden_states, self.q_proj.weight).view(B, S, self.num_heads, self.head_dim)
        k = ops.linear(hidden_states, self.k_proj.weight).view(B, S, self.num_key_value_heads, self.head_dim)
        v = ops.linear(hidden_states, self.v_proj.weight).view(B, S, self.num_key_value_heads, self.head_dim)
        past = past_key_value[0].shape[-2] if past_key_value is not None else 0
        cos, sin = self.rotary_emb.tables(S + past, hidden_states.device)
        if position_ids is None:
            position_ids = torch.arange(past, past + S, device=hidden_states.device)[None]
        q, k = ops.rope(q, k, cos, sin, position_ids)
        if past_key_value is not None:
            k = torch.cat([past_key_value[0].transpose(1, 2), k], dim=1)
            v = torch.cat([past_key_value[1].transpose(1, 2), v], dim=1)
        present = (k.transpose(1, 2), v.transpose(1, 2)) if use_cache else None
        seqstart, seqlens = kwargs.get("seqstart"), kwargs.get("seqlens")
        if seqstart is None and seqlens is None:
            seqstart, seqlens = _mask_to_spans(attention_mask, q_len=S)
        o = ops.flash_attn(q, k, v, causal=True, seqlens=seqlens, seqstart=seqstart)
        out = ops.linear(o.reshape(B, S, self.hidden_size), self.o_proj.weight)
        return out, None, present


DreamLLMFlashAttention2 = DreamLLMAttention  # same module: flash semantics are the only execution path


class _MaskHasHoles(ValueError):
    """A 2-D mask whose valid tokens are not one contiguous run per row (see `_mask_to_spans`)."""


def _mask4d_to_2d(mask4d):
    """[B, 1, Sq, Sk] mask in the format the reference's eager attention receives from `_prepare_4d_causal_attention_mask`
    (modeling_dreamllm.py:35,965-967: additive, 0 = attend / dtype-min = masked; a boolean "attend" mask is accepted too) -> the 2-D key
    mask [B, Sk] it was built from.  Only the form the reference itself builds is representable on the flash path -- causal over
    the (past + new) keys AND key padding: the newest query row sees every valid key, so it IS the key mask; the other rows are
    checked against causal & key-mask (rows whose own token is padding are not compared: their content differs between
    transformers versions and their output is discarded).  Anything else (bidirectional, sliding window, arbitrary biases)
    raises ValueError; nothing is silently approximated."""
    if mask4d.dim() != 4 or mask4d.shape[1] != 1:
        raise ValueError(f"attention_mask must be [B, Sk] or [B, 1, Sq, Sk], got {tuple(mask4d.shape)}")
    allowed = mask4d[:, 0] if mask4d.dtype == torch.bool else (mask4d[:, 0] == 0)
    B, Sq, Sk = allowed.shape
    if Sk < Sq:
        raise ValueError("4-D attention_mask: fewer keys than queries")
    key_valid = allowed[:, -1, :]
    if not (allowed.is_cuda and torch.cuda.is_current_stream_capturing()):
        past = Sk - Sq
        qi = torch.arange(Sq, device=allowed.device)[:, None] + past
        causal = torch.arange(Sk, device=allowed.device)[None, :] <= qi
        expect = causal[None] & key_valid[:, None, :]
        q_valid = key_valid[:, past:]
        # additive form: entries are 0 (attend) or a large negative (dtype-min / -inf); anything else is a bias, not a mask
        is_mask = True if mask4d.dtype == torch.bool else bool(((mask4d == 0) | (mask4d <= -1e4)).all())
        if not is_mask or not bool(((allowed == expect) | ~q_valid[:, :, None]).all()):
            raise ValueError("a 4-D attention_mask is supported in the form the reference builds (causal + key padding, "
                             "_prepare_4d_causal_attention_mask); other patterns cannot run on the flash-attention path")
    return key_valid.to(torch.long)


def _mask_to_spans(attention_mask, q_len=None):
    """2-D padding mask [B, Sk] -> (seqstart, seqlens): int32 [B] device tensors (or None) describing the ONE contiguous run
    of valid tokens of every row, which is what the flash kernels take instead of the reference's unpad / pad round trip
    (`_get_unpad_data` / `_upad_input`, modeling_dreamllm.py:69-74,553-583).

    * right padding (collator, builder_dreamllm.py:466-482; tokenizer padding_side "right", train.py:74): start 0;
    * left padding (inference callers: omni/eval/vqa/vqa_inference.py:276, omni/eval/text2img/ddp_sample_coco.py:64,
      projects/dreamllm/cli_stable_diffusion_pipeline.py:19): start = number of pad tokens in front;
    * `q_len < Sk` (a KV cache is attached): the run has to reach the last key -- the new tokens are valid by
      construction -- and only `seqstart` is returned.
    A mask with holes or more than one run cannot be expressed as a span and raises `_MaskHasHoles` (a ValueError):
    `DreamLLMModel._forward` then takes the compaction path (valid tokens gathered to the front in order, original positions kept
    for RoPE -- what `_upad_input` / `pad_input` do around the reference's flash kernel, modeling_dreamllm.py:523-545,553-583);
    silently attending to pad tokens is never an option.  A 4-D mask in the reference's eager format (causal + key padding) is
    reduced to its key mask first (`_mask4d_to_2d`); any other 4-D pattern is rejected.  The check reads
    one flag back from the device (skipped under stream capture); callers that already know the spans pass
    `seqlens=` / `seqstart=` instead of a mask and stay sync-free (bench.py, data.collate_interleaved)."""
    if attention_mask is None:
        return None, None
    if attention_mask.dim() == 4:   # the reference's eager-path format: recover the key mask it was built from
        if q_len is None:
            q_len = attention_mask.shape[2]
        attention_mask = _mask4d_to_2d(attention_mask)
    if attention_mask.dim() != 2:
        raise ValueError(f"attention_mask must be [B, Sk] or [B, 1, Sq, Sk], got {tuple(attention_mask.shape)}")
    m = attention_mask != 0
    B, Sk = m.shape
    lens = m.sum(dim=-1, dtype=torch.int32)
    start = torch.argmax(m.to(torch.int8), dim=-1).to(torch.int32)  # first valid position (0 for an all-pad row)
    with_cache = q_len is not None and q_len != Sk
    if not (m.is_cuda and torch.cuda.is_current_stream_capturing()):
        ar = torch.arange(Sk, device=m.device, dtype=torch.int32)[None]
        run = (ar >= start[:, None]) & (ar < (start + lens)[:, None])
        flags = torch.stack([(run == m).all(), (start == 0).all(), (lens == Sk).all(),
                             ((start + lens == Sk) | (lens == 0)).all()]).tolist()
        if not flags[0]:
            raise _MaskHasHoles("attention_mask must mark ONE contiguous run of valid tokens per row (left or right padding) "
                                "when a KV cache is attached; masks with holes are only supported without a cache")
        if with_cache and not flags[3]:
            raise ValueError("with past_key_values the valid keys must extend to the newest token (left padding only)")
        if flags[1] and flags[2]:
            return None, None  # dense batch
        if flags[1]:
            start = None
    if with_cache:
        return (start.contiguous() if start is not None else None), None
    return (start.contiguous() if start is not None else None), lens.contiguous()


class DreamLLMDecoderLayer(nn.Module):
    """modeling_dreamllm.py:586-654."""

    def __init__(self, config: DreamLLMConfig):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.self_attn = DreamLLMAttention(config=config)
        self.mlp = DreamLLMMLP(config)
        self.input_layernorm = DreamLLMRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = DreamLLMRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self._pack_key = None

    def pack_weights(self):
        """q/k/v and gate/up weights as row blocks of one buffer each => one GEMM per group (`pack_linear_weights`).  Skipped
        when an optimizer already owns the parameters' storage (`distributed.ShardedGradAdamW` lays its flat buffers out in
        registration order, which keeps an existing packing intact)."""
        a, m = self.self_attn, self.mlp
        ws = [a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, m.gate_proj.weight, m.up_proj.weight]
        if any(getattr(w, "_dllm_flat_owned", False) for w in ws):
            return False
        pack_linear_weights(a.q_proj, a.k_proj, a.v_proj)
        pack_linear_weights(m.gate_proj, m.up_proj)
        return True

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None, output_attentions=False,
                use_cache=False, **kwargs):
        if output_attentions:
            raise ValueError("output_attentions is not available on the flash-attention path (modeling_dreamllm.py:934-936)")
        a = self.self_attn
        if torch.compiler.is_compiling() and not torch.is_grad_enabled() and past_key_value is None:
            # torch.compile(model), inference without a cache: ONE registered op per layer (torch_ops.decoder_layer), no graph break
            B, S, _ = hidden_states.shape
            # (the model passes the tables it built once for the whole forward: inside a traced graph the per-device cache of
            # `RotaryEmbedding.tables` is not visible, and every layer would re-trace the sinusoid)
            cos, sin = kwargs.get("rope_tables") or a.rotary_emb.tables(S, hidden_states.device)
            seqlens, seqstart = kwargs.get("seqlens", None), kwargs.get("seqstart", None)
            if seqlens is None and seqstart is None and attention_mask is not None:
                seqstart, seqlens = _mask_to_spans(attention_mask)   # (reads a flag back: Dynamo breaks the graph here, once)
            pos = position_ids.expand(B, S).contiguous().view(-1).long() if position_ids is not None else None
            args = (hidden_states, self.input_layernorm.weight, a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, a.o_proj.weight,
                    self.post_attention_layernorm.weight, self.mlp.gate_proj.weight, self.mlp.up_proj.weight,
                    self.mlp.down_proj.weight, cos, sin, pos, seqlens, seqstart, a.num_heads, a.num_key_value_heads,
                    float(self.input_layernorm.variance_epsilon))
            if use_cache:
                y, k, v = torch.ops.dreamllm.decoder_layer_kv(*args)
                return (y, (k.transpose(1, 2), v.transpose(1, 2)))
            return (torch.ops.dreamllm.decoder_layer(*args),)
        if hidden_states.is_cuda and getattr(a.config, "pack_projection_weights", True):
            # cheap per-forward check (five pointer reads): `.to(dtype/device)`, `.half()` or a re-assigned parameter drop the
            # packing; re-pack then instead of silently falling back to five GEMMs (0.4 GB of copies per 7B layer, once)
            key = (a.q_proj.weight.data_ptr(), a.k_proj.weight.data_ptr(), a.v_proj.weight.data_ptr(),
                   self.mlp.gate_proj.weight.data_ptr(), self.mlp.up_proj.weight.data_ptr())
            if key != self._pack_key:
                if not self.pack_weights() and _packed_view(a.q_proj.weight.data, a.k_proj.weight.data, a.v_proj.weight.data) is None:
                    logger.warning("DreamLLMDecoderLayer: q|k|v / gate|up weights are owned by a flat optimizer buffer that splits "
                                   "them: running one GEMM per projection (pass `atomic_groups=packed_parameter_groups(model)` "
                                   "to ShardedGradAdamW)")
                self._pack_key = (a.q_proj.weight.data_ptr(), a.k_proj.weight.data_ptr(), a.v_proj.weight.data_ptr(),
                                  self.mlp.gate_proj.weight.data_ptr(), self.mlp.up_proj.weight.data_ptr())
        if past_key_value is None:
            B, S, _ = hidden_states.shape
            cos, sin = a.rotary_emb.tables(S, hidden_states.device)
            seqlens, seqstart = kwargs.get("seqlens", None), kwargs.get("seqstart", None)
            if seqlens is None and seqstart is None:
                seqstart, seqlens = _mask_to_spans(attention_mask)
            pos = None
            if position_ids is not None:
                pos = position_ids.expand(B, S).contiguous().view(-1).long()
            y, k, v = _DecoderLayerFn.apply(
                hidden_states.contiguous(), self.input_layernorm.weight, a.q_proj.weight, a.k_proj.weight, a.v_proj.weight,
                a.o_proj.weight, self.post_attention_layernorm.weight, self.mlp.gate_proj.weight, self.mlp.up_proj.weight,
                self.mlp.down_proj.weight, cos, sin, pos, seqlens, seqstart, a.num_heads, a.num_key_value_heads,
                self.input_layernorm.variance_epsilon, bool(use_cache),
                bool(kwargs.get("recompute", False)) and self.training and torch.is_grad_enabled(), kwargs.get("pack"))
            outputs = (y,)
            if use_cache:
                outputs += ((k.transpose(1, 2), v.transpose(1, 2)),)
            return outputs
        # incremental decoding with a KV cache: module-by-module path (same kernels)
        residual = hidden_states
        h = self.input_layernorm(hidden_states)
        h, _, present = a(h, attention_mask=attention_mask, position_ids=position_ids, past_key_value=past_key_value,
                          use_cache=use_cache, seqstart=kwargs.get("seqstart"), seqlens=kwargs.get("seqlens"))
        hidden_states = ops.add(residual, h)
        residual = hidden_states
        h = self.mlp(self.post_attention_layernorm(hidden_states))
        hidden_states = ops.add(residual, h)
        outputs = (hidden_states,)
        if use_cache:
            outputs += (present,)
        return outputs


@dataclass
class BaseModelOutputWithPast(ModelOutput):
    """modeling_dreamllm.py:763-800."""

    last_hidden_state: torch.FloatTensor = None
    past_key_values: tuple[tuple[torch.FloatTensor]] | None = None
    hidden_states: tuple[torch.FloatTensor] | None = None
    attentions: tuple[torch.FloatTensor] | None = None
    additional_log_info: dict[str, Any] | None = None


@dataclass
class CausalLMOutputWithPast(ModelOutput):
    """modeling_dreamllm.py:1172-1206.

    `logits` may be LAZY: the training forward computes the LM loss with the fused lm_head + cross-entropy unit
    (`ops.LMHeadCEFn`), which never holds the [B, S, V] fp32 logits (4.2 GB at the stage-II shape).  A caller that does read
    `out.logits` / `out["logits"]` gets them materialised on first access (one GEMM on the saved hidden states, detached --
    the loss gradient does not flow through them); callers that only use the loss, as the trainer does
    (omni/train/trainer.py:1061), never pay for them."""

    loss: torch.FloatTensor | None = None
    logits: torch.FloatTensor = None
    past_key_values: tuple[tuple[torch.FloatTensor]] | None = None
    hidden_states: tuple[torch.FloatTensor] | None = None
    attentions: tuple[torch.FloatTensor] | None = None
    additional_log_info: dict[str, Any] | None = None

    def set_lazy_logits(self, thunk, bulk_views=True):
        """`bulk_views=False` (set by the TRAINING forward, grad enabled): the mapping views that plumbing walks every step --
        `items()` / `values()` / `keys()` / iteration / `len()`: accelerate's `convert_outputs_to_fp32`, DDP's output flattening --
        leave pending logits out, exactly as for a `None` field, so the [B, S, V] GEMM never runs for a caller that only reads the
        loss (omni/train/trainer.py:1083,1092 read `outputs["loss"]` / `outputs["additional_log_info"]`).  Explicit requests still
        materialise them: `.logits`, `["logits"]`, `"logits" in out`, integer / slice indexing and `to_tuple()`.  With
        `bulk_views=True` (eval / no_grad: `Trainer.prediction_step` walks `outputs.items()` FOR the logits) the bulk views
        materialise too and show `logits` in slot 1 as the reference returns it."""
        object.__setattr__(self, "_logits_thunk", thunk)
        object.__setattr__(self, "_logits_bulk", bool(bulk_views))
        return self

    def _pending(self):
        return self.__dict__.get("_logits_thunk") is not None

    def _materialize_logits(self):
        thunk = self.__dict__.get("_logits_thunk")
        if thunk is None:
            return None
        object.__setattr__(self, "_logits_thunk", None)
        val = thunk()
        self.logits = val          # ModelOutput.__setattr__ registers the key -- at the END of the ordered dict
        for f in fields(self):     # restore the dataclass order: logits stays in slot 1, as the reference returns it (:1500-1509)
            if f.name in ("loss", "logits"):
                continue
            if OrderedDict.__contains__(self, f.name):
                self.move_to_end(f.name)
        return val

    def _bulk(self):
        if self._pending() and self.__dict__.get("_logits_bulk", True):
            self._materialize_logits()

    def __getattribute__(self, name):
        if name == "logits":
            val = super().__getattribute__("logits")
            if val is None and self.__dict__.get("_logits_thunk") is not None:
                return self._materialize_logits()
            return val
        return super().__getattribute__(name)

    # String keys other than "logits" (`out["loss"]`, `out["additional_log_info"]`: all the reference trainer reads) and attribute
    # access never touch the thunk.  Positional access (`out[1]`, slices, `to_tuple()`) is a request for the reference's tuple
    # layout and materialises; the bulk mapping views follow `bulk_views` (see set_lazy_logits).
    def keys(self):
        self._bulk()
        return super().keys()

    def values(self):
        self._bulk()
        return super().values()

    def items(self):
        self._bulk()
        return super().items()

    def __iter__(self):
        self._bulk()
        return super().__iter__()

    def __len__(self):
        self._bulk()
        return super().__len__()

    def __contains__(self, k):
        if k == "logits":
            self._materialize_logits()
        return super().__contains__(k)

    def __getitem__(self, k):
        if not isinstance(k, str) or k == "logits":
            self._materialize_logits()
        return super().__getitem__(k)

    def to_tuple(self):
        self._materialize_logits()
        return super().to_tuple()


class DreamLLMPreTrainedModel(PreTrainedModel, FSDPMixin):
    """modeling_dreamllm.py:657-757."""

    config_class = DreamLLMConfig
    base_model_prefix = "model"
    supports_gradient_checkpointing = True
    _no_split_modules = ["DreamLLMDecoderLayer"]
    _skip_keys_device_placement = "past_key_values"
    _supports_flash_attn_2 = True
    _keys_to_ignore_on_save = []

    def init_plugin_modules(self):
        pass

    def _init_weights(self, module):
        """modeling_dreamllm.py:674-683."""
        std = self.config.initializer_range
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()


def _extend_ignore(model, keys):
    """plugin weights are saved separately as `{save_model_name}.bin` (modeling_dreamllm.py:828-830,1232-1234); the
    attribute is a list in transformers 4.35 and a set in 5.x."""
    cur = model._keys_to_ignore_on_save
    if isinstance(cur, set):
        model._keys_to_ignore_on_save = set(cur) | set(keys)
    else:
        model._keys_to_ignore_on_save = list(cur or []) + list(keys)


def _special_id(config, token, nested=True):
    d = config.special_tokens2ids_dict
    return d["additional_special_tokens"][token] if nested else d[token]


def _slot_indices(input_ids, start_id, length, max_slots=None):
    """Flat row indices (into [B*S]) of the `length` positions that follow every `start_id` token, in (batch, position)
    order -- the order the reference's Python loops visit them (modeling_dreamllm.py:1085-1098,1110-1139)."""
    B, S = input_ids.shape
    starts = torch.nonzero(input_ids.reshape(-1) == start_id, as_tuple=False).flatten()
    if max_slots is not None:
        starts = starts[:max_slots]
    if starts.numel() > 0:
        assert int(((starts % S) + length).max()) < S, "multimodal slot runs past the end of the sequence"
    idx = starts[:, None] + 1 + torch.arange(length, device=input_ids.device)[None]
    return idx.reshape(-1), starts.numel()


# Ragged training batches (right-padded rows, `seqlens` known): the decoder runs on COMPACT rows -- the valid tokens of all rows back to
# back, filled up to a multiple of 256 rows with copies of one pad position -- so that the GEMMs, norms and element-wise passes (83 % of the
# step) do not compute on padding; only attention sees the padded grid (`_DecoderLayerFn._run_forward`).  The reference's flash path unpads
# around attention alone (modeling_dreamllm.py:523-545) and computes everything else on the padded grid; results at valid positions are the
# same, pad positions of the returned hidden states are zeros.  PACK_RAGGED_MIN_SAVING: below this share of padding it is not worth it.
PACK_RAGGED = os.environ.get("DREAMLLM_PACK_RAGGED", "1") != "0"
PACK_RAGGED_MIN_SAVING = 0.06


def _token_pack(seqlens, B, S, device):
    """(vidx int64 [Tvp] device, tv, B, S, pos int64 [Tvp] device) or None.  One device->host read of the B lengths per forward."""
    lens = [max(0, min(int(v), S)) for v in seqlens.tolist()]
    tv = sum(lens)
    if tv == 0 or tv > (1.0 - PACK_RAGGED_MIN_SAVING) * B * S:
        return None
    tvp = (tv + 255) // 256 * 256
    if tvp >= B * S:
        return None
    b_pad = next(b for b, L in enumerate(lens) if L < S)
    idx = torch.cat([torch.arange(L, dtype=torch.int64) + b * S for b, L in enumerate(lens)] +
                    [torch.full((tvp - tv,), b_pad * S + lens[b_pad], dtype=torch.int64)])
    pos = torch.cat([torch.arange(L, dtype=torch.int64) for L in lens] + [torch.zeros(tvp - tv, dtype=torch.int64)])
    return idx.to(device, non_blocking=True), tv, B, S, pos.to(device, non_blocking=True)


class DreamLLMModel(DreamLLMPreTrainedModel):
    """modeling_dreamllm.py:803-1169."""

    def __init__(self, config: DreamLLMConfig):
        super().__init__(config)
        self.padding_idx = config.pad_token_id
        self.vocab_size = config.vocab_size
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, self.padding_idx)
        self.layers = nn.ModuleList([DreamLLMDecoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.norm = DreamLLMRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.gradient_checkpointing = False
        self.post_init()

    def init_plugin_modules(self):
        """modeling_dreamllm.py:822-831."""
        for name, init_kwargs in self.config.plugins_init_kwargs.items():
            if self.config.plugins_type[name] == "embedding":
                setattr(self, name, deep_instantiate(init_kwargs).to(self.device, dtype=self.dtype))
                keys_to_ignore = [f"model.{name}.{key}" for key in getattr(self, name).state_dict().keys()]
                _extend_ignore(self, keys_to_ignore)
                logger.info(f"Added the prefix keys of `model.{name}` to the list of keys to ignore on save.")

    def fsdp_ignored_modules(self) -> list:
        ignored_modules = []
        for name, _ in self.config.plugins_init_kwargs.items():
            if self.config.plugins_type[name] == "embedding":
                ignored_modules += getattr(self, name).fsdp_ignored_modules()
        return ignored_modules

    def get_input_embeddings(self):
        return self.embed_tokens

    def set_input_embeddings(self, value):
        self.embed_tokens = value

    def embed(self, input_ids):
        if torch.compiler.is_compiling() and not torch.is_grad_enabled():
            return torch.ops.dreamllm.embedding(self.embed_tokens.weight, input_ids)
        return ops.embedding(self.embed_tokens.weight, input_ids)

    def _forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                 use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, seqlens=None,
                 seqstart=None):
        """modeling_dreamllm.py:846-1043.  `seqlens` / `seqstart` (int32 [B], device) describe the valid span of every row
        directly and replace the mask (sync-free); otherwise the spans are derived from `attention_mask` once per forward."""
        output_hidden_states = output_hidden_states if output_hidden_states is not None else self.config.output_hidden_states
        use_cache = use_cache if use_cache is not None else self.config.use_cache
        return_dict = return_dict if return_dict is not None else getattr(self.config, "return_dict", True)
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
        elif input_ids is not None:
            batch_size, seq_length = input_ids.shape[:2]
        elif inputs_embeds is not None:
            batch_size, seq_length = inputs_embeds.shape[:2]
        else:
            raise ValueError("You have to specify either input_ids or inputs_embeds")
        past_len = past_key_values[0][0].shape[2] if past_key_values is not None else 0
        if inputs_embeds is None:
            inputs_embeds = self.embed(input_ids)
        if position_ids is None and past_len > 0:
            position_ids = torch.arange(past_len, seq_length + past_len, dtype=torch.long, device=inputs_embeds.device)[None]
        unperm_order = None
        if seqlens is None and seqstart is None and attention_mask is not None:
            if attention_mask.shape[-1] != seq_length + past_len:
                raise ValueError(f"attention_mask covers {attention_mask.shape[-1]} positions, expected past + new = "
                                 f"{past_len} + {seq_length} (modeling_dreamllm.py:960-967)")
            if attention_mask.dim() == 4:   # the eager-path format (modeling_dreamllm.py:965-967) -> the key mask it was built from
                if attention_mask.shape[2] != seq_length:
                    raise ValueError(f"4-D attention_mask has {attention_mask.shape[2]} query rows, expected {seq_length}")
                attention_mask = _mask4d_to_2d(attention_mask)
            try:
                seqstart, seqlens = _mask_to_spans(attention_mask, q_len=seq_length)  # once per forward, not per layer
            except _MaskHasHoles:
                if past_len > 0 or use_cache:
                    raise
                # masks with holes (`_get_unpad_data`, modeling_dreamllm.py:69-74, handles any 0/1 mask): compact every row's valid
                # tokens to the front, in order; they keep their ORIGINAL positions for RoPE (position_ids default to arange,
                # :950-955, independent of the mask); run the decoder on the right-padded compact batch; un-permute at the end.
                # Rows at masked positions are don't-care in the reference as well (the loss never reads them).
                valid = attention_mask != 0
                unperm_order = torch.argsort((~valid).to(torch.int8), dim=1, stable=True)
                pos = (position_ids if position_ids is not None else
                       torch.arange(seq_length, device=inputs_embeds.device)[None]).expand(batch_size, seq_length)
                position_ids = torch.gather(pos, 1, unperm_order)
                inputs_embeds = torch.gather(inputs_embeds, 1, unperm_order[:, :, None].expand(-1, -1, inputs_embeds.shape[-1]))
                seqstart, seqlens = None, valid.sum(-1, dtype=torch.int32).contiguous()
            attention_mask = None
        if self.training and use_cache:
            use_cache = False
        hidden_states = inputs_embeds
        all_hidden_states = () if output_hidden_states else None
        next_decoder_cache = () if use_cache else None
        additional_log_info = {}
        extra = {}
        if self.gradient_checkpointing and self.training:
            # modeling_dreamllm.py:972-976,994-1003 (`gradient_checkpointing_enable()`, stage2/base.py:99): whole-layer activation
            # recompute, done inside _DecoderLayerFn (only each layer's input stays resident) instead of torch.utils.checkpoint
            use_cache = False
            next_decoder_cache = None
            extra["recompute"] = True
        if torch.compiler.is_compiling() and not torch.is_grad_enabled() and past_key_values is None and len(self.layers) > 0:
            extra["rope_tables"] = self.layers[0].self_attn.rotary_emb.tables(seq_length, hidden_states.device)
        pack = None
        if (PACK_RAGGED and self.training and torch.is_grad_enabled() and hidden_states.is_cuda and past_key_values is None and not use_cache
                and not output_hidden_states and seqlens is not None and seqstart is None and unperm_order is None and batch_size > 1
                and getattr(self.config, "pack_ragged_tokens", True) and not torch.compiler.is_compiling()):
            pack = _token_pack(seqlens, batch_size, seq_length, hidden_states.device)
        if pack is not None:
            vidx, tv, _, _, ppos = pack
            if position_ids is not None:   # explicit positions travel with their tokens
                ppos = position_ids.expand(batch_size, seq_length).reshape(-1)[vidx].long()
            hidden_states = ops.PackRowsFn.apply(hidden_states.reshape(batch_size * seq_length, -1), vidx, tv)[None]
            position_ids = ppos[None]
            extra["pack"] = (vidx, tv, batch_size, seq_length)
        for idx, decoder_layer in enumerate(self.layers):
            if output_hidden_states:
                all_hidden_states += (hidden_states,)
            past_key_value = past_key_values[idx] if past_key_values is not None else None
            layer_outputs = decoder_layer(hidden_states, attention_mask=attention_mask, position_ids=position_ids,
                                          past_key_value=past_key_value, use_cache=use_cache, seqlens=seqlens,
                                          seqstart=seqstart, **extra)
            hidden_states = layer_outputs[0]
            if use_cache:
                next_decoder_cache += (layer_outputs[1],)
        hidden_states = self.norm(hidden_states)
        if pack is not None:   # back to the padded grid the caller indexes (zeros at pad positions)
            hidden_states = ops.UnpackRowsFn.apply(hidden_states[0], pack[0], pack[1], batch_size * seq_length).view(batch_size, seq_length, -1)
        if output_hidden_states:
            all_hidden_states += (hidden_states,)
        if unperm_order is not None:  # back to the caller's token order
            back = lambda t: torch.zeros_like(t).scatter_(1, unperm_order[:, :, None].expand(-1, -1, t.shape[-1]), t)  # noqa: E731
            hidden_states = back(hidden_states)
            if output_hidden_states:
                all_hidden_states = tuple(back(t) for t in all_hidden_states)
        next_cache = next_decoder_cache if use_cache else None
        if not return_dict:
            return tuple(v for v in [hidden_states, next_cache, all_hidden_states, None, additional_log_info] if v is not None)
        return BaseModelOutputWithPast(last_hidden_state=hidden_states, past_key_values=next_cache,
                                       hidden_states=all_hidden_states, attentions=None,
                                       additional_log_info=additional_log_info)

    def forward(self, input_ids=None, images=None, images_dm=None, attention_mask=None, position_ids=None,
                past_key_values=None, inputs_embeds=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, dream_index=None, image_index=None, seqlens=None, seqstart=None):
        """modeling_dreamllm.py:1045-1158.  `dream_index` / `image_index` (flat row indices from the data pipeline) are an
        optional fast path that skips the device->host sync of locating the slots; semantics are unchanged."""
        embed_tokens_backup = getattr(self, "embed_tokens_backup", None)
        if embed_tokens_backup is not None:  # modeling_dreamllm.py:1059-1064
            with torch.no_grad():
                self.embed_tokens.weight[: -self.num_added_tokens] = embed_tokens_backup[: -self.num_added_tokens].data
        if inputs_embeds is None and input_ids is not None:
            inputs_embeds = self.embed(input_ids)
        image_start_id = _special_id(self.config, DEFAULT_IMAGE_START_TOKEN)
        dream_start_id = _special_id(self.config, DEFAULT_DREAM_START_TOKEN)
        if (images is not None and not self.training and past_key_values is not None
                and (input_ids is None or not bool((input_ids == image_start_id).any()))):
            images = None  # modeling_dreamllm.py:1072-1079
        B, S, H = inputs_embeds.shape if inputs_embeds is not None else (0, 0, 0)
        # replace diffusion query tokens (modeling_dreamllm.py:1081-1099)
        if images_dm is not None and inputs_embeds is not None and hasattr(self, "dream_embedding"):
            nq = self.dream_embedding.embed_len
            if dream_index is None:
                dream_index, n_slots = _slot_indices(input_ids, dream_start_id, nq)
            else:
                n_slots = dream_index.numel() // nq
            if n_slots > 0:
                rows = self.dream_embedding(1).expand(n_slots, nq, H).reshape(n_slots * nq, H)
                inputs_embeds = ops.scatter_rows(inputs_embeds.reshape(B * S, H), dream_index, rows).view(B, S, H)
        # CLIP features (modeling_dreamllm.py:1102); the zero-image dummy pass is only needed to give every trainable
        # parameter a gradient, so it is taken in training only (the reference re-runs CLIP on zeros per decoded token)
        image_features = None
        if hasattr(self, "clip_vision_embedding") and (images is not None or self.training):
            image_features = self.clip_vision_embedding(images)
        if inputs_embeds is not None:
            if images is not None and image_features is not None:
                npatch = image_features.shape[1]
                if image_index is None:
                    image_index, n_img = _slot_indices(input_ids, image_start_id, npatch, max_slots=image_features.shape[0])
                else:
                    n_img = image_index.numel() // npatch
                rows = image_features[:n_img].reshape(n_img * npatch, H)
                inputs_embeds = ops.scatter_rows(inputs_embeds.reshape(B * S, H), image_index, rows).view(B, S, H)
            elif self.training and image_features is not None:
                inputs_embeds = inputs_embeds + image_features  # dummy 0-valued term (modeling_dreamllm.py:1142-1144)
        else:
            inputs_embeds = image_features.unsqueeze(0)
        return self._forward(input_ids=None, attention_mask=attention_mask, position_ids=position_ids,
                             past_key_values=past_key_values, inputs_embeds=inputs_embeds, use_cache=use_cache,
                             output_attentions=output_attentions, output_hidden_states=output_hidden_states,
                             return_dict=return_dict, seqlens=seqlens, seqstart=seqstart)

    def prepare_dream_queries_with_special_token(self, batch_size: int = 1):
        """modeling_dreamllm.py:1161-1169."""
        dream_start_id = _special_id(self.config, DEFAULT_DREAM_START_TOKEN)
        dream_end_id = _special_id(self.config, DEFAULT_DREAM_END_TOKEN)
        sp = self.embed(torch.as_tensor([[dream_start_id, dream_end_id]], device=self.device))
        dream_queries = torch.cat([sp[..., :1, :], self.dream_embedding(), sp[..., 1:, :]], 1)
        return dream_queries.repeat(batch_size, 1, 1).to(self.device)


class DreamLLMForCausalMLM(DreamLLMPreTrainedModel):
    """modeling_dreamllm.py:1209-1509 (+ prompt encoding / pipeline front-end 1598-1880)."""

    _tied_weights_keys = {}
    _base_model_class = None                                  # set below (DreamLLMModel); the SDXL variant swaps it
    _dream_patch_token = DEFAULT_IMAGE_PATCH_TOKEN            # token filling the dream slots of the unconditional prompt
    _loss_scale_twice = False                                 # the SDXL variant divides by loss_scale twice (see there)

    def __init__(self, config: DreamLLMConfig):
        super().__init__(config)
        self.model = (self._base_model_class or DreamLLMModel)(config)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.loss_weight_lm = config.loss_weight_lm
        self.loss_weight_vm = config.loss_weight_vm
        self.post_init()

    def init_plugin_modules(self):
        """modeling_dreamllm.py:1224-1235."""
        self.model.init_plugin_modules()
        _extend_ignore(self, self.model._keys_to_ignore_on_save or [])  # the reference shares one class-level list
        for name, init_kwargs in self.config.plugins_init_kwargs.items():
            if self.config.plugins_type[name] == "head":
                setattr(self, name, deep_instantiate(init_kwargs).to(self.device, dtype=self.dtype))
                keys_to_ignore = [f"{name}.{key}" for key in getattr(self, name).state_dict().keys()]
                _extend_ignore(self, keys_to_ignore)
                logger.info(f"Added the prefix keys of `{name}` to the list of keys to ignore on save.")

    def fsdp_ignored_modules(self) -> list:
        ignored_modules = self.model.fsdp_ignored_modules()
        for name, _ in self.config.plugins_init_kwargs.items():
            if self.config.plugins_type[name] == "head":
                ignored_modules += getattr(self, name).fsdp_ignored_modules()
        return ignored_modules

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, tokenizer=None, *model_args, config=None, cache_dir=None,
                        ignore_mismatched_sizes: bool = False, force_download: bool = False, local_files_only: bool = False,
                        token=None, revision: str = "main", use_safetensors: bool = None,
                        reset_plugin_model_name_or_path: bool = False, **kwargs):
        """modeling_dreamllm.py:1244-1333 -- the positional `tokenizer` override that `projects/dreamllm/train.py:130-137`
        calls: load the LLM weights (an initial LLaMA/Vicuna checkpoint or an exported DreamLLM one), grow the embeddings to
        the tokenizer's vocabulary, point every plugin without an explicit checkpoint at the model folder, and only then
        build the plugins (`init_plugin_modules`, after the resize so `_init_weights` cannot touch them, :1328-1329).

        `use_flash_attention_2` (train.py:135) is accepted and ignored: the flash kernels are the only attention path here,
        so the reference's ImportError for a missing flash_attn wheel (:706-710) cannot occur."""
        assert tokenizer is not None, "tokenizer should not be None"
        kwargs.pop("use_flash_attention_2", None)
        kwargs.pop("attn_implementation", None)
        if config is None or isinstance(config, (str, os.PathLike)):
            config_path = config if config is not None else pretrained_model_name_or_path
            config = cls.config_class.from_pretrained(config_path, cache_dir=cache_dir, force_download=force_download,
                                                      local_files_only=local_files_only, token=token, revision=revision)
        model = super().from_pretrained(pretrained_model_name_or_path, *model_args, config=config, cache_dir=cache_dir,
                                        ignore_mismatched_sizes=ignore_mismatched_sizes, force_download=force_download,
                                        local_files_only=local_files_only, token=token, revision=revision,
                                        use_safetensors=use_safetensors, **kwargs)
        config = model.config
        if reset_plugin_model_name_or_path:
            config.reset_plugins_init_kwargs()
        if len(tokenizer) > model.config.vocab_size:
            logger.info(f"The tokenizer vocabulary size {len(tokenizer)} is larger than the model vocabulary size "
                        f"{model.config.vocab_size}. Resizing token embedding of model...")
            model.resize_token_embeddings(len(tokenizer))
            model.vocab_size = model.model.vocab_size = len(tokenizer)
        elif len(tokenizer) < model.config.vocab_size:
            logger.warning(f"The tokenizer vocabulary size {len(tokenizer)} is smaller than the model vocabulary size "
                           f"{model.config.vocab_size}. Carefully check the configuration to avoid potential issues.")
        logger.info(f"Now, the tokenizer and model vocabulary sizes are both {len(tokenizer)}.")
        # HACK (reference): add all pretrained plugins path if not specified
        for _, init_kwargs in config.plugins_init_kwargs.items():
            if init_kwargs.get("pretrained_model_name_or_path", None) is None:
                init_kwargs["pretrained_model_name_or_path"] = pretrained_model_name_or_path
        model.init_plugin_modules()
        return model

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def set_input_embeddings(self, value):
        self.model.embed_tokens = value

    def get_output_embeddings(self):
        return self.lm_head

    def set_output_embeddings(self, new_embeddings):
        self.lm_head = new_embeddings

    def set_decoder(self, decoder):
        self.model = decoder

    def get_decoder(self):
        return self.model

    def _head_loss(self, head, images_dm, enc, u_enc):
        """modeling_dreamllm.py:1441."""
        return head(images_dm, enc, u_enc)

    def _head_dummy(self, head, images_dm):
        """modeling_dreamllm.py:1445: keeps every trainable parameter in the autograd graph when a batch has no dream image."""
        return head(images_dm, None, None, self.model.dream_embedding())

    def forward(self, input_ids=None, images=None, images_dm=None, attention_mask=None, position_ids=None,
                past_key_values=None, inputs_embeds=None, labels=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, dream_index=None, image_index=None, seqlens=None, seqstart=None,
                loss_index=None):
        """modeling_dreamllm.py:1353-1509.  `loss_index` (optional, from the data pipeline: flat rows of [B*S] whose SHIFTED label is not
        -100, ascending): the fused lm_head + CE then skips the rows that carry no loss (image / dream patch slots, padding)."""
        if input_ids is not None:
            assert (
                input_ids.shape[1] <= self.config.max_position_embeddings
            ), f"the sequence length should be less than model max length {self.config.max_position_embeddings}"
        output_hidden_states = output_hidden_states if output_hidden_states is not None else self.config.output_hidden_states
        return_dict = return_dict if return_dict is not None else getattr(self.config, "return_dict", True)
        if dream_index is None and images_dm is not None and input_ids is not None and hasattr(self.model, "dream_embedding"):
            dream_index, _ = _slot_indices(input_ids, _special_id(self.config, DEFAULT_DREAM_START_TOKEN),
                                           self.model.dream_embedding.embed_len)
        outputs = self.model(input_ids=input_ids, images=images, images_dm=images_dm, attention_mask=attention_mask,
                             position_ids=position_ids, past_key_values=past_key_values, inputs_embeds=inputs_embeds,
                             use_cache=use_cache, output_attentions=output_attentions,
                             output_hidden_states=output_hidden_states, return_dict=True, dream_index=dream_index,
                             image_index=image_index, seqlens=seqlens, seqstart=seqstart)
        hidden_states = outputs.last_hidden_state
        B, S, H = hidden_states.shape

        # Let's train diffusion!  (modeling_dreamllm.py:1397-1445)
        vm_loss = 0.0
        head = getattr(self, "stable_diffusion_head", None)
        if self.training and images_dm is not None and head is not None:
            nq = self.model.dream_embedding.embed_len
            n_slots = min(dream_index.numel() // nq, images_dm.shape[0])
            enc = ops.gather_rows_unique(hidden_states.reshape(B * S, H), dream_index[: n_slots * nq]).view(n_slots, nq, H)
            u_enc = None
            if head.drop_prob is not None:  # modeling_dreamllm.py:1420-1439
                bos_id = _special_id(self.config, DEFAULT_BOS_TOKEN, nested=False)
                eos_id = _special_id(self.config, DEFAULT_EOS_TOKEN, nested=False)
                ds = _special_id(self.config, DEFAULT_DREAM_START_TOKEN)
                de = _special_id(self.config, DEFAULT_DREAM_END_TOKEN)
                dp = _special_id(self.config, self._dream_patch_token)
                u_ids = torch.tensor([[bos_id, ds] + [dp] * nq + [de, eos_id]], device=hidden_states.device)
                u_out = self.model(input_ids=u_ids, attention_mask=torch.ones_like(u_ids), use_cache=False, return_dict=True)
                u_enc = u_out.last_hidden_state[:, 2: 2 + nq, :].repeat(n_slots, 1, 1)
            vm_loss = self._head_loss(head, images_dm, enc, u_enc)
        elif self.training and head is not None:
            vm_loss = self._head_dummy(head, images_dm)

        lm_loss = 0.0
        logits = lazy_logits = None
        if labels is not None:
            # shift so that tokens < n predict n: row (b, s) is scored against labels[b, s+1]; the last row is ignored
            shift = torch.cat([labels[:, 1:], labels.new_full((B, 1), -100)], dim=1).reshape(-1)
            if getattr(self.config, "fused_lm_head_ce", True) and return_dict:
                # fused lm_head + CE: loss (and, in the same pass, its gradients) without the [B,S,V] fp32 logits
                lm_loss = ops.lm_head_ce(hidden_states.reshape(B * S, H), self.lm_head.weight, shift, rows=loss_index)
                hs, w = hidden_states.detach(), self.lm_head.weight
                lazy_logits = lambda: ops.linear_fwd(hs, ops._pad_vocab(w.detach()), out_dtype=torch.float32)[..., : w.shape[0]]  # noqa: E731
            else:
                lm_loss, logits = ops.lm_head_ce(hidden_states.reshape(B * S, H), self.lm_head.weight, shift, return_logits=True)
                logits = logits.view(B, S, -1)
        elif torch.compiler.is_compiling() and not torch.is_grad_enabled():
            logits = torch.ops.dreamllm.linear(hidden_states, self.lm_head.weight, None, True)
        else:
            logits = ops.linear(hidden_states, self.lm_head.weight, out_fp32=True)

        if self.config.loss_scale_schedule == "l1_norm":
            loss_scale = self.loss_weight_lm + self.loss_weight_vm
        elif self.config.loss_scale_schedule == "l2_norm":
            loss_scale = math.sqrt(self.loss_weight_lm**2 + self.loss_weight_vm**2)
        else:
            loss_scale = 1
        # NaN guards (modeling_dreamllm.py:1479-1486) evaluated on device: no .cpu().item() sync in the step
        if self.training and images is not None and torch.is_tensor(lm_loss):
            lm_term = torch.where(torch.isnan(lm_loss), torch.zeros_like(lm_loss), lm_loss * self.loss_weight_lm)
        else:
            lm_term = lm_loss * self.loss_weight_lm
        if self.training and images_dm is not None and torch.is_tensor(vm_loss):
            vm_term = torch.where(torch.isnan(vm_loss), torch.zeros_like(vm_loss), vm_loss * self.loss_weight_vm)
        else:
            vm_term = vm_loss * self.loss_weight_vm
        loss = (vm_term + lm_term) / loss_scale
        # The reference's SDXL model file divides by loss_scale a second time (modeling_dreamllm_sdxl.py:1485-1487).  Reproduced by default
        # for that class (`_loss_scale_twice`), and switchable per model: `config.loss_scale_twice = False` gives the single division.
        if getattr(self.config, "loss_scale_twice", self._loss_scale_twice):
            loss = loss / loss_scale
        if not torch.is_tensor(loss):
            loss = None if labels is None and not self.training else torch.as_tensor(loss, device=hidden_states.device)

        if not return_dict:
            output = (logits,) + tuple(v for v in (outputs.past_key_values, outputs.hidden_states) if v is not None)
            return (loss,) + output if loss is not None else output
        additional_log_info = {
            "lm_loss": lm_loss.detach() if torch.is_tensor(lm_loss) else lm_loss,
            "vm_loss": vm_loss.detach() if torch.is_tensor(vm_loss) else vm_loss,
        }
        additional_log_info.update(outputs.additional_log_info or {})
        out = CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=outputs.past_key_values,
                                     hidden_states=outputs.hidden_states, attentions=None,
                                     additional_log_info=additional_log_info)
        if lazy_logits is not None:
            out.set_lazy_logits(lazy_logits, bulk_views=not (self.training and torch.is_grad_enabled()))
        return out

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, **kwargs):
        """modeling_dreamllm.py:1511-1547."""
        if past_key_values:
            input_ids = input_ids[:, -1:]
        position_ids = kwargs.get("position_ids", None)
        if attention_mask is not None and position_ids is None:
            position_ids = attention_mask.long().cumsum(-1) - 1
            position_ids.masked_fill_(attention_mask == 0, 1)
            if past_key_values:
                position_ids = position_ids[:, -1].unsqueeze(-1)
        if inputs_embeds is not None and past_key_values is None:
            model_inputs = {"inputs_embeds": inputs_embeds}
        else:
            model_inputs = {"input_ids": input_ids}
        model_inputs.update({"position_ids": position_ids, "past_key_values": past_key_values,
                             "use_cache": kwargs.get("use_cache"), "attention_mask": attention_mask,
                             "images": kwargs.get("images", None)})
        return model_inputs

    @torch.no_grad()
    def greedy_generate(self, input_ids, max_new_tokens, images=None, fast=None, use_graph=True, attention_mask=None,
                        pad_token_id=None, vocab_limit="auto"):
        """Greedy decode with a KV cache: the loop of omni/eval/language_eval/modeling_dreamllm.py:47-109 with
        temperature == 0 (argmax over `logits[..., :32000]`, :79,86,92 -- `vocab_limit`).

        * `pad_token_id` given: the reference loop's ragged-batch protocol -- `input_ids` is RIGHT-padded with
          `pad_token_id`, the prefill covers the shortest prompt, and rows still inside their prompt are teacher-forced
          (`input_text_mask`, :72,93-94).  Returns the `tokens` buffer [B, max_prompt + max_new_tokens].
        * `attention_mask` given (LEFT-padded, HF-generate callers such as omni/eval/vqa/vqa_inference.py:276): mask-aware
          position ids (`prepare_inputs_for_generation`), pad keys masked.  Returns cat([input_ids, new tokens]).
        * neither: a dense batch.
        `vocab_limit="auto"`: 32000 for the language-eval protocols (the loop's own slice), the whole vocabulary when an
        `attention_mask` selects HF-generate semantics (HF's greedy search does not slice).
        `fast` (default: batch <= 8) runs the token steps on the decode kernels (GEMV + cache attention, one hipGraph replay
        per token: `decode.GreedyDecodeSession`); `fast=False` re-enters the model forward per token like the reference."""
        B, S = input_ids.shape
        dev = input_ids.device
        if vocab_limit == "auto":
            vocab_limit = None if attention_mask is not None else 32000
        V = min(int(vocab_limit), self.config.vocab_size) if vocab_limit is not None else self.config.vocab_size
        if fast is None:
            fast = B <= 8
        tokens = forced = fmask = None
        if pad_token_id is not None:
            if attention_mask is not None:
                raise ValueError("pass either pad_token_id (right-padded ragged prompts) or attention_mask (left-padded), not both")
            text_mask = input_ids != pad_token_id
            plen = text_mask.sum(-1)
            if not bool((text_mask == (torch.arange(S, device=dev)[None] < plen[:, None])).all()):
                raise ValueError("pad_token_id protocol: prompts must be right-padded")
            s0 = int(plen.min())
            total = S + max_new_tokens
            tokens = torch.full((B, total), pad_token_id, dtype=torch.long, device=dev)
            tokens[:, :S] = input_ids
            fmask = torch.zeros(B, total, dtype=torch.bool, device=dev)
            fmask[:, :S] = text_mask
            forced, fm = tokens[:, s0:], fmask[:, s0:]
            n_new = total - s0
            prompt = input_ids[:, :s0]
        else:
            prompt, n_new, fm = input_ids, max_new_tokens, None
        if fast:
            from .decode import GreedyDecodeSession
            key = (B, prompt.shape[1] + n_new + 1, use_graph, V)
            sess = getattr(self, "_decode_session", None)
            if sess is None or sess[0] != key:
                sess = (key, GreedyDecodeSession(self, B, key[1], use_graph=use_graph, vocab_limit=V))
                self._decode_session = sess
            sess = sess[1]
            first = sess.prefill(prompt, images=images, attention_mask=attention_mask, forced_tokens=forced, forced_mask=fm)
            rest = sess.generate(n_new - 1)
            new = torch.cat([first[:, None], rest], dim=1)
        else:
            position_ids = None
            if attention_mask is not None:
                position_ids = (attention_mask.long().cumsum(-1) - 1).masked_fill(attention_mask == 0, 1)
            out = self(input_ids=prompt, images=images, attention_mask=attention_mask, position_ids=position_ids, use_cache=True,
                       return_dict=True)
            past, mask, new = out.past_key_values, attention_mask, []
            for i in range(n_new):
                nxt = out.logits[:, -1, :V].argmax(-1)
                if fm is not None:
                    nxt = torch.where(fm[:, i], forced[:, i], nxt)
                new.append(nxt)
                if i + 1 == n_new:
                    break
                pos = None
                if mask is not None:
                    mask = torch.cat([mask, mask.new_ones(B, 1)], dim=1)
                    pos = (mask.long().sum(-1) - 1)[:, None]
                out = self(input_ids=nxt[:, None], past_key_values=past, attention_mask=mask, position_ids=pos, use_cache=True,
                           return_dict=True)
                past = out.past_key_values
            new = torch.stack(new, 1)
        if tokens is not None:
            tokens[:, s0:] = new
            return tokens
        return torch.cat([input_ids, new], dim=1)

    @torch.no_grad()
    def get_prompt_embeds(self, input_ids, attention_mask=None, images=None):
        """Prompt -> dream-query hidden states (modeling_dreamllm.py:1598-1672): prefill the text with a KV cache, then run
        [<dream_start>, 64 queries, <dream_end>] against it with the mask `cat([text_mask, ones])` (:1656-1657) and take the
        query positions of the last hidden state.  Batched prompts of unequal length come LEFT-padded from the reference's
        callers (padding_side="left": ddp_sample_coco.py:64, cli_stable_diffusion_pipeline.py:19); positions are the padded
        indices, as in the reference (default `position_ids`, :950-955)."""
        out = self(input_ids=input_ids, attention_mask=attention_mask, images=images, use_cache=True, return_dict=True)
        B = input_ids.shape[0]
        dq = self.model.prepare_dream_queries_with_special_token(B).to(self.dtype)
        mask2 = None
        if attention_mask is not None:
            mask2 = torch.cat([attention_mask, attention_mask.new_ones(B, dq.shape[1])], dim=1)
        out2 = self.model._forward(inputs_embeds=dq, attention_mask=mask2, past_key_values=out.past_key_values, use_cache=False,
                                   output_hidden_states=True, return_dict=True)
        return out2.hidden_states[-1][:, 1:-1, :]

    @torch.no_grad()
    def stable_diffusion_pipeline(self, prompt_ids, negative_prompt_ids, guidance_scale=7.5, num_inference_steps=50,
                                  height=None, width=None, generator=None, latents=None, output_type="latent",
                                  guidance_rescale=0.0, prompt_attention_mask=None, negative_prompt_attention_mask=None, **kw):
        """modeling_dreamllm.py:1766-1880 on token ids (tokenisation is host-side, out of scope); the masks are what the
        tokenizer returns next to the ids (`tokenizer(prompt, padding=True)`, :1609-1611)."""
        prompt_embeds = self.get_prompt_embeds(prompt_ids, attention_mask=prompt_attention_mask)
        negative = None
        if guidance_scale > 1.0:
            negative = self.get_prompt_embeds(negative_prompt_ids, attention_mask=negative_prompt_attention_mask)
        return self.stable_diffusion_head.pipeline(
            height=height, width=width, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
            generator=generator, latents=latents, prompt_embeds=prompt_embeds, negative_prompt_embeds=negative,
            output_type=output_type, guidance_rescale=guidance_rescale, **kw)
