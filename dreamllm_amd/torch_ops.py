"""`torch.library` registration of the inference-path operators (SURVEY.md §8-b1 threading note: the reference's inference scripts
call `torch.compile(model)`, projects/dreamllm/inference.py:70, omni/eval/vqa/vqa_inference.py:297).

Every kernel launch of this package goes through ctypes, which Dynamo cannot trace.  Round 2 marked the operator entry points opaque
(`ops.make_dynamo_opaque()`: a graph break around every call).  Here the operators of the no-grad, no-cache forward -- token
embedding, RMSNorm, the fused decoder layer, Linear / lm_head -- are registered as custom ops `torch.ops.dreamllm.*` with fake
(meta) kernels, so `torch.compile(model)` captures the text forward of `DreamLLMForCausalMLM` as ONE graph whose nodes are these
ops, with no graph break; the modules route through them only while Dynamo is tracing (`torch.compiler.is_compiling()`) and
autograd is off -- eager execution and training are untouched (the autograd Functions of ops.py stay the training path).  Paths
that are not registered (KV-cache decode, the multimodal splice, the diffusion head) keep working under `torch.compile` through
`ops.make_dynamo_opaque()`.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor
from torch.library import custom_op

from . import ops


@custom_op("dreamllm::embedding", mutates_args=())
def embedding(weight: Tensor, ids: Tensor) -> Tensor:
    with torch.no_grad():
        return ops.embedding(weight, ids)


@embedding.register_fake
def _(weight, ids):
    return weight.new_empty(*ids.shape, weight.shape[1])


@custom_op("dreamllm::rmsnorm", mutates_args=())
def rmsnorm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    with torch.no_grad():
        return ops.rmsnorm(x, weight, eps)


@rmsnorm.register_fake
def _(x, weight, eps):
    return torch.empty_like(x)


@custom_op("dreamllm::linear", mutates_args=())
def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], out_fp32: bool) -> Tensor:
    with torch.no_grad():
        return ops.linear(x, weight, bias, None, out_fp32)


@linear.register_fake
def _(x, weight, bias, out_fp32):
    return x.new_empty(*x.shape[:-1], weight.shape[0], dtype=torch.float32 if out_fp32 else x.dtype)


@custom_op("dreamllm::decoder_layer", mutates_args=())
def decoder_layer(x: Tensor, w_in: Tensor, wq: Tensor, wk: Tensor, wv: Tensor, wo: Tensor, w_post: Tensor, wg: Tensor, wu: Tensor,
                  wd: Tensor, cos: Tensor, sin: Tensor, pos: Optional[Tensor], seqlens: Optional[Tensor],
                  seqstart: Optional[Tensor], n_heads: int, n_kv: int, eps: float) -> Tensor:
    """DreamLLMDecoderLayer.forward without a KV cache (modeling_dreamllm.py:599-654), inference: the fused layer function."""
    from .modeling_dreamllm import _DecoderLayerFn
    with torch.no_grad():
        return _DecoderLayerFn.apply(x.contiguous(), w_in, wq, wk, wv, wo, w_post, wg, wu, wd, cos, sin, pos, seqlens, seqstart,
                                     n_heads, n_kv, eps, False)[0]


@decoder_layer.register_fake
def _(x, w_in, wq, wk, wv, wo, w_post, wg, wu, wd, cos, sin, pos, seqlens, seqstart, n_heads, n_kv, eps):
    return torch.empty_like(x)


@custom_op("dreamllm::decoder_layer_kv", mutates_args=())
def decoder_layer_kv(x: Tensor, w_in: Tensor, wq: Tensor, wk: Tensor, wv: Tensor, wo: Tensor, w_post: Tensor, wg: Tensor, wu: Tensor,
                     wd: Tensor, cos: Tensor, sin: Tensor, pos: Optional[Tensor], seqlens: Optional[Tensor],
                     seqstart: Optional[Tensor], n_heads: int, n_kv: int, eps: float) -> tuple[Tensor, Tensor, Tensor]:
    """The prefill form (`use_cache=True`, no past): also returns this layer's rotated keys and values [B, S, H_kv, D]."""
    from .modeling_dreamllm import _DecoderLayerFn
    with torch.no_grad():
        y, k, v = _DecoderLayerFn.apply(x.contiguous(), w_in, wq, wk, wv, wo, w_post, wg, wu, wd, cos, sin, pos, seqlens, seqstart,
                                        n_heads, n_kv, eps, True)
    return y, k, v


@decoder_layer_kv.register_fake
def _(x, w_in, wq, wk, wv, wo, w_post, wg, wu, wd, cos, sin, pos, seqlens, seqstart, n_heads, n_kv, eps):
    B, S, H = x.shape
    hd = H // n_heads
    return torch.empty_like(x), x.new_empty(B, S, n_kv, hd), x.new_empty(B, S, n_kv, hd)


def tracing_inference() -> bool:
    """True while Dynamo traces a forward that needs no gradient: the modules then call `torch.ops.dreamllm.*`."""
    return torch.compiler.is_compiling() and not torch.is_grad_enabled()
