"""TEST INFRASTRUCTURE -- generates tests/golden/err_ref.pt: the `err_ref` yard-sticks of the tolerance contract, obtained
by EXECUTING the reference classes in bfloat16 on CPU (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_errref

Contract (tests/test_model_gpu.py): storage is bf16, so every tensor check is
`err_ours <= 1.5 * err_ref + 2e-3` with `err_ref = ||reference_run_in_bf16 - reference_run_in_fp32|| / ||fp32||` for the SAME
quantity, and every fp32 scalar (loss) check is `|ours - fp32| <= 1.5 * |bf16_run - fp32| + 1e-3 * |fp32|`.  This script
loads the committed fp32 fixtures (decoder_layer.pt, causal_mlm.pt, causal_mlm_sdxl.pt, model_forward.pt), rebuilds the
reference modules from their stored state dicts, runs them with parameters and activations in bfloat16
(`module.to(torch.bfloat16)`, the reference's own inference/training dtype) and stores the per-quantity errors.  Nothing is
restated here: both runs are the reference's code.
"""
from __future__ import annotations

import importlib
import os

import torch
import torch.nn as nn

from . import ref_loader
from .make_golden import HID, OUT, TINY, rel, save, special_tokens2ids_dict
from .make_golden_sdxl import special_tokens2ids_dict as special_tokens2ids_dict_sdxl

BF = torch.bfloat16


def _load(name):
    return torch.load(os.path.join(OUT, name), map_location="cpu", weights_only=False)


class FakeDream(nn.Module):
    embed_len = 4

    def __init__(self):
        super().__init__()
        self.dream_queries = nn.Parameter(torch.zeros(1, 4, HID))

    def forward(self, batch_size=1):
        return self.dream_queries.repeat(batch_size, 1, 1)


class FakeClip(nn.Module):
    embed_len = 6

    def __init__(self, with_proj=True):
        super().__init__()
        if with_proj:
            self.proj = nn.Linear(8, HID)

    def forward(self, images=None):
        if not hasattr(self, "proj"):
            return torch.zeros(())
        if images is None:
            return (0.0 * self.proj(torch.zeros(1, 6, 8, dtype=self.proj.weight.dtype))).sum()
        return self.proj(images)


class FakeHead(nn.Module):
    drop_prob = None

    def forward(self, images, encoder_hidden_states, u=None, dream_embeddings=None):
        if images is None:
            return (0.0 * dream_embeddings).sum()
        return (encoder_hidden_states.float() * images).pow(2).mean()


class FakeHeadXL(nn.Module):
    drop_prob = 0.1

    def forward(self, images, encoder_hidden_states, u=None, add_time_ids=None, dream_embeddings=None):
        if images is None:
            return (0.0 * dream_embeddings).sum()
        t = (add_time_ids.float() / 100.0).sum(-1)[:, None, None]
        return ((encoder_hidden_states.float() * images).pow(2) * t).mean() + 0.5 * (u.float() * images).pow(2).mean()


def _scalar(bf, fp):
    return dict(fp32=float(fp), bf16=float(bf), abs_err=abs(float(bf) - float(fp)))


def main():
    m = ref_loader.load_modeling()
    from transformers.modeling_attn_mask_utils import _prepare_4d_causal_attention_mask
    E = {}

    # ---- decoder layer fwd + bwd
    g = _load("decoder_layer.pt")
    cfg = ref_loader.make_config(**{**TINY, "num_key_value_heads": g["cfg"]["num_key_value_heads"]})
    layer = m.DreamLLMDecoderLayer(cfg)
    layer.load_state_dict({k: v.float() for k, v in g["sd"].items()})
    layer = layer.to(BF)
    B, S, _ = g["x"].shape
    x = g["x"].to(BF).requires_grad_(True)
    mask = _prepare_4d_causal_attention_mask(None, (B, S), x, 0)
    y = layer(x, attention_mask=mask, position_ids=torch.arange(S)[None])[0]
    y.backward(g["dy"].to(BF))
    E["decoder_layer"] = dict(y=rel(y.detach().float(), g["y"]), dx=rel(x.grad.float(), g["dx"]),
                              grads={n: rel(p.grad.float(), g["grads"][n].float()) for n, p in layer.named_parameters()})

    # ---- DreamLLMForCausalMLM (stage-II interleaved step)
    g = _load("causal_mlm.pt")
    cfg = ref_loader.make_config(**TINY)
    cfg.special_tokens2ids_dict = special_tokens2ids_dict()
    lm = m.DreamLLMForCausalMLM(cfg)
    lm.model.dream_embedding = FakeDream()
    lm.model.clip_vision_embedding = FakeClip()
    lm.stable_diffusion_head = FakeHead()
    res = lm.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in g["sd"].items()}, strict=False)
    assert not res.unexpected_keys and all("inv_freq" in k for k in res.missing_keys), res
    lm = lm.to(BF).train()
    out = lm(input_ids=g["input_ids"], images=g["images"].to(BF), images_dm=g["images_dm"].to(BF),
             attention_mask=g["attention_mask"], labels=g["labels"], return_dict=True)
    out.loss.backward()
    am = g["attention_mask"]
    E["causal_mlm"] = dict(
        logits=[rel(out.logits[b, : int(am[b].sum())].detach().float(), g["logits"][b, : int(am[b].sum())]) for b in range(am.shape[0])],
        lm_loss=_scalar(out.additional_log_info["lm_loss"], g["lm_loss"]),
        vm_loss=_scalar(out.additional_log_info["vm_loss"], g["vm_loss"]),
        loss=_scalar(out.loss.detach(), g["loss"]),
        grad_dream=rel(lm.model.dream_embedding.dream_queries.grad.float(), g["grad_dream"]),
        grad_lm_head=rel(lm.lm_head.weight.grad.float(), g["grad_lm_head"].float()),
        grad_q0=rel(lm.model.layers[0].self_attn.q_proj.weight.grad.float(), g["grad_q0"].float()),
        grad_embed=rel(lm.model.embed_tokens.weight.grad.float(), g["grad_embed"].float()),
        grad_clip_proj=rel(lm.model.clip_vision_embedding.proj.weight.grad.float(), g["grad_clip_proj"]))

    # ---- DreamLLM-SDXL model file
    g = _load("causal_mlm_sdxl.pt")
    mx = importlib.import_module("omni.models.dreamllm_sdxl.modeling_dreamllm_sdxl")
    cfgmod = importlib.import_module("omni.models.dreamllm_sdxl.configuration_dreamllm_sdxl")
    cfgx = cfgmod.DreamLLMSDXLConfig(**{**TINY, "vocab_size": 168})
    cfgx.rope_scaling = None
    cfgx.special_tokens2ids_dict = special_tokens2ids_dict_sdxl()
    cfgx.loss_weight_lm, cfgx.loss_weight_vm, cfgx.loss_scale_schedule = g["loss_weight_lm"], g["loss_weight_vm"], g["loss_scale_schedule"]
    lx = mx.DreamLLMSDXLForCausalMLM(cfgx)
    lx.model.dream_embedding = FakeDream()
    lx.model.clip_vision_embedding = FakeClip(with_proj=False)
    lx.stable_diffusion_head = FakeHeadXL()
    res = lx.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in g["sd"].items()}, strict=False)
    assert not res.unexpected_keys and not res.missing_keys, res
    lx = lx.to(BF).train()
    out = lx(input_ids=g["input_ids"], images=None, images_dm=g["images_dm"].to(BF), add_time_ids=g["add_time_ids"],
             attention_mask=g["attention_mask"], labels=g["labels"], return_dict=True)
    out.loss.backward()
    E["causal_mlm_sdxl"] = dict(
        logits=rel(out.logits.detach().float(), g["logits"]),
        lm_loss=_scalar(torch.as_tensor(out.additional_log_info["lm_loss"]).float(), g["lm_loss"]),
        vm_loss=_scalar(torch.as_tensor(out.additional_log_info["vm_loss"]).float(), g["vm_loss"]),
        loss=_scalar(out.loss.detach(), g["loss"]),
        grad_dream=rel(lx.model.dream_embedding.dream_queries.grad.float(), g["grad_dream"]),
        grad_q0=rel(lx.model.layers[0].self_attn.q_proj.weight.grad.float(), g["grad_q0"].float()))
    lx.zero_grad()
    out2 = lx(input_ids=g["input_ids"], images=None, images_dm=None, add_time_ids=None, attention_mask=g["attention_mask"],
              labels=g["labels"], return_dict=True)
    E["causal_mlm_sdxl"]["loss_dummy"] = _scalar(out2.loss.detach(), g["loss_dummy"])

    def show(d, ind=0):
        for k, v in d.items():
            if isinstance(v, dict) and "abs_err" not in v:
                print(" " * ind + k)
                show(v, ind + 2)
            else:
                print(" " * ind + f"{k}: {v}")

    show(E)
    save("err_ref.pt", E)


if __name__ == "__main__":
    main()
