"""TEST INFRASTRUCTURE -- generates tests/golden/causal_mlm_sdxl.pt by EXECUTING the reference's DreamLLM-SDXL model file
(omni/models/dreamllm_sdxl/modeling_dreamllm_sdxl.py, needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_sdxl

Pins what differs from the base model file (see dreamllm_amd/modeling_dreamllm_sdxl.py): `add_time_ids` reaching the head as
4th positional argument, the dummy head call with `None` in that position, `<dream_patch>` in the unconditional prompt of the
CFG-drop pass (head.drop_prob set), the second division by `loss_scale` (l1_norm schedule so that it is visible), and
`inv_freq` absent from the state_dict.  Fake plugins stand in for the diffusers-based ones, exactly as in make_golden.py.
"""
from __future__ import annotations

import importlib

import torch
import torch.nn as nn

from . import ref_loader
from .make_golden import HID, TINY, bf16r, init_params, rel, save

SPECIAL = {"pad": 150, "image": 151, "im_patch": 152, "im_start": 153, "im_end": 154, "dream": 155, "dream_patch": 156,
           "dream_start": 157, "dream_end": 158}


def special_tokens2ids_dict():
    return {"<s>": 1, "</s>": 2, "<unk>": 0, "[PAD]": SPECIAL["pad"],
            "additional_special_tokens": {"<image>": SPECIAL["image"], "<im_patch>": SPECIAL["im_patch"],
                                          "<im_start>": SPECIAL["im_start"], "<im_end>": SPECIAL["im_end"],
                                          "<dream>": SPECIAL["dream"], "<dream_patch>": SPECIAL["dream_patch"],
                                          "<dream_start>": SPECIAL["dream_start"], "<dream_end>": SPECIAL["dream_end"]}}


def main():
    torch.manual_seed(7)
    ref_loader.install()
    m = importlib.import_module("omni.models.dreamllm_sdxl.modeling_dreamllm_sdxl")
    cfgmod = importlib.import_module("omni.models.dreamllm_sdxl.configuration_dreamllm_sdxl")
    cfg = cfgmod.DreamLLMSDXLConfig(**{**TINY, "vocab_size": 168})
    cfg.rope_scaling = None
    cfg.special_tokens2ids_dict = special_tokens2ids_dict()
    cfg.loss_weight_lm, cfg.loss_weight_vm, cfg.loss_scale_schedule = 1.0, 3.0, "l1_norm"  # loss_scale = 4, applied twice

    class FakeDream(nn.Module):
        embed_len = 4

        def __init__(self):
            super().__init__()
            self.dream_queries = nn.Parameter(bf16r(torch.randn(1, 4, HID) * 0.05))

        def forward(self, batch_size=1):
            return self.dream_queries.repeat(batch_size, 1, 1)

    class FakeClip(nn.Module):
        embed_len = 6

        def forward(self, images=None):
            return torch.zeros(())  # creation-only batches: no comprehension images

    class FakeHead(nn.Module):
        drop_prob = 0.1  # not None => the model runs the unconditional (<dream_patch>) pass and hands its states over

        def forward(self, images, encoder_hidden_states, u=None, add_time_ids=None, dream_embeddings=None):
            if images is None:
                assert add_time_ids is None
                return (0.0 * dream_embeddings).sum()
            t = (add_time_ids.float() / 100.0).sum(-1)[:, None, None]
            return ((encoder_hidden_states.float() * images).pow(2) * t).mean() + 0.5 * (u.float() * images).pow(2).mean()

    lm = m.DreamLLMSDXLForCausalMLM(cfg)
    init_params(lm)
    lm.model.dream_embedding = FakeDream()
    lm.model.clip_vision_embedding = FakeClip()
    lm.stable_diffusion_head = FakeHead()
    lm.train()
    assert not any("inv_freq" in k for k in lm.state_dict()), "sdxl variant registers inv_freq non-persistent"
    B, S = 2, 24
    sp = SPECIAL
    ids = torch.randint(3, 150, (B, S))
    ids[:, 0] = 1
    for b, at in ((0, 5), (1, 11)):
        ids[b, at: at + 6] = torch.tensor([sp["dream_start"]] + [sp["dream_patch"]] * 4 + [sp["dream_end"]])
    labels = ids.clone()
    for t_ in (sp["dream_patch"], sp["dream_end"]):
        labels[ids == t_] = -100
    am = torch.ones(B, S, dtype=torch.long)
    images_dm = bf16r(torch.randn(B, 4, HID))
    tids = torch.tensor([[128., 128, 0, 0, 128, 128], [200., 160, 8, 16, 128, 128]])
    out = lm(input_ids=ids, images=None, images_dm=images_dm, add_time_ids=tids, attention_mask=am, labels=labels, return_dict=True)
    out.loss.backward()
    lmv = torch.as_tensor(out.additional_log_info["lm_loss"]).detach().float()
    vmv = torch.as_tensor(out.additional_log_info["vm_loss"]).detach().float()
    # the reference divides by loss_scale twice (modeling_dreamllm_sdxl.py:1485-1487)
    expect = (vmv * 3.0 + lmv * 1.0) / 4.0 / 4.0
    assert abs(float(expect) - float(out.loss)) < 1e-5 * abs(float(out.loss)), (float(expect), float(out.loss))
    b16 = lambda v: v.to(torch.bfloat16) if v.is_floating_point() else v
    g = dict(cfg=dict(TINY, vocab_size=168, special=SPECIAL), sd={k: b16(v.detach().clone()) for k, v in lm.state_dict().items()},
             input_ids=ids, labels=labels, attention_mask=am, images_dm=images_dm, add_time_ids=tids,
             loss=out.loss.detach(), logits=out.logits.detach(), lm_loss=lmv.detach(), vm_loss=vmv.detach(),
             grad_dream=lm.model.dream_embedding.dream_queries.grad.detach().clone(),
             grad_q0=b16(lm.model.layers[0].self_attn.q_proj.weight.grad.detach().clone()),
             loss_weight_lm=1.0, loss_weight_vm=3.0, loss_scale_schedule="l1_norm")
    # dummy branch (no dream images): head(images_dm, None, None, None, dream_queries)
    lm.zero_grad()
    out2 = lm(input_ids=ids, images=None, images_dm=None, add_time_ids=None, attention_mask=am, labels=labels, return_dict=True)
    g["loss_dummy"] = out2.loss.detach()
    assert rel(out2.logits.detach(), out.logits.detach()) > 1e-3  # dream slots are NOT overwritten without images_dm
    save("causal_mlm_sdxl.pt", g)


if __name__ == "__main__":
    main()
