"""TEST INFRASTRUCTURE -- generates tests/golden/*.pt by EXECUTING the real reference (needs /root/reference).

Run in the authoring container:   PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden
The fixtures are small seeded input/output pairs of the reference's own classes (fp32 on CPU); tests/test_oracle.py
replays them against the restatement in oracle/llm_ref.py, and the `-m gpu` tests replay them against the HIP path.
"""
from __future__ import annotations

import os
import sys

import torch
import torch.nn as nn

from . import llm_ref, ref_loader

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# head_dim = 64 (the HIP attention kernels support 64 and 128)
TINY = dict(vocab_size=160, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
            max_position_embeddings=128, rms_norm_eps=1e-6)
HID = TINY["hidden_size"]


def bf16r(t):
    """round to bf16-representable fp32 so the HIP path (bf16 storage) sees bit-identical inputs/weights"""
    return t.to(torch.bfloat16).to(torch.float32)


def init_params(module):
    for p in module.parameters():
        if p.dim() >= 2:
            p.data = bf16r(torch.randn_like(p) * 0.05)
        else:
            p.data = bf16r(1.0 + 0.1 * torch.randn_like(p))
# special ids laid out like tokenization_dreamllm.py:78-94 appended after a base vocab of 150 (+ [PAD])
SPECIAL = {"pad": 150, "image": 151, "im_patch": 152, "im_start": 153, "im_end": 154, "dream": 155, "dream_start": 156,
           "dream_end": 157}


def special_tokens2ids_dict():
    return {
        "<s>": 1, "</s>": 2, "<unk>": 0, "[PAD]": SPECIAL["pad"],
        "additional_special_tokens": {
            "<image>": SPECIAL["image"], "<im_patch>": SPECIAL["im_patch"], "<im_start>": SPECIAL["im_start"],
            "<im_end>": SPECIAL["im_end"], "<dream>": SPECIAL["dream"], "<dream_start>": SPECIAL["dream_start"],
            "<dream_end>": SPECIAL["dream_end"],
        },
    }


def cfg_dict(cfg):
    return dict(num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                num_key_value_heads=cfg.num_key_value_heads, rms_norm_eps=cfg.rms_norm_eps,
                max_position_embeddings=cfg.max_position_embeddings, rope_theta=cfg.rope_theta,
                hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
                special_ids={"dream_start": SPECIAL["dream_start"], "im_start": SPECIAL["im_start"]})


def save(name, obj):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    torch.save(obj, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def main():
    torch.manual_seed(0)
    m = ref_loader.load_modeling()
    from transformers.modeling_attn_mask_utils import _prepare_4d_causal_attention_mask

    # ---- 1. RMSNorm (fp32 and bf16 reference behaviour)
    norm = m.DreamLLMRMSNorm(96, eps=1e-6)
    norm.weight.data = 1.0 + 0.1 * torch.randn(96)
    x = torch.randn(5, 96)
    g = dict(x=x, w=norm.weight.data.clone(), eps=1e-6, y=norm(x).detach(),
             y_bf16=norm.to(torch.bfloat16)(x.to(torch.bfloat16)).detach())
    norm.float()
    assert rel(llm_ref.rmsnorm(x, g["w"], 1e-6), g["y"]) < 1e-6
    save("rmsnorm.pt", g)

    # ---- 2. RoPE
    rot = m.RotaryEmbedding(32, max_position_embeddings=64)
    q, k = torch.randn(2, 4, 16, 32), torch.randn(2, 4, 16, 32)
    pos = torch.stack([torch.arange(16), torch.arange(3, 19)])
    cos, sin = rot(q, seq_len=64)
    qo, ko = m.apply_rotary_pos_emb(q, k, cos, sin, pos)
    c2, s2 = llm_ref.rope_tables(32, 64)
    q2, k2 = llm_ref.apply_rope(q, k, c2, s2, pos)
    assert rel(q2, qo) < 1e-6 and rel(k2, ko) < 1e-6
    save("rope.pt", dict(q=q, k=k, pos=pos, q_out=qo, k_out=ko))

    # ---- 3. decoder layer forward + backward
    cfg_gqa = ref_loader.make_config(**{**TINY, "num_key_value_heads": 1})
    cfg = ref_loader.make_config(**TINY)
    cfg.special_tokens2ids_dict = special_tokens2ids_dict()
    layer = m.DreamLLMDecoderLayer(cfg_gqa)
    init_params(layer)
    B, S = 2, 24
    x = bf16r(torch.randn(B, S, HID)).requires_grad_(True)
    mask = _prepare_4d_causal_attention_mask(None, (B, S), x, 0)
    pos = torch.arange(S)[None]
    y = layer(x, attention_mask=mask, position_ids=pos)[0]
    dy = bf16r(torch.randn_like(y))
    y.backward(dy)
    sd = {k_: v.detach().clone() for k_, v in layer.state_dict().items()}
    grads = {n: p.grad.detach().clone() for n, p in layer.named_parameters()}
    cd = cfg_dict(cfg)
    cdg = cfg_dict(cfg_gqa)
    c2, s2 = llm_ref.rope_tables(64, 128)
    y2 = llm_ref.decoder_layer(x.detach(), sd, "", cdg, c2, s2, pos, llm_ref.causal_mask_4d(None, B, S, torch.float32))
    assert rel(y2, y.detach()) < 1e-5, rel(y2, y.detach())
    save("decoder_layer.pt", dict(cfg=cdg, sd={k_: v.to(torch.bfloat16) for k_, v in sd.items()}, x=x.detach(), y=y.detach(),
                                  dy=dy, dx=x.grad.detach(), grads={k_: v.to(torch.bfloat16) for k_, v in grads.items()}))

    # ---- 4. DreamLLMModel._forward with right padding
    model = m.DreamLLMModel(cfg)
    init_params(model)
    emb = bf16r(torch.randn(B, S, HID))
    am = torch.ones(B, S, dtype=torch.long)
    am[1, 17:] = 0
    out = model._forward(inputs_embeds=emb, attention_mask=am).last_hidden_state.detach()
    sdm = {"model." + k_: v.detach().clone() for k_, v in model.state_dict().items()}
    out2 = llm_ref.model_forward(emb, sdm, cd, attention_mask=am)
    assert rel(out2[1, :17], out[1, :17]) < 1e-5 and rel(out2[0], out[0]) < 1e-5
    save("model_forward.pt", dict(cfg=cd, sd={k_: (v.to(torch.bfloat16) if v.is_floating_point() else v) for k_, v in sdm.items()},
                                  emb=emb, attention_mask=am, out=out))

    # ---- 5. DreamLLMForCausalMLM.forward with fake plugins (pins splice + dream-state gather + loss mix)
    class FakeDream(nn.Module):
        embed_len = 4

        def __init__(self):
            super().__init__()
            self.dream_queries = nn.Parameter(bf16r(torch.randn(1, 4, HID) * 0.05))

        def forward(self, batch_size=1):
            return self.dream_queries.repeat(batch_size, 1, 1)

    class FakeClip(nn.Module):
        embed_len = 6

        def __init__(self):
            super().__init__()
            self.proj = nn.Linear(8, HID)

        def forward(self, images=None):
            if images is None:
                return (0.0 * self.proj(torch.zeros(1, 6, 8))).sum()
            return self.proj(images)  # images here are already [n_img, 6, 8] "patch features"

    class FakeHead(nn.Module):
        drop_prob = None

        def forward(self, images, encoder_hidden_states, u=None, dream_embeddings=None):
            if images is None:
                return (0.0 * dream_embeddings).sum()
            return (encoder_hidden_states.float() * images).pow(2).mean()

    lm = m.DreamLLMForCausalMLM(cfg)
    init_params(lm)
    lm.model.dream_embedding = FakeDream()
    lm.model.clip_vision_embedding = FakeClip()
    init_params(lm.model.clip_vision_embedding)
    lm.model.clip_vision_embedding.proj.bias.data = bf16r(0.1 * torch.randn(HID))
    lm.stable_diffusion_head = FakeHead()
    lm.train()
    S2 = 40
    ids = torch.randint(3, 150, (B, S2))
    sp = SPECIAL

    def put(row, at, toks):
        ids[row, at: at + len(toks)] = torch.tensor(toks)

    put(0, 2, [sp["dream_start"]] + [sp["im_patch"]] * 4 + [sp["dream_end"]] + [sp["im_start"]] + [sp["im_patch"]] * 6 + [sp["im_end"]])
    put(0, 22, [sp["dream_start"]] + [sp["im_patch"]] * 4 + [sp["dream_end"]] + [sp["im_start"]] + [sp["im_patch"]] * 6 + [sp["im_end"]])
    put(1, 5, [sp["im_start"]] + [sp["im_patch"]] * 6 + [sp["im_end"]] + [sp["dream_start"]] + [sp["im_patch"]] * 4 + [sp["dream_end"]])
    ids[:, 0] = 1
    labels = ids.clone()
    for t_ in (sp["im_patch"], sp["im_end"], sp["dream_end"]):
        labels[ids == t_] = -100
    am = torch.ones(B, S2, dtype=torch.long)
    am[1, 33:] = 0
    labels[am == 0] = -100
    images = bf16r(torch.randn(3, 6, 8))
    images_dm = bf16r(torch.randn(3, 4, HID))
    out = lm(input_ids=ids, images=images, images_dm=images_dm, attention_mask=am, labels=labels, return_dict=True)
    out.loss.backward()
    sdl = {k_: v.detach().clone() for k_, v in lm.state_dict().items()}
    b16 = lambda v: v.to(torch.bfloat16) if v.is_floating_point() else v
    g5 = dict(cfg=cd, sd={k_: b16(v) for k_, v in sdl.items()}, input_ids=ids, labels=labels, attention_mask=am, images=images, images_dm=images_dm,
              loss=out.loss.detach(), logits=out.logits.detach(), lm_loss=out.additional_log_info["lm_loss"],
              vm_loss=out.additional_log_info["vm_loss"],
              grad_dream=lm.model.dream_embedding.dream_queries.grad.detach().clone(),
              grad_embed=b16(lm.model.embed_tokens.weight.grad.detach().clone()),
              grad_lm_head=b16(lm.lm_head.weight.grad.detach().clone()),
              grad_q0=b16(lm.model.layers[0].self_attn.q_proj.weight.grad.detach().clone()),
              grad_clip_proj=lm.model.clip_vision_embedding.proj.weight.grad.detach().clone(),
              loss_weight_lm=cfg.loss_weight_lm, loss_weight_vm=cfg.loss_weight_vm)
    # restatement check
    with torch.no_grad():
        feats = lm.model.clip_vision_embedding(images)
        emb2 = llm_ref.splice_inputs(ids, sdl, cd, dream_queries=sdl["model.dream_embedding.dream_queries"][0],
                                     image_features=feats)
        h2 = llm_ref.model_forward(emb2, sdl, cd, attention_mask=am)
        enc = llm_ref.gather_dream_states(h2, ids, cd, 4, images_dm.shape[0])
        vm = (enc.float() * images_dm).pow(2).mean()
        lml, logits2 = llm_ref.lm_loss(h2, sdl["lm_head.weight"], labels)
        loss2 = vm * cfg.loss_weight_vm + lml * cfg.loss_weight_lm
    assert rel(logits2[0], out.logits[0].detach()) < 1e-5
    assert abs(loss2.item() - out.loss.item()) < 1e-4 * abs(out.loss.item()), (loss2.item(), out.loss.item())
    save("causal_mlm.pt", g5)

    # ---- 6. projectors
    pb = ref_loader.load_projector()
    lin = pb.build_projector(dict(projector="linear", freeze_projector=False, depth=1, save_model_name="clip",
                                  model_name_or_path=None), 48, 64, bias=True)
    mlp = pb.build_projector(dict(projector="mlp", freeze_projector=False, depth=2, save_model_name="sd",
                                  model_name_or_path=None), 64, 32, bias=False)
    init_params(lin)
    init_params(mlp)
    lin.projector.bias.data = bf16r(0.1 * torch.randn(64))
    xp = bf16r(torch.randn(3, 7, 48))
    xq = bf16r(torch.randn(3, 7, 64))
    yl = lin(xp)[-1].detach()
    ym = mlp(xq)[-1].detach()
    assert rel(llm_ref.linear_projector(xp, lin.projector.weight, lin.projector.bias), yl) < 1e-6
    assert rel(llm_ref.mlp_projector(xq, [mlp.projector[0].weight, mlp.projector[2].weight], [None, None]), ym) < 1e-6
    save("projectors.pt", dict(lin_sd={k_: v.detach().clone() for k_, v in lin.state_dict().items()},
                               mlp_sd={k_: v.detach().clone() for k_, v in mlp.state_dict().items()},
                               x_lin=xp, y_lin=yl, x_mlp=xq, y_mlp=ym))

    # ---- 7. greedy decode (BASELINE config 1 plumbing): reference class + KV cache, argmax
    lm.eval()
    prompt = torch.randint(3, 150, (1, 9))
    prompt[0, 0] = 1
    with torch.no_grad():
        o = lm(input_ids=prompt, use_cache=True, return_dict=True)
        past = o.past_key_values
        toks = [o.logits[:, -1].argmax(-1)]
        for _ in range(7):
            o = lm(input_ids=toks[-1][:, None], past_key_values=past, use_cache=True, return_dict=True)
            past = o.past_key_values
            toks.append(o.logits[:, -1].argmax(-1))
    gen = torch.cat([prompt, torch.stack(toks, 1)], 1)
    gen2 = llm_ref.greedy_decode(prompt, sdl, cd, 8)
    assert torch.equal(gen, gen2), (gen, gen2)
    save("greedy_decode.pt", dict(prompt=prompt, tokens=gen))
    print("all restatement checks passed")


if __name__ == "__main__":
    sys.exit(main())
