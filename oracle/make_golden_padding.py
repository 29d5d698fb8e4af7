"""TEST INFRASTRUCTURE -- generates tests/golden/padding.pt by EXECUTING the reference on padded / ragged batches (needs
/root/reference).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_padding

Pins what the reference's own inference callers do with batches of unequal length:
  1. `left_prefill`: `DreamLLMForCausalMLM.forward` (modeling_dreamllm.py:1353) on a LEFT-padded batch with its default
     position ids (arange, :950-955) -- `padding_side="left"` callers: omni/eval/vqa/vqa_inference.py:276,
     omni/eval/text2img/ddp_sample_coco.py:64, projects/dreamllm/cli_stable_diffusion_pipeline.py:19;
  2. `left_generate`: HF-generate semantics on the same batch -- the reference's `prepare_inputs_for_generation` (:1511-1547,
     mask-aware position ids) driven for 6 greedy steps with a growing mask;
  3. `prompt_embeds`: the reference's `get_prompt_embeds` (:1598-1672) EXECUTED with a duck tokenizer that returns the
     left-padded ids/mask: KV-cache prefill, then the dream queries against the cache with mask cat([text_mask, ones]);
  4. `ragged_generate`: the reference's language-eval greedy loop `generate` (omni/eval/language_eval/modeling_dreamllm.py:47-109)
     EXECUTED (with `.cuda()` / `device="cuda"` neutralised) on right-padded ragged prompts, temperature 0: prefill of the
     shortest prompt, teacher forcing inside longer prompts, argmax over `logits[..., :32000]`.  The model has the full
     32008-token vocabulary and lm_head rows of the 8 added tokens boosted x4, so the `:32000` slice decides most tokens.

Weights come from `llm_ref.random_state_dict(cfg, seed, ...)` (seeded, not stored).  Eager attention (flash_attn is absent),
fp32, CPU.
"""
from __future__ import annotations

import importlib
import types

import torch
import torch.nn as nn

from . import llm_ref, ref_loader
from .make_golden import bf16r, rel, save, special_tokens2ids_dict

CFG = dict(vocab_size=160, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
           num_key_value_heads=2, max_position_embeddings=128, rms_norm_eps=1e-6)
CFG_BIG = dict(CFG, vocab_size=32008)
SEED, SEED_BIG = 101, 102
PAD = 150  # [PAD] id of the tiny vocabulary (oracle/make_golden.SPECIAL)


class FakeDream(nn.Module):
    embed_len = 4

    def __init__(self, q):
        super().__init__()
        self.dream_queries = nn.Parameter(q)

    def forward(self, batch_size=1):
        return self.dream_queries.repeat(batch_size, 1, 1)


class NoClip(nn.Module):
    embed_len = 6

    def forward(self, images=None):
        return torch.zeros(())


def build(m, cfgd, seed, **sdkw):
    cfg = ref_loader.make_config(**cfgd)
    cfg.special_tokens2ids_dict = special_tokens2ids_dict()
    lm = m.DreamLLMForCausalMLM(cfg)
    sd = llm_ref.random_state_dict(cfgd, seed, **sdkw)
    res = lm.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all("inv_freq" in k for k in res.missing_keys), res
    g = torch.Generator().manual_seed(seed + 1000)
    lm.model.dream_embedding = FakeDream(bf16r(torch.randn(1, 4, cfgd["hidden_size"], generator=g) * 0.05))
    lm.model.clip_vision_embedding = NoClip()
    return lm.eval()


def main():
    m = ref_loader.load_modeling()
    lm = build(m, CFG, SEED)
    g = torch.Generator().manual_seed(7)
    B, S = 3, 12
    pads = [0, 3, 5]
    ids = torch.randint(3, 150, (B, S), generator=g)
    mask = torch.ones(B, S, dtype=torch.long)
    for b, p in enumerate(pads):
        ids[b, :p] = PAD
        mask[b, :p] = 0
        ids[b, p] = 1  # bos
    out = {"cfg": CFG, "seed": SEED, "dream_seed": SEED + 1000, "input_ids": ids, "attention_mask": mask, "pads": pads}

    with torch.no_grad():
        # ---- 1. left-padded prefill, default positions
        o = lm(input_ids=ids, attention_mask=mask, use_cache=True, return_dict=True)
        out["left_prefill"] = dict(logits=o.logits.clone())
        # sanity: the pad tokens' identity is irrelevant to the valid positions (they are masked as keys)
        ids2 = ids.clone()
        ids2[mask == 0] = 7
        o2 = lm(input_ids=ids2, attention_mask=mask, return_dict=True)
        for b, p in enumerate(pads):
            assert rel(o2.logits[b, p:], o.logits[b, p:]) < 1e-5

        # ---- 2. HF-generate semantics through the reference's prepare_inputs_for_generation
        cur, am, past, toks, margins = ids, mask, None, [], []
        for step in range(7):
            inp = lm.prepare_inputs_for_generation(cur, past_key_values=past, attention_mask=am, use_cache=True)
            o = lm(**inp, return_dict=True)
            past = o.past_key_values
            if step == 0:
                out["left_generate"] = dict(prefill_logits=o.logits.clone(), position_ids=inp["position_ids"].clone())
            nxt = o.logits[:, -1].argmax(-1)
            top2 = o.logits[:, -1].topk(2, dim=-1).values
            margins.append(top2[:, 0] - top2[:, 1])
            toks.append(nxt)
            cur = torch.cat([cur, nxt[:, None]], 1)
            am = torch.cat([am, am.new_ones(B, 1)], 1)
        out["left_generate"]["tokens"] = torch.stack(toks, 1)
        out["left_generate"]["margins"] = torch.stack(margins, 1)  # top-1 minus top-2 logit of every step
        # margin of every argmax (bf16 runs can only flip near-ties): recorded so the GPU test can skip ambiguous steps
        print("left_generate tokens", out["left_generate"]["tokens"].tolist())

        # ---- 3. the reference's get_prompt_embeds, executed
        class Tok:
            def __call__(self, prompt, padding=True, return_tensors="pt"):
                assert isinstance(prompt, list) and len(prompt) == B
                return types.SimpleNamespace(input_ids=ids, attention_mask=mask)

        pe = lm.get_prompt_embeds(Tok(), ["a", "b", "c"], "cpu")
        out["prompt_embeds"] = pe.clone()
        assert pe.shape == (B, 4, CFG["hidden_size"])

    # ---- 4. the reference's language-eval greedy loop, executed
    ref_loader._STUB_ROOTS.add("fairscale")
    le = importlib.import_module("omni.eval.language_eval.modeling_dreamllm")
    big = build(m, CFG_BIG, SEED_BIG, n_added=8, added_boost=4.0)
    g2 = torch.Generator().manual_seed(9)
    plens = [5, 9, 7]
    prompts = [[1] + torch.randint(3, 32000, (n - 1,), generator=g2).tolist() for n in plens]
    PADB = 32000  # [PAD] is the first added token (tokenization_dreamllm.py:78-94)

    class Tok2:
        pad_token_id, eos_token_id = PADB, 2

        def encode(self, x, bos=True, eos=False):
            return list(prompts[int(x)])

        def decode(self, t):
            return list(t)

    rag_margins = []
    big_forward = big.forward

    def logged_forward(*a, **k):
        o_ = big_forward(*a, **k)
        t2 = o_.logits[:, -1, :32000].topk(2, dim=-1).values
        rag_margins.append(t2[:, 0] - t2[:, 1])
        return o_

    big.forward = logged_forward
    orig_cuda, orig_ones = torch.Tensor.cuda, torch.ones
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.ones = lambda *a, **k: orig_ones(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    try:
        with torch.no_grad():
            dec = le.generate(big, ["0", "1", "2"], Tok2(), max_seq_len=64, max_gen_len=6, temperature=0.0)
    finally:
        torch.Tensor.cuda, torch.ones = orig_cuda, orig_ones
        big.forward = big_forward
    with torch.no_grad():  # how often the slice mattered: argmax over the full vocabulary at the prefill position
        full = big(input_ids=torch.tensor([p[:5] for p in prompts]), return_dict=True).logits[:, -1]
    print("ragged decode (cut at eos / max_gen_len):", dec)
    print("full-vocab argmax at the prefill:", full.argmax(-1).tolist(), " sliced:", full[:, :32000].argmax(-1).tolist())
    assert all(t < 32000 for row in dec for t in row)
    assert (full.argmax(-1) >= 32000).any(), "the boosted special tokens should win without the slice"
    maxp = max(plens)
    inp = torch.full((len(prompts), maxp), PADB, dtype=torch.long)
    for b, p in enumerate(prompts):
        inp[b, : len(p)] = torch.tensor(p)
    out["ragged_generate"] = dict(cfg=CFG_BIG, seed=SEED_BIG, n_added=8, added_boost=4.0, input_ids=inp, pad_token_id=PADB,
                                  max_gen_len=6, decoded=dec, prompt_lens=plens,
                                  margins=torch.stack(rag_margins, 1))  # [B, positions min_prompt .. total_len-1]
    save("padding.pt", out)


if __name__ == "__main__":
    main()
