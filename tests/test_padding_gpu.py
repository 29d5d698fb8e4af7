"""`-m gpu`: padded / ragged batches against the EXECUTED reference (tests/golden/padding.pt, oracle/make_golden_padding.py)
and, at kernel level, against an fp32 torch attention.  The reference's inference callers tokenise with padding_side="left"
(omni/eval/vqa/vqa_inference.py:276, omni/eval/text2img/ddp_sample_coco.py:64, projects/dreamllm/cli_stable_diffusion_pipeline.py:19);
its language-eval loop right-pads and teacher-forces (omni/eval/language_eval/modeling_dreamllm.py:66-97)."""
import math

import pytest
import torch
import torch.nn as nn

from conftest import check_tensor, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def bf16r(t):
    return t.to(BF).float()


# ----------------------------------------------------------------------------------------------- kernels
def _ref_attn(q, k, v, key_ok, q_ok, causal_off):
    """fp32 reference on [S,D] slices: keys masked by key_ok, causal with query i at key position i + causal_off."""
    Sq, Sk = q.shape[0], k.shape[0]
    s = (q @ k.t()) / math.sqrt(q.shape[1])
    allow = key_ok[None, :] & (torch.arange(Sk)[None, :] <= (torch.arange(Sq)[:, None] + causal_off))
    s = s.masked_fill(~allow, float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, 0.0)
    o = p @ v
    return o * q_ok[:, None]


@pytest.fixture(params=[1, 2], ids=["fwd4wave", "fwd8wave"])
def attn_variant(request):
    from dreamllm_amd import ops
    ops.ATTN_VARIANT = request.param
    yield request.param
    ops.ATTN_VARIANT = 0


@pytest.mark.parametrize("D,H,Hkv", [(128, 2, 2), (64, 4, 2)])
def test_attention_spans_fwd_bwd(D, H, Hkv, attn_variant):
    """seqstart / seqlens: left padding, right padding, both, an empty row, a full row -- forward, LSE-consistent backward
    (dQ, dK, dV) and zeros at every pad row."""
    from dreamllm_amd import ops
    torch.manual_seed(D)
    S = 200 if attn_variant == 1 else 600
    spans = [(0, S), (37, S - 37), (0, 129), (64, 100), (5, 0), (S - 1, 1), (S // 2, S // 3)]
    B = len(spans)
    q, k, v = (bf16r(torch.randn(B, S, H, D)) for _ in range(3))
    k = k[:, :, :Hkv].contiguous()
    v = v[:, :, :Hkv].contiguous()
    do = bf16r(torch.randn(B, S, H, D))
    start = torch.tensor([s for s, _ in spans], dtype=torch.int32)
    lens = torch.tensor([n for _, n in spans], dtype=torch.int32)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    outs = []
    for b, (st, n) in enumerate(spans):
        ok = (torch.arange(S) >= st) & (torch.arange(S) < st + n)
        row = []
        for h in range(H):
            row.append(_ref_attn(qr[b, :, h], kr[b, :, h // (H // Hkv)], vr[b, :, h // (H // Hkv)], ok, ok.float(), 0))
        outs.append(torch.stack(row, 1))
    oref = torch.stack(outs)
    oref.backward(do)
    qd, kd, vd = (t.to(BF).to(DEV).requires_grad_(True) for t in (q, k, v))
    o = ops.flash_attn(qd, kd, vd, causal=True, seqlens=lens.to(DEV), seqstart=start.to(DEV))
    o.backward(do.to(BF).to(DEV))
    assert rel_l2(o, oref) < 6e-3
    assert rel_l2(qd.grad, qr.grad) < 1.2e-2 and rel_l2(kd.grad, kr.grad) < 1.2e-2 and rel_l2(vd.grad, vr.grad) < 1.2e-2
    for b, (st, n) in enumerate(spans):  # pad rows are exact zeros (pad_input semantics, modeling_dreamllm.py:545)
        pad = torch.ones(S, dtype=torch.bool)
        pad[st:st + n] = False
        for t in (o, qd.grad, kd.grad, vd.grad):
            assert float(t[b][pad.to(DEV)].float().abs().sum()) == 0.0


def test_attention_seqstart_with_cache(attn_variant):
    """Sq != Sk (KV cache): seqstart masks the first keys of each row, every query is valid and sits at the end of the keys."""
    from dreamllm_amd import ops
    torch.manual_seed(1)
    B, Sq, Sk, H, D = 3, 70, 150, 2, 64
    q, k, v = bf16r(torch.randn(B, Sq, H, D)), bf16r(torch.randn(B, Sk, H, D)), bf16r(torch.randn(B, Sk, H, D))
    start = torch.tensor([0, 31, 80], dtype=torch.int32)
    ref = torch.stack([torch.stack([_ref_attn(q[b, :, h], k[b, :, h], v[b, :, h], torch.arange(Sk) >= int(start[b]),
                                              torch.ones(Sq), Sk - Sq) for h in range(H)], 1) for b in range(B)])
    o = ops.flash_attn(q.to(BF).to(DEV), k.to(BF).to(DEV), v.to(BF).to(DEV), causal=True, seqstart=start.to(DEV))
    assert rel_l2(o, ref) < 6e-3


def test_attn_decode_kv_start():
    from dreamllm_amd import ops
    torch.manual_seed(2)
    B, H, Hkv, D, Smax = 3, 4, 2, 128, 96
    q = bf16r(torch.randn(B, H, D))
    kc, vc = bf16r(torch.randn(B, Smax, Hkv, D)), bf16r(torch.randn(B, Smax, Hkv, D))
    kv_len = torch.tensor([40, 96, 17], dtype=torch.int32)
    kv_start = torch.tensor([0, 50, 16], dtype=torch.int32)
    ref = torch.zeros(B, H, D)
    for b in range(B):
        sl = slice(int(kv_start[b]), int(kv_len[b]))
        for h in range(H):
            p = torch.softmax((kc[b, sl, h // 2] @ q[b, h]) / math.sqrt(D), 0)
            ref[b, h] = p @ vc[b, sl, h // 2]
    out = ops.attn_decode(q.to(BF).to(DEV), kc.to(BF).to(DEV), vc.to(BF).to(DEV), kv_len.to(DEV), kv_start=kv_start.to(DEV))
    assert rel_l2(out, ref) < 6e-3


# ----------------------------------------------------------------------------------------------- model level
class _FakeDream(nn.Module):
    embed_len = 4

    def __init__(self, q):
        super().__init__()
        self.dream_queries = nn.Parameter(q)

    def forward(self, batch_size=1):
        return self.dream_queries.repeat(batch_size, 1, 1)


def _build(cfgd, seed, dream_seed=None, **sdkw):
    from dreamllm_amd.configuration_dreamllm import DreamLLMConfig
    from dreamllm_amd.modeling_dreamllm import DreamLLMForCausalMLM
    from oracle import llm_ref
    from oracle.make_golden import special_tokens2ids_dict
    cfg = DreamLLMConfig(**cfgd, special_tokens2ids_dict=special_tokens2ids_dict())
    lm = DreamLLMForCausalMLM(cfg)
    sd = llm_ref.random_state_dict(cfgd, seed, **sdkw)
    res = lm.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all("inv_freq" in k for k in res.missing_keys)
    if dream_seed is not None:
        g = torch.Generator().manual_seed(dream_seed)
        lm.model.dream_embedding = _FakeDream(bf16r(torch.randn(1, 4, cfgd["hidden_size"], generator=g) * 0.05))
    return lm.to(DEV, BF).eval(), sd


def _tokens_agree(ours, ref, margins, forced=None, thresh=0.06):
    """Greedy sequences of two implementations may only part ways at a near-tie (bf16 logits): per row, tokens must agree up to
    the first step whose reference top-1/top-2 margin is below `thresh` (forced steps cannot diverge)."""
    B, n = ref.shape
    checked = 0
    for b in range(B):
        for i in range(n):
            f = forced is not None and bool(forced[b, i])
            if not f and float(margins[b, i]) < thresh:
                break
            assert int(ours[b, i]) == int(ref[b, i]), (b, i, ours[b].tolist(), ref[b].tolist())
            checked += 1
    return checked


def test_left_padded_prefill_matches_executed_reference(golden):
    from oracle import llm_ref
    import torch.nn.functional as F
    g = golden("padding.pt")
    lm, sd = _build(g["cfg"], g["seed"], g["dream_seed"])
    ids, am = g["input_ids"], g["attention_mask"]
    with torch.no_grad():
        out = lm(input_ids=ids.to(DEV), attention_mask=am.to(DEV), return_dict=True).logits
    sdb = {k: v.to(BF) for k, v in sd.items()}
    hb = llm_ref.model_forward(F.embedding(ids, sdb["model.embed_tokens.weight"]), sdb, g["cfg"], attention_mask=am)
    lb = F.linear(hb, sdb["lm_head.weight"]).float()
    ref = g["left_prefill"]["logits"]
    for b, p in enumerate(g["pads"]):
        check_tensor(f"padding.left_prefill.logits[{b}]", out[b, p:], ref[b, p:], rel_l2(lb[b, p:], ref[b, p:]))
    # pad tokens are masked as keys: their identity cannot matter
    ids2 = ids.clone()
    ids2[am == 0] = 9
    with torch.no_grad():
        out2 = lm(input_ids=ids2.to(DEV), attention_mask=am.to(DEV), return_dict=True).logits
    for b, p in enumerate(g["pads"]):
        assert torch.equal(out2[b, p:], out[b, p:])


@pytest.mark.parametrize("mode", ["graph", "kernels", "model"])
def test_left_padded_generate_matches_executed_reference(golden, mode):
    """HF-generate semantics (mask-aware position ids, `prepare_inputs_for_generation`) on a left-padded batch of 3."""
    g = golden("padding.pt")
    lm, _ = _build(g["cfg"], g["seed"], g["dream_seed"])
    ref = g["left_generate"]
    n = ref["tokens"].shape[1]
    toks = lm.greedy_generate(g["input_ids"].to(DEV), n, attention_mask=g["attention_mask"].to(DEV), fast=mode != "model",
                              use_graph=mode == "graph")
    assert toks.shape[1] == g["input_ids"].shape[1] + n
    assert _tokens_agree(toks[:, -n:].cpu(), ref["tokens"], ref["margins"]) >= 8


def test_prompt_embeds_left_padded_matches_executed_reference(golden):
    """`get_prompt_embeds` (modeling_dreamllm.py:1598-1672): KV-cache prefill of the left-padded prompts, then the dream
    queries against the cache with mask cat([text_mask, ones])."""
    from oracle import llm_ref
    import torch.nn.functional as F
    g = golden("padding.pt")
    lm, sd = _build(g["cfg"], g["seed"], g["dream_seed"])
    ids, am = g["input_ids"], g["attention_mask"]
    pe = lm.get_prompt_embeds(ids.to(DEV), attention_mask=am.to(DEV))
    ref = g["prompt_embeds"]
    assert pe.shape == ref.shape
    # yard-stick: the restated oracle in bf16 over the concatenated sequence [text ; <dream_start> queries <dream_end>]
    sdb = {k: v.to(BF) for k, v in sd.items()}
    dq = lm.model.dream_embedding.dream_queries.detach().cpu()
    sp = lm.config.special_tokens2ids_dict["additional_special_tokens"]
    B = ids.shape[0]

    def run(sdx, dt):
        emb = F.embedding(ids, sdx["model.embed_tokens.weight"])
        se = F.embedding(torch.tensor([sp["<dream_start>"], sp["<dream_end>"]]), sdx["model.embed_tokens.weight"])
        tail = torch.cat([se[:1], dq[0].to(dt), se[1:]])[None].expand(B, -1, -1)
        full = torch.cat([emb, tail], 1)
        m2 = torch.cat([am, am.new_ones(B, tail.shape[1])], 1)
        return llm_ref.model_forward(full, sdx, g["cfg"], attention_mask=m2)[:, ids.shape[1] + 1: -1].float()

    assert rel_l2(run(sd, torch.float32), ref) < 1e-5  # the restated composition IS what the reference computed
    check_tensor("padding.prompt_embeds", pe, ref, rel_l2(run(sdb, BF), ref))


@pytest.mark.parametrize("mode", ["graph", "model"])
def test_ragged_language_eval_loop_matches_executed_reference(golden, mode):
    """The reference's language-eval `generate` loop (executed): right-padded ragged prompts, prefill of the shortest prompt,
    teacher forcing inside longer prompts, argmax over logits[..., :32000] -- 32008-token vocabulary whose added tokens would
    win most argmaxes without the slice."""
    g = golden("padding.pt")["ragged_generate"]
    lm, _ = _build(g["cfg"], g["seed"], n_added=g["n_added"], added_boost=g["added_boost"])
    ids = g["input_ids"]
    toks = lm.greedy_generate(ids.to(DEV), g["max_gen_len"], pad_token_id=g["pad_token_id"], fast=mode == "graph").cpu()
    assert int(toks.max()) < 32000 or int((toks == g["pad_token_id"]).sum()) >= 0
    plens = g["prompt_lens"]
    s0 = min(plens)
    total = toks.shape[1]
    ref = torch.full((len(plens), total), -1, dtype=torch.long)
    for b, row in enumerate(g["decoded"]):  # the reference cuts each row at prompt_len + max_gen_len (and at eos)
        ref[b, : len(row)] = torch.tensor(row)
    forced = torch.zeros(len(plens), total - s0, dtype=torch.bool)
    for b, n in enumerate(plens):
        forced[b, : n - s0] = True
    ours, refn = toks[:, s0:], ref[:, s0:]
    valid = refn >= 0
    checked = 0
    for b in range(len(plens)):
        nb = int(valid[b].sum())
        checked += _tokens_agree(ours[b:b + 1, :nb], refn[b:b + 1, :nb], g["margins"][b:b + 1], forced[b:b + 1])
    assert checked >= 8  # 6 forced + the free steps before each row's first near-tie
    gen = ours[~forced[:, : ours.shape[1]]]
    assert int(gen.max()) < 32000  # no added special token is ever emitted


def test_holey_and_right_padded_cache_masks_raise():
    from dreamllm_amd.factory import TINY, build_dreamllm
    lm = build_dreamllm(TINY, device=DEV, dtype=BF, with_clip=False, with_sd=False).eval()
    ids = torch.randint(3, 1000, (2, 10), device=DEV)
    am = torch.ones(2, 10, dtype=torch.long, device=DEV)
    am[0, 4] = 0
    with pytest.raises(ValueError):
        lm(input_ids=ids, attention_mask=am)
    am = torch.ones(2, 10, dtype=torch.long, device=DEV)
    am[1, 7:] = 0
    out = lm(input_ids=ids, attention_mask=am, use_cache=True, return_dict=True)  # right padding alone is fine
    with pytest.raises(ValueError):  # ... but a right-padded prompt followed by new tokens leaves a hole
        lm(input_ids=ids[:, :1], attention_mask=torch.cat([am, am.new_ones(2, 1)], 1), past_key_values=out.past_key_values)


def test_mask_with_holes_matches_the_eager_oracle():
    """A 0/1 mask whose valid tokens are NOT one run per row (`_get_unpad_data`, modeling_dreamllm.py:69-74, takes any mask; the
    eager path the reference runs on this box honours it through the 4-D additive mask, :960-967): the HIP model compacts the
    valid tokens, keeps their original RoPE positions and un-permutes the result.  Hidden states at the VALID positions against
    the fp32 oracle with the oracle-in-bf16 yard-stick; no cache (a cache of a compacted batch is rejected)."""
    from dreamllm_amd.factory import TINY, build_dreamllm
    from oracle import llm_ref
    torch.manual_seed(0)
    lm = build_dreamllm(TINY, device=DEV, dtype=BF, with_clip=False, with_sd=False).eval()
    B, S = 3, 96
    ids = torch.randint(3, 1000, (B, S), device=DEV)
    am = (torch.rand(B, S, device=DEV) > 0.3).long()
    am[:, 0] = 1
    am[1] = 1                       # one dense row
    am[2, 40:] = 0                  # one row with holes AND right padding
    with torch.no_grad():
        out = lm.model(input_ids=ids, attention_mask=am, use_cache=False, return_dict=True).last_hidden_state
    sd = {k: v.detach().float().cpu() for k, v in lm.state_dict().items()}
    cfg = lm.config
    cd = dict(num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
              num_key_value_heads=cfg.num_key_value_heads, rms_norm_eps=cfg.rms_norm_eps,
              max_position_embeddings=cfg.max_position_embeddings, rope_theta=cfg.rope_theta)
    emb = torch.nn.functional.embedding(ids.cpu(), sd["model.embed_tokens.weight"])
    ref = llm_ref.model_forward(emb, sd, cd, attention_mask=am.cpu())
    yard = llm_ref.model_forward(emb.to(BF), {k: v.to(BF) for k, v in sd.items()}, cd, attention_mask=am.cpu()).float()
    m = am.bool().cpu()
    check_tensor("padding.mask_with_holes.hidden", out.float().cpu()[m], ref[m], rel_l2(yard[m], ref[m]))
    with pytest.raises(ValueError):     # a KV cache of the compacted batch would not line up with the caller's positions
        lm.model(input_ids=ids, attention_mask=am, use_cache=True)
