"""`-m gpu`: the parity holes VERDICT r03 listed (weak #1c-#1f, "next" #4), closed at the sizes the bench legs run:

  (a) FULL DEPTH: a 32-layer Vicuna-7B-dims DreamLLM (CLIP-L/14 spliced in) at B = 1, S = 514 with one image = BASELINE config 2's
      prompt: hidden states after 2 / 8 / 32 layers and the logits against the fp32 oracle (error growth with depth next to the
      growth of the oracle run in bf16), then 16 greedy tokens on the KV-cache decode kernels (BASELINE config 1 "full 7 B dims once")
      teacher-forced with the ORACLE's greedy tokens: every step's logits and every argmax whose oracle margin exceeds what a bf16
      program can resolve (modeling_dreamllm.py:846-1043,1353-1509; omni/eval/language_eval/modeling_dreamllm.py:76-97);
  (b) SDXL at the size config 5 times: SDXL UNet context gradient AND `text_embeds` gradient at 2.567 G parameters / 128 x 128
      latents, and one DreamLLM-SDXL stage-I step (2 layers, frozen LLM, dgrad through the layers into the dream queries, both
      projectors) against the oracle composition (omni/models/dreamllm_sdxl/modeling_plugins.py:151-236);
  (c) the SD-2.1 UNet forward at batch 16 -- the kernel selection the B_img = 8 denoise leg and the training step take (256-row
      pipelined tiles, stream-K small grids, three-launch GroupNorm), which the batch-2 test never reaches;
  (d) STRUCTURED inputs at model level: a decoder stack whose attention is sharply peaked (a few keys carry most of the
      probability mass: the online-softmax rescale path inside the model, not only in the kernel tests) and a UNet input with a large
      per-channel offset (GroupNorm statistics with mean >> std), instead of the near-uniform softmax rows / near-Gaussian GroupNorm
      inputs that seeded N(0, sigma) weights produce.

Oracles run in fp32 ON the GPU as the checker, and once more in bf16 as the yard-stick (conftest.check_tensor).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import check_scalar, check_tensor, rel_l2

pytestmark = pytest.mark.gpu
DEV, BF = "cuda", torch.bfloat16


def _randn(*shape, seed=0, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, device=DEV, generator=g) * scale).to(BF).float()


def _to(sd, dtype):
    return {k: v.to(dtype) for k, v in sd.items()}


# ------------------------------------------------------------------------------------------------------ (a) full depth
def _oracle_stack(emb, sd_bf16, cd, dtype, taps, attention_mask=None):
    """llm_ref.model_forward with the layer loop opened: hidden states after the layers in `taps` (pre-norm, as
    `output_hidden_states` returns them) and the final normed state; one layer's weights are converted at a time (the fp32 copy
    of the whole 7 B stack would be 27 GB next to the model under test)."""
    from oracle import llm_ref
    B, S, H = emb.shape
    hd = H // cd["num_attention_heads"]
    cos, sin = llm_ref.rope_tables(hd, max(cd["max_position_embeddings"], S), device=emb.device)
    pos = torch.arange(S, device=emb.device)[None]
    mask4d = llm_ref.causal_mask_4d(attention_mask, B, S, dtype, device=emb.device)
    x = emb.to(dtype)
    out = {}
    for i in range(cd["num_hidden_layers"]):
        pre = f"model.layers.{i}."
        lsd = {k: v.to(dtype) for k, v in sd_bf16.items() if k.startswith(pre)}
        x = llm_ref.decoder_layer(x, lsd, pre, cd, cos, sin, pos, mask4d)   # fp32 tables in both runs, as llm_ref.model_forward
        if i + 1 in taps:
            out[i + 1] = x.float()
    final = llm_ref.rmsnorm(x, sd_bf16["model.norm.weight"].to(dtype), cd["rms_norm_eps"])
    return out, final


def test_vicuna7b_full_depth_hidden_states_logits_and_greedy_decode_vs_oracle():
    from dreamllm_amd.decode import GreedyDecodeSession
    from dreamllm_amd.factory import VICUNA_7B, build_dreamllm
    from dreamllm_amd.synthetic import make_interleaved_batch
    from oracle import llm_ref
    L, S, NEW = 32, 514, 16
    model = build_dreamllm(VICUNA_7B, device=DEV, seed=9, with_sd=False).eval()
    b = make_interleaved_batch(batch_size=1, seq_len=S, images_per_sample=1, seed=5, device=DEV)
    ids, img = b["input_ids"], b["images"]
    sd = {k: v.detach() for k, v in model.state_dict().items()}     # bf16 views: both sides hold identical values
    sp = model.config.special_tokens2ids_dict["additional_special_tokens"]
    cd = dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=L, num_attention_heads=32, num_key_value_heads=32,
              rms_norm_eps=model.config.rms_norm_eps, max_position_embeddings=2048, vocab_size=model.config.vocab_size,
              special_ids=dict(dream_start=sp["<dream_start>"], im_start=sp["<im_start>"]))
    with torch.no_grad():
        out = model(input_ids=ids, images=img, output_hidden_states=True, return_dict=True)
        feats = model.model.clip_vision_embedding(img).float()      # CLIP parity has its own full-size test
        emb_sd = {"model.embed_tokens.weight": sd["model.embed_tokens.weight"].float()}
        # (the dream slot of the prompt keeps its token embeddings: queries are spliced only when `images_dm` is given, :1081-1099)
        emb = llm_ref.splice_inputs(ids, emb_sd, cd, None, feats)
        taps = (2, 8, 32)
        ref, ref_final = _oracle_stack(emb, sd, cd, torch.float32, taps)
        yard, yard_final = _oracle_stack(emb, sd, cd, BF, taps)
        w_head = sd["lm_head.weight"]
        ref_logits = F.linear(ref_final, w_head.float()).float()
        yard_logits = F.linear(yard_final, w_head).float()
    hs = out.hidden_states
    assert len(hs) == L + 1
    growth = {}
    for d in (2, 8):
        growth[d] = (check_tensor(f"depth.hidden_after_{d}_layers", hs[d], ref[d], rel_l2(yard[d], ref[d])), rel_l2(yard[d], ref[d]))
    growth[32] = (check_tensor("depth.hidden_after_32_layers(normed)", hs[L], ref_final, rel_l2(yard_final.float(), ref_final)),
                  rel_l2(yard_final.float(), ref_final))
    check_tensor("depth.logits_32_layers", out.logits, ref_logits, rel_l2(yard_logits, ref_logits))
    # error growth with depth, ours next to the reference-in-bf16 (pairs land in gpurun_out/parity_report.json)
    print("depth -> (err_ours, err_ref):", {d: (f"{a:.3e}", f"{b:.3e}") for d, (a, b) in growth.items()})

    # ---- config 1 at full dims: 16 greedy tokens.  Oracle tokens by sequential fp32 greedy decoding (full recompute per token,
    # reference loop :76-97 with the [:32000] slice); ours on the KV-cache decode kernels, teacher-forced with those tokens.
    VL = 32000
    seq = ids.clone()
    step_ref = []
    with torch.no_grad():
        cur_emb = emb
        for j in range(NEW):
            _, fin = _oracle_stack(cur_emb, sd, cd, torch.float32, ())
            lg = F.linear(fin[:, -1], w_head.float()).float()
            step_ref.append(lg[0])
            nxt = lg[:, :VL].argmax(-1, keepdim=True)
            seq = torch.cat([seq, nxt], dim=1)
            cur_emb = torch.cat([cur_emb, F.embedding(nxt, emb_sd["model.embed_tokens.weight"])], dim=1)
        # yard-stick: ONE bf16 pass over prompt + oracle tokens (causal: position S - 1 + j sees exactly the step-j prefix)
        _, finb = _oracle_stack(cur_emb[:, :-1], sd, cd, BF, ())
        step_yard = [F.linear(finb[:, S - 1 + j], w_head).float()[0] for j in range(NEW)]
    oracle_tokens = seq[:, S:]                                      # [1, NEW]
    sess = GreedyDecodeSession(model, 1, S + NEW + 8)
    forced = oracle_tokens.clone()
    first = sess.prefill(ids, images=img, forced_tokens=forced, forced_mask=torch.ones_like(forced, dtype=torch.bool))
    assert int(first) == int(oracle_tokens[0, 0])                   # (forced)
    agree, decided = 0, 0
    prompt_last = out.logits[0, -1].float()
    ours_steps = [prompt_last]
    for j in range(NEW - 1):
        sess.generate(1)
        ours_steps.append(sess.logits[0].float().clone())
    for j in range(NEW):
        r, y, o = step_ref[j], step_yard[j], ours_steps[j]
        check_tensor(f"depth.greedy_step_{j:02d}.logits", o[:VL], r[:VL], rel_l2(y[:VL], r[:VL]))
        top2 = r[:VL].topk(2).values
        margin = float(top2[0] - top2[1])
        noise = float((y[:VL] - r[:VL]).abs().max())               # what the reference itself loses in bf16 at this position
        if margin > 2.0 * noise:
            decided += 1
            agree += int(int(o[:VL].argmax()) == int(r[:VL].argmax()))
    assert agree == decided, f"greedy argmax differs from the oracle on {decided - agree} of {decided} decidable steps"
    check_scalar("depth.greedy_decidable_steps_agree", agree, decided)
    # VERDICT r04 weak #1d: with N(0, 0.02) weights the logits over 32000 tokens are nearly flat, the top-2 margin is seldom above what
    # a bf16 program resolves and the check above may hold vacuously (`decided` is logged).  The same comparison over the FIRST v
    # tokens of the vocabulary (a greedy step restricted to v candidates: the margin between the best two of v Gaussians grows as v
    # shrinks, the bf16 noise does not) must agree on every decidable step too -- and at v = 16 at least half of the steps ARE decidable.
    counts = {VL: (decided, agree)}
    for v in (4096, 512, 64, 16):
        dec_v = agr_v = 0
        for j in range(NEW):
            r, y, o = step_ref[j][:v], step_yard[j][:v], ours_steps[j][:v]
            top2 = r.topk(2).values
            if float(top2[0] - top2[1]) > 2.0 * float((y - r).abs().max()):
                dec_v += 1
                agr_v += int(int(o.argmax()) == int(r.argmax()))
        counts[v] = (dec_v, agr_v)
        assert agr_v == dec_v, f"argmax over the first {v} tokens differs from the oracle on {dec_v - agr_v} of {dec_v} decidable steps"
        check_scalar(f"depth.greedy_decidable_steps_agree.vocab{v}", agr_v, dec_v)
    print("greedy: vocabulary -> (decidable steps of 16, agreeing):", counts)
    assert counts[16][0] >= NEW // 2, f"only {counts[16][0]} of {NEW} steps decidable even over 16 candidates"
    del sess, model
    torch.cuda.empty_cache()


def test_vicuna7b_dims_8_layer_backward_gradient_error_growth_vs_oracle():
    """VERDICT r04 missing #5 / weak #1e ("loss + selected grad-norm parity on step 0"): the deepest
    oracle-checked BACKWARD was 2 layers.  An 8-layer Vicuna-7B-dims text-only training step (B = 1, S = 1024, fused lm_head + CE):
    loss, and for EVERY layer the gradients of the packed q|k|v, o_proj, gate|up, down_proj and both norm weights against the fp32
    oracle's autograd, beside the oracle run in bf16 -- the gradient error as a function of the distance from the loss
    (modeling_dreamllm.py:599-654,986-1022,1452-1470)."""
    from dreamllm_amd.factory import VICUNA_7B, build_dreamllm
    from oracle import llm_ref
    L, S = 8, 1024
    model = build_dreamllm(dict(VICUNA_7B, num_hidden_layers=L), device=DEV, seed=17, with_sd=False, with_clip=False).train()
    g = torch.Generator(device=DEV).manual_seed(23)
    ids = torch.randint(3, 32000, (1, S), device=DEV, generator=g)
    labels = ids.clone()
    labels[:, :7] = -100
    model.zero_grad(set_to_none=True)
    out = model(input_ids=ids, labels=labels, return_dict=True)
    out.loss.backward()
    full = {k: v.detach().float() for k, v in model.state_dict().items()}
    cd = dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=L, num_attention_heads=32, num_key_value_heads=32,
              rms_norm_eps=model.config.rms_norm_eps, max_position_embeddings=2048, vocab_size=model.config.vocab_size, special_ids={})
    llm_keys = [k for k in full if k.startswith("model.layers.") or k in ("model.embed_tokens.weight", "model.norm.weight", "lm_head.weight")]

    def oracle(dtype):
        sd = {k: full[k].detach().to(dtype).clone().requires_grad_(True) for k in llm_keys}   # fresh leaves per run
        emb = F.embedding(ids, sd["model.embed_tokens.weight"])
        hidden = llm_ref.model_forward(emb, sd, cd, attention_mask=None)
        lm, _ = llm_ref.lm_loss(hidden, sd["lm_head.weight"], labels)
        lm.backward()
        return float(lm), {k: v.grad.float() for k, v in sd.items() if v.grad is not None}

    lr_, gr = oracle(torch.float32)
    lb_, gb = oracle(BF)
    check_scalar("depth8.bwd.loss", out.loss, lr_, abs(lb_ - lr_))
    params = dict(model.named_parameters())
    growth = {}
    for i in range(L):
        pre = f"model.layers.{i}."
        errs = []
        for n in ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.o_proj.weight",
                  "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight", "input_layernorm.weight",
                  "post_attention_layernorm.weight"):
            k = pre + n
            e = check_tensor("depth8.bwd.grad." + k, params[k].grad, gr[k], rel_l2(gb[k], gr[k]))
            errs.append((e, rel_l2(gb[k], gr[k])))
        # grad-norm parity of the whole layer
        ours_n = math.sqrt(sum(float(params[pre + n].grad.float().square().sum()) for n in
                               ("self_attn.q_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight")))
        ref_n = math.sqrt(sum(float(gr[pre + n].square().sum()) for n in
                              ("self_attn.q_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight")))
        yard_n = math.sqrt(sum(float(gb[pre + n].square().sum()) for n in
                               ("self_attn.q_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight")))
        check_scalar(f"depth8.bwd.grad_norm.layer{i}", ours_n, ref_n, abs(yard_n - ref_n), rtol=2e-3)
        growth[i] = (max(a for a, _ in errs), max(b for _, b in errs))
    for k in ("model.embed_tokens.weight", "lm_head.weight", "model.norm.weight"):
        check_tensor("depth8.bwd.grad." + k, params[k].grad, gr[k], rel_l2(gb[k], gr[k]))
    print("layer -> (max grad err ours, max grad err of the oracle in bf16):", {i: (f"{a:.3e}", f"{b:.3e}") for i, (a, b) in growth.items()})
    del model
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------------ (b) SDXL gradients
def test_sdxl_unet_full_size_context_and_text_embeds_gradient():
    from dreamllm_amd.unet import HipUNet2DConditionModel, load_unet_config
    from oracle import unet_ref
    cfg = dict(unet_ref.SDXL_BASE)
    sd = {k: v.to(BF).float() for k, v in unet_ref.random_state_dict(cfg, seed=12, device=DEV).items()}
    with torch.device(DEV):
        m = HipUNet2DConditionModel(load_unet_config(cfg))
    m.load_state_dict(sd, strict=True)
    m = m.to(BF).requires_grad_(False)
    N = 1
    x, ctx, te, dy = _randn(N, 4, 128, 128, seed=4), _randn(N, 196, 2048, seed=5), _randn(N, 1280, seed=6), _randn(N, 4, 128, 128, seed=7)
    tid = torch.tensor([[1024., 1024, 0, 0, 1024, 1024]] * N, device=DEV)
    t = torch.tensor([500], device=DEV)

    def oracle(dtype):
        c = ctx.detach().clone().to(dtype).requires_grad_(True)
        e = te.detach().clone().to(dtype).requires_grad_(True)
        y = unet_ref.unet_forward(x.to(dtype), t, c, _to(sd, dtype), cfg, dict(text_embeds=e, time_ids=tid.to(dtype)))
        y.backward(dy.to(dtype))
        return y.detach().float(), c.grad.float(), e.grad.float()

    yr, gcr, ger = oracle(torch.float32)
    yb, gcb, geb = oracle(BF)
    del sd
    torch.cuda.empty_cache()
    cd, ed = ctx.to(BF).requires_grad_(True), te.to(BF).requires_grad_(True)
    y = m(x.to(BF), t, cd, added_cond_kwargs=dict(text_embeds=ed, time_ids=tid)).sample
    check_tensor("fullsize.sdxl_unet.forward(grad run)", y, yr, rel_l2(yb, yr))
    y.backward(dy.to(BF))
    check_tensor("fullsize.sdxl_unet.grad_ctx", cd.grad, gcr, rel_l2(gcb, gcr))
    check_tensor("fullsize.sdxl_unet.grad_text_embeds", ed.grad, ger, rel_l2(geb, ger))


def test_dreamllm_sdxl_stage1_step_2layer_vs_oracle():
    """BASELINE config 5's step at full width, 2 of the 32 layers, B = 2, S = 256 (caption + 196 dream queries), 1024 px targets:
    frozen LLM forward, SDXL head (VAE encode -> add_noise -> local projector on the dream states, global projector on their mean ->
    `text_embeds`, time ids -> SDXL UNet -> MSE), backward through the UNet (dgrad only) and the frozen layers into the dream
    queries and both projectors (dreamllm_sdxl/modeling_plugins.py:151-236)."""
    from dreamllm_amd.factory import VICUNA_7B, build_dreamllm_sdxl
    from dreamllm_amd.synthetic import make_creation_batch
    from dreamllm_amd.utils import replay_draws
    from oracle import llm_ref, sched_ref, unet_ref, vae_ref
    B, S, NQ = 2, 256, 196
    model = build_dreamllm_sdxl(dict(VICUNA_7B, num_hidden_layers=2), device=DEV, seed=4, with_clip=False).train()
    batch = make_creation_batch(batch_size=B, seq_len=S, n_dream=NQ, seed=31, device=DEV, dm_size=1024)
    head = model.stable_diffusion_head
    vae_noise, noise = _randn(B, 4, 128, 128, seed=51), _randn(B, 4, 128, 128, seed=52)
    ts = torch.tensor([777, 21], device=DEV)
    model.zero_grad(set_to_none=True)
    with replay_draws([("randn", vae_noise), ("randn_like", noise), ("randint", ts)]):
        out = model(**batch, return_dict=True)
    out.loss.backward()

    full = {k: v.detach() for k, v in model.state_dict().items()}
    sp = model.config.special_tokens2ids_dict["additional_special_tokens"]
    cd = dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=2, num_attention_heads=32, num_key_value_heads=32,
              rms_norm_eps=model.config.rms_norm_eps, max_position_embeddings=2048, vocab_size=model.config.vocab_size,
              special_ids=dict(dream_start=sp["<dream_start>"], im_start=sp.get("<im_start>", -1)))
    usd = {k[len("stable_diffusion_head.unet."):]: v for k, v in full.items() if k.startswith("stable_diffusion_head.unet.")}
    vsd = {k[len("stable_diffusion_head.vae."):]: v for k, v in full.items() if k.startswith("stable_diffusion_head.vae.")}
    ucfg, vcfg = dict(unet_ref.SDXL_BASE), dict(head.vae.config.to_dict())
    ac = torch.tensor(sched_ref.alphas_cumprod(), device=DEV)
    names = dict(model.named_parameters())
    leaf_names = [n for n, p in names.items() if p.requires_grad]
    assert any("dream_queries" in n for n in leaf_names) and any("global_projector" in n for n in leaf_names), leaf_names
    llm_keys = [k for k in full if k.startswith("model.layers.") or k in ("model.embed_tokens.weight", "model.norm.weight")]
    ids = batch["input_ids"]

    def oracle(dtype):
        sd = {k: full[k].to(dtype) for k in llm_keys}
        leaves = {n: full[n].detach().clone().to(dtype).requires_grad_(True) for n in leaf_names}
        dq = leaves[next(n for n in leaf_names if "dream_queries" in n)]
        emb = llm_ref.splice_inputs(ids, sd, cd, dq[0], None)
        hidden = llm_ref.model_forward(emb, sd, cd, attention_mask=batch["attention_mask"])
        ds = llm_ref.gather_dream_states(hidden, ids, cd, NQ, B)
        lw = leaves[next(n for n in leaf_names if n.endswith("stable_diffusion_head.projector.projector.weight"))]
        gw = leaves[next(n for n in leaf_names if "global_projector" in n and n.endswith("weight"))]
        ctx = F.linear(ds, lw)
        te = F.linear(ds.mean(dim=1), gw)
        with torch.no_grad():
            mom = vae_ref.encode_moments(batch["images_dm"].to(dtype), _to(vsd, dtype), vcfg)
            lat = vae_ref.sample_latents(mom, vae_noise, vcfg["scaling_factor"]).to(dtype)
            a = ac[ts].to(dtype)[:, None, None, None]
            noisy = a.sqrt() * lat + (1 - a).sqrt() * noise.to(dtype)
        pred = unet_ref.unet_forward(noisy, ts, ctx, _to(usd, dtype), ucfg,
                                     dict(text_embeds=te, time_ids=batch["add_time_ids"].to(dtype)))
        vm = F.mse_loss(pred.float(), noise.float())
        vm.backward()
        return float(vm), {n: t.grad.float() for n, t in leaves.items() if t.grad is not None}

    lr, gr = oracle(torch.float32)
    lb, gb = oracle(BF)
    check_scalar("fullsize.sdxl_step.vm_loss", out.additional_log_info["vm_loss"], lr, abs(lb - lr))
    checked = 0
    for n in leaf_names:
        if n in gr and names[n].grad is not None:
            check_tensor("fullsize.sdxl_step.grad." + n, names[n].grad, gr[n], rel_l2(gb[n], gr[n]))
            checked += 1
    assert checked >= 3, (checked, leaf_names)


# ------------------------------------------------------------------------------------------------------ (c) UNet at batch 16
def test_sd21_unet_full_size_forward_batch16():
    from dreamllm_amd.unet import HipUNet2DConditionModel, load_unet_config
    from oracle import unet_ref
    cfg = dict(unet_ref.SD21_BASE)
    sd = {k: v.to(BF).float() for k, v in unet_ref.random_state_dict(cfg, seed=11, device=DEV).items()}
    with torch.device(DEV):
        m = HipUNet2DConditionModel(load_unet_config(cfg))
    m.load_state_dict(sd, strict=True)
    m = m.to(BF).requires_grad_(False)
    N = 16
    x, ctx = _randn(N, 4, 64, 64, seed=1), _randn(N, 64, 1024, seed=2)
    t = torch.tensor([981, 41, 500, 7] * 4, device=DEV)
    with torch.no_grad():
        y = m(x.to(BF), t, ctx.to(BF)).sample
        yg = m(x.to(BF), t, ctx.to(BF), context_cache=m.prepare_context(ctx.to(BF))).sample
        sdb = _to(sd, BF)
        # the oracle in chunks of 4 (its fp32 attention at 64 x 64 holds [N, heads, 4096, 4096] scores)
        yr = torch.cat([unet_ref.unet_forward(x[i:i + 4], t[i:i + 4], ctx[i:i + 4], sd, cfg).float() for i in range(0, N, 4)])
        yb = torch.cat([unet_ref.unet_forward(x[i:i + 4].to(BF), t[i:i + 4], ctx[i:i + 4].to(BF), sdb, cfg).float() for i in range(0, N, 4)])
    check_tensor("fullsize.sd21_unet.forward_batch16", y, yr, rel_l2(yb, yr))
    assert torch.equal(y, yg)


# ------------------------------------------------------------------------------------------------------ (d) structured inputs
def test_decoder_stack_with_peaked_attention_vs_oracle():
    """4 Vicuna-width layers whose q/k projections are scaled 1.8x so that the pre-softmax scores have a standard deviation of ~5
    (seeded N(0, 0.02) weights give ~1.6: near-uniform rows): each query puts almost all of its mass on a handful of keys, the
    running maximum jumps by far more than the kernel's deferred-rescale threshold from tile to tile, and the backward's
    P = exp2(S - lse) spans 30+ binades.  Forward hidden state and the gradient w.r.t. the input embeddings."""
    from dreamllm_amd.factory import VICUNA_7B, build_dreamllm
    from oracle import llm_ref
    L, B, S = 4, 2, 1024
    model = build_dreamllm(dict(VICUNA_7B, num_hidden_layers=L), device=DEV, seed=21, with_clip=False, with_sd=False).train()
    with torch.no_grad():
        for layer in model.model.layers:
            layer.self_attn.q_proj.weight.mul_(1.8)
            layer.self_attn.k_proj.weight.mul_(1.8)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    cd = dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=L, num_attention_heads=32, num_key_value_heads=32,
              rms_norm_eps=model.config.rms_norm_eps, max_position_embeddings=2048, vocab_size=model.config.vocab_size)
    emb0 = _randn(B, S, 4096, seed=61, scale=1.0)
    dy = _randn(B, S, 4096, seed=62)

    def oracle(dtype):
        e = emb0.detach().clone().to(dtype).requires_grad_(True)
        lsd = {k: v.to(dtype) for k, v in sd.items() if k.startswith("model.layers.") or k == "model.norm.weight"}
        h = llm_ref.model_forward(e, lsd, cd)
        h.backward(dy.to(dtype))
        return h.detach().float(), e.grad.float()

    hr, gr = oracle(torch.float32)
    hb, gb = oracle(BF)
    # the premise of the test: the oracle's first-layer attention rows are sharply peaked
    with torch.no_grad():
        x0 = llm_ref.rmsnorm(emb0[:1], sd["model.layers.0.input_layernorm.weight"].float(), cd["rms_norm_eps"])
        q = F.linear(x0, sd["model.layers.0.self_attn.q_proj.weight"].float()).view(1, S, 32, 128)[:, :, 0]
        k = F.linear(x0, sd["model.layers.0.self_attn.k_proj.weight"].float()).view(1, S, 32, 128)[:, :, 0]
        sc = (q @ k.transpose(1, 2)) / math.sqrt(128)
        sc = sc.masked_fill(torch.triu(torch.ones(S, S, device=DEV, dtype=torch.bool), 1), float("-inf"))
        pmax = sc.softmax(-1).amax(-1)[0, S // 2:].mean()
    assert float(pmax) > 0.2, f"attention not peaked: mean max-probability {float(pmax):.3f}"   # uniform rows would give ~1 / 768
    e = emb0.to(BF).requires_grad_(True)
    out = model.model(inputs_embeds=e, return_dict=True)
    h = out.last_hidden_state
    check_tensor("structured.peaked_attention.hidden", h, hr, rel_l2(hb, hr))
    h.backward(dy.to(BF))
    check_tensor("structured.peaked_attention.grad_emb", e.grad, gr, rel_l2(gb, gr))


def test_sd21_unet_with_channel_offset_input_vs_oracle():
    """GroupNorm with mean >> std: the UNet's first ResBlocks see conv_in(x) where x carries a constant per-channel offset of 6
    standard deviations (a saturated latent), and conv_in's bias is large; variance by E[x^2] - E[x]^2 in fp32 loses ~2 digits there.
    Forward at the real SD-2.1 dimensions, batch 2."""
    from dreamllm_amd.unet import HipUNet2DConditionModel, load_unet_config
    from oracle import unet_ref
    cfg = dict(unet_ref.SD21_BASE)
    sd = {k: v.to(BF).float() for k, v in unet_ref.random_state_dict(cfg, seed=17, device=DEV).items()}
    sd["conv_in.bias"] = (sd["conv_in.bias"] + 4.0 * torch.sign(_randn(320, seed=71))).to(BF).float()
    with torch.device(DEV):
        m = HipUNet2DConditionModel(load_unet_config(cfg))
    m.load_state_dict(sd, strict=True)
    m = m.to(BF).requires_grad_(False)
    N = 2
    off = torch.tensor([6.0, -6.0, 3.0, -3.0], device=DEV)[None, :, None, None]
    x = (_randn(N, 4, 64, 64, seed=72) + off).to(BF).float()
    ctx = _randn(N, 64, 1024, seed=73)
    t = torch.tensor([650, 3], device=DEV)
    with torch.no_grad():
        yr = unet_ref.unet_forward(x, t, ctx, sd, cfg).float()
        yb = unet_ref.unet_forward(x.to(BF), t, ctx.to(BF), _to(sd, BF), cfg).float()
        y = m(x.to(BF), t, ctx.to(BF)).sample
    check_tensor("structured.gn_offset.sd21_unet.forward", y, yr, rel_l2(yb, yr))
