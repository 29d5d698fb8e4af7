"""`-m gpu` module-level parity: the HIP DreamLLM decoder against the golden vectors produced by EXECUTING the reference
(tests/golden/*.pt, oracle/make_golden.py) and against the fp32 oracle restatement (oracle/llm_ref.py).

Tolerance contract (SURVEY.md §7 "Numerical contract"): storage is bf16, so an element-wise 1e-3 is unattainable by ANY
bf16 implementation -- including the reference's own.  Every check therefore measures
    err_ours = ||ours - fp32_reference|| / ||fp32_reference||
next to the yard-stick
    err_ref  = ||reference_run_in_bf16 - fp32_reference|| / ||fp32_reference||   (oracle evaluated in bf16 on CPU)
and requires  err_ours <= 1.15 * err_ref + 5e-4 (conftest.check_tensor; round 2: 1.5 / 2e-3).  fp32 scalars (loss values) are held to
|ours - fp32| <= 1.15 * |reference_in_bf16 - fp32| + 1e-3 * |fp32| (conftest.check_scalar): 1e-3 relative, the bound
north_star states, plus the same yard-stick allowance.  err_ref comes from EXECUTING the reference in bf16
(tests/golden/err_ref.pt, oracle/make_golden_errref.py).  Every measured (err, err_ref) pair of a run is written to
gpurun_out/parity_report.json; the round's copy is profiles/r03_parity_report.json.
"""
import pytest
import torch
import torch.nn as nn

from conftest import bound as _bound, check_scalar, check_tensor, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _cfg(cd, **kw):
    from dreamllm_amd.configuration_dreamllm import DreamLLMConfig
    return DreamLLMConfig(vocab_size=cd["vocab_size"], hidden_size=cd["hidden_size"], intermediate_size=cd["intermediate_size"],
                          num_hidden_layers=cd["num_hidden_layers"], num_attention_heads=cd["num_attention_heads"],
                          num_key_value_heads=cd["num_key_value_heads"], rms_norm_eps=cd["rms_norm_eps"],
                          max_position_embeddings=cd["max_position_embeddings"], **kw)


def _bf16_sd(sd):
    return {k: (v.to(BF) if v.is_floating_point() else v) for k, v in sd.items()}


def test_decoder_layer_golden(golden):
    from dreamllm_amd.modeling_dreamllm import DreamLLMDecoderLayer
    from oracle import llm_ref
    g = golden("decoder_layer.pt")
    cd = g["cfg"]
    layer = DreamLLMDecoderLayer(_cfg(cd))
    sd = {k: v for k, v in g["sd"].items()}
    missing = layer.load_state_dict(sd, strict=True)
    layer = layer.to(DEV, BF)
    x = g["x"].to(BF).to(DEV).requires_grad_(True)
    y = layer(x)[0]
    y.backward(g["dy"].to(BF).to(DEV))
    # yard-stick: the oracle itself in bf16
    B, S, _ = g["x"].shape
    sdb = _bf16_sd(g["sd"])
    cos, sin = llm_ref.rope_tables(64, 128)
    yb = llm_ref.decoder_layer(g["x"].to(BF), sdb, "", cd, cos, sin, torch.arange(S)[None],
                               llm_ref.causal_mask_4d(None, B, S, BF))
    e_ref = rel_l2(yb, g["y"])
    e = rel_l2(y, g["y"])
    assert e <= _bound(e_ref), (e, e_ref)
    er = golden("err_ref.pt")["decoder_layer"]  # the reference layer EXECUTED in bf16 (oracle/make_golden_errref.py)
    check_tensor("decoder_layer.y", y, g["y"], er["y"])
    check_tensor("decoder_layer.dx", x.grad, g["dx"], er["dx"])
    for name, p in layer.named_parameters():
        check_tensor("decoder_layer.grad." + name, p.grad, g["grads"][name].float(), er["grads"][name])


def test_model_forward_padding_golden(golden):
    from dreamllm_amd.modeling_dreamllm import DreamLLMModel
    from oracle import llm_ref
    g = golden("model_forward.pt")
    cd = g["cfg"]
    model = DreamLLMModel(_cfg(cd))
    sd = {k[len("model."):]: v for k, v in g["sd"].items()}
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV, BF).eval()
    am = g["attention_mask"]
    with torch.no_grad():
        out = model._forward(inputs_embeds=g["emb"].to(BF).to(DEV), attention_mask=am.to(DEV), use_cache=False).last_hidden_state
    ob = llm_ref.model_forward(g["emb"].to(BF), _bf16_sd(g["sd"]), cd, attention_mask=am)
    L = int(am[1].sum())
    for b, n in ((0, am.shape[1]), (1, L)):
        e_ref = rel_l2(ob[b, :n], g["out"][b, :n])
        e = rel_l2(out[b, :n], g["out"][b, :n])
        assert e <= _bound(e_ref), (b, e, e_ref)
    # the reference's eager-path mask format (4-D additive, modeling_dreamllm.py:965-967) reduces to the same spans: same launches,
    # bit-identical output -- through the model and through a direct DreamLLMDecoderLayer.forward call
    B, S = am.shape
    m4 = llm_ref.causal_mask_4d(am, B, S, torch.float32).to(DEV)
    with torch.no_grad():
        out4 = model._forward(inputs_embeds=g["emb"].to(BF).to(DEV), attention_mask=m4, use_cache=False).last_hidden_state
        assert torch.equal(out4, out)
        x = g["emb"].to(BF).to(DEV)
        y2 = model.layers[0](x, attention_mask=am.to(DEV))[0]
        y4 = model.layers[0](x, attention_mask=m4.to(BF))[0]
        assert torch.equal(y4, y2)


class _FakeDream(nn.Module):
    embed_len = 4

    def __init__(self, hid):
        super().__init__()
        self.dream_queries = nn.Parameter(torch.zeros(1, 4, hid))

    def forward(self, batch_size=1):
        return self.dream_queries.repeat(batch_size, 1, 1)


class _FakeClip(nn.Module):
    embed_len = 6

    def __init__(self, hid):
        super().__init__()
        from dreamllm_amd.projector import HipLinear
        self.proj = HipLinear(8, hid)

    def forward(self, images=None):
        if images is None:
            return (0.0 * self.proj(torch.zeros(1, 6, 8, device=self.proj.weight.device, dtype=self.proj.weight.dtype))).sum()
        return self.proj(images)


class _FakeHead(nn.Module):
    drop_prob = None

    def forward(self, images, encoder_hidden_states, u=None, dream_embeddings=None):
        if images is None:
            return (0.0 * dream_embeddings).sum()
        return (encoder_hidden_states.float() * images.float()).pow(2).mean()


def _build_lm(g):
    from dreamllm_amd.modeling_dreamllm import DreamLLMForCausalMLM
    from oracle.make_golden import special_tokens2ids_dict
    cd = g["cfg"]
    cfg = _cfg(cd, special_tokens2ids_dict=special_tokens2ids_dict())
    lm = DreamLLMForCausalMLM(cfg)
    lm.model.dream_embedding = _FakeDream(cd["hidden_size"])
    lm.model.clip_vision_embedding = _FakeClip(cd["hidden_size"])
    lm.stable_diffusion_head = _FakeHead()
    res = lm.load_state_dict(g["sd"], strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all("inv_freq" in k for k in res.missing_keys), res.missing_keys
    return lm.to(DEV, BF)


def test_causal_mlm_golden(golden):
    """Interleaved forward/backward: splice of dream queries + image features, dream-state gather, SD-head loss hook,
    lm_head + shifted masked CE, loss mix 10*vm + 1*lm (modeling_dreamllm.py:1353-1509)."""
    g = golden("causal_mlm.pt")
    lm = _build_lm(g).train()
    out = lm(input_ids=g["input_ids"].to(DEV), images=g["images"].to(BF).to(DEV), images_dm=g["images_dm"].to(BF).to(DEV),
             attention_mask=g["attention_mask"].to(DEV), labels=g["labels"].to(DEV), return_dict=True)
    out.loss.backward()
    am = g["attention_mask"]
    assert out.logits.dtype == torch.float32 and out.logits.shape == g["logits"].shape
    er = golden("err_ref.pt")["causal_mlm"]  # the reference model EXECUTED in bf16 (oracle/make_golden_errref.py)
    for b in range(am.shape[0]):
        n = int(am[b].sum())
        check_tensor(f"causal_mlm.logits[{b}]", out.logits[b, :n], g["logits"][b, :n], er["logits"][b])
    check_scalar("causal_mlm.lm_loss", out.additional_log_info["lm_loss"], g["lm_loss"], er["lm_loss"]["abs_err"])
    check_scalar("causal_mlm.vm_loss", out.additional_log_info["vm_loss"], g["vm_loss"], er["vm_loss"]["abs_err"])
    check_scalar("causal_mlm.loss", out.loss, g["loss"], er["loss"]["abs_err"])
    check_tensor("causal_mlm.grad_dream", lm.model.dream_embedding.dream_queries.grad, g["grad_dream"], er["grad_dream"])
    check_tensor("causal_mlm.grad_lm_head", lm.lm_head.weight.grad, g["grad_lm_head"].float(), er["grad_lm_head"])
    check_tensor("causal_mlm.grad_q0", lm.model.layers[0].self_attn.q_proj.weight.grad, g["grad_q0"].float(), er["grad_q0"])
    check_tensor("causal_mlm.grad_embed", lm.model.embed_tokens.weight.grad, g["grad_embed"].float(), er["grad_embed"])
    check_tensor("causal_mlm.grad_clip_proj", lm.model.clip_vision_embedding.proj.weight.grad, g["grad_clip_proj"],
                 er["grad_clip_proj"])


def test_causal_mlm_fast_index_path_matches(golden):
    """Precomputed slot indices (sync-free path) give bit-identical results to locating the slots from input_ids."""
    from dreamllm_amd.modeling_dreamllm import _slot_indices
    g = golden("causal_mlm.pt")
    lm = _build_lm(g).train()
    ids = g["input_ids"].to(DEV)
    kw = dict(input_ids=ids, images=g["images"].to(BF).to(DEV), images_dm=g["images_dm"].to(BF).to(DEV),
              attention_mask=g["attention_mask"].to(DEV), labels=g["labels"].to(DEV), return_dict=True)
    a = lm(**kw)
    sp = lm.config.special_tokens2ids_dict["additional_special_tokens"]
    di, _ = _slot_indices(ids, sp["<dream_start>"], 4)
    ii, _ = _slot_indices(ids, sp["<im_start>"], 6)
    b = lm(**kw, dream_index=di, image_index=ii)
    assert torch.equal(a.logits, b.logits) and torch.equal(a.loss, b.loss)


@pytest.mark.parametrize("mode", ["graph", "kernels", "unfused", "model"])
def test_greedy_decode_golden(golden, mode):
    """BASELINE config 1 plumbing: KV-cache prefill + 8 greedy steps reproduce the reference's token ids -- through the
    decode kernels replayed as a hipGraph ("graph"), the same kernels launched eagerly ("kernels"), and the per-token model
    forward ("model")."""
    g = golden("causal_mlm.pt")
    gd = golden("greedy_decode.pt")
    lm = _build_lm(g).eval()
    if mode == "unfused":  # the 17-launch-per-layer token step (every operator on its own)
        from dreamllm_amd.decode import GreedyDecodeSession
        sess = GreedyDecodeSession(lm, 1, gd["prompt"].shape[1] + 9, use_graph=False, fused=False)
        first = sess.prefill(gd["prompt"].to(DEV))
        toks = torch.cat([gd["prompt"].to(DEV), first[:, None], sess.generate(7)], 1)
    else:
        toks = lm.greedy_generate(gd["prompt"].to(DEV), 8, fast=mode != "model", use_graph=mode == "graph")
    # bf16 logits can flip an argmax only on near-ties; require the reference sequence (seeded, no ties in the fixture)
    assert torch.equal(toks.cpu(), gd["tokens"]), (toks.cpu(), gd["tokens"])


def test_decode_session_matches_model_forward_logits():
    """Token-step logits of GreedyDecodeSession (GEMV + cache attention, hipGraph replay) against a full model forward over
    the same prefix, batch 3, for 6 consecutive steps (teacher-forced on the session's own tokens, so near-ties of the
    random-weight logits cannot make the two paths diverge); a second `prefill` on the same session reproduces the run."""
    from dreamllm_amd.decode import GreedyDecodeSession
    from dreamllm_amd.factory import TINY, build_dreamllm
    lm = build_dreamllm(TINY, device=DEV, dtype=BF, with_clip=False, with_sd=False).eval()
    torch.manual_seed(3)
    ids = torch.randint(3, 30000, (3, 21), device=DEV)
    sess = GreedyDecodeSession(lm, 3, 64, use_graph=True)
    first = sess.prefill(ids)
    seq = torch.cat([ids, first[:, None]], 1)
    toks = []
    for _ in range(6):
        nxt = sess.generate(1)
        with torch.no_grad():
            full = lm(input_ids=seq, return_dict=True).logits[:, -1]
        assert rel_l2(sess.logits, full) < 2e-2
        tok = sess.logits.argmax(-1)
        assert torch.equal(nxt[:, 0], tok)  # repeated generate() calls continue where the previous one stopped
        top2 = full.float().topk(2, dim=-1).values
        clear = (top2[:, 0] - top2[:, 1]) > 0.05
        assert torch.equal(tok[clear], full.argmax(-1)[clear])
        toks.append(tok)
        seq = torch.cat([seq, tok[:, None]], 1)
    assert torch.equal(sess.out_tokens[:6].t(), torch.stack(toks, 1))
    first2 = sess.prefill(ids)
    rest2 = sess.generate(6)
    assert torch.equal(first2, first) and torch.equal(rest2, torch.stack(toks, 1))


def test_projectors_golden(golden):
    from dreamllm_amd.projector import build_projector
    g = golden("projectors.pt")
    lin = build_projector(dict(projector="linear", freeze_projector=False, depth=1, save_model_name="clip",
                               model_name_or_path=None), 48, 64, bias=True)
    lin.load_state_dict(g["lin_sd"])
    y = lin.to(DEV, BF)(g["x_lin"].to(BF).to(DEV))
    assert isinstance(y, list) and rel_l2(y[-1], g["y_lin"]) <= 4e-3
    mlp = build_projector(dict(projector="mlp", freeze_projector=False, depth=2, save_model_name="sd",
                               model_name_or_path=None), 64, 32, bias=False)
    mlp.load_state_dict(g["mlp_sd"])
    y = mlp.to(DEV, BF)(g["x_mlp"].to(BF).to(DEV))
    assert rel_l2(y[-1], g["y_mlp"]) <= 8e-3
    # freeze_projector=True: no graph is built (mlp_projector.py:26,49)
    frz = build_projector(dict(projector="linear", freeze_projector=True, depth=1, save_model_name="clip",
                               model_name_or_path=None), 48, 64, bias=True).to(DEV, BF)
    assert not frz(g["x_lin"].to(BF).to(DEV).requires_grad_(True))[-1].requires_grad


def test_sdxl_model_stage1_step_matches_manual_composition():
    """DreamLLMSDXLForCausalMLM (omni/models/dreamllm_sdxl/modeling_dreamllm_sdxl.py:1353-1509), stage-I freeze policy:
    the model's loss equals [decoder -> gather dream-query outputs -> StableDiffusionXLHead(add_time_ids)] composed by
    hand with the same RNG stream; only the dream queries and the head's two projectors receive gradients."""
    from dreamllm_amd.factory import TINY, build_dreamllm_sdxl
    from dreamllm_amd.synthetic import make_creation_batch
    from oracle import unet_ref
    sd = dict(unet=unet_ref.tiny_config(64, sdxl=True), vae=dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1))
    m = build_dreamllm_sdxl(TINY, device=DEV, dtype=BF, with_clip=False, diffusion=sd, num_dream_queries=8,
                            global_condition_hidden_size=40)
    m.train()
    assert m.config.vocab_size == 32009 and not any("inv_freq" in k for k in m.state_dict())
    b = make_creation_batch(batch_size=3, seq_len=32, n_dream=8, device=DEV, dtype=BF, dm_size=128)
    b["add_time_ids"][1] = torch.tensor([200., 160, 8, 16, 128, 128], device=DEV)
    torch.manual_seed(11)
    out = m(**b)
    out.loss.backward()
    trainable = {n for n, p in m.named_parameters() if p.requires_grad}
    assert trainable == {"model.dream_embedding.dream_queries", "stable_diffusion_head.projector.projector.weight",
                         "stable_diffusion_head.global_projector.projector.weight"}
    got = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    assert set(got) == trainable and all(float(g.float().abs().sum()) > 0 for g in got.values())
    assert float(out.additional_log_info["lm_loss"]) > 0  # computed and logged, weight 0 in stage I
    m.zero_grad(set_to_none=True)
    hs = m.model(input_ids=b["input_ids"], images_dm=b["images_dm"], attention_mask=b["attention_mask"],
                 dream_index=b["dream_index"], return_dict=True).last_hidden_state
    enc = hs.reshape(-1, hs.shape[-1])[b["dream_index"]].view(3, 8, -1)
    torch.manual_seed(11)
    ref = m.stable_diffusion_head(b["images_dm"], enc, None, b["add_time_ids"])
    assert abs(out.loss.item() - ref.item()) <= 1e-3 * abs(ref.item()) + 1e-6, (out.loss.item(), ref.item())
    ref.backward()
    for n, p in m.named_parameters():
        if p.grad is not None:
            assert rel_l2(p.grad, got[n]) <= 2e-2, n


def test_torch_compile_coexistence():
    """The reference's inference scripts wrap the model in torch.compile (projects/dreamllm/inference.py:70).  After
    `ops.make_dynamo_opaque()` Dynamo must run the ctypes-launched kernels eagerly around its graph breaks and reproduce the
    eager logits exactly (backend "eager": no code generation needed on the test box).  Runs in a subprocess because the
    opaque wrappers are process-global."""
    import subprocess
    import sys
    code = """
import torch
from dreamllm_amd import ops
from dreamllm_amd.factory import TINY, build_dreamllm
lm = build_dreamllm(TINY, device="cuda", dtype=torch.bfloat16, with_clip=False, with_sd=False).eval()
ids = torch.randint(3, 30000, (2, 40), device="cuda")
with torch.no_grad():
    ref = lm(input_ids=ids, return_dict=True).logits
ops.make_dynamo_opaque()
clm = torch.compile(lm, backend="eager")
with torch.no_grad():
    out = clm(input_ids=ids, return_dict=True).logits
assert torch.equal(out, ref), (out - ref).abs().max()
print("COMPILE_OK")
"""
    code_registered = """
import torch
from dreamllm_amd.factory import TINY, build_dreamllm
lm = build_dreamllm(TINY, device="cuda", dtype=torch.bfloat16, with_clip=False, with_sd=False).eval()
ids = torch.randint(3, 30000, (2, 40), device="cuda")
with torch.no_grad():
    ref = lm(input_ids=ids, return_dict=True)
    ex = torch._dynamo.explain(lm)(input_ids=ids, return_dict=True)
assert ex.graph_count == 1 and ex.graph_break_count == 0, ex.break_reasons
names = [str(n.target) for g in ex.graphs for n in g.graph.nodes if n.op == "call_function"]
assert names.count("dreamllm.decoder_layer_kv") == len(lm.model.layers) and "dreamllm.linear" in names, names
torch._dynamo.reset()
clm = torch.compile(lm, backend="eager")      # NO make_dynamo_opaque(): the registered torch.ops.dreamllm.* carry the forward
with torch.no_grad():
    out = clm(input_ids=ids, return_dict=True)
assert torch.equal(out.logits, ref.logits), (out.logits - ref.logits).abs().max()
for (k0, v0), (k1, v1) in zip(ref.past_key_values, out.past_key_values):
    assert torch.equal(k0, k1) and torch.equal(v0, v1)
print("REGISTERED_OK")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       cwd=__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    assert "COMPILE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    # torch.library registration (dreamllm_amd/torch_ops.py): the text forward is ONE Dynamo graph of torch.ops.dreamllm.* nodes and
    # gives bit-identical logits and KV cache to eager execution on the HIP kernels
    r = subprocess.run([sys.executable, "-c", code_registered], capture_output=True, text=True, timeout=600,
                       cwd=__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    assert "REGISTERED_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


class _FakeXLHead(nn.Module):
    """the fake head of oracle/make_golden_sdxl.py (drop_prob set => the model must run the unconditional pass)"""
    drop_prob = 0.1

    def forward(self, images, encoder_hidden_states, u=None, add_time_ids=None, dream_embeddings=None):
        if images is None:
            assert add_time_ids is None
            return (0.0 * dream_embeddings).sum()
        t = (add_time_ids.float() / 100.0).sum(-1)[:, None, None]
        return ((encoder_hidden_states.float() * images.float()).pow(2) * t).mean() + 0.5 * (u.float() * images.float()).pow(2).mean()


class _NoClip(nn.Module):
    embed_len = 6

    def forward(self, images=None):
        return torch.zeros((), device=DEV)


def test_causal_mlm_sdxl_golden(golden):
    """DreamLLMSDXLForCausalMLM against the EXECUTED reference (omni/models/dreamllm_sdxl/modeling_dreamllm_sdxl.py, fixture
    from oracle/make_golden_sdxl.py): add_time_ids handed to the head, unconditional pass over <dream_patch> tokens
    (head.drop_prob set), loss divided by loss_scale twice (l1_norm schedule => /16), dummy branch, no inv_freq keys."""
    from dreamllm_amd.modeling_dreamllm_sdxl import DreamLLMSDXLConfig, DreamLLMSDXLForCausalMLM
    from oracle.make_golden_sdxl import special_tokens2ids_dict
    g = golden("causal_mlm_sdxl.pt")
    cd = g["cfg"]
    cfg = DreamLLMSDXLConfig(vocab_size=cd["vocab_size"], hidden_size=cd["hidden_size"], intermediate_size=cd["intermediate_size"],
                             num_hidden_layers=cd["num_hidden_layers"], num_attention_heads=cd["num_attention_heads"],
                             rms_norm_eps=cd["rms_norm_eps"], max_position_embeddings=cd["max_position_embeddings"],
                             special_tokens2ids_dict=special_tokens2ids_dict(), loss_weight_lm=g["loss_weight_lm"],
                             loss_weight_vm=g["loss_weight_vm"], loss_scale_schedule=g["loss_scale_schedule"])
    lm = DreamLLMSDXLForCausalMLM(cfg)
    lm.model.dream_embedding = _FakeDream(cd["hidden_size"])
    lm.model.clip_vision_embedding = _NoClip()
    lm.stable_diffusion_head = _FakeXLHead()
    res = lm.load_state_dict(g["sd"], strict=False)
    assert not res.unexpected_keys and not res.missing_keys, (res.unexpected_keys, res.missing_keys)  # no inv_freq either side
    lm = lm.to(DEV, BF).train()
    kw = dict(input_ids=g["input_ids"].to(DEV), attention_mask=g["attention_mask"].to(DEV), labels=g["labels"].to(DEV), return_dict=True)
    out = lm(images_dm=g["images_dm"].to(BF).to(DEV), add_time_ids=g["add_time_ids"].to(DEV), **kw)
    out.loss.backward()
    er = golden("err_ref.pt")["causal_mlm_sdxl"]  # the reference SDXL model file EXECUTED in bf16
    check_tensor("causal_mlm_sdxl.logits", out.logits, g["logits"], er["logits"])
    check_scalar("causal_mlm_sdxl.lm_loss", out.additional_log_info["lm_loss"], g["lm_loss"], er["lm_loss"]["abs_err"])
    check_scalar("causal_mlm_sdxl.vm_loss", out.additional_log_info["vm_loss"], g["vm_loss"], er["vm_loss"]["abs_err"])
    check_scalar("causal_mlm_sdxl.loss", out.loss, g["loss"], er["loss"]["abs_err"])   # (3 vm + lm) / 4 / 4
    check_tensor("causal_mlm_sdxl.grad_dream", lm.model.dream_embedding.dream_queries.grad, g["grad_dream"], er["grad_dream"])
    check_tensor("causal_mlm_sdxl.grad_q0", lm.model.layers[0].self_attn.q_proj.weight.grad, g["grad_q0"].float(), er["grad_q0"])
    lm.zero_grad(set_to_none=True)
    out2 = lm(images_dm=None, add_time_ids=None, **kw)
    check_scalar("causal_mlm_sdxl.loss_dummy", out2.loss, g["loss_dummy"], er["loss_dummy"]["abs_err"])


def test_collator_indices_match_the_model_path(golden):
    """The data bridge (dreamllm_amd/data.py): a batch collated from per-sample dicts, carrying seqlens / dream_index /
    image_index, gives bit-identical loss and logits to the same batch without them (the model then derives the slots and the
    spans from input_ids / attention_mask on the device)."""
    from types import SimpleNamespace
    from dreamllm_amd.data import DataCollatorForDreamLLMDataset
    g = golden("causal_mlm.pt")
    lm = _build_lm(g).train()
    am = g["attention_mask"]
    ids = g["input_ids"]
    sp = lm.config.special_tokens2ids_dict["additional_special_tokens"]
    exs, ic, dc = [], 0, 0
    for b in range(ids.shape[0]):
        n = int(am[b].sum())
        ni = int((ids[b, :n] == sp["<im_start>"]).sum())
        nd = int((ids[b, :n] == sp["<dream_start>"]).sum())
        exs.append(dict(input_ids=ids[b, :n], attention_mask=am[b, :n], labels=g["labels"][b, :n],
                        images=g["images"][ic:ic + ni] if ni else None, images_dm=g["images_dm"][dc:dc + nd] if nd else None))
        ic, dc = ic + ni, dc + nd
    col = DataCollatorForDreamLLMDataset.from_model(SimpleNamespace(pad_token_id=150), lm)
    assert (col.n_dream, col.n_patch) == (4, 6)
    batch = col(exs)
    valid = am.bool()  # the fixture's pad positions hold arbitrary tokens; the collator writes pad_token_id there
    assert torch.equal(batch["input_ids"][valid], ids[valid]) and torch.equal(batch["labels"], g["labels"])
    assert torch.equal(batch["attention_mask"], am)
    dev = lambda v: v.to(DEV) if torch.is_tensor(v) and not v.is_floating_point() else (v.to(BF).to(DEV) if torch.is_tensor(v) else v)
    full = {k: dev(v) for k, v in batch.items()}
    a = lm(**full, return_dict=True)
    bare = {k: v for k, v in full.items() if k not in ("seqlens", "dream_index", "image_index")}
    b = lm(**bare, return_dict=True)
    assert torch.equal(a.loss, b.loss) and torch.equal(a.logits, b.logits)
    # `loss_index` (rows whose shifted label carries a loss): the fused lm_head + CE skips the ignored rows -- same loss and the same
    # gradients as scoring every row (the ignored rows contribute exact zeros); only the fp32 summation order over rows changes
    shift = torch.cat([batch["labels"][:, 1:], torch.full((ids.shape[0], 1), -100)], 1).reshape(-1)
    assert torch.equal(batch["loss_index"], torch.nonzero(shift != -100).flatten()) and 0 < batch["loss_index"].numel() < shift.numel()
    lm.zero_grad(set_to_none=True)
    a.loss.backward()
    ga = {n: p.grad.clone() for n, p in lm.named_parameters() if p.grad is not None}
    lm.zero_grad(set_to_none=True)
    c = lm(**{k: v for k, v in full.items() if k != "loss_index"}, return_dict=True)
    c.loss.backward()
    assert abs(float(a.loss) - float(c.loss)) <= 1e-6 * abs(float(c.loss)) and torch.equal(a.logits, c.logits)
    for n, p in lm.named_parameters():
        if p.grad is not None:
            assert rel_l2(ga[n], p.grad) < 2e-3, n   # (lm_head dW: fp32 chunk sums in a different order, then one bf16 rounding)


@pytest.mark.parametrize("n_kv", [2, 1])
def test_decoder_layer_packed_weights_match_unpacked(n_kv):
    """q/k/v and gate/up as ONE GEMM each over packed weights (`pack_linear_weights`: parameters keep their identity and
    state_dict keys) against one GEMM per projection: forward, input gradient and every weight gradient; GQA included."""
    from dreamllm_amd.configuration_dreamllm import DreamLLMConfig
    from dreamllm_amd.modeling_dreamllm import DreamLLMDecoderLayer, _packed_view
    res = {}
    for packed in (False, True):
        cfg = DreamLLMConfig(vocab_size=64, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
                             num_key_value_heads=n_kv, max_position_embeddings=128)
        cfg.pack_projection_weights = packed
        torch.manual_seed(3)
        layer = DreamLLMDecoderLayer(cfg)
        keys = list(layer.state_dict())
        layer = layer.to(DEV, BF)
        torch.manual_seed(4)
        x = torch.randn(2, 96, 128, device=DEV).to(BF).requires_grad_(True)
        lens = torch.tensor([96, 50], dtype=torch.int32, device=DEV)
        y = layer(x, seqlens=lens)[0]
        y.backward(torch.randn_like(y))
        a = layer.self_attn
        assert (_packed_view(a.q_proj.weight, a.k_proj.weight, a.v_proj.weight) is not None) == packed
        assert list(layer.state_dict()) == keys and layer.self_attn.k_proj.weight.shape == (n_kv * 64, 128)
        res[packed] = (y.detach(), x.grad, {n: p.grad for n, p in layer.named_parameters()})
    assert torch.equal(res[True][0], res[False][0])              # same per-element dot products in the forward
    assert rel_l2(res[True][1], res[False][1]) < 4e-3           # dh: one K = 3H GEMM vs three accumulating ones (extra roundings)
    for n in res[True][2]:  # the norm-weight gradients sum the (differently rounded) dh over all tokens
        assert rel_l2(res[True][2][n], res[False][2][n]) < (8e-3 if "layernorm" in n else 4e-3), n


def test_glu_bwd_emits_forward_product():
    from dreamllm_amd import ops
    torch.manual_seed(0)
    g, u, d = (torch.randn(300, 512, device=DEV).to(BF) for _ in range(3))
    act = torch.empty_like(g)
    dg, du = ops.glu_bwd(d, g, u, 0, act_out=act)
    dg2, du2 = ops.glu_bwd(d, g, u, 0)
    assert torch.equal(act, ops.glu_fwd(g, u, 0)) and torch.equal(dg, dg2) and torch.equal(du, du2)


def test_gradient_checkpointing_recomputes_the_layer_and_matches_bit_for_bit():
    """`gradient_checkpointing_enable()` (the reference's stage-II recipe, projects/dreamllm/configs/stage2/base.py:99;
    modeling_dreamllm.py:994-1003) = whole-layer activation recompute inside _DecoderLayerFn: only each layer's input stays resident,
    the backward re-runs the forward launches.  Same kernels on the same inputs: loss and EVERY gradient are bit-identical to the
    keeping path, and the activations held between forward and backward shrink."""
    from dreamllm_amd.factory import TINY, TINY_CLIP, TINY_DIFFUSION, build_dreamllm
    from dreamllm_amd.synthetic import make_interleaved_batch
    m = build_dreamllm(dict(TINY, num_hidden_layers=4), device=DEV, seed=0, clip=TINY_CLIP, diffusion=TINY_DIFFUSION, num_dream_queries=8).train()
    batch = make_interleaved_batch(2, 256, 1, n_dream=8, n_patch=16, seed=11, device=DEV, image_size=56, dm_size=128)

    def run():
        m.zero_grad(set_to_none=True)
        torch.manual_seed(5)                      # the SD head draws its noise / timesteps from the default generator
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        out = m(**batch, return_dict=True)
        held = torch.cuda.memory_allocated() - base
        out.loss.backward()
        return out.loss.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, held

    loss0, g0, held0 = run()
    m.gradient_checkpointing_enable()
    assert m.model.gradient_checkpointing
    loss1, g1, held1 = run()
    m.gradient_checkpointing_disable()
    assert not m.model.gradient_checkpointing
    assert torch.equal(loss0, loss1)
    assert g0.keys() == g1.keys() and len(g0) > 30
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    assert held1 < held0, (held0, held1)
    # eval / no-grad forwards ignore the flag (the reference checks `self.training`, modeling_dreamllm.py:994)
    m.gradient_checkpointing_enable()
    m.eval()
    with torch.no_grad():
        a = m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], return_dict=True).logits
    m.gradient_checkpointing_disable()
    with torch.no_grad():
        b = m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], return_dict=True).logits
    assert torch.equal(a, b)


def test_training_step_is_bit_reproducible():
    """The same batch and seed twice: loss (both terms) and every gradient bit-identical.  Round 6 found the SD head's MSE summed its
    per-block partials with float atomics (one ulp of run-to-run jitter in vm_loss, 12 of 40 repetitions); every reduction on the
    path is now a fixed-order sum, which is also what keeps data-parallel replicas bit-identical."""
    from dreamllm_amd.factory import TINY, TINY_CLIP, TINY_DIFFUSION, build_dreamllm
    from dreamllm_amd.synthetic import make_interleaved_batch
    m = build_dreamllm(TINY, device=DEV, seed=0, clip=TINY_CLIP, diffusion=TINY_DIFFUSION, num_dream_queries=8).train()
    batch = make_interleaved_batch(2, 256, 1, n_dream=8, n_patch=16, seed=11, device=DEV, image_size=56, dm_size=128)

    def run():
        m.zero_grad(set_to_none=True)
        torch.manual_seed(5)
        out = m(**batch, return_dict=True)
        out.loss.backward()
        return out.loss.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    loss0, g0 = run()
    for rep in range(12):
        loss, g = run()
        assert torch.equal(loss, loss0), rep
        for n in g0:
            assert torch.equal(g[n], g0[n]), (rep, n)


def test_training_step_has_the_same_bits_on_either_gemm_kernel_family():
    """Round 6: the four-wave GEMM kernel (csrc/gemm_w4.hip) takes the decoder's linears from 256 full tiles on; it adds the products in the order
    of the 8-wave kernel, so a training step must not change by one bit.  A 2-layer model wide enough for the library to pick it (hidden 1024,
    MLP 4096, 8192 tokens: 128 ... 512 tiles per launch -- the packed q|k|v, gate|up + SwiGLU, down-dgrad + SwiGLU-backward and weight-gradient
    launches of >= 256 tiles run on it), stepped with DREAMLLM_W4M on and off: loss and every gradient bit-identical."""
    from dreamllm_amd import ops
    from dreamllm_amd.factory import TINY_CLIP, TINY_DIFFUSION, build_dreamllm
    from dreamllm_amd.synthetic import make_interleaved_batch
    cfg = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=2, num_attention_heads=8, max_position_embeddings=2048, rms_norm_eps=1e-6)
    m = build_dreamllm(cfg, device=DEV, seed=0, clip=TINY_CLIP, diffusion=TINY_DIFFUSION, num_dream_queries=8).train()
    batch = make_interleaved_batch(4, 2048, 1, n_dream=8, n_patch=16, seed=11, device=DEV, image_size=56, dm_size=128)

    def run(w4m):
        prev, ops.W4M = ops.W4M, w4m
        try:
            m.zero_grad(set_to_none=True)
            torch.manual_seed(5)
            out = m(**batch, return_dict=True)
            out.loss.backward()
            return out.loss.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            ops.W4M = prev

    l1, g1 = run(True)
    l0, g0 = run(False)
    assert torch.equal(l1, l0)
    assert g1.keys() == g0.keys() and len(g1) > 10
    for n in g1:
        assert torch.equal(g1[n], g0[n]), n


def test_ragged_batch_on_compact_rows_matches_the_padded_grid():
    """Round 6: a right-padded ragged training batch with `seqlens` runs the decoder on COMPACT rows (valid tokens back to back, attention
    alone on the padded grid).  Loss, both loss terms and every gradient must agree with the padded-grid path (same kernels on the same
    rows; only split-K choices and the weight gradients' token order may differ: bf16-level tolerance), hidden states at pad positions
    are zeros, and a dense batch is untouched."""
    from dreamllm_amd import modeling_dreamllm as MD
    from dreamllm_amd.factory import TINY, TINY_CLIP, TINY_DIFFUSION, build_dreamllm
    from dreamllm_amd.synthetic import make_interleaved_batch
    m = build_dreamllm(dict(TINY, num_hidden_layers=3), device=DEV, seed=0, clip=TINY_CLIP, diffusion=TINY_DIFFUSION, num_dream_queries=8).train()
    batch = make_interleaved_batch(6, 512, 1, n_dream=8, n_patch=16, seed=3, device=DEV, image_size=56, dm_size=128, ragged=True)
    lens = batch["seqlens"].tolist()
    assert sum(lens) < 0.9 * 6 * 512

    def run(pack):
        MD.PACK_RAGGED = pack
        try:
            m.zero_grad(set_to_none=True)
            torch.manual_seed(5)
            out = m(**batch, return_dict=True)
            out.loss.backward()
            return out.loss.detach().float().item(), {n: p.grad.float().clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            MD.PACK_RAGGED = True

    l0, g0 = run(False)
    l1, g1 = run(True)
    m.gradient_checkpointing_enable()          # whole-layer recompute on compact rows: the same bits as keeping them
    l2, g2 = run(True)
    m.gradient_checkpointing_disable()
    assert l2 == l1 and all(torch.equal(g1[n], g2[n]) for n in g1)
    assert abs(l0 - l1) <= 2e-3 * abs(l0), (l0, l1)
    assert g0.keys() == g1.keys()
    worst = max(((g0[n] - g1[n]).norm() / (g0[n].norm() + 1e-12)).item() for n in g0)
    assert worst < 2e-2, worst
    # the model's hidden states: valid rows agree, pad rows are zeros
    with torch.enable_grad():
        MD.PACK_RAGGED = True
        h1 = m.model(input_ids=batch["input_ids"], images=batch["images"], seqlens=batch["seqlens"], image_index=batch["image_index"],
                     dream_index=batch["dream_index"], return_dict=True).last_hidden_state
        MD.PACK_RAGGED = False
        h0 = m.model(input_ids=batch["input_ids"], images=batch["images"], seqlens=batch["seqlens"], image_index=batch["image_index"],
                     dream_index=batch["dream_index"], return_dict=True).last_hidden_state
        MD.PACK_RAGGED = True
    for b, L in enumerate(lens):
        assert rel_l2(h1[b, :L], h0[b, :L].float()) < 1e-2
        assert torch.count_nonzero(h1[b, L:]) == 0
