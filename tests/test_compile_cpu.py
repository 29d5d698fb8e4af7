"""CPU: `torch.compile` traceability of the inference forward through the registered custom ops (`dreamllm_amd/torch_ops.py`,
SURVEY.md §8-b1: the reference's inference scripts compile the model, projects/dreamllm/inference.py:70).

The HIP kernels cannot run here, so the four operators the traced forward reaches (`ops.embedding`, `ops.rmsnorm`, `ops.linear`, the
fused `_DecoderLayerFn`) are replaced -- in this test only -- by the CPU oracle (`oracle/llm_ref.py`); what is checked is the part
that does not need a GPU: Dynamo captures `DreamLLMForCausalMLM.forward` (text prompt, prefill with and without a KV cache) as ONE
graph whose compute nodes are `torch.ops.dreamllm.*`, with no graph break before the output object is built, and the compiled
module returns the oracle's logits.  The GPU twin (tests/test_model_gpu.py::test_torch_compile_coexistence) runs the real kernels."""
import pytest
import torch
import torch.nn.functional as F


@pytest.fixture()
def cpu_kernels(monkeypatch):
    from dreamllm_amd import modeling_dreamllm as M
    from dreamllm_amd import ops
    from oracle import llm_ref

    class OracleLayerFn:
        @staticmethod
        def apply(x, w_in, wq, wk, wv, wo, w_post, wg, wu, wd, cos, sin, pos, seqlens, seqstart, nh, nkv, eps, want_kv):
            sd = {"input_layernorm.weight": w_in, "self_attn.q_proj.weight": wq, "self_attn.k_proj.weight": wk,
                  "self_attn.v_proj.weight": wv, "self_attn.o_proj.weight": wo, "post_attention_layernorm.weight": w_post,
                  "mlp.gate_proj.weight": wg, "mlp.up_proj.weight": wu, "mlp.down_proj.weight": wd}
            B, S, H = x.shape
            c, s = torch.cat([cos, cos], -1), torch.cat([sin, sin], -1)
            y = llm_ref.decoder_layer(x, sd, "", dict(num_attention_heads=nh, num_key_value_heads=nkv, rms_norm_eps=eps), c, s,
                                      torch.arange(S)[None], llm_ref.causal_mask_4d(None, B, S, x.dtype))
            hd = H // nh
            return y, x.new_zeros(B, S, nkv, hd), x.new_zeros(B, S, nkv, hd)

    monkeypatch.setattr(ops, "embedding", lambda w, ids: F.embedding(ids, w))
    monkeypatch.setattr(ops, "rmsnorm", lambda x, w, eps: llm_ref.rmsnorm(x, w, eps))
    monkeypatch.setattr(ops, "linear", lambda x, w, b=None, r=None, f32=False: F.linear(x, w, b).float() if f32 else F.linear(x, w, b))
    monkeypatch.setattr(M, "_DecoderLayerFn", OracleLayerFn)
    return llm_ref


@pytest.mark.parametrize("use_cache", [False, True])
def test_text_forward_compiles_to_one_graph_of_registered_ops(cpu_kernels, use_cache):
    from dreamllm_amd.configuration_dreamllm import DreamLLMConfig
    from dreamllm_amd.modeling_dreamllm import DreamLLMForCausalMLM
    from dreamllm_amd.tokenization_dreamllm import default_special_tokens2ids
    llm_ref = cpu_kernels
    torch.manual_seed(0)
    cfg = DreamLLMConfig(special_tokens2ids_dict=default_special_tokens2ids(50), vocab_size=64, hidden_size=128, intermediate_size=256,
                         num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=64)
    lm = DreamLLMForCausalMLM(cfg).eval()
    ids = torch.randint(3, 50, (2, 16))
    torch._dynamo.reset()
    with torch.no_grad():
        ex = torch._dynamo.explain(lm)(input_ids=ids, use_cache=use_cache, return_dict=True)
    assert ex.graph_count == 1 and ex.graph_break_count == 0, ex.break_reasons
    names = [str(n.target) for g in ex.graphs for n in g.graph.nodes if n.op == "call_function"]
    layer_op = "dreamllm.decoder_layer_kv" if use_cache else "dreamllm.decoder_layer"
    assert names.count(layer_op) == 2 and names.count("dreamllm.embedding") == 1
    assert names.count("dreamllm.rmsnorm") == 1 and names.count("dreamllm.linear") == 1       # final norm + lm_head
    assert sum(n.startswith("dreamllm.") for n in names) == 5
    # the compiled module computes what the oracle computes
    clm = torch.compile(lm, backend="eager")
    with torch.no_grad():
        out = clm(input_ids=ids, use_cache=use_cache, return_dict=True)
    sd = {k: v.detach() for k, v in lm.state_dict().items()}
    cd = dict(num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2, rms_norm_eps=cfg.rms_norm_eps,
              max_position_embeddings=64, rope_theta=cfg.rope_theta)
    ref = F.linear(llm_ref.model_forward(F.embedding(ids, sd["model.embed_tokens.weight"]), sd, cd), sd["lm_head.weight"])
    assert (out.logits - ref).abs().max() < 1e-4 * ref.abs().max()
    assert (out.past_key_values is not None) == use_cache
    torch._dynamo.reset()


def test_registered_ops_have_schemas_and_fake_kernels():
    import dreamllm_amd.torch_ops  # noqa: F401
    for name in ("embedding", "rmsnorm", "linear", "decoder_layer", "decoder_layer_kv"):
        op = getattr(torch.ops.dreamllm, name)
        assert op.default._schema.name == f"dreamllm::{name}"
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x, w = torch.empty(2, 8, 128), torch.empty(64, 128)
        assert torch.ops.dreamllm.linear(x, w, None, True).shape == (2, 8, 64)
        assert torch.ops.dreamllm.linear(x, w, None, True).dtype == torch.float32
        assert torch.ops.dreamllm.embedding(w, torch.empty(2, 5, dtype=torch.long)).shape == (2, 5, 128)
