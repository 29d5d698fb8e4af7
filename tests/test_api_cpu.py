"""CPU: drop-in surface of the reference: config fields, plugin contract, state_dict keys, registry,
synthetic batch layout, host logic of the splice indices.  No kernels are launched."""
import pytest
import torch

from dreamllm_amd.configuration_dreamllm import DreamLLMConfig, create_config_init_kwargs
from dreamllm_amd.factory import TINY, build_dreamllm, plugin_configs
from dreamllm_amd.modeling_dreamllm import DreamLLMDecoderLayer, DreamLLMForCausalMLM, DreamLLMModel, DreamLLMRMSNorm, _slot_indices
from dreamllm_amd.modeling_plugins import (CLIPVisionEmbedding, DreamEmbedding, MultimodalEmbedding, MultimodalHead, PluginBase,
                                            StableDiffusionHead)
from dreamllm_amd.projector import build_projector
from dreamllm_amd.synthetic import make_interleaved_batch
from oracle import unet_ref

TINY_CLIP = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=56)
TINY_SD = dict(unet=unet_ref.tiny_config(64), vae=dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1))


@pytest.fixture(scope="module")
def tiny_model():
    return build_dreamllm(TINY, device="cpu", dtype=torch.float32, clip=TINY_CLIP, diffusion=TINY_SD, num_dream_queries=8)


def test_config_defaults_and_plugin_registry():
    cfg = DreamLLMConfig()
    assert (cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_key_value_heads) == (4096, 11008, 32, 32)
    assert cfg.loss_weight_vm == 10.0 and cfg.loss_weight_lm == 1.0 and cfg.rope_scaling is None
    with pytest.raises(ValueError):
        create_config_init_kwargs(dict(_class_="notatype", _name_="x", _plugin_type_="head"))
    name = cfg.update_plugins(plugin_configs(4096)[0])
    assert name == "dream_embedding" and cfg.plugins_type[name] == "embedding"
    assert cfg.plugins_init_kwargs[name]["_target_"].endswith("modeling_plugins.DreamEmbedding")
    with pytest.raises(ValueError):
        DreamLLMConfig(rope_scaling={"type": "bogus", "factor": 2.0})
    assert DreamLLMConfig.from_dict(cfg.to_dict()).plugins_type == cfg.plugins_type  # JSON round trip


def test_state_dict_keys_match_reference_layout(tiny_model):
    keys = set(tiny_model.state_dict().keys())
    for k in ("model.embed_tokens.weight", "model.norm.weight", "lm_head.weight", "model.layers.0.self_attn.q_proj.weight",
              "model.layers.0.self_attn.rotary_emb.inv_freq", "model.layers.1.mlp.down_proj.weight",
              "model.layers.0.input_layernorm.weight", "model.layers.0.post_attention_layernorm.weight",
              "model.dream_embedding.dream_queries", "model.clip_vision_embedding.projector.projector.weight",
              "model.clip_vision_embedding.projector.projector.bias",
              "model.clip_vision_embedding.clip_vision_model.vision_model.embeddings.class_embedding",
              "model.clip_vision_embedding.clip_vision_model.vision_model.pre_layrnorm.weight",
              "stable_diffusion_head.projector.projector.weight", "stable_diffusion_head.unet.conv_in.weight",
              "stable_diffusion_head.unet.mid_block.attentions.0.transformer_blocks.0.attn2.to_k.weight",
              "stable_diffusion_head.vae.encoder.conv_in.weight", "stable_diffusion_head.vae.quant_conv.weight"):
        assert k in keys, k
    assert "stable_diffusion_head.projector.projector.bias" not in keys  # bias=False on the SD side (modeling_plugins.py:389-391)


def test_plugin_contract(tiny_model, tmp_path):
    de, ce, sh = tiny_model.model.dream_embedding, tiny_model.model.clip_vision_embedding, tiny_model.stable_diffusion_head
    assert isinstance(de, MultimodalEmbedding) and isinstance(ce, MultimodalEmbedding) and isinstance(sh, MultimodalHead)
    assert issubclass(MultimodalHead, PluginBase) and PluginBase.initializer_range == 0.02
    assert (de.plugin_type, ce.plugin_type, sh.plugin_type) == ("embedding", "embedding", "head")
    assert de.embed_len == 8 and de.embed_dim == 256 and ce.embed_len == 16 and ce.embed_dim == 256
    assert de(3).shape == (3, 8, 256)
    assert de.fsdp_ignored_modules() == [] and ce.fsdp_ignored_modules() == [ce.clip_vision_model]
    assert sh.fsdp_ignored_modules() == [sh.vae, sh.unet]
    assert not any(p.requires_grad for p in sh.unet.parameters()) and sh.projector.projector.weight.requires_grad
    assert set(de.config) == {"pretrained_model_name_or_path", "num_dream_queries", "embed_len", "embed_dim", "freeze_dream_queries"}
    # save / load round trip in the reference's file layout ({save_model_name}.bin)
    for p in (de, ce, sh):
        p.save_model(str(tmp_path))
    assert sorted(f.name for f in tmp_path.iterdir()) == ["clip_vision_embedding.bin", "dream_embedding.bin", "stable_diffusion_head.bin"]
    de2 = DreamEmbedding(pretrained_model_name_or_path=str(tmp_path), num_dream_queries=8, embed_hidden_size=256)
    assert torch.equal(de2.dream_queries, de.dream_queries)
    with pytest.raises(NotImplementedError):
        CLIPVisionEmbedding(TINY_CLIP, embed_hidden_size=64, freeze_clip_vision_model=False)
    with pytest.raises(NotImplementedError):
        StableDiffusionHead(TINY_SD, embed_hidden_size=64, freeze_unet=False)
    with pytest.raises(ValueError):
        sh.check_inputs(100, 128, 1, prompt_embeds=torch.zeros(1, 8, 256))


def test_sdxl_head_contract(tmp_path):
    """StableDiffusionXLHead (omni/models/dreamllm_sdxl/modeling_plugins.py:48-149): constructor kwargs, config keys,
    state_dict keys, save/load file name, SDXLDataProcessor time ids."""
    from dreamllm_amd.modeling_plugins_sdxl import SDXLDataProcessor, StableDiffusionXLHead
    sd = dict(unet=unet_ref.tiny_config(64, sdxl=True), vae=dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1))
    h = StableDiffusionXLHead(sd, embed_hidden_size=64, global_condition_hidden_size=40)
    assert isinstance(h, StableDiffusionHead) and h.plugin_type == "head" and h.save_model_name == "stable_diffusion_xl_head"
    assert h.projector.projector.weight.shape == (64, 64) and h.global_projector.projector.weight.shape == (40, 64)
    assert set(h.config) == {"diffusion_name_or_path", "pretrained_model_name_or_path", "embed_hidden_size",
                             "global_condition_hidden_size", "drop_prob", "noise_offset", "input_perturbation", "snr_gamma",
                             "freeze_vae", "freeze_unet", "freeze_projector"}
    assert h.fsdp_ignored_modules() == [h.vae, h.unet]
    hf = StableDiffusionXLHead(sd, embed_hidden_size=64, global_condition_hidden_size=40, freeze_projector=True)
    assert hf.fsdp_ignored_modules() == [hf.vae, hf.unet, hf.projector, hf.global_projector]
    assert not hf.global_projector.projector.weight.requires_grad
    h.save_model(str(tmp_path))
    assert [f.name for f in tmp_path.iterdir()] == ["stable_diffusion_xl_head.bin"]
    h2 = StableDiffusionXLHead(sd, pretrained_model_name_or_path=str(tmp_path), embed_hidden_size=64, global_condition_hidden_size=40)
    assert torch.equal(h2.global_projector.projector.weight, h.global_projector.projector.weight)
    proc = h.processor
    assert isinstance(proc, SDXLDataProcessor) and proc.resolution == 1024
    img, ids = SDXLDataProcessor(resolution=64, center_crop=True)(torch.rand(3, 100, 150))
    assert img.shape == (3, 64, 64) and ids == [100, 150, 0, 16, 64, 64] and -1.0 <= float(img.min()) and float(img.max()) <= 1.0


def test_projector_registry():
    cfg = dict(projector="linear", freeze_projector=False, depth=1, save_model_name="x", model_name_or_path=None)
    p = build_projector(cfg, 32, 64, bias=True)
    assert set(p.state_dict()) == {"projector.weight", "projector.bias"} and p.save_model_name == "x_projector"
    m = build_projector(dict(cfg, projector="mlp", depth=3), 32, 64, bias=False)
    assert set(m.state_dict()) == {"projector.0.weight", "projector.2.weight", "projector.4.weight"}
    with pytest.raises(AssertionError):
        build_projector(cfg, 32, 64, bias=None)
    with pytest.raises(AssertionError):
        build_projector(dict(cfg, projector="mlp", depth=1), 32, 64, bias=False)
    with pytest.raises(ValueError):
        build_projector(dict(cfg, projector="nope"), 32, 64, bias=False)


def test_model_api_surface(tiny_model):
    assert isinstance(tiny_model.model, DreamLLMModel) and isinstance(tiny_model.model.layers[0], DreamLLMDecoderLayer)
    assert tiny_model.get_decoder() is tiny_model.model and tiny_model.get_output_embeddings() is tiny_model.lm_head
    from transformers.pytorch_utils import ALL_LAYERNORM_LAYERS
    assert DreamLLMRMSNorm in ALL_LAYERNORM_LAYERS
    assert DreamLLMForCausalMLM.config_class is DreamLLMConfig and DreamLLMForCausalMLM._no_split_modules == ["DreamLLMDecoderLayer"]
    ign = set(tiny_model._keys_to_ignore_on_save)
    assert "stable_diffusion_head.projector.projector.weight" in ign and "model.dream_embedding.dream_queries" in ign
    with pytest.raises(RuntimeError):  # CPU tensors never reach a silent fallback
        tiny_model(input_ids=torch.ones(1, 8, dtype=torch.long))


def test_synthetic_batch_layout_and_slot_indices():
    b = make_interleaved_batch(3, 512, 1, n_dream=8, n_patch=16, seed=3, with_pixels=False, ragged=True)
    ids, lab, am = b["input_ids"], b["labels"], b["attention_mask"]
    sp = {"<dream_start>": 32006, "<im_start>": 32003, "<im_patch>": 32002, "<im_end>": 32004, "<dream_end>": 32007}
    assert (ids[:, 0] == 1).all() and ((am[:, :-1] - am[:, 1:]) >= 0).all()  # bos first, right padding only
    assert (lab[ids == sp["<im_patch>"]] == -100).all() and (lab[ids == sp["<im_end>"]] == -100).all()
    assert (lab[ids == sp["<dream_start>"]] == sp["<dream_start>"]).all() and (lab[am == 0] == -100).all()
    di, n = _slot_indices(ids, sp["<dream_start>"], 8)
    ii, m = _slot_indices(ids, sp["<im_start>"], 16)
    assert n == 3 and m == 3 and torch.equal(di, b["dream_index"]) and torch.equal(ii, b["image_index"])
    assert (ids.view(-1)[di] == sp["<im_patch>"]).all() and (ids.view(-1)[ii + 16 - 15 * 0][-1] != -1)
    full = make_interleaved_batch(2, 2048, 2, with_pixels=False)
    assert full["attention_mask"].all() and full["dream_index"].numel() == 2 * 2 * 64 and full["image_index"].numel() == 2 * 2 * 256


def test_sdxl_model_classes_and_creation_batch():
    """DreamLLM-SDXL host side (omni/models/dreamllm_sdxl/*): 8 additional special tokens with <dream_patch> between <dream>
    and <dream_start> (vocab 32009), class names, non-persistent inv_freq, stage-I freeze policy, creation-only batch layout."""
    from dreamllm_amd.factory import build_dreamllm_sdxl
    from dreamllm_amd.modeling_dreamllm_sdxl import (DreamLLMSDXLConfig, DreamLLMSDXLForCausalMLM, DreamLLMSDXLModel,
                                                      additional_special_tokens, default_special_tokens2ids)
    from dreamllm_amd.synthetic import make_creation_batch
    ids = default_special_tokens2ids(32000)
    add = ids["additional_special_tokens"]
    assert additional_special_tokens.index("<dream_patch>") == additional_special_tokens.index("<dream>") + 1
    assert add["<dream_patch>"] == 32006 and add["<dream_start>"] == 32007 and add["<dream_end>"] == 32008 and ids["[PAD]"] == 32000
    sd = dict(unet=unet_ref.tiny_config(64, sdxl=True), vae=dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1))
    m = build_dreamllm_sdxl(TINY, device="cpu", dtype=torch.float32, with_clip=False, diffusion=sd, num_dream_queries=8,
                            global_condition_hidden_size=40)
    assert isinstance(m, DreamLLMSDXLForCausalMLM) and isinstance(m.model, DreamLLMSDXLModel) and isinstance(m.config, DreamLLMSDXLConfig)
    assert m.config.vocab_size == 32009 and m.lm_head.weight.shape[0] == 32009
    assert m._dream_patch_token == "<dream_patch>" and m._loss_scale_twice
    assert not any("inv_freq" in k for k in m.state_dict())
    assert sorted(n for n, p in m.named_parameters() if p.requires_grad) == [
        "model.dream_embedding.dream_queries", "stable_diffusion_head.global_projector.projector.weight",
        "stable_diffusion_head.projector.projector.weight"]
    assert (m.config.loss_weight_lm, m.config.loss_weight_vm) == (0.0, 1.0)
    b = make_creation_batch(batch_size=3, seq_len=32, n_dream=8, with_pixels=False)
    ii = b["input_ids"]
    assert ii.shape == (3, 32) and (ii[:, 0] == 1).all() and (ii[:, -1] == 2).all()
    assert (ii == add["<dream_patch>"]).sum() == 3 * 8 and (ii == add["<dream_start>"]).sum() == 3
    di, _ = _slot_indices(ii, add["<dream_start>"], 8)
    assert torch.equal(di, b["dream_index"])
    assert (b["labels"][ii == add["<dream_patch>"]] == -100).all() and (b["labels"][ii == add["<dream_end>"]] == -100).all()


def test_from_pretrained_with_tokenizer_resizes_and_builds_plugins(tmp_path):
    """`DreamLLMForCausalMLM.from_pretrained(path, tokenizer, config=..., use_flash_attention_2=..., torch_dtype=...)` as
    projects/dreamllm/train.py:130-137 calls it (modeling_dreamllm.py:1244-1333): weights loaded, embeddings grown to the
    tokenizer, plugins pointed at the model folder and built AFTER the resize; `average_init_token_embeddings`
    (omni/utils/tokenizer_utils.py:70-80) for the added rows."""
    import torch
    from dreamllm_amd.configuration_dreamllm import DreamLLMConfig, create_config_init_kwargs
    from dreamllm_amd.modeling_dreamllm import DreamLLMForCausalMLM
    from dreamllm_amd.modeling_plugins import DreamEmbedding
    from dreamllm_amd.utils import average_init_token_embeddings
    cfg = DreamLLMConfig(vocab_size=64, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
                         max_position_embeddings=32)
    lm = DreamLLMForCausalMLM(cfg)
    lm.save_pretrained(str(tmp_path))
    de = DreamEmbedding(num_dream_queries=4, embed_hidden_size=128)
    de.save_model(str(tmp_path))
    cfg2 = DreamLLMConfig.from_pretrained(str(tmp_path))
    name = cfg2.update_plugins(create_config_init_kwargs(dict(_class_=DreamEmbedding, _name_="dream_embedding", _plugin_type_="embedding",
                                                              pretrained_model_name_or_path=None, num_dream_queries=4,
                                                              embed_hidden_size=128)))
    assert name == "dream_embedding"

    class Tok:
        def __len__(self):
            return 72

    with pytest.raises(AssertionError):
        DreamLLMForCausalMLM.from_pretrained(str(tmp_path))  # tokenizer is mandatory (:1265)
    m = DreamLLMForCausalMLM.from_pretrained(str(tmp_path), Tok(), config=cfg2, local_files_only=True, use_flash_attention_2=True,
                                             torch_dtype=torch.bfloat16)
    assert m.config.vocab_size == 72 and m.model.embed_tokens.weight.shape == (72, 128) and m.lm_head.weight.shape == (72, 128)
    assert m.dtype == torch.bfloat16
    assert torch.equal(m.model.embed_tokens.weight[:64].float(), lm.model.embed_tokens.weight.to(torch.bfloat16).float())
    assert torch.equal(m.model.dream_embedding.dream_queries.float(), de.dream_queries.to(torch.bfloat16).float())
    assert any(k.startswith("model.dream_embedding.") for k in m._keys_to_ignore_on_save)
    average_init_token_embeddings(m, 8)
    assert torch.allclose(m.model.embed_tokens.weight[-1].float(), m.model.embed_tokens.weight[:64].float().mean(0), atol=1e-2)


def test_data_collator_matches_reference_padding_and_model_slot_order():
    """`data.DataCollatorForDreamLLMDataset` = the reference collator (builder_dreamllm.py:465-482: right padding with
    pad_token_id / 0 / -100, images concatenated, None dropped) + seqlens and splice indices identical to what the model derives
    from `input_ids` on its own (`_slot_indices`, the visiting order of modeling_dreamllm.py:1085-1098,1110-1139)."""
    from types import SimpleNamespace
    from dreamllm_amd.data import DataCollatorForDreamLLMDataset, DataCollatorForDreamLLMSDXLDataset
    DS, IS, PATCH, PAD = 900, 901, 902, 999
    tok = SimpleNamespace(pad_token_id=PAD)

    def sample(n_text, with_img, with_dm, seed):
        g = torch.Generator().manual_seed(seed)
        ids = [1] + torch.randint(3, 800, (n_text,), generator=g).tolist()
        if with_dm:
            ids += [DS] + [PATCH] * 4 + [903]
        if with_img:
            ids += [IS] + [PATCH] * 6 + [904]
        ids += torch.randint(3, 800, (3,), generator=g).tolist() + [2]
        t = torch.tensor(ids)
        return dict(input_ids=t, attention_mask=torch.ones_like(t), labels=t.clone(),
                    images=torch.randn(1, 3, 8, 8, generator=g) if with_img else None,
                    images_dm=torch.randn(1, 3, 16, 16, generator=g) if with_dm else None)

    exs = [sample(5, True, True, 0), sample(11, False, True, 1), sample(2, True, False, 2), sample(7, False, False, 3)]
    col = DataCollatorForDreamLLMDataset(tok, dream_start_id=DS, image_start_id=IS, n_dream=4, n_patch=6)
    b = col(exs)
    S = max(len(e["input_ids"]) for e in exs)
    assert b["input_ids"].shape == (4, S) and b["images"].shape[0] == 2 and b["images_dm"].shape[0] == 2
    for i, e in enumerate(exs):
        n = len(e["input_ids"])
        assert torch.equal(b["input_ids"][i, :n], e["input_ids"]) and (b["input_ids"][i, n:] == PAD).all()
        assert (b["attention_mask"][i, n:] == 0).all() and (b["labels"][i, n:] == -100).all()
        assert int(b["seqlens"][i]) == n
    di, nd = _slot_indices(b["input_ids"], DS, 4)
    ii, ni = _slot_indices(b["input_ids"], IS, 6, max_slots=2)
    assert torch.equal(b["dream_index"], di) and torch.equal(b["image_index"], ii) and nd == 2 and ni == 2
    assert (b["input_ids"].reshape(-1)[b["dream_index"]] == PATCH).all()
    # no images at all in the batch: keys are None / absent, as in the reference
    b2 = col([exs[3], exs[3]])
    assert b2["images"] is None and b2["images_dm"] is None and "dream_index" not in b2
    # SDXL variant concatenates add_time_ids
    for e in exs:
        e["add_time_ids"] = torch.ones(1, 6) if e["images_dm"] is not None else None
    b3 = DataCollatorForDreamLLMSDXLDataset(tok, dream_start_id=DS, image_start_id=IS, n_dream=4, n_patch=6)(exs)
    assert b3["add_time_ids"].shape == (2, 6)


def test_lazy_logits_keep_the_model_output_contract():
    """with the fused lm_head + CE path `logits` is lazy, but every positional / mapping view must still show it in
    slot 1 as the reference returns it (modeling_dreamllm.py:1500-1509): keys / items / to_tuple / out[1] / dict(out) /
    HF Trainer.prediction_step's `outputs.items()`.  Reading only `.loss` never materialises them."""
    from dreamllm_amd.modeling_dreamllm import CausalLMOutputWithPast
    calls = []

    def thunk():
        calls.append(1)
        return torch.full((2, 3), 7.0)

    def make():
        return CausalLMOutputWithPast(loss=torch.tensor(1.0), hidden_states=(torch.zeros(1),),
                                      additional_log_info={"lm_loss": 1.0}).set_lazy_logits(thunk)

    o = make()
    assert float(o.loss) == 1.0 and o.additional_log_info["lm_loss"] == 1.0 and not calls
    assert list(o.keys()) == ["loss", "logits", "hidden_states", "additional_log_info"] and len(calls) == 1
    o = make()
    assert o[1].shape == (2, 3) and float(o[1][0, 0]) == 7.0 and o[0] is o.loss
    o = make()
    assert [k for k, _ in o.items()] == ["loss", "logits", "hidden_states", "additional_log_info"] and len(o.to_tuple()) == 4
    o = make()
    assert "logits" in o and o["logits"].shape == (2, 3) and len(o) == 4
    o = make()
    assert list(dict(o)) == ["loss", "logits", "hidden_states", "additional_log_info"]
    n = len(calls)
    assert o.logits is o["logits"] and len(calls) == n      # materialised once
    eager = CausalLMOutputWithPast(loss=torch.tensor(1.0), logits=torch.zeros(1), past_key_values=((1,),))
    assert list(eager.keys()) == ["loss", "logits", "past_key_values"]


def test_lazy_logits_stay_lazy_for_the_training_plumbing():
    """the reference trainer reads `outputs["loss"]` / `outputs["additional_log_info"]` (omni/train/trainer.py:1083,1092)
    and accelerate's `convert_outputs_to_fp32` / DDP walk `items()` / `values()` every step: none of that may run the [B, S, V]
    lm_head GEMM.  The training forward marks its output `bulk_views=False`; explicit requests still materialise."""
    from dreamllm_amd.modeling_dreamllm import CausalLMOutputWithPast
    calls = []

    def thunk():
        calls.append(1)
        return torch.full((2, 3), 7.0)

    def make(bulk):
        return CausalLMOutputWithPast(loss=torch.tensor(1.0), hidden_states=(torch.zeros(1),),
                                      additional_log_info={"lm_loss": 1.0}).set_lazy_logits(thunk, bulk_views=bulk)

    o = make(False)
    assert float(o["loss"]) == 1.0 and o["additional_log_info"]["lm_loss"] == 1.0 and not calls
    assert [k for k, _ in o.items()] == ["loss", "hidden_states", "additional_log_info"] and len(o) == 3 and not calls
    assert len(list(o.values())) == 3 and list(o) == list(o.keys()) and "loss" in o and not calls
    try:   # accelerate.utils.operations.convert_to_fp32 is what `convert_outputs_to_fp32` applies to a training forward's output
        from accelerate.utils.operations import convert_to_fp32
        conv = convert_to_fp32(o)
        assert float(conv["loss"]) == 1.0 and not calls
    except ImportError:
        pass
    assert o["logits"].shape == (2, 3) and len(calls) == 1                 # an explicit request still gets them, in slot 1
    assert list(o.keys()) == ["loss", "logits", "hidden_states", "additional_log_info"]
    o = make(False)
    assert o[1].shape == (2, 3) and len(o.to_tuple()) == 4 and len(calls) == 2
    o = make(False)
    assert "logits" in o and o.logits.shape == (2, 3) and len(calls) == 3
    o = make(True)   # eval / no_grad outputs: the bulk views materialise (Trainer.prediction_step walks items() for the logits)
    assert [k for k, _ in o.items()] == ["loss", "logits", "hidden_states", "additional_log_info"] and len(calls) == 4


def test_rope_scaling_variants_match_the_executed_reference(golden):
    """`rope_scaling = {"type": "linear" | "dynamic", "factor": f}` (modeling_dreamllm.py:131-173, selected at :279-304): the cos / sin
    tables of `RotaryEmbedding.forward` against the executed reference classes, in a call order that includes sequences beyond
    `max_position_embeddings` and shorter ones afterwards (the reference keeps serving the table it rebuilt for the longest
    sequence seen; dynamic NTK therefore depends on the call history, and so does this implementation)."""
    from dreamllm_amd.modeling_dreamllm import RotaryEmbedding
    g = golden("rope_scaling.pt")
    x = torch.zeros(1, 1, 1, g["dim"])
    for c in g["cases"]:
        rot = RotaryEmbedding(g["dim"], g["max_position_embeddings"], base=g["base"], scaling_factor=c["factor"],
                              scaling_type=None if c["type"] == "none" else c["type"])
        for st in c["steps"]:
            cos, sin = rot(x, seq_len=st["seq_len"])
            assert cos.shape == st["cos"].shape
            assert (cos - st["cos"]).abs().max() < 2e-5 and (sin - st["sin"]).abs().max() < 2e-5, (c["type"], c["factor"], st["seq_len"])


def test_unet_time_bias_layout_and_mask_span_errors():
    """Host-side pieces of round 3 that need no kernel: (i) the flat time-bias buffer of `HipUNet2DConditionModel.precompute_time_bias`
    -- one [N, cout] slice per ResnetBlock2D in execution order, contiguous, 22 blocks for the SD-2.1 layout; (ii) `_mask_to_spans`:
    left / right padding become spans, a mask with holes raises `_MaskHasHoles` (the model then compacts), 4-D masks are rejected."""
    import pytest
    from dreamllm_amd.modeling_dreamllm import _mask_to_spans, _MaskHasHoles
    from dreamllm_amd.unet import SD21_BASE, HipUNet2DConditionModel, load_unet_config
    with torch.device("meta"):
        unet = HipUNet2DConditionModel(load_unet_config(dict(SD21_BASE)))
    res = unet._resnets()
    assert len(res) == 22 and res[0] is unet.down_blocks[0].resnets[0] and res[-1] is unet.up_blocks[-1].resnets[-1]
    offs, total = unet.time_bias_layout(2)
    assert offs[0] == (0, 320) and total == 2 * sum(c for _, c in offs)
    for (o0, c0), (o1, _) in zip(offs, offs[1:]):
        assert o1 == o0 + 2 * c0
    assert sorted({c for _, c in offs}) == [320, 640, 1280]
    am = torch.ones(3, 10, dtype=torch.long)
    assert _mask_to_spans(am) == (None, None)
    am[1, 7:] = 0
    start, lens = _mask_to_spans(am)
    assert start is None and lens.tolist() == [10, 7, 10]
    am2 = torch.ones(2, 10, dtype=torch.long)
    am2[0, :3] = 0
    start, lens = _mask_to_spans(am2)
    assert start.tolist() == [3, 0] and lens.tolist() == [7, 10]
    am2[1, 4] = 0
    with pytest.raises(_MaskHasHoles):
        _mask_to_spans(am2)
    assert issubclass(_MaskHasHoles, ValueError)
    with pytest.raises(ValueError):
        _mask_to_spans(torch.zeros(2, 1, 4, 4))      # 4-D, not causal: not representable on the flash path


def test_4d_causal_mask_reduces_to_the_same_spans_as_its_2d_mask():
    """The reference's eager attention receives `_prepare_4d_causal_attention_mask(mask_2d, ...)` (modeling_dreamllm.py:35,965-967).
    A caller that hands that 4-D mask to the HIP decoder gets exactly the spans of the 2-D mask it was built from -- right padding,
    left padding, with a KV cache, boolean form -- and any 4-D pattern that is not causal + key padding raises."""
    import warnings
    import pytest
    from dreamllm_amd.modeling_dreamllm import _mask4d_to_2d, _mask_to_spans
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from transformers.modeling_attn_mask_utils import _prepare_4d_causal_attention_mask as prep

        def same(a, b):
            return all((x is None and y is None) or (x is not None and y is not None and torch.equal(x, y)) for x, y in zip(a, b))

        B, S = 3, 10
        x = torch.zeros(B, S, 8)
        right = torch.ones(B, S, dtype=torch.long)
        right[1, 7:] = 0
        left = torch.ones(B, S, dtype=torch.long)
        left[0, :3] = 0
        left[2, :1] = 0
        for m2 in (right, left):
            m4 = prep(m2, (B, S), x, 0)
            assert m4.shape == (B, 1, S, S)
            assert torch.equal(_mask4d_to_2d(m4), m2)
            assert same(_mask_to_spans(m4), _mask_to_spans(m2))
            assert same(_mask_to_spans(m4 == 0), _mask_to_spans(m2))          # boolean "attend" form
            assert same(_mask_to_spans(m4.to(torch.bfloat16)), _mask_to_spans(m2))
        past = 6
        cache = torch.ones(B, S + past, dtype=torch.long)
        cache[2, :4] = 0
        m4 = prep(cache, (B, S), x, past)
        assert m4.shape == (B, 1, S, S + past)
        assert same(_mask_to_spans(m4), _mask_to_spans(cache, q_len=S))
    bidir = torch.zeros(B, 1, S, S)                                           # everything attends: not causal
    with pytest.raises(ValueError):
        _mask_to_spans(bidir)
    window = prep(right, (B, S), x, 0).clone()
    window[:, :, 5, 0] = torch.finfo(torch.float32).min                       # a sliding-window-like hole in a valid row
    with pytest.raises(ValueError):
        _mask_to_spans(window)
    with pytest.raises(ValueError):
        _mask_to_spans(torch.zeros(B, 2, S, S))                               # per-head masks
