"""CPU: the data bridge against the EXECUTED reference (tests/golden/data_bridge.pt, written by
oracle/make_golden_data.py from the reference's `_interleaved_to_dict`, `UnifiedInterleavedITWebdataset.to_return_type`,
`DreamLLMDataset.__getitem__` and the two collators).  Token ids and labels must be IDENTICAL (integer work: bit-exact), image
tensors identical, the collated batches identical on every key the reference emits; the extra keys this package adds
(`seqlens`, `dream_index`, `image_index`) must address exactly the slots the reference's model loops visit."""
import torch

from dreamllm_amd import data as D
from oracle.make_golden_data import FakeTokenizer, clip_proc, dream_proc


def _same(a, b):
    if a is None or b is None:
        return a is None and b is None
    return a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b)


def _builder(g, m):
    sp = g["special"]
    special = {k: sp[f"<{k}>"] for k in ("im_patch", "im_start", "im_end", "dream_start", "dream_end")}
    return D.InterleavedExampleBuilder(FakeTokenizer(m["max_len"]), special, clip_proc, dream_proc, n_patch=m["n_patch"],
                                       n_dream=m["n_dream"], comprehension_only=m["comprehension_only"],
                                       creation_only=m["creation_only"])


def test_webdataset_sample_filters_and_ordering_match_the_reference(golden):
    g = golden("data_bridge.pt")
    assert g["dropped"] == [False, False, False, False, True, True, True]
    b = _builder(g, g["modes"][0])
    for s, dropped, lists in zip(g["samples"], g["dropped"], g["lists"]):
        assert (b.from_wds_sample(s) is None) == dropped
        if dropped:
            continue
        text, images, index, sim = D.interleaved_sample_to_lists(D.interleaved_to_dict(s))
        assert text == lists["text_list"] and index == lists["matched_text_index"] and sim == lists["matched_sim"]
        assert len(images) == len(lists["image_list"]) and all(torch.equal(a, b_) for a, b_ in zip(images, lists["image_list"]))
    assert D.has_text_and_images(b'{"text_list": [], "image_info": []}') and not D.has_text_and_images(b'{"text_list": []}')


def test_interleaved_examples_match_the_executed_reference(golden):
    """joint / comprehension-only / creation-only modes and a short `model_max_length` that truncates documents at a text piece
    and at an image slot; a corrupted image drops its slot tokens."""
    g = golden("data_bridge.pt")
    kept = [s for s, d in zip(g["samples"], g["dropped"]) if not d]
    for m in g["modes"]:
        b = _builder(g, m)
        for s, ref in zip(kept, g["examples"][m["name"]]):
            ex = b.from_wds_sample(s)
            assert set(ex) == set(ref)
            for k in ref:
                assert _same(ex[k], ref[k]), (m["name"], s["__key__"], k)
    # the corrupted image of sample "c": one slot fewer than images named in its json
    ref_c = g["examples"]["joint"][2]
    assert ref_c["images"].shape[0] == 1 and int((ref_c["input_ids"] == g["special"]["<dream_start>"]).sum()) == 1


def test_collators_match_the_executed_reference(golden):
    g = golden("data_bridge.pt")
    sp = g["special"]
    for m in g["modes"]:
        col = D.DataCollatorForDreamLLMDataset(FakeTokenizer(m["max_len"]), dream_start_id=sp["<dream_start>"],
                                               image_start_id=sp["<im_start>"], n_dream=m["n_dream"], n_patch=m["n_patch"])
        exs = g["examples"][m["name"]]
        ours, ref = col([dict(e) for e in exs]), g["collated"][m["name"]]
        assert set(ref) <= set(ours)
        for k in ref:
            assert _same(ours[k], ref[k]), (m["name"], k)
        # the added keys: spans = mask row sums; slot indices = the rows after each start token, in (batch, position) order
        assert torch.equal(ours["seqlens"].long(), ref["attention_mask"].sum(-1))
        ids = ref["input_ids"]
        S = ids.shape[1]
        if ref["images_dm"] is not None:
            rows = [b * S + p + 1 + j for b in range(ids.shape[0]) for p in torch.where(ids[b] == sp["<dream_start>"])[0].tolist()
                    for j in range(m["n_dream"])]                                   # modeling_dreamllm.py:1085-1098
            assert ours["dream_index"].tolist() == rows and len(rows) == ref["images_dm"].shape[0] * m["n_dream"]
        if ref["images"] is not None:
            rows = [b * S + p + 1 + j for b in range(ids.shape[0]) for p in torch.where(ids[b] == sp["<im_start>"])[0].tolist()
                    for j in range(m["n_patch"])]                                   # modeling_dreamllm.py:1110-1139
            assert ours["image_index"].tolist() == rows
    ours = D.DataCollatorForDreamLLMSDXLDataset(FakeTokenizer())([dict(e) for e in g["sdxl_examples"]])
    for k, v in g["sdxl_collated"].items():
        assert _same(ours[k], v), k


def test_synthetic_batch_labels_follow_the_dataset_rule():
    """bench.py's synthetic documents mask exactly what `DreamLLMDataset` masks (builder_dreamllm.py:285-288): <im_patch>,
    <im_start>, <im_end>, <dream_end> (and padding); only <dream_start> of the special tokens is learned."""
    from dreamllm_amd.synthetic import make_interleaved_batch
    from dreamllm_amd.tokenization_dreamllm import default_special_tokens2ids
    add = default_special_tokens2ids(32000)["additional_special_tokens"]
    b = make_interleaved_batch(2, 512, 1, n_dream=8, n_patch=16, with_pixels=False, ragged=True)
    ids, lab = b["input_ids"], b["labels"]
    for t in ("<im_patch>", "<im_start>", "<im_end>", "<dream_end>"):
        assert bool((lab[ids == add[t]] == -100).all())
    assert bool((lab[ids == add["<dream_start>"]] == add["<dream_start>"]).all())
    assert bool((lab[b["attention_mask"] == 0] == -100).all())
