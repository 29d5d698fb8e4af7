"""TEST INFRASTRUCTURE -- golden vectors for the data bridge from EXECUTING the reference's own code:

  * `webdataset.filters._interleaved_to_dict`                      (third_party/webdataset/webdataset/filters.py:413-445)
  * `UnifiedInterleavedITWebdataset.to_return_type`                 (omni/data/datasets/unified_it_interleaved_webdataset.py:49-75)
  * `DreamLLMDataset.__getitem__`, InterleavedImageText branch      (omni/data/builders/builder_dreamllm.py:130-141,232-288,438-455)
  * `DataCollatorForDreamLLMDataset` / `...SDXLDataset.__call__`    (omni/data/builders/builder_dreamllm.py:465-505)

on synthetic tar samples (text sentences + tensor "images"), with a deterministic word-hash tokenizer and tensor image
processors that both sides share (`FakeTokenizer`, `clip_proc`, `dream_proc` below; tests import them from here).  Run in the
authoring container only (needs /root/reference):  python -m oracle.make_golden_data  -> tests/golden/data_bridge.pt
"""
from __future__ import annotations

import os
import sys
import types
import zlib

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "data_bridge.pt")

SPECIAL = {"<image>": 32001, "<im_patch>": 32002, "<im_start>": 32003, "<im_end>": 32004, "<dream>": 32005,
           "<dream_start>": 32006, "<dream_end>": 32007}


class _Enc:
    def __init__(self, ids):
        self.input_ids = ids


class FakeTokenizer:
    """Whitespace words -> crc32-hashed ids in [3, 32000); BOS first (LLaMA convention), no EOS (`add_eos_token=False`)."""
    bos_token_id, eos_token_id, pad_token_id = 1, 2, 32000
    add_eos_token = False

    def __init__(self, model_max_length=2048):
        self.model_max_length = model_max_length

    def __call__(self, text, max_length=None, truncation=False):
        ids = [self.bos_token_id] + [3 + zlib.crc32(w.encode()) % 31997 for w in text.split()]
        if truncation and max_length is not None:
            ids = ids[:max_length]
        return _Enc(ids)

    def convert_tokens_to_ids(self, toks):
        return [SPECIAL[t] for t in toks]


class _ClipProc:
    """`CLIPImageProcessor.preprocess(image, return_tensors="pt")["pixel_values"][0]` stand-in on tensor images."""

    def preprocess(self, image, return_tensors="pt"):
        if image is None or (torch.is_tensor(image) and image.numel() == 1):
            raise ValueError("corrupted image")
        return {"pixel_values": [image.float().mean() + torch.arange(3 * 4 * 4, dtype=torch.float32).view(3, 4, 4)]}

    def __call__(self, image):
        return self.preprocess(image)["pixel_values"][0]


def dream_proc(image):
    if image is None or (torch.is_tensor(image) and image.numel() == 1):
        raise ValueError("corrupted image")
    return image.float()[:, :6, :6] * 2 - 1


clip_proc = _ClipProc()


def make_samples():
    """Synthetic tar samples: out-of-order matched indices, ties, a trailing text remainder, a missing `matched_sim`, a png-named
    image restored as jpg, a corrupted image, a document that overflows `model_max_length`, and three samples the reference's
    pipeline drops (foreign extension, image missing from the tar, empty image list)."""
    g = torch.Generator().manual_seed(0)
    words = "the quick brown fox jumps over a lazy dog while seven wizards quietly vex bold jam packed boxes".split()

    def sent(n, k):
        return " ".join(words[(k + i * 3) % len(words)] for i in range(n)) + ("  " if k % 2 else "")

    def img():
        return torch.rand(3, 8, 8, generator=g)

    S = []
    S.append({"__key__": "a", "json": {"text_list": [sent(5, 0), sent(7, 1), sent(4, 2), sent(6, 3)],
                                       "image_info": [{"image_name": "i1.jpg", "matched_text_index": 2, "matched_sim": 0.31},
                                                      {"image_name": "i0.png", "matched_text_index": 0}]},
              "i0.jpg": img(), "i1.jpg": img()})
    S.append({"__key__": "b", "json": {"text_list": [sent(3, 4), sent(9, 5), sent(2, 6)],
                                       "image_info": [{"image_name": "x.jpg", "matched_text_index": 1, "matched_sim": 0.2},
                                                      {"image_name": "y.jpg", "matched_text_index": 1, "matched_sim": 0.9},
                                                      {"image_name": "z.jpg", "matched_text_index": 2, "matched_sim": 0.5}]},
              "x.jpg": img(), "y.jpg": img(), "z.jpg": img()})
    S.append({"__key__": "c", "json": {"text_list": [sent(4, 7), sent(4, 8)],
                                       "image_info": [{"image_name": "bad.jpg", "matched_text_index": 0, "matched_sim": 0.4},
                                                      {"image_name": "ok.jpg", "matched_text_index": 1, "matched_sim": 0.4}]},
              "bad.jpg": torch.zeros(1), "ok.jpg": img()})                                           # corrupted first image
    S.append({"__key__": "d", "json": {"text_list": [sent(30, 9), sent(40, 10), sent(50, 11), sent(20, 12)],
                                       "image_info": [{"image_name": f"p{i}.jpg", "matched_text_index": i, "matched_sim": 0.3}
                                                      for i in range(4)]},
              **{f"p{i}.jpg": img() for i in range(4)}})                                                # overflows a short max length
    S.append({"__key__": "e", "json": {"text_list": [sent(3, 1)], "image_info": [{"image_name": "q.jpg", "matched_text_index": 0}]},
              "q.jpg": img(), "notes.txt": "x"})                                                        # dropped: foreign extension
    S.append({"__key__": "f", "json": {"text_list": [sent(3, 2)], "image_info": [{"image_name": "gone.jpg", "matched_text_index": 0}]}})
    S.append({"__key__": "g", "json": {"text_list": [sent(3, 3)], "image_info": []}})
    return S


MODES = [dict(name="joint", comprehension_only=False, creation_only=False, max_len=2048, n_patch=16, n_dream=8),
         dict(name="comprehension", comprehension_only=True, creation_only=False, max_len=2048, n_patch=16, n_dream=8),
         dict(name="creation", comprehension_only=False, creation_only=True, max_len=2048, n_patch=16, n_dream=8),
         dict(name="joint_short", comprehension_only=False, creation_only=False, max_len=96, n_patch=16, n_dream=8)]


def main():
    from oracle import ref_loader
    ref_loader._STUB_ROOTS.discard("webdataset")          # the vendored third_party/webdataset is the real thing
    ref_loader._STUB_ROOTS.update({"braceexpand", "isodate"})
    ref_loader.install()
    sys.path.insert(0, os.path.join(ref_loader.REFERENCE_ROOT, "third_party", "webdataset"))
    st = types.ModuleType("omni.data.constants")          # the dataset registry (imports every dataset class); only the name is needed
    st.DataManager = object
    sys.modules["omni.data.constants"] = st
    import importlib
    filters = importlib.import_module("webdataset.filters")
    builder = importlib.import_module("omni.data.builders.builder_dreamllm")
    uni = importlib.import_module("omni.data.datasets.unified_it_interleaved_webdataset")
    dtype_mod = importlib.import_module("omni.data.manager.dataset_type")

    samples = make_samples()
    swallow = lambda exn: True                             # `wds.warn_and_continue`: drop the sample, keep going
    fake_self = types.SimpleNamespace(dataset_type=dtype_mod.DatasetType.InterleavedImageText)
    returned = []                                         # per input sample: the reference's InterleavedImageTextReturnType or None
    for s in samples:
        got = list(filters._interleaved_to_dict([s], "json;jpg;png;jpeg", handler=swallow))
        if not got:
            returned.append(None)
            continue
        try:
            returned.append(uni.UnifiedInterleavedITWebdataset.to_return_type(fake_self, got[0]))
        except Exception:                                  # `wds.map(..., handler=warn_and_continue)`
            returned.append(None)
    out = dict(samples=samples, special=SPECIAL, modes=MODES, dropped=[r is None for r in returned],
               lists=[None if r is None else dict(text_list=r.text_list, image_list=r.image_list,
                                                  matched_text_index=r.matched_text_index, matched_sim=r.matched_sim) for r in returned],
               examples={}, collated={})

    class _Inner:
        def __init__(self, items):
            self.items = items

        def __getitem__(self, i):
            return self.items[i]

        def __len__(self):
            return len(self.items)

    kept = [r for r in returned if r is not None]
    for m in MODES:
        ds = builder.DreamLLMDataset.__new__(builder.DreamLLMDataset)     # bypass __init__ (DataManager, registry); set what __getitem__ reads
        ds.inner_dataset = _Inner(kept)
        ds.tokenizer = FakeTokenizer(m["max_len"])
        ds.clip_vision_embedding_processor = clip_proc
        ds.stable_diffusion_head_processor = dream_proc
        ds.clip_vision_embedding_len, ds.dream_embedding_len = m["n_patch"], m["n_dream"]
        ds.comprehension_only, ds.creation_only = m["comprehension_only"], m["creation_only"]
        ds.use_sdxl_head, ds.use_image_start_and_end, ds.use_dream_start_and_end, ds.conv_template = False, True, True, None
        exs = [ds[i] for i in range(len(kept))]
        out["examples"][m["name"]] = exs
        col = builder.DataCollatorForDreamLLMDataset(ds.tokenizer)(exs)
        out["collated"][m["name"]] = col
    # SDXL collator on ragged examples with micro-conditioning rows (one example without a dream image)
    exs = [dict(e) for e in out["examples"]["creation"]]
    for i, e in enumerate(exs):
        e["add_time_ids"] = None if e["images_dm"] is None else torch.tensor([[1024., 1024, 0, i, 1024, 1024]] * e["images_dm"].shape[0])
    out["sdxl_examples"] = exs
    out["sdxl_collated"] = builder.DataCollatorForDreamLLMSDXLDataset(FakeTokenizer())(exs)
    torch.save(out, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", "dropped:", out["dropped"])
    for m in MODES:
        print(m["name"], [int(e["input_ids"].numel()) for e in out["examples"][m["name"]]],
              [None if e["images"] is None else tuple(e["images"].shape) for e in out["examples"][m["name"]]])


if __name__ == "__main__":
    main()
