"""Data bridge between the reference's dataset objects and the HIP decoder.

`DreamLLMDataset.__getitem__` (omni/data/builders/builder_dreamllm.py:232-455) yields per-sample dicts
`{input_ids, attention_mask, labels, images, images_dm[, add_time_ids]}`; the reference's collators
(`DataCollatorForDreamLLMDataset` :466-482, `DataCollatorForDreamLLMSDXLDataset` :485-505) right-pad and concatenate them.
The collators here do exactly that and ADD what the HIP model otherwise has to derive from the batch on the device, with a
device->host sync each (`torch.where` / `nonzero` over `input_ids`, the mask reduction):

  * `seqlens`      int32 [B]   valid tokens per (right-padded) row  -> the flash kernels' span, no `_upad_input`;
  * `dream_index`  int64 [N_dm * n_dream]   flat rows (into [B*S]) of the dream-query slots after each <dream_start>;
  * `image_index`  int64 [N_img * n_patch]  flat rows of the image-patch slots after each <im_start>;
  * `loss_index`   int64 [n_valid]          flat rows whose shifted label is not -100 (the rows the LM loss is computed on);

in the (batch, position) order in which the reference's Python loops visit the slots (modeling_dreamllm.py:1085-1098,
1110-1139), so `DreamLLMForCausalMLM.forward(**batch)` runs sync-free.  Everything is computed on the CPU inside the DataLoader
worker (the reference uses 8 of them, stage2/base.py:89); the model's semantics do not change: dropping the three keys gives the
same result, bit for bit (tests/test_model_gpu.py::test_collator_indices_match_the_model_path).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

IGNORE_INDEX = -100  # omni/constants.py:48


def batch_dict(list_dict: list[dict]) -> dict:
    """builder_dreamllm.py:457-462."""
    return {key: [d[key] for d in list_dict] for key in list_dict[0].keys()}


def slot_indices(input_ids: torch.Tensor, start_id: int, length: int, max_slots: int | None = None):
    """Flat row indices of the `length` positions following every `start_id`, (batch, position) order; CPU or device."""
    B, S = input_ids.shape
    starts = torch.nonzero(input_ids.reshape(-1) == start_id, as_tuple=False).flatten()
    if max_slots is not None:
        starts = starts[:max_slots]
    if starts.numel() > 0 and int(((starts % S) + length).max()) >= S:
        raise ValueError("a multimodal slot runs past the end of its sequence (truncated sample?)")
    idx = starts[:, None] + 1 + torch.arange(length, device=input_ids.device)[None]
    return idx.reshape(-1), int(starts.numel())


def _cat_or_none(items):
    items = [x for x in items if x is not None]
    return torch.cat(items, 0) if len(items) > 0 else None


@dataclass
class DataCollatorForDreamLLMDataset:
    """builder_dreamllm.py:465-482 + the splice indices.  `dream_start_id` / `image_start_id` / `n_dream` / `n_patch` come from
    the model (`from_model`)."""

    tokenizer: object
    dream_start_id: int | None = None
    image_start_id: int | None = None
    n_dream: int = 64
    n_patch: int = 256
    with_time_ids: bool = False

    @classmethod
    def from_model(cls, tokenizer, model, **kw):
        sp = model.config.special_tokens2ids_dict["additional_special_tokens"]
        core = model.get_decoder() if hasattr(model, "get_decoder") else model
        n_dream = core.dream_embedding.embed_len if hasattr(core, "dream_embedding") else 64
        n_patch = core.clip_vision_embedding.embed_len if hasattr(core, "clip_vision_embedding") else 256
        return cls(tokenizer, dream_start_id=sp["<dream_start>"], image_start_id=sp["<im_start>"], n_dream=n_dream,
                   n_patch=n_patch, **kw)

    def __call__(self, examples: list[dict]) -> dict:
        ex = batch_dict(examples)
        pad = torch.nn.utils.rnn.pad_sequence
        ex["input_ids"] = pad(ex["input_ids"], batch_first=True, padding_value=self.tokenizer.pad_token_id)
        ex["attention_mask"] = pad(ex["attention_mask"], batch_first=True, padding_value=0)
        ex["labels"] = pad(ex["labels"], batch_first=True, padding_value=IGNORE_INDEX)
        ex["images"] = _cat_or_none(ex["images"])
        ex["images_dm"] = _cat_or_none(ex["images_dm"])
        if self.with_time_ids or "add_time_ids" in ex:
            ex["add_time_ids"] = _cat_or_none(ex.get("add_time_ids", []))
        # ---- the bridge: spans and splice indices, computed here so that the model step needs no device->host sync
        mask = ex["attention_mask"] != 0
        lens = mask.sum(-1)
        if bool((mask == (torch.arange(mask.shape[1])[None] < lens[:, None])).all()):
            ex["seqlens"] = lens.to(torch.int32)
        # rows of the flattened [B*S] batch whose SHIFTED label (labels[b, s+1]) carries a loss: the fused lm_head + CE runs on these only
        lab = ex["labels"]
        shift = torch.cat([lab[:, 1:], lab.new_full((lab.shape[0], 1), IGNORE_INDEX)], dim=1).reshape(-1)
        ex["loss_index"] = torch.nonzero(shift != IGNORE_INDEX, as_tuple=False).flatten()
        ids = ex["input_ids"]
        if self.dream_start_id is not None and ex["images_dm"] is not None:
            ex["dream_index"], n = slot_indices(ids, self.dream_start_id, self.n_dream)
            if n != ex["images_dm"].shape[0]:
                raise ValueError(f"{n} <dream_start> tokens but {ex['images_dm'].shape[0]} dream images in the batch")
        if self.image_start_id is not None and ex["images"] is not None:
            ex["image_index"], _ = slot_indices(ids, self.image_start_id, self.n_patch, max_slots=ex["images"].shape[0])
        return ex


@dataclass
class DataCollatorForDreamLLMSDXLDataset(DataCollatorForDreamLLMDataset):
    """builder_dreamllm.py:485-505: additionally concatenates the SDXL micro-conditioning rows (`add_time_ids`)."""

    with_time_ids: bool = True


# ---------------------------------------------------------------------------------------------------------------------------
# webdataset interleaved-document format -> DreamLLM training example (the last piece of the data bridge)
#
#   tar sample  {"__key__": ..., "json": {"text_list": [...], "image_info": [{"image_name", "matched_text_index"[, "matched_sim"]}]},
#                "<name>.jpg": image, ...}
#     --interleaved_to_dict-->            third_party/webdataset/webdataset/filters.py:413-445
#     --interleaved_sample_to_lists-->    omni/data/datasets/unified_it_interleaved_webdataset.py:49-75 (`to_return_type`)
#     --InterleavedExampleBuilder-->      omni/data/builders/builder_dreamllm.py:232-288,438-455 (`DatasetType.InterleavedImageText`)
#   -> {input_ids, attention_mask, labels, images, images_dm}  -> DataCollatorForDreamLLMDataset (above) -> model(**batch)
#
# Reading tar shards, shuffling and image decoding stay with webdataset / PIL (host I/O, out of scope); what is restated here is
# the part that decides WHICH tokens and images the model sees, so that the accelerated path consumes the reference's data format
# unchanged.  Pinned by tests/test_data_bridge_cpu.py against the executed reference (tests/golden/data_bridge.pt).
def interleaved_to_dict(sample: dict, patterns: str = "json;jpg;png;jpeg"):
    """One sample through `wds.interleaved_to_dict("json;jpg;png;jpeg")` (filters.py:413-445): keep every non-dunder key whose
    name ends with one of the extensions; a key with any other extension makes the whole sample invalid (the reference raises
    inside the pipeline and `warn_and_continue` drops it) -> returns None."""
    exts = patterns.split(";")
    out = {}
    for k, v in sample.items():
        if k.startswith("__"):
            continue
        if any(k.endswith(e) for e in exts) and v is not None:
            out[k] = v
        else:
            return None
    return out


def has_text_and_images(raw_json: bytes) -> bool:
    """`filter_no_text_or_no_image` (unified_it_interleaved_webdataset.py:13-14) on the still-encoded json member."""
    return (b"text_list" in raw_json) and (b"image_info" in raw_json)


def interleaved_sample_to_lists(sample: dict):
    """`UnifiedInterleavedITWebdataset.to_return_type` (:49-75): images are looked up as `<stem>.jpg`, then (index, image, sim)
    triples are sorted by the index of the sentence they follow (ties keep document order).
    -> (text_list, image_list, matched_text_index, matched_sim)"""
    meta = sample["json"]
    rows = []
    for seq, info in enumerate(meta["image_info"]):
        name = info["image_name"].split(".")[0] + ".jpg"  # images are restored as jpg (:55)
        rows.append((info["matched_text_index"], seq, sample[name], info.get("matched_sim")))
    rows.sort(key=lambda r: (r[0], r[1]))
    return (meta["text_list"], [r[2] for r in rows], [r[0] for r in rows], [r[3] for r in rows])


def merge_text_list(text_list, matched_text_index):
    """`DreamLLMDataset._merge_text_list` (builder_dreamllm.py:100-108): sentences up to and including each matched index are
    joined into the text piece that precedes that image; a remainder becomes the trailing piece."""
    out, prev = [], 0
    for index in matched_text_index:
        out.append(" ".join(text_list[prev: index + 1]))
        prev = index + 1
    if prev != len(text_list):
        out.append(" ".join(text_list[prev:]))
    return out


@dataclass
class InterleavedExampleBuilder:
    """The `DatasetType.InterleavedImageText` branch of `DreamLLMDataset.__getitem__` (builder_dreamllm.py:232-288,438-455).

    tokenizer: callable `tok(text).input_ids` (with BOS first), `bos_token_id`, `eos_token_id`, `model_max_length`;
    `special`: ids of <im_patch>, <im_start>, <im_end>, <dream_start>, <dream_end> (`from_model` reads them from the model config);
    `clip_processor(image)` / `dream_processor(image)` -> tensors [3,224,224] / [3,512,512] (a failing processor drops the image
    AND its slot tokens, as the reference's try/except does)."""

    tokenizer: object
    special: dict
    clip_processor: object
    dream_processor: object
    n_patch: int = 256
    n_dream: int = 64
    comprehension_only: bool = False
    creation_only: bool = False
    use_image_start_and_end: bool = True
    use_dream_start_and_end: bool = True

    @classmethod
    def from_model(cls, tokenizer, model, clip_processor, dream_processor, **kw):
        sp = model.config.special_tokens2ids_dict["additional_special_tokens"]
        core = model.get_decoder() if hasattr(model, "get_decoder") else model
        special = {k: sp[f"<{k}>"] for k in ("im_patch", "im_start", "im_end", "dream_start", "dream_end")}
        return cls(tokenizer, special, clip_processor, dream_processor, n_patch=core.clip_vision_embedding.embed_len,
                   n_dream=core.dream_embedding.embed_len, **kw)

    def _image_ids(self):   # builder_dreamllm.py:110-117
        s = self.special
        ids = [s["im_patch"]] * self.n_patch
        return [s["im_start"]] + ids + [s["im_end"]] if self.use_image_start_and_end else ids

    def _dream_ids(self):   # :119-128 (the dream slots reuse <im_patch>; the SDXL variant has its own <dream_patch>)
        s = self.special
        ids = [s.get("dream_patch", s["im_patch"])] * self.n_dream
        return [s["dream_start"]] + ids + [s["dream_end"]] if self.use_dream_start_and_end else ids

    def __call__(self, text_list, image_list, matched_text_index, matched_sim=None) -> dict:
        assert not (self.comprehension_only and self.creation_only)
        tok, s = self.tokenizer, self.special
        pieces = merge_text_list([t.strip() for t in text_list], matched_text_index)
        max_len = tok.model_max_length
        input_ids, images, images_dm = [], [], []
        for idx, text in enumerate(pieces):
            cur = tok(text).input_ids[1:]                                   # drop BOS (:252)
            if len(input_ids) + len(cur) + 2 > max_len:                      # +2: BOS and EOS are added at the end (:253)
                break
            input_ids = input_ids + cur
            if idx < len(image_list):
                if self.comprehension_only:
                    append = self._image_ids()
                elif self.creation_only:
                    append = self._dream_ids()
                else:
                    append = self._dream_ids() + self._image_ids()          # dream slot first, then the image itself (:264)
                if len(input_ids) + len(append) + 2 > max_len:
                    break
                try:
                    if not self.creation_only:
                        images.append(self.clip_processor(image_list[idx]))
                    if not self.comprehension_only:
                        images_dm.append(self.dream_processor(image_list[idx]))
                except Exception:  # noqa: BLE001 -- the reference's bare `except:` (:281): corrupted image -> no slot
                    append = []
                input_ids = input_ids + append
        input_ids = [tok.bos_token_id] + input_ids + [tok.eos_token_id]
        ignore = {s["im_patch"], s["im_start"], s["im_end"], s.get("dream_patch", s["im_patch"]), s["dream_end"]}
        labels = [IGNORE_INDEX if x in ignore else x for x in input_ids]    # only <dream_start> is learned (:285-288)
        return {
            "input_ids": torch.tensor(input_ids),
            "attention_mask": torch.tensor([1] * len(input_ids)),
            "labels": torch.tensor(labels),
            "images": torch.stack(images, 0) if len(images) > 0 else None,
            "images_dm": torch.stack(images_dm, 0) if len(images_dm) > 0 else None,
        }

    def from_wds_sample(self, sample: dict):
        """tar sample (json already decoded to a dict, images decoded) -> training example, or None when the pipeline's
        filters would have dropped it."""
        d = interleaved_to_dict(sample)
        if d is None or "text_list" not in d.get("json", {}) or "image_info" not in d.get("json", {}):
            return None
        if len(d["json"]["image_info"]) == 0:       # `zip(*sorted_pairs)` of nothing raises in the reference (:68) -> dropped
            return None
        try:
            lists = interleaved_sample_to_lists(d)
        except KeyError:                            # an image named in the json is not in the tar sample (:56) -> dropped
            return None
        return self(*lists)
