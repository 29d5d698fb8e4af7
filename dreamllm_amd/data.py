"""Data bridge between the reference's dataset objects and the HIP decoder.

`DreamLLMDataset.__getitem__` (omni/data/builders/builder_dreamllm.py:232-455) yields per-sample dicts
`{input_ids, attention_mask, labels, images, images_dm[, add_time_ids]}`; the reference's collators
(`DataCollatorForDreamLLMDataset` :466-482, `DataCollatorForDreamLLMSDXLDataset` :485-505) right-pad and concatenate them.
The collators here do exactly that and ADD what the HIP model otherwise has to derive from the batch on the device, with a
device->host sync each (`torch.where` / `nonzero` over `input_ids`, the mask reduction):

  * `seqlens`      int32 [B]   valid tokens per (right-padded) row  -> the flash kernels' span, no `_upad_input`;
  * `dream_index`  int64 [N_dm * n_dream]   flat rows (into [B*S]) of the dream-query slots after each <dream_start>;
  * `image_index`  int64 [N_img * n_patch]  flat rows of the image-patch slots after each <im_start>;

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
