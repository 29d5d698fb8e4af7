"""Synthetic interleaved text-image batches with the tensor layout of the reference's data pipeline.

Mirrors `DreamLLMDataset` joint mode + `DataCollatorForDreamLLMDataset` (omni/data/builders/builder_dreamllm.py:232-288,
438-482): per sample `<s>` + K x ( text ++ <dream_start> <im_patch>x64 <dream_end> ++ <im_start> <im_patch>x256 <im_end> )
++ text ++ `</s>`, right-padded with [PAD] (mask 0, labels -100); labels = ids with <im_patch>/<im_end>/<dream_end>
positions set to -100 (:285-288); `images` ~ N(0,1) [B*K,3,224,224] (CLIP-normalised), `images_dm` ~ U(-1,1)
[B*K,3,512,512].  Also emits the flat slot indices the model's sync-free splice path consumes.  SURVEY.md §8(d).
"""
from __future__ import annotations

import torch

from .tokenization_dreamllm import default_special_tokens2ids


def make_interleaved_batch(batch_size=16, seq_len=2048, images_per_sample=2, n_dream=64, n_patch=256, base_vocab=32000,
                           seed=1234, device="cpu", dtype=torch.bfloat16, ragged=False, image_size=224, dm_size=512,
                           with_pixels=True):
    g = torch.Generator().manual_seed(seed)
    sp = default_special_tokens2ids(base_vocab)
    add = sp["additional_special_tokens"]
    pad, bos, eos = sp["[PAD]"], sp["<s>"], sp["</s>"]
    K = images_per_sample
    slot = (1 + n_dream + 1) + (1 + n_patch + 1)
    ids = torch.full((batch_size, seq_len), pad, dtype=torch.long)
    mask = torch.zeros(batch_size, seq_len, dtype=torch.long)
    dream_pos, image_pos = [], []
    for b in range(batch_size):
        L = seq_len if not ragged else int(torch.randint(max(seq_len // 2, K * slot + 2 + K), seq_len + 1, (1,), generator=g))
        text_total = L - 2 - K * slot
        assert text_total >= K, "sequence too short for the requested number of image slots"
        cuts = torch.sort(torch.randint(1, text_total, (K,), generator=g))[0].tolist() if text_total > K else list(range(1, K + 1))
        lens = [cuts[0]] + [cuts[i] - cuts[i - 1] for i in range(1, K)] + [text_total - cuts[-1]]
        row = [bos]
        for k in range(K):
            row += torch.randint(3, base_vocab, (lens[k],), generator=g).tolist()
            dream_pos.append(b * seq_len + len(row))
            row += [add["<dream_start>"]] + [add["<im_patch>"]] * n_dream + [add["<dream_end>"]]
            image_pos.append(b * seq_len + len(row))
            row += [add["<im_start>"]] + [add["<im_patch>"]] * n_patch + [add["<im_end>"]]
        row += torch.randint(3, base_vocab, (lens[K],), generator=g).tolist() + [eos]
        assert len(row) == L, (len(row), L)
        ids[b, :L] = torch.tensor(row)
        mask[b, :L] = 1
    labels = ids.clone()
    for t in ("<im_patch>", "<im_start>", "<im_end>", "<dream_end>"):   # builder_dreamllm.py:285-288: only <dream_start> is learned
        labels[ids == add[t]] = -100
    labels[mask == 0] = -100
    dpos, ipos = torch.tensor(dream_pos), torch.tensor(image_pos)
    shift = torch.cat([labels[:, 1:], labels.new_full((batch_size, 1), -100)], dim=1).reshape(-1)
    batch = dict(
        input_ids=ids.to(device), attention_mask=mask.to(device), labels=labels.to(device),
        dream_index=(dpos[:, None] + 1 + torch.arange(n_dream)[None]).reshape(-1).to(device),
        image_index=(ipos[:, None] + 1 + torch.arange(n_patch)[None]).reshape(-1).to(device),
        # valid tokens per (right-padded) row, as data.DataCollatorForDreamLLMDataset emits them: the step then needs no
        # device->host sync to validate / reduce the mask
        seqlens=mask.sum(-1).to(torch.int32).to(device),
        # rows whose shifted label carries a loss (data.DataCollatorForDreamLLMDataset emits the same key)
        loss_index=torch.nonzero(shift != -100, as_tuple=False).flatten().to(device),
    )
    if with_pixels:
        n_img = batch_size * K
        batch["images"] = torch.randn(n_img, 3, image_size, image_size, generator=g).to(dtype).to(device)
        batch["images_dm"] = (torch.rand(n_img, 3, dm_size, dm_size, generator=g) * 2 - 1).to(dtype).to(device)
    return batch


def make_creation_batch(batch_size=16, seq_len=256, n_dream=196, base_vocab=32000, seed=1234, device="cpu", dtype=torch.bfloat16,
                        dm_size=1024, with_pixels=True):
    """Creation-only batch of DreamLLM-SDXL stage I (projects/dreamllm_sdxl/configs/stage1/base.py:60-65: `creation_only`):
    per sample `[bos] caption <dream_start> <dream_patch>*n_dream <dream_end> [eos]`, one target image per sample and its
    six SDXL micro-conditioning numbers (original_size + crop_top_left + target_size, as `SDXLDataProcessor` returns)."""
    from .modeling_dreamllm_sdxl import default_special_tokens2ids as sdxl_ids
    g = torch.Generator().manual_seed(seed)
    sp = sdxl_ids(base_vocab)
    add = sp["additional_special_tokens"]
    pad, bos, eos = sp["[PAD]"], sp["<s>"], sp["</s>"]
    n_text = seq_len - (n_dream + 2) - 2
    assert n_text >= 1, "sequence too short for the dream slot"
    ids = torch.full((batch_size, seq_len), pad, dtype=torch.long)
    dream_pos = []
    for b in range(batch_size):
        row = [bos] + torch.randint(3, base_vocab, (n_text,), generator=g).tolist()
        dream_pos.append(b * seq_len + len(row))
        row += [add["<dream_start>"]] + [add["<dream_patch>"]] * n_dream + [add["<dream_end>"]] + [eos]
        ids[b] = torch.tensor(row)
    mask = torch.ones_like(ids)
    labels = ids.clone()
    for t in ("<dream_patch>", "<dream_end>"):
        labels[ids == add[t]] = -100
    dpos = torch.tensor(dream_pos)
    batch = dict(input_ids=ids.to(device), attention_mask=mask.to(device), labels=labels.to(device),
                 dream_index=(dpos[:, None] + 1 + torch.arange(n_dream)[None]).reshape(-1).to(device))
    if with_pixels:
        batch["images_dm"] = (torch.rand(batch_size, 3, dm_size, dm_size, generator=g) * 2 - 1).to(dtype).to(device)
        batch["add_time_ids"] = torch.tensor([[dm_size, dm_size, 0, 0, dm_size, dm_size]] * batch_size, dtype=torch.float32,
                                             device=device)
    return batch
