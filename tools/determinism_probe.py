"""Run-to-run determinism of the tiny training forward: the same batch and seed twice, compared stage by stage (hidden states of every
layer, CLIP features, logits, the two loss terms).  Prints the first stage whose bits differ.  python tools/determinism_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd.factory import TINY, TINY_CLIP, TINY_DIFFUSION, build_dreamllm  # noqa: E402
from dreamllm_amd.synthetic import make_interleaved_batch  # noqa: E402

DEV = "cuda"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
m = build_dreamllm(dict(TINY, num_hidden_layers=4), device=DEV, seed=0, clip=TINY_CLIP, diffusion=TINY_DIFFUSION, num_dream_queries=8).train()
batch = make_interleaved_batch(2, 256, 1, n_dream=8, n_patch=16, seed=11, device=DEV, image_size=56, dm_size=128)


def run():
    torch.manual_seed(5)
    out = m(**batch, return_dict=True, output_hidden_states=True)
    st = {f"hidden{i}": h.detach().clone() for i, h in enumerate(out.hidden_states)}
    st["loss"] = out.loss.detach().clone()
    for k, v in (out.additional_log_info or {}).items():
        if torch.is_tensor(v):
            st["log." + k] = v.detach().clone()
    with torch.no_grad():
        st["clip"] = m.model.clip_vision_embedding(batch["images"]).detach().clone()
    return st


ref = run()
print("stages:", list(ref.keys()))
bad = 0
for r in range(reps):
    cur = run()
    diffs = [k for k in ref if not torch.equal(ref[k], cur[k])]
    if diffs:
        bad += 1
        k = diffs[0]
        d = (ref[k].float() - cur[k].float()).abs()
        print(f"rep {r}: differs at {diffs}; first: {k} max abs {d.max().item():.3e} count {(d > 0).sum().item()} of {d.numel()}")
print(f"{bad} of {reps} repetitions differ")
