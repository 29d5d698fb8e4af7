#!/usr/bin/env python
"""Batch-1 greedy decode of the 7B decoder alone (no plugins): tokens/s and the weight-stream rate; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel table of a token step.   python tools/decode_bench.py [new_tokens=64] [ctx=514] [ab]
`ab`: both session forms in one process, interleaved -- the split-KV partials merged inside the o projection's GEMV (round 6, 5 launches per
layer) against the separate combine launch (6 per layer)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd.decode import GreedyDecodeSession  # noqa: E402
from dreamllm_amd.factory import VICUNA_7B, build_dreamllm  # noqa: E402

NEW = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CTX = int(sys.argv[2]) if len(sys.argv) > 2 else 514
model = build_dreamllm(VICUNA_7B, device="cuda", with_clip=False, with_sd=False).eval()
ids = torch.randint(3, 32000, (1, CTX), device="cuda")
AB = len(sys.argv) > 3 and sys.argv[3] == "ab"
wbytes = sum(p.numel() for n, p in model.named_parameters() if n.startswith("model.layers") or n.startswith("lm_head") or n == "model.norm.weight") * 2
sessions = {"merged in o-proj" if m else "combine launch": GreedyDecodeSession(model, 1, CTX + 2 * NEW + 16, merge_in_oproj=m)
            for m in ((True, False) if AB else (True,))}
for sess in sessions.values():
    sess.prefill(ids)
    sess.generate(4)
ts = {k: [] for k in sessions}
for _ in range(3):
    for name, sess in sessions.items():
        sess.prefill(ids)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sess.generate(NEW)
        torch.cuda.synchronize()
        ts[name].append((time.perf_counter() - t0) / NEW)
for name, v in ts.items():
    t = sorted(v)[1]
    print(f"decode [{name}]: {1 / t:.1f} tok/s, {t * 1e3:.3f} ms/token, weight stream {wbytes / t / 1e9:.0f} GB/s ({wbytes / t / 8e12 * 100:.1f} % of 8 TB/s), runs {[round(x * 1e3, 3) for x in v]}")
