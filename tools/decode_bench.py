#!/usr/bin/env python
"""Batch-1 greedy decode of the 7B decoder alone (no plugins): tokens/s and the weight-stream rate; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel table of a token step.   python tools/decode_bench.py [new_tokens=64] [ctx=514]"""
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
sess = GreedyDecodeSession(model, 1, CTX + 2 * NEW + 16)
sess.prefill(ids)
sess.generate(4)
sess.prefill(ids)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    sess.prefill(ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sess.generate(NEW)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / NEW)
t = sorted(ts)[1]
wbytes = sum(p.numel() for n, p in model.named_parameters() if n.startswith("model.layers") or n.startswith("lm_head") or n == "model.norm.weight") * 2
print(f"decode: {1 / t:.1f} tok/s, {t * 1e3:.3f} ms/token, weight stream {wbytes / t / 1e9:.0f} GB/s ({wbytes / t / 8e12 * 100:.1f} % of 8 TB/s), runs {[round(x * 1e3, 3) for x in ts]}")
