#!/usr/bin/env python
"""The UNet's self-attention shapes at batch 2 (the denoise loop at B_img = 1): 4-wave (128-query blocks) vs 8-wave (256-query blocks)
forward kernels -- at this batch the 8-wave kernel launches 160 / 80 work-groups on 256 CUs.   python tools/attn_small_batch.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

BF = torch.bfloat16


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (2, 16):
    for name, S, H in (("64x64 C320", 4096, 5), ("32x32 C640", 1024, 10), ("16x16 C1280", 256, 20)):
        q, k, v = (torch.randn(B, S, H, 64, device="cuda").to(BF) for _ in range(3))
        res = {}
        for var in (0, 1, 2):
            ops.ATTN_VARIANT = var
            res[var] = statistics.median([timed(lambda: ops.attn_fwd(q, k, v, False, need_lse=False)) for _ in range(5)])
        ops.ATTN_VARIANT = 0
        fl = 4.0 * B * H * S * S * 64
        print(f"B={B:2d} {name:12s} S={S:4d} H={H:2d}: auto {res[0]:6.1f} us | 4-wave {res[1]:6.1f} us ({fl / res[1] / 1e6:5.0f} TF) | "
              f"8-wave {res[2]:6.1f} us ({fl / res[2] / 1e6:5.0f} TF)", flush=True)
