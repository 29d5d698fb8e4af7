#!/usr/bin/env python
"""A few direct launches of the ring-buffered 128 x 128 kernel on shapes of the denoising loop, for rocprofv3 --pmc passes
(MFMA busy share, LDS wait share, bank conflicts of its K loop: DESIGN.md §12).  Four-stage (variant 267) and two-stage (268) forms.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d out -- python tools/ring_once.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import _lib, ops  # noqa: E402

BF = torch.bfloat16


def linear(M, N, K, variant, reps=3):
    x = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.03).to(BF)
    b = torch.zeros(N, device="cuda", dtype=BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    for _ in range(reps):
        _lib.check("dllm_gemm_bf16_splitk", ops._p(x), ops._p(w), ops._p(out), ops._p(b), None, M, N, K, K, K, N, 0, 0, 0, 0, 0, 0, 1.0,
                   1, None, None, variant, ops._stream())


def conv(NB, H, C, CO, variant, reps=3):
    x = torch.randn(NB, H, H, C, device="cuda").to(BF)
    w = (torch.randn(CO, 9 * C, device="cuda") * 0.02).to(BF)
    b = torch.zeros(CO, device="cuda", dtype=BF)
    out = torch.empty(NB, H, H, CO, device="cuda", dtype=BF)
    for _ in range(reps):
        _lib.check("dllm_conv2d_nhwc_bf16_splitk", ops._p(x), ops._p(w), ops._p(out), ops._p(b), None, None, NB, H, H, C, H, H, CO, 3, 3,
                   1, 1, 0, 0, 0, 0, 1, None, None, variant, ops._stream())


for v in (267, 268):
    linear(65536, 320, 320, v)      # 64x64 level, UNet batch 16: 1536 blocks, 5 K tiles
    linear(4096, 10240, 1280, v)    # ff1 of the 16x16 level, batch 16: 2560 blocks, 20 K tiles
    linear(8192, 320, 320, v)       # 64x64 level, batch 2: 192 blocks
    conv(2, 64, 320, 320, v)        # [8192, 320, 2880]: 192 blocks, 45 K tiles
    conv(16, 64, 320, 320, v)       # [65536, 320, 2880]: 1536 blocks
torch.cuda.synchronize()
print("done")
