#!/usr/bin/env python
"""dQ of the ping-pong kernel (ATTN_VARIANT 3, csrc/attn_bwd_pp.hip) against the 8-wave kernel (variant 2) and an fp32 reference on
small shapes, then the backward timings of both at the bench shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

BF = torch.bfloat16


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()


for (B, H, Hkv, Sq, Sk, D, causal) in [(1, 2, 2, 777, 777, 128, True), (1, 5, 5, 600, 600, 128, False), (2, 4, 2, 2048, 2048, 128, True),
                                       (1, 2, 2, 40, 600, 128, True), (1, 2, 2, 512, 512, 128, True)]:
    torch.manual_seed(0)
    q, do = (torch.randn(B, Sq, H, D, device="cuda").to(BF) for _ in range(2))
    k, v = (torch.randn(B, Sk, Hkv, D, device="cuda").to(BF) for _ in range(2))
    ops.ATTN_VARIANT = 0
    o, lse = ops.attn_fwd(q, k, v, causal)
    out = {}
    for var in (2, 3):
        ops.ATTN_VARIANT = var
        out[var] = ops.attn_bwd(do, q, k, v, o, lse, causal)
    torch.cuda.synchronize()
    print(f"B{B} H{H}/{Hkv} Sq{Sq} Sk{Sk} causal={causal}: dq v3 vs v2 {rel(out[3][0], out[2][0]):.3e}  dk {rel(out[3][1], out[2][1]):.3e}  "
          f"dv {rel(out[3][2], out[2][2]):.3e}  finite {bool(torch.isfinite(out[3][0].float()).all())}", flush=True)
ops.ATTN_VARIANT = 0
