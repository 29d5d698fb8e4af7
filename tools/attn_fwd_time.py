#!/usr/bin/env python
"""Forward attention (automatic kernel choice = ping-pong) at the bench shapes: best-of timings and the difference to the 8-wave kernel's
output (a correctness tripwire for A/B builds).  DREAMLLM_HIP_LIB selects the library."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

BF = torch.bfloat16
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, B, S, H, D, causal in [("llm S2048 d128 causal", 16, 2048, 32, 128, True), ("llm S1536 d128 causal", 16, 1536, 32, 128, True),
                                 ("unet S4096 d64", 16, 4096, 5, 64, False), ("unet S1024 d64", 16, 1024, 10, 64, False),
                                 ("odd B3 H5 S777 d128 causal", 3, 777, 5, 128, True)]:
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, S, H, D, device="cuda").to(BF) for _ in range(3))
    ops.ATTN_VARIANT = 2
    oref, lref = ops.attn_fwd(q, k, v, causal)
    ops.ATTN_VARIANT = 3
    o, l = ops.attn_fwd(q, k, v, causal)
    err = (o.float() - oref.float()).abs().max().item()
    lerr = (l - lref).abs().max().item()
    best = 1e9
    for _ in range(4):
        e0.record()
        for _ in range(5):
            ops.attn_fwd(q, k, v, causal)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    fl = 4 * S * S * D * H * B / (2 if causal else 1)
    print(f"fwd {name:28s} {best:.3f} ms {fl / best / 1e9:6.0f} TF   max|o - o_8wave| {err:.2e}  max|lse diff| {lerr:.2e}", flush=True)
ops.ATTN_VARIANT = 0
