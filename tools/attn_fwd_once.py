"""A few launches of ONE attention forward variant at the bench shape (for rocprofv3 --pmc passes): python tools/attn_fwd_once.py <variant>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

ops.ATTN_VARIANT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, S, H, D = 16, 2048, 32, 128
q, k, v = (torch.randn(B, S, H, D, device="cuda").to(torch.bfloat16) for _ in range(3))
for _ in range(3):
    ops.attn_fwd(q, k, v, True)
torch.cuda.synchronize()
print("done")
