"""A few launches of the LLM attention forward/backward shape (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

BF = torch.bfloat16
B, S, H, D = 16, 2048, 32, 128
q, k, v, do = (torch.randn(B, S, H, D, device="cuda").to(BF) for _ in range(4))
for _ in range(3):
    o, lse = ops.attn_fwd(q, k, v, True)
    ops.attn_bwd(do, q, k, v, o, lse, True)
torch.cuda.synchronize()
print("done")
