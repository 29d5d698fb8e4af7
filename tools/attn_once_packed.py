"""LLM attention forward/backward on the PACKED layout the decoder layer uses (q, k, v = views of one [B,S,3,H,D] buffer, dq/dk/dv
written into one packed dQKV buffer) -- for rocprofv3 kernel traces: python tools/attn_once_packed.py [packed|plain]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

BF = torch.bfloat16
B, S, H, D = 16, 2048, 32, 128
packed = (sys.argv[1] if len(sys.argv) > 1 else "packed") == "packed"
if packed:
    qkv = torch.randn(B, S, 3, H, D, device="cuda").to(BF)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]
else:
    q, k, v = (torch.randn(B, S, H, D, device="cuda").to(BF) for _ in range(3))
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
do = torch.randn(B, S, H, D, device="cuda").to(BF)
for _ in range(5):
    o, lse = ops.attn_fwd(q, k, v, True)
    ops.attn_bwd(do, q, k, v, o, lse, True, dq=dq, dk=dk, dv=dv)
torch.cuda.synchronize()
print("done")
