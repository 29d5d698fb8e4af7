"""A few launches of the packed gate|up forward GEMM [32768 x 22016 x 4096] per kernel variant (default 259 = 8-wave, 280 = four-wave; 261 = the MFMA 32x32x16 experiment), for rocprofv3 --pmc
passes that compare kernel families on one box: python tools/gemm_variant_once.py [variants]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

BF = torch.bfloat16
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "259,280").split(",")]
x = torch.randn(32768, 4096, device="cuda").to(BF)
w = (torch.randn(22016, 4096, device="cuda") * 0.02).to(BF)
for _ in range(3):
    for v in variants:
        with ops.gemm_variant(v):
            ops.linear_fwd(x, w)
torch.cuda.synchronize()
print("done")
