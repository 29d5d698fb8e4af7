"""A handful of launches of the dominant GEMM shapes (for rocprofv3 --pmc passes: FETCH_SIZE / WRITE_SIZE per launch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

BF = torch.bfloat16
T = 32768
x = torch.randn(T, 4096, device="cuda").to(BF)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 22016   # packed gate|up projection of the decoder layer (round 2); 11008 = one of them
w = (torch.randn(N, 4096, device="cuda") * 0.02).to(BF)
dy = torch.randn(T, N, device="cuda").to(BF)
for _ in range(3):
    ops.linear_fwd(x, w)       # gate|up forward   [32768 x N x 4096]
    ops.linear_dgrad(dy, w)    # dgrad
    ops.linear_wgrad(dy, x)    # wgrad
torch.cuda.synchronize()
print("done")
