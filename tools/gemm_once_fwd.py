"""Three launches of the dominant forward GEMM (packed gate|up [32768 x 22016 x 4096]) for a rocprofv3 --pmc pass; DREAMLLM_GEMM_PERSIST
selects the walk.  (tools/gemm_once.py runs fwd + dgrad + wgrad.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

x = torch.randn(32768, 4096, device="cuda").to(torch.bfloat16)
w = (torch.randn(22016, 4096, device="cuda") * 0.02).to(torch.bfloat16)
for _ in range(3):
    ops.linear_fwd(x, w)
torch.cuda.synchronize()
print("done")
