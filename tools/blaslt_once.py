"""A few torch.matmul launches of the packed gate|up forward shape (hipBLASLt's hand-written kernel) for a rocprofv3 --pmc pass: how much does
the yard-stick fetch through the fabric for the same problem?  (tools only; rocprofv3 needs several minutes to load hipBLASLt's code objects)"""
import torch
BF = torch.bfloat16
a = torch.randn(32768, 4096, device="cuda").to(BF)
w = torch.randn(22016, 4096, device="cuda").to(BF)
for _ in range(4):
    y = a @ w.t()
torch.cuda.synchronize()
