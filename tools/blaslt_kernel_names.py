"""Which hipBLASLt kernels does torch.matmul pick for the step's three big GEMM layouts?  (tools only: a yard-stick, never the product)
Kernel names (= Tensile solution names) and durations through torch.profiler."""
import torch
from torch.profiler import ProfilerActivity, profile
BF = torch.bfloat16
M, N, K = 32768, 22016, 4096
a = torch.zeros(M, K, device="cuda", dtype=BF)
w = torch.zeros(N, K, device="cuda", dtype=BF)
g = torch.zeros(M, N, device="cuda", dtype=BF)
for _ in range(2):
    y = a @ w.t(); dx = g @ w; dw = g.t() @ a
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        y = a @ w.t()        # forward  (NT)
        dx = g @ w           # dgrad    (NN)
        dw = g.t() @ a       # wgrad    (TN)
    torch.cuda.synchronize()
for e in prof.key_averages():
    if e.device_time_total > 0 and e.device_type.name != "CPU":
        print(f"{e.device_time_total / max(e.count, 1):10.1f} us x{e.count}  {e.key}")
