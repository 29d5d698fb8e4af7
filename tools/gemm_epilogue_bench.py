import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops
BF = torch.bfloat16
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
M = 32768
for name, N, K in [("o fwd+res", 4096, 4096), ("down fwd+res", 4096, 11008)]:
    x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF); r = torch.randn(M, N, device="cuda").to(BF)
    a = t(lambda: ops.linear_fwd(x, w)); b = t(lambda: ops.linear_fwd(x, w, residual=r))
    print(f"{name}: plain {2*M*N*K/a/1e9:.0f} TF, residual {2*M*N*K/b/1e9:.0f} TF")
dy = torch.randn(M, 4096, device="cuda").to(BF); w = (torch.randn(4096, 4096, device="cuda") * 0.02).to(BF)
out = torch.zeros(M, 4096, device="cuda", dtype=BF)
a = t(lambda: ops.gemm(dy, w, M, 4096, 4096, 4096, 4096, 0, 1, out=out)); b = t(lambda: ops.gemm(dy, w, M, 4096, 4096, 4096, 4096, 0, 1, out=out, accumulate=True))
print(f"qkv dgrad: plain {2*M*4096*4096/a/1e9:.0f} TF, accumulate {2*M*4096*4096/b/1e9:.0f} TF")
