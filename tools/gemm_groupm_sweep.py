import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import _lib, ops
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
shapes = [("qkv fwd", 4096, 4096), ("gate fwd", 11008, 4096), ("down fwd", 4096, 11008)]
ten = {}
for name, N, K in shapes:
    ten[name] = (torch.randn(M, K, device="cuda").to(BF), (torch.randn(N, K, device="cuda") * 0.02).to(BF), torch.randn(M, N, device="cuda").to(BF))
for gm in (8, 4, 16, 2, 32, 8):
    ops.GEMM_VARIANT = gm << 16  # GROUP_M rides in bits 16-23 of the per-call variant
    out = []
    for name, N, K in shapes:
        x, w, dy = ten[name]
        a = t(lambda: ops.linear_fwd(x, w)); b = t(lambda: ops.linear_dgrad(dy, w)); c = t(lambda: ops.linear_wgrad(dy, x))
        out.append(f"{name}: {2*M*N*K/a/1e9:.0f}/{2*M*N*K/b/1e9:.0f}/{2*M*N*K/c/1e9:.0f}")
    print(f"GROUP_M={gm:2d}  " + "   ".join(out), flush=True)
