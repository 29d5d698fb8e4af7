"""K sweep of the forward GEMM at fixed M, N: the intercept of time(K) is the per-tile fixed cost (prologue + epilogue + launch),
the slope the steady-state K-loop rate.   python tools/gemm_ksweep.py [--tile 0]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tile", type=int, default=0)
ap.add_argument("--M", type=int, default=32768)
ap.add_argument("--N", type=int, default=4096)
a = ap.parse_args()
ops.GEMM_VARIANT = a.tile  # per-call kernel variant (diagnostic modes 258/260/263/265 need a -DDLLM_BENCH_MODES build)
BF = torch.bfloat16
M, N = a.M, a.N
res = []
for K in (512, 1024, 2048, 4096, 8192, 16384):
    x = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    for _ in range(3):
        ops.linear_fwd(x, w)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.linear_fwd(x, w)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    res.append((K, ms))
    print(f"tile={a.tile} M={M} N={N} K={K:6d}  {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TF", flush=True)
rounds = -(-(M // 256 * -(-N // 256)) // 256)
(k0, t0), (k1, t1) = res[2], res[-1]
slope = (t1 - t0) / (k1 - k0) * 64  # ms per 64-k tile per kernel
icpt = t0 - slope * k0 / 64
print(f"rounds/CU={rounds}: per-K-tile {slope/rounds*1e3:.3f} us, fixed per output tile {icpt/rounds*1e3:.2f} us (kernel intercept {icpt*1e3:.1f} us)")
