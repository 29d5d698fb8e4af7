"""GEMM kernel variants on the shapes of the headline step, interleaved in one process (box-to-box variance is larger than the
differences): python tools/gemm_bench.py [variants, default 259,280].  259 = 8-wave pipelined kernel, 280 = four-wave kernel (261: its MFMA 32x32x16
experiment).  SHORT bursts: the clock has not settled on the power limit and the ranking differs from sustained runs -- use gemm_ab_sustained.py to decide."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

BF = torch.bfloat16
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "259,280").split(",")]


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


T = 32768
shapes = [("qkv fwd", "fwd", T, 12288, 4096), ("o fwd", "fwd", T, 4096, 4096), ("gate/up fwd", "fwd", T, 22016, 4096),
          ("down fwd", "fwd", T, 4096, 11008), ("lm_head fwd", "fwd", 4096, 32008, 4096),
          ("qkv dgrad", "dgrad", T, 12288, 4096), ("gate/up dgrad", "dgrad", T, 22016, 4096), ("down dgrad", "dgrad", T, 4096, 11008),
          ("qkv wgrad", "wgrad", T, 12288, 4096), ("gate/up wgrad", "wgrad", T, 22016, 4096), ("down wgrad", "wgrad", T, 4096, 11008),
          ("square 4096", "fwd", 4096, 4096, 4096), ("square 8192", "fwd", 8192, 8192, 8192)]
for name, kind, M, N, K in shapes:
    x = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    dy = torch.randn(M, N, device="cuda").to(BF)
    if kind == "fwd":
        fn = lambda: ops.linear_fwd(x, w)  # noqa: E731
    elif kind == "dgrad":
        fn = lambda: ops.linear_dgrad(dy, w)  # noqa: E731
    else:
        fn = lambda: ops.linear_wgrad(dy, x)  # noqa: E731
    res = {}
    for rnd in range(2):
        for v in variants:
            with ops.gemm_variant(v):
                res.setdefault(v, []).append(timed(fn))
    flops = 2.0 * M * N * K
    print(f"{name:16s} M={M:6d} N={N:6d} K={K:6d}  " + "  ".join(f"[{v}] {min(ts):7.3f} ms {flops / min(ts) / 1e9:6.0f} TF" for v, ts in res.items()), flush=True)
    del x, w, dy
