"""Sustained (power-limited) A/B of GEMM kernel families on the twelve big linears of the headline step, each point ~1.5 s back to back so that
the clock has settled on the socket power limit (short bursts run 10-15 % faster and rank the kernels differently), GROUP_M from the shipped
table, two interleaved rounds, HIP-event time of the second half of every run.
  python tools/gemm_ab_sustained.py [tile codes, default 259,280] [seconds per point, default 1.5]
259 = the 8-wave kernel (gemm_pipe_kernel), 280 = the four-wave kernel (gemm_w4m_kernel), 0 = the launcher's own choice."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

BF = torch.bfloat16
codes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "259,280").split(",")]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
T = 32768
shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate|up", 22016, 4096), ("down", 4096, 11008)]


def sustained(fn):
    fn()
    torch.cuda.synchronize()
    t0, times = time.perf_counter(), []
    while time.perf_counter() - t0 < secs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1) / 10)
    late = sorted(times[len(times) // 2:])
    return late[len(late) // 2]


tot = {c: 0.0 for c in codes}
for name, N, K in shapes:
    x = torch.randn(T, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    dy = torch.randn(T, N, device="cuda").to(BF)
    fns = {"fwd": (lambda: ops.linear_fwd(x, w), (0, 0, T, N, K)), "dgrad": (lambda: ops.linear_dgrad(dy, w), (0, 1, T, K, N)),
           "wgrad": (lambda: ops.linear_wgrad(dy, x), (1, 1, N, K, T))}
    flops = 2.0 * T * N * K
    for kind, (fn, key) in fns.items():
        gm = ops._group_m_for(*key)
        res = {c: [] for c in codes}
        for _ in range(2):
            for c in codes:
                with ops.gemm_variant(c, gm):
                    res[c].append(sustained(fn))
        for c in codes:
            tot[c] += min(res[c])
        print(f"{name:8s} {kind:6s} gm{gm}  " + "  ".join(f"[{c}] {min(res[c]):7.3f} ms {flops / min(res[c]) / 1e9:6.0f} TF" for c in codes), flush=True)
    del x, w, dy
# the fused launches of the decoder layer (family by gemm_variant: 259 / 280 reach the fused entry points through bits 8-9 of group_m)
from oracle import llm_ref  # noqa: E402  (tools may use the oracle's RoPE tables)
D = 128
x = torch.randn(T, 4096, device="cuda").to(BF)
wqkv = (torch.randn(12288, 4096, device="cuda") * 0.02).to(BF)
wgu = (torch.randn(22016, 4096, device="cuda") * 0.02).to(BF)
wd = (torch.randn(4096, 11008, device="cuda") * 0.02).to(BF)
dy = torch.randn(T, 4096, device="cuda").to(BF)
gu = torch.randn(T, 22016, device="cuda").to(BF)
cos, sin = llm_ref.rope_tables(D, 2048)
ct, st = cos[:, : D // 2].contiguous().cuda(), sin[:, : D // 2].contiguous().cuda()
fused = {"qkv + RoPE fwd": (lambda: ops.linear_rope_qkv(x, wqkv, ct, st, None, 64, D, 2048), 2.0 * T * 12288 * 4096),
         "gate|up + SwiGLU fwd": (lambda: ops.linear_swiglu_fwd(x, wgu), 2.0 * T * 22016 * 4096),
         "down dgrad + SwiGLU bwd": (lambda: ops.linear_dgrad_swiglu(dy, wd, gu), 2.0 * T * 11008 * 4096)}
for name, (fn, flops) in fused.items():
    res = {c: [] for c in codes}
    for _ in range(2):
        for c in codes:
            with ops.gemm_variant(c):
                assert fn() is not None
                res[c].append(sustained(fn))
    print(f"{name:24s}  " + "  ".join(f"[{c}] {min(res[c]):7.3f} ms {flops / min(res[c]) / 1e9:6.0f} TF" for c in codes), flush=True)
print("sum of the twelve  " + "  ".join(f"[{c}] {tot[c]:8.3f} ms" for c in codes))
