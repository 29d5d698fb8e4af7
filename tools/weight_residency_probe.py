#!/usr/bin/env python
"""How much of a small-M layer's time is the weight fetch?  One UNet linear / conv shape at UNet batch 2, launched back to back from a
hipGraph with the weights rotating over 1 copy (L2-resident), enough copies for ~100 MB (Infinity-Cache-resident: 256 MB) and enough
for ~1 GB (every launch streams its weights from HBM, as in the real forward: 1.73 GB of weights per step).  If HBM-cold is much
slower than cache-warm, prefetching the next layers' weights on a side stream would pay.   python tools/weight_residency_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import _lib, ops  # noqa: E402

BF = torch.bfloat16
REPS = 40


def graph_time(fn_of_i, reps=REPS):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(2):
            fn_of_i(i)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            fn_of_i(i)
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts[1:])[1]


def linear(M, N, K, ncopies):
    x = torch.randn(M, K, device="cuda").to(BF)
    ws_ = [(torch.randn(N, K, device="cuda") * 0.03).to(BF) for _ in range(ncopies)]
    out = torch.empty(M, N, device="cuda", dtype=BF)
    sk = _lib.call("dllm_gemm_splitk_hint", M, N, K)
    ws = torch.empty(sk * M * N, dtype=torch.float32, device="cuda") if sk > 1 else None

    def fn(i):
        _lib.check("dllm_gemm_bf16_splitk", ops._p(x), ops._p(ws_[i % ncopies]), ops._p(out), None, None, M, N, K, K, K, N, 0, 0, 0, 0, 0, 0,
                   1.0, sk, ops._p(ws), None, 0, ops._stream())
    return graph_time(fn)


def conv(NB, H, C, CO, ncopies):
    x = torch.randn(NB, H, H, C, device="cuda").to(BF)
    ws_ = [(torch.randn(CO, 9 * C, device="cuda") * 0.02).to(BF) for _ in range(ncopies)]
    out = torch.empty(NB, H, H, CO, device="cuda", dtype=BF)
    M = NB * H * H
    sk = _lib.call("dllm_gemm_splitk_hint", M, CO, 9 * C)
    ws = torch.empty(sk * M * CO, dtype=torch.float32, device="cuda") if sk > 1 else None

    def fn(i):
        _lib.check("dllm_conv2d_nhwc_bf16_splitk", ops._p(x), ops._p(ws_[i % ncopies]), ops._p(out), None, None, None, NB, H, H, C, H, H, CO,
                   3, 3, 1, 1, 0, 0, 0, 0, sk, ops._p(ws), None, 0, ops._stream())
    return graph_time(fn)


for name, f, wbytes in [("lin M=512 N=1280 K=1280", lambda c: linear(512, 1280, 1280, c), 1280 * 1280 * 2),
                        ("lin M=512 N=10240 K=1280 (ff1)", lambda c: linear(512, 10240, 1280, c), 10240 * 1280 * 2),
                        ("lin M=512 N=1280 K=5120 (ff2)", lambda c: linear(512, 1280, 5120, c), 1280 * 5120 * 2),
                        ("lin M=128 N=1280 K=1280", lambda c: linear(128, 1280, 1280, c), 1280 * 1280 * 2),
                        ("lin M=2048 N=640 K=640", lambda c: linear(2048, 640, 640, c), 640 * 640 * 2),
                        ("conv 16x16 1280->1280", lambda c: conv(2, 16, 1280, 1280, c), 9 * 1280 * 1280 * 2),
                        ("conv 8x8 1280->1280", lambda c: conv(2, 8, 1280, 1280, c), 9 * 1280 * 1280 * 2),
                        ("conv 32x32 640->640", lambda c: conv(2, 32, 640, 640, c), 9 * 640 * 640 * 2)]:
    res = {}
    for tag, total in (("L2 (1 copy)", 0), ("MALL (~100 MB)", 100e6), ("HBM (~1 GB)", 1000e6)):
        c = 1 if total == 0 else max(2, int(total // wbytes))
        res[tag] = f(c)
    print(f"{name:34s} weights {wbytes / 1e6:6.1f} MB | " + " | ".join(f"{k}: {v:6.1f} us" for k, v in res.items()), flush=True)
