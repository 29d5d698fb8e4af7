#!/usr/bin/env python
"""A/B of the stream-K tail (ops.STREAMK) on the GEMM shapes of the headline step whose grids end in a partial round of 256-tiles:
interleaved rounds in ONE process (on, off, on, off, ...), median per variant.   python tools/streamk_ab.py [rounds]"""
import os
import sys
import statistics

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import _lib, ops  # noqa: E402

T, d, F = 32768, 4096, 11008
BF, DEV = torch.bfloat16, "cuda"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7


def timed(fn, n=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


x = torch.randn(T, d, device=DEV).to(BF)
dgu = torch.randn(T, 2 * F, device=DEV).to(BF)
a = torch.randn(T, F, device=DEV).to(BF)
dy = torch.randn(T, d, device=DEV).to(BF)
wd = (torch.randn(d, F, device=DEV) * 0.02).to(BF)
cases = [("gate|up wgrad [22016x4096xT]", lambda: ops.linear_wgrad(dgu, x), 2.0 * 2 * F * d * T, (2 * F, d, T, 1, 1)),
         ("down wgrad [4096x11008xT]", lambda: ops.linear_wgrad(dy, a), 2.0 * F * d * T, (d, F, T, 1, 1)),
         ("down dgrad [Tx11008x4096]", lambda: ops.linear_dgrad(dy, wd), 2.0 * F * d * T, (T, F, d, 0, 1))]
for name, fn, flops, hint in cases:
    if _lib.call("dllm_gemm_streamk_hint", *hint) != 1:   # the library's plan declines the shape (measured loss: see streamk_plan)
        print(f"{name:32s} not planned for stream-K (hint 0)", flush=True)
        continue
    res = {True: [], False: []}
    for v in (True, False):
        ops.STREAMK = v
        timed(fn, 2)
    for _ in range(rounds):
        for v in (True, False):
            ops.STREAMK = v
            res[v].append(timed(fn))
    on, off = statistics.median(res[True]), statistics.median(res[False])
    print(f"{name:32s} stream-K {on:.3f} ms = {flops / on / 1e9:7.1f} TF | whole tiles {off:.3f} ms = {flops / off / 1e9:7.1f} TF | "
          f"{100 * (off / on - 1):+.1f} %", flush=True)
ops.STREAMK = True
