#!/usr/bin/env python
"""A/B of the XCD-synchronised persistent walk (ops.GEMM_PERSIST, variant bit 24) on the large GEMM shapes of the headline step:
bit-identical results, interleaved timing rounds in ONE process (on, off, on, off, ...), median per variant.
    python tools/persist_ab.py [rounds]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

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


g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, device=DEV, generator=g) * sc).to(BF)
x, a, dy, dgu, dqkv = rn(T, d), rn(T, F), rn(T, d), rn(T, 2 * F), rn(T, 3 * d)
wqkv, wo, wgu, wd = rn(3 * d, d, sc=0.02), rn(d, d, sc=0.02), rn(2 * F, d, sc=0.02), rn(d, F, sc=0.02)
cases = [("qkv fwd   [T x 12288 x 4096]", lambda: ops.linear_fwd(x, wqkv), 2.0 * T * 3 * d * d),
         ("o fwd     [T x 4096 x 4096]", lambda: ops.linear_fwd(x, wo), 2.0 * T * d * d),
         ("gate|up fwd [T x 22016 x 4096]", lambda: ops.linear_fwd(x, wgu), 2.0 * T * 2 * F * d),
         ("down fwd  [T x 4096 x 11008]", lambda: ops.linear_fwd(a, wd), 2.0 * T * F * d),
         ("qkv dgrad [T x 4096 x 12288]", lambda: ops.linear_dgrad(dqkv, wqkv), 2.0 * T * 3 * d * d),
         ("gate|up dgrad [T x 4096 x 22016]", lambda: ops.linear_dgrad(dgu, wgu), 2.0 * T * 2 * F * d),
         ("down dgrad [T x 11008 x 4096]", lambda: ops.linear_dgrad(dy, wd), 2.0 * T * F * d),
         ("gate|up wgrad [22016 x 4096 x T]", lambda: ops.linear_wgrad(dgu, x), 2.0 * T * 2 * F * d)]
tot = {True: 0.0, False: 0.0}
for name, fn, flops in cases:
    res = {True: [], False: []}
    outs = {}
    for v in (True, False):
        ops.GEMM_PERSIST = v
        outs[v] = fn()
        timed(fn, 2)
    same = torch.equal(outs[True], outs[False])
    del outs
    for _ in range(rounds):
        for v in (True, False):
            ops.GEMM_PERSIST = v
            res[v].append(timed(fn))
    on, off = statistics.median(res[True]), statistics.median(res[False])
    tot[True] += on
    tot[False] += off
    print(f"{name:34s} persist {on:.3f} ms = {flops / on / 1e9:7.1f} TF | per-tile blocks {off:.3f} ms = {flops / off / 1e9:7.1f} TF | "
          f"{100 * (off / on - 1):+.1f} %  identical={same}", flush=True)
print(f"sum: persist {tot[True]:.3f} ms, per-tile {tot[False]:.3f} ms, {100 * (tot[False] / tot[True] - 1):+.2f} %")
ops.GEMM_PERSIST = False
