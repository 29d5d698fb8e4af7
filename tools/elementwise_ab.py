#!/usr/bin/env python
"""A/B of the non-temporal SwiGLU kernels (ops.GLU_NT_MIN_BYTES) and timing of the 8-wide AdamW kernel at the headline shapes;
interleaved rounds in one process, medians.   python tools/elementwise_ab.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

T, F = 32768, 11008
BF, DEV = torch.bfloat16, "cuda"


def timed(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


gu = torch.randn(T, 2 * F, device=DEV).to(BF)
d = torch.randn(T, F, device=DEV).to(BF)
dgu = torch.empty(T, 2 * F, dtype=BF, device=DEV)
act = torch.empty(T, F, dtype=BF, device=DEV)
g, u = gu[:, :F], gu[:, F:]
cases = [("glu_fwd  (6 B/elt)", lambda: ops.glu_fwd(g, u, 0), 3 * T * F * 2),
         ("glu_bwd  (12 B/elt, emits act)", lambda: ops.glu_bwd(d, g, u, 0, da=dgu[:, :F], db=dgu[:, F:], act_out=act), 6 * T * F * 2)]
for name, fn, nbytes in cases:
    res = {True: [], False: []}
    outs = {}
    for nt in (True, False):
        ops.GLU_NT_MIN_BYTES = 0 if nt else 1 << 62
        r = fn()
        outs[nt] = (r if torch.is_tensor(r) else act).clone()
        timed(fn, 3)
    for _ in range(7):
        for nt in (True, False):
            ops.GLU_NT_MIN_BYTES = 0 if nt else 1 << 62
            res[nt].append(timed(fn))
    a, b = statistics.median(res[True]), statistics.median(res[False])
    print(f"{name:32s} nt {a * 1e3:7.1f} us = {nbytes / a / 1e9:6.2f} TB/s | plain {b * 1e3:7.1f} us = {nbytes / b / 1e9:6.2f} TB/s | "
          f"{100 * (b / a - 1):+.1f} %  identical={torch.equal(outs[True], outs[False])}", flush=True)
ops.GLU_NT_MIN_BYTES = 64 << 20
# AdamW: the largest parameter tensor of the 7B model (packed gate|up: 22016 x 4096) -- 8-wide non-temporal kernel vs the scalar one
n = 22016 * 4096
for off, tag in ((0, "vec8 nt"), (1, "scalar")):
    p = torch.randn(n + 8, device=DEV).to(BF)[off:off + n]
    gr = torch.randn(n + 8, device=DEV).to(BF)[off:off + n]
    m = torch.zeros(n + 8, dtype=BF, device=DEV)[off:off + n]
    v = torch.zeros(n + 8, dtype=BF, device=DEV)[off:off + n]
    fn = lambda: ops.adamw_(p, gr, m, v, 1e-4, 0.9, 0.98, 1e-8, 0.0, 3)
    timed(fn, 3)
    t = statistics.median([timed(fn) for _ in range(7)])
    print(f"adamw {tag:8s} n={n}: {t * 1e3:7.1f} us = {14 * n / t / 1e9:6.2f} TB/s (14 B/param)", flush=True)
