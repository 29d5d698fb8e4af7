#!/usr/bin/env python
"""s_memtime timeline of one work-group of the ping-pong dQ kernel (csrc/attn_bwd_pp.hip) at the bench shape: what each of the four
intervals of a key tile (C12 of half tile 0, E + C3 of half tile 0, C12 of half tile 1, E + C3 of half tile 1) costs wave 0 (group A)
and wave 4 (group B, one interval behind) and how long each barrier holds them; then the kernel's time alone.
Needs the bench library:  DLLM_BENCH_MODES=1 python -m dreamllm_amd.build ;  python tools/attn_bwd_pp_timeline.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DREAMLLM_HIP_LIB", os.path.join(ROOT, "dreamllm_amd", "libdreamllm_hip_bench.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamllm_amd import _lib, ops  # noqa: E402

BF = torch.bfloat16
B, S, H, D = 16, 2048, 32, 128
q, k, v, do = (torch.randn(B, S, H, D, device="cuda").to(BF) for _ in range(4))
o, lse = ops.attn_fwd(q, k, v, True)
dq = torch.empty_like(q)
delta = torch.empty(3, B, H, S, dtype=torch.float32, device="cuda")
stamps = torch.zeros(2048, dtype=torch.int64, device="cuda")
fn = _lib.lib().dllm_attn_bwd_dq_pp_timeline
fn.argtypes = [ctypes.c_void_p] * 9 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
fn.restype = ctypes.c_int


def run(st):
    rc = fn(ops._p(do), ops._p(q), ops._p(k), ops._p(v), ops._p(o), ops._p(lse), ops._p(delta), ops._p(dq), st, B, H, S, ops._stream())
    assert rc == 0, rc


for _ in range(3):
    run(ops._p(stamps))
torch.cuda.synchronize()
st = stamps.cpu().tolist()
PH = ["start", "requested", "delta", "landed+barrier", "loop", "loop-end", "last-barrier", "stored"]
for w, pb in ((0, 1024), (4, 1088)):
    ph = st[pb:pb + 17]
    for p in (0, 1):
        t = [ph[1 + 8 * p + i] - ph[0] for i in range(8)]
        print(f"wave {w} pass {p}: " + "  ".join(f"{n} {x}" for n, x in zip(PH, t)) + "   | steps: " + " ".join(str(t[i + 1] - t[i]) for i in range(7)))
names = ["C12(0)", "E+C3(0)", "C12(1)", "E+C3(1)"]
for w, base in ((0, 0), (4, 512)):
    s = [x for x in st[base:base + 512] if x != 0]
    n = (len(s) - 1) // 8
    ph0 = st[1024 + (64 if w == 4 else 0)]
    print(f"   first stamp at {s[0] - ph0}, last at {s[-1] - ph0} clk from kernel entry; every 32nd stamp: " + " ".join(str(x - ph0) for x in s[::32]))
    print(f"wave {w}: {len(s)} stamps, {n} tiles, first->last {s[-1] - s[0]} clk ({(s[-1] - s[0]) / max(n, 1):.0f} per tile)")
    print("   deltas of stamps 64..112: " + " ".join(str(s[i + 1] - s[i]) for i in range(64, 112)))
    work = {nm: [] for nm in names}
    bar = {nm: [] for nm in names}
    for j in range(1, n - 1):
        b = 8 * j  # stamps of tile j: [8j + 2p] end of interval p, [8j + 2p + 1] start of the next one
        for p, nm in enumerate(names):
            work[nm].append(s[b + 2 * p] - s[b + 2 * p - 1])
            bar[nm].append(s[b + 2 * p + 1] - s[b + 2 * p])

    def med(x):
        x = sorted(x)
        return x[len(x) // 2] if x else 0
    tot = 0
    for nm in names:
        print(f"   {nm:8s} work {med(work[nm]):5d}  (min {min(work[nm]):5d} max {max(work[nm]):5d})   barrier wait {med(bar[nm]):5d}")
        tot += med(work[nm]) + med(bar[nm])
    print(f"   sum of medians {tot} clk per tile")

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    e0.record()
    for _ in range(5):
        run(None)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 5)
print(f"dQ kernel alone: {best:.3f} ms  ({2 * 3 * B * H * S * S * D / 2 / best / 1e9:.0f} TF executed-unit rate over its 3 GEMM units)")

for var in (2, 3):  # the whole backward (dQ, dK, dV) through the product entry point
    ops.ATTN_VARIANT = var
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(3):
            ops.attn_bwd(do, q, k, v, o, lse, True)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 3)
    print(f"backward variant {var}: {best:.3f} ms  ({2.5 * 4 * B * H * S * S * D / 2 / best / 1e9:.0f} TF algorithmic)")
