#!/usr/bin/env python
"""Per-shape timing of EVERY GEMM-shaped launch of one SD-2.1 UNet forward (linears of the transformer blocks + 3x3 / 1x1 convs) at a
given UNet batch, for a set of kernel choices: the automatic one, the register-staged 128-tile kernel (tile code 128), the
ring-buffered 128-tile kernel (264), the pipelined 256-tile kernels (259 / 262), each at several split-K factors.

Launches of 5-50 us cannot be timed from Python launch by launch (ctypes + torch overhead is ~10 us per call): every measurement is a
hipGraph of REPS back-to-back launches replayed 3 times (median).  The weights rotate over enough copies to exceed the 256 MB
Infinity Cache (in the real forward every layer's weights come from HBM once per step); the activation is reused (in the real
forward it was just written by the previous kernel).

    python tools/unet_gemm_bench.py [batch=2] [--quick]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import _lib, ops  # noqa: E402

BF = torch.bfloat16
args = [a for a in sys.argv[1:] if not a.startswith("--")]
NB = int(args[0]) if args else 2
QUICK = "--quick" in sys.argv
CONVS_ONLY = "--convs-only" in sys.argv
REPS = 20

# transformer linears per level: (blocks, HW, C)
LEVELS = [(5, 64 * 64, 320), (5, 32 * 32, 640), (5, 16 * 16, 1280), (1, 8 * 8, 1280)]
LINEARS = []  # (count, M, N, K, tag)
for nblk, hw, C in LEVELS:
    M = NB * hw
    LINEARS += [(3 * nblk, M, C, C, "proj_in/to_q/proj_out"), (nblk, M, 3 * C, C, "qkv"), (2 * nblk, M, C, C, "to_out(+res)"),
                (nblk, M, 8 * C, C, "ff1"), (nblk, M, C, 4 * C, "ff2(+res)")]
# (count per forward, H, C, CO, k, stride, up2)
CONVS = [(7, 64, 320, 320, 3, 1, 0), (2, 64, 640, 320, 3, 1, 0), (1, 64, 960, 320, 3, 1, 0), (6, 32, 640, 640, 3, 1, 0),
         (1, 32, 1920, 640, 3, 1, 0), (1, 32, 1280, 640, 3, 1, 0), (1, 32, 960, 640, 3, 1, 0), (1, 32, 320, 640, 3, 1, 0),
         (6, 16, 1280, 1280, 3, 1, 0), (2, 16, 2560, 1280, 3, 1, 0), (1, 16, 1920, 1280, 3, 1, 0), (1, 16, 640, 1280, 3, 1, 0),
         (11, 8, 1280, 1280, 3, 1, 0), (3, 8, 2560, 1280, 3, 1, 0),
         (1, 64, 320, 320, 3, 2, 0), (1, 32, 640, 640, 3, 2, 0), (1, 16, 1280, 1280, 3, 2, 0),
         (1, 8, 1280, 1280, 3, 1, 1), (1, 16, 1280, 1280, 3, 1, 1), (1, 32, 640, 640, 3, 1, 1),
         (2, 16, 2560, 1280, 1, 1, 0), (2, 64, 640, 320, 1, 1, 0), (3, 8, 2560, 1280, 1, 1, 0)]


def graph_time(fn_of_i, reps=REPS):
    """us per launch: REPS launches captured in one hipGraph, median of 3 replays."""
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
    del g
    return sorted(ts[1:])[1]


def copies(nbytes):
    return max(2, min(16, int(300e6 // max(nbytes, 1)) + 1))


def run_linear(M, N, K, variant, sk, geglu=False):
    x = (torch.randn(M, K, device="cuda")).to(BF)
    R = copies(N * K * 2)
    ws_ = [(torch.randn(N, K, device="cuda") * 0.03).to(BF) for _ in range(R)]
    b = torch.zeros(N, device="cuda", dtype=BF)
    NO = N // 2 if geglu else N
    out = torch.empty(M, NO, device="cuda", dtype=BF)
    ws = torch.empty(max(sk, 1) * M * N, dtype=torch.float32, device="cuda") if sk > 1 else None

    def fn(i):
        _lib.check("dllm_gemm_bf16_splitk", ops._p(x), ops._p(ws_[i % R]), ops._p(out), ops._p(b), None, M, NO, K, K, K, NO, 0, 0, 0,
                   4 if geglu else 0, 0, 0, 1.0, sk, ops._p(ws), None, variant, ops._stream())
    return graph_time(fn)


def run_conv(N, H, C, CO, k, stride, up, variant, sk):
    x = torch.randn(N, H, H, C, device="cuda").to(BF)
    R = copies(CO * k * k * C * 2)
    ws_ = [(torch.randn(CO, k * k * C, device="cuda") * 0.02).to(BF) for _ in range(R)]
    b = torch.zeros(CO, device="cuda", dtype=BF)
    OH = H * (2 if up else 1) // stride
    out = torch.empty(N, OH, OH, CO, device="cuda", dtype=BF)
    M = N * OH * OH
    ws = torch.empty(max(sk, 1) * M * CO, dtype=torch.float32, device="cuda") if sk > 1 else None

    def fn(i):
        _lib.check("dllm_conv2d_nhwc_bf16_splitk", ops._p(x), ops._p(ws_[i % R]), ops._p(out), ops._p(b), None, None, N, H, H, C, OH, OH,
                   CO, k, k, stride, k // 2, int(up), 0, 0, 0, sk, ops._p(ws), None, variant, ops._stream())
    return graph_time(fn)


def sk_candidates(M, N, K, la=0):
    hint = _lib.call("dllm_gemm_splitk_hint", M, N, K, la, 0)   # (round 6: the hint takes the layouts; la = 2: the NHWC conv gather)
    tiles = -(-M // 128) * -(-N // 128)
    kt = K // 64
    cands = {1, hint}
    if not QUICK:
        for want in (128, 192, 256, 384, 512):
            s = max(1, min(-(-want // tiles), kt // 4, 32))
            cands.add(s)
    return hint, sorted(cands)


def main():
    tot = {}

    def acc(name, cnt, us):
        tot[name] = tot.get(name, 0.0) + cnt * us

    print(f"# UNet batch {NB}: linears")
    for cnt, M, N, K, tag in ([] if CONVS_ONLY else LINEARS):
        hint, cands = sk_candidates(M, N, K)
        res = {}
        run_linear(M, N, K, 0, hint)               # (untimed: the first measurement of a shape runs on clocks that are still ramping)
        res["auto"] = run_linear(M, N, K, 0, hint)
        res["old128"] = run_linear(M, N, K, 128, hint)
        for sk in cands:
            res[f"ring/sk{sk}"] = run_linear(M, N, K, 264, sk)
        res["ring4st"] = run_linear(M, N, K, 267, 1)   # four-stage ring forced (one block per CU)
        res["ring2st"] = run_linear(M, N, K, 268, 1)   # two-stage ring forced (two blocks per CU)
        if -(-M // 256) * -(-N // 256) >= 32:
            res["pipe256"] = run_linear(M, N, K, 259, 1)
        if tag == "ff1":
            res["ring+geglu"] = run_linear(M, N, K, 264, 1, geglu=True)
        best_ring = min(v for k, v in res.items() if k.startswith("ring/"))
        acc("auto", cnt, res["auto"]); acc("old128", cnt, res["old128"]); acc("ring_hint", cnt, res[f"ring/sk{hint}"]); acc("ring_best", cnt, best_ring)
        acc("best_any", cnt, min(v for k, v in res.items() if k != "ring+geglu"))
        acc("ring4st", cnt, res["ring4st"]); acc("ring2st", cnt, res["ring2st"])
        fl = 2.0 * M * N * K
        print(f"x{cnt:2d} lin {tag:22s} M={M:6d} N={N:5d} K={K:5d} hint={hint:2d} | " +
              " ".join(f"{k}={v:6.1f}" for k, v in res.items()) + f" | best {fl / min(res.values()) / 1e6:6.1f} TF", flush=True)
    lin_tot = dict(tot)
    print("# linears, us per forward: " + " ".join(f"{k}={v:8.1f}" for k, v in lin_tot.items()), flush=True)
    print(f"# UNet batch {NB}: convs")
    for cnt, H, C, CO, k, stride, up in CONVS:
        OH = H * (2 if up else 1) // stride
        M, K = NB * OH * OH, k * k * C
        hint, cands = sk_candidates(M, CO, K, 2)
        res = {}
        run_conv(NB, H, C, CO, k, stride, up, 0, hint)
        res["auto"] = run_conv(NB, H, C, CO, k, stride, up, 0, hint)
        res["old128"] = run_conv(NB, H, C, CO, k, stride, up, 128, hint)
        for sk in cands:
            res[f"ring/sk{sk}"] = run_conv(NB, H, C, CO, k, stride, up, 264, sk)
        if -(-M // 256) * -(-CO // 256) >= 32:
            res["pipe256"] = run_conv(NB, H, C, CO, k, stride, up, 259, 1)
            res["pipe256x128"] = run_conv(NB, H, C, CO, k, stride, up, 262, 1)
        res["ring4st"] = run_conv(NB, H, C, CO, k, stride, up, 267, 1)
        res["ring2st"] = run_conv(NB, H, C, CO, k, stride, up, 268, 1)
        best_ring = min(v for k_, v in res.items() if k_.startswith("ring/"))
        acc("auto", cnt, res["auto"]); acc("old128", cnt, res["old128"]); acc("ring_hint", cnt, res[f"ring/sk{hint}"]); acc("ring_best", cnt, best_ring)
        acc("best_any", cnt, min(res.values()))
        acc("ring4st", cnt, res["ring4st"]); acc("ring2st", cnt, res["ring2st"])
        fl = 2.0 * M * CO * K
        print(f"x{cnt:2d} conv {H:2d}x{H:<2d} C{C:4d}->{CO:4d} k{k} s{stride} up{up} M={M:6d} K={K:5d} hint={hint:2d} | " +
              " ".join(f"{k_}={v:6.1f}" for k_, v in res.items()) + f" | best {fl / min(res.values()) / 1e6:6.1f} TF", flush=True)
    print("# convs, us per forward: " + " ".join(f"{k}={tot[k] - lin_tot.get(k, 0.0):8.1f}" for k in tot), flush=True)
    print("# all GEMM-shaped launches, us per forward: " + " ".join(f"{k}={v:8.1f}" for k, v in tot.items()), flush=True)


if __name__ == "__main__":
    main()
