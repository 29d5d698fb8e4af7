#!/usr/bin/env python
"""Experiment (round 4): the 16-wave form of the ring kernel (tile code 272, csrc/gemm_ring.hip gemm_ring16_kernel) against the
shipped 8-wave kernel (264 automatic stages, 267 four-stage forced): results (rel-L2 between the two, both against an fp32 reference)
and time per launch on shapes of the denoising loop at UNet batch 2 (single-round grids) and 16.   python tools/ring16_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import _lib, ops  # noqa: E402
from tools.unet_gemm_bench import graph_time  # noqa: E402

BF = torch.bfloat16
torch.manual_seed(0)


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def linear(M, N, K, sk=1):
    x = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    b = torch.randn(N, device="cuda").to(BF)
    res = torch.randn(M, N, device="cuda").to(BF)
    ws = torch.empty(sk * M * N, dtype=torch.float32, device="cuda") if sk > 1 else None
    ref = (x.float() @ w.float().t() + b.float() + res.float())
    outs, ts = {}, {}
    for v in (267, 273, 268, 274, 272):
        out = torch.zeros(M, N, device="cuda", dtype=BF)

        def fn(i, v=v, out=out):
            _lib.check("dllm_gemm_bf16_splitk", ops._p(x), ops._p(w), ops._p(out), ops._p(b), ops._p(res), M, N, K, K, K, N, N, 0, 0, 0, 0, 0,
                       1.0, sk, ops._p(ws), None, v, ops._stream())
        fn(0)
        torch.cuda.synchronize()
        outs[v] = out.clone()
        ts[v] = graph_time(fn, reps=10)
    print(f"lin  M={M:6d} N={N:5d} K={K:5d} sk={sk:2d} | " + " ".join(f"[{v}] {ts[v]:7.1f}us err {rel(outs[v], ref):.2e}" for v in outs) +
          f" | 273 vs 267 {rel(outs[273], outs[267]):.1e} 274 vs 268 {rel(outs[274], outs[268]):.1e} 272 vs 267 {rel(outs[272], outs[267]):.1e}", flush=True)


def conv(NB, H, C, CO, stride=1, up=0, sk=1):
    x = torch.randn(NB, H, H, C, device="cuda").to(BF)
    w = (torch.randn(CO, 9 * C, device="cuda") * 0.02).to(BF)
    b = torch.randn(CO, device="cuda").to(BF)
    OH = H * (2 if up else 1) // stride
    res = torch.randn(NB, OH, OH, CO, device="cuda").to(BF)
    M = NB * OH * OH
    ws = torch.empty(sk * M * CO, dtype=torch.float32, device="cuda") if sk > 1 else None
    xin = x.permute(0, 3, 1, 2).float()
    if up:
        xin = torch.nn.functional.interpolate(xin, scale_factor=2, mode="nearest")
    ref = torch.nn.functional.conv2d(xin, w.float().view(CO, 3, 3, C).permute(0, 3, 1, 2), b.float(), stride=stride, padding=1)
    ref = ref.permute(0, 2, 3, 1) + res.float()
    outs, ts = {}, {}
    for v in (267, 273, 268, 274, 272):
        out = torch.zeros(NB, OH, OH, CO, device="cuda", dtype=BF)

        def fn(i, v=v, out=out):
            _lib.check("dllm_conv2d_nhwc_bf16_splitk", ops._p(x), ops._p(w), ops._p(out), ops._p(b), ops._p(res), None, NB, H, H, C, OH, OH,
                       CO, 3, 3, stride, 1, int(up), 0, 0, 0, sk, ops._p(ws), None, v, ops._stream())
        fn(0)
        torch.cuda.synchronize()
        outs[v] = out.clone()
        ts[v] = graph_time(fn, reps=10)
    print(f"conv N={NB:2d} {H:2d}x{H:<2d} C{C:4d}->{CO:4d} s{stride} up{up} sk={sk:2d} M={M:6d} K={9 * C:5d} | " +
          " ".join(f"[{v}] {ts[v]:7.1f}us err {rel(outs[v], ref):.2e}" for v in outs) +
          f" | 273 vs 267 {rel(outs[273], outs[267]):.1e} 274 vs 268 {rel(outs[274], outs[268]):.1e} 272 vs 267 {rel(outs[272], outs[267]):.1e}", flush=True)


linear(1000, 328, 640)          # partial tiles in both directions
linear(8192, 320, 320)
linear(8192, 960, 320)
linear(8192, 320, 1280)
linear(2048, 640, 640)
linear(2048, 640, 2560)
linear(512, 1280, 1280)
linear(512, 1280, 5120, sk=4)
linear(128, 1280, 5120, sk=8)
conv(2, 64, 320, 320)
conv(2, 64, 640, 320)
conv(2, 32, 640, 640, sk=3)
conv(2, 16, 1280, 1280, sk=6)
conv(2, 8, 1280, 1280, sk=20)
conv(2, 64, 320, 320, stride=2)
conv(2, 8, 1280, 1280, up=1, sk=5)
conv(16, 64, 320, 320)
linear(65536, 320, 320)
linear(4096, 10240, 1280)
