#!/usr/bin/env python
"""Per-shape timing of the distinct 3x3 / 1x1 convolutions of one SD-2.1 UNet forward at a given batch (the denoise loop runs batch
2 * B_img): which shapes sit below the family's speed, with the kernel family / split-K the library picks for them.
    python tools/unet_conv_bench.py [batch=16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import _lib, ops  # noqa: E402

BF = torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
# (count per forward, H, C, CO, k, stride, up2)  -- profiles/r03_unet_conv_shapes_b16.txt
SHAPES = [(7, 64, 320, 320, 3, 1, 0), (2, 64, 640, 320, 3, 1, 0), (1, 64, 960, 320, 3, 1, 0), (6, 32, 640, 640, 3, 1, 0),
          (1, 32, 1920, 640, 3, 1, 0), (1, 32, 1280, 640, 3, 1, 0), (1, 32, 960, 640, 3, 1, 0), (1, 32, 320, 640, 3, 1, 0),
          (6, 16, 1280, 1280, 3, 1, 0), (2, 16, 2560, 1280, 3, 1, 0), (1, 16, 1920, 1280, 3, 1, 0), (1, 16, 640, 1280, 3, 1, 0),
          (11, 8, 1280, 1280, 3, 1, 0), (3, 8, 2560, 1280, 3, 1, 0),
          (1, 64, 320, 320, 3, 2, 0), (1, 32, 640, 640, 3, 2, 0), (1, 16, 1280, 1280, 3, 2, 0),
          (1, 8, 1280, 1280, 3, 1, 1), (1, 16, 1280, 1280, 3, 1, 1), (1, 32, 640, 640, 3, 1, 1),
          (2, 16, 2560, 1280, 1, 1, 0), (2, 64, 640, 320, 1, 1, 0), (3, 8, 2560, 1280, 1, 1, 0)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


tot_ms = tot_fl = 0.0
for cnt, H, C, CO, k, stride, up in SHAPES:
    x = torch.randn(N, H, H, C, device="cuda").to(BF)
    w = (torch.randn(CO, k * k * C, device="cuda") * 0.02).to(BF)
    b = torch.zeros(CO, device="cuda", dtype=BF)
    OH = H * (2 if up else 1) // stride
    fn = lambda: ops.conv2d_nhwc(x, w, CO, k, k, stride=stride, pad=k // 2, bias=b, up2=bool(up))
    ms = timeit(fn)
    M = N * OH * OH
    fl = 2.0 * M * CO * k * k * C
    sk = _lib.call("dllm_gemm_splitk_hint", M, CO, k * k * C, 2, 0)
    tot_ms += cnt * ms
    tot_fl += cnt * fl
    print(f"x{cnt:2d} {H:2d}x{H:<2d} C{C:4d}->{CO:4d} k{k} s{stride} up{up}  M={M:6d} K={k * k * C:5d} tiles256={-(-M // 256) * -(-CO // 256):4d} "
          f"splitk={sk:2d}  {ms * 1e3:7.1f} us  {fl / ms / 1e9:6.1f} TF", flush=True)
print(f"sum over one forward at batch {N}: {tot_ms:.3f} ms, {tot_fl / tot_ms / 1e9:.1f} TF average")
