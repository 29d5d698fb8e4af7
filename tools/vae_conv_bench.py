#!/usr/bin/env python
"""3x3 convolutions of the SD VAE encoder at the training batch (32 images of 512 x 512) under each kernel family:
auto, 256 x 256 pipelined (259), 256 x 128 pipelined (262), ring four-stage (267), ring two-stage (268).  us per launch (graph of
5 launches, median of 3 replays).      python tools/vae_conv_bench.py [batch=32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import _lib, ops  # noqa: E402
from tools.unet_gemm_bench import graph_time  # noqa: E402

BF = torch.bfloat16
NB = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 32
SHAPES = [(512, 128, 128), (256, 128, 256), (256, 256, 256), (128, 256, 512), (128, 512, 512), (64, 512, 512)]
for H, C, CO in SHAPES:
    x = torch.randn(NB, H, H, C, device="cuda").to(BF)
    w = (torch.randn(CO, 9 * C, device="cuda") * 0.02).to(BF)
    b = torch.zeros(CO, device="cuda", dtype=BF)
    out = torch.empty(NB, H, H, CO, device="cuda", dtype=BF)
    res = {}
    for name, v in (("auto", 0), ("pipe256", 259), ("pipe256x128", 262), ("ring4st", 267), ("ring2st", 268)):
        def fn(i, v=v):
            _lib.check("dllm_conv2d_nhwc_bf16_splitk", ops._p(x), ops._p(w), ops._p(out), ops._p(b), None, None, NB, H, H, C, H, H, CO, 3, 3,
                       1, 1, 0, 0, 0, 0, 1, None, None, v, ops._stream())
        res[name] = graph_time(fn, reps=5)
    fl = 2.0 * NB * H * H * CO * 9 * C
    print(f"conv {H:3d}x{H:<3d} C{C:4d}->{CO:4d} M={NB * H * H:8d} K={9 * C:5d} | " + " ".join(f"{k}={v:8.1f}" for k, v in res.items()) +
          f" | best {fl / min(res.values()) / 1e6:6.1f} TF", flush=True)
    del x, w, out
