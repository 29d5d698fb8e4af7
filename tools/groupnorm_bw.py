"""GroupNorm(+SiLU) forward / backward on the tensors of the training step (UNet at 32 images, VAE encoder at 32 x 512 px), effective bandwidth
counted as 3 passes forward (statistics read, apply read, write) and 5 backward: python tools/groupnorm_bw.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

BF = torch.bfloat16


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, N, HW, C in [("unet 64^2 320", 32, 4096, 320), ("unet 64^2 640 (cat)", 32, 4096, 640), ("unet 32^2 640", 32, 1024, 640),
                       ("unet 16^2 1280", 32, 256, 1280), ("vae 512^2 128", 32, 262144, 128), ("vae 256^2 256", 32, 65536, 256),
                       ("vae 128^2 512", 32, 16384, 512), ("vae 64^2 512", 32, 4096, 512)]:
    x = torch.randn(N, HW, C, device="cuda").to(BF)
    g, b = torch.ones(C, device="cuda", dtype=BF), torch.zeros(C, device="cuda", dtype=BF)
    t = timed(lambda: ops.groupnorm_fwd(x, g, b, 32, 1e-5, True))
    by = x.numel() * 2
    dy = torch.randn_like(x)
    y, mean, rstd = ops.groupnorm_fwd(x, g, b, 32, 1e-5, True)[:3]
    t2 = timed(lambda: ops.groupnorm_bwd(dy, x, g, b, mean, rstd, 32, True))
    print(f"{name:22s} {by / 1e6:8.1f} MB  fwd {t * 1e3:8.1f} us = {3 * by / t / 1e9:5.2f} TB/s   bwd {t2 * 1e3:8.1f} us = {5 * by / t2 / 1e9:5.2f} TB/s", flush=True)
    del x, dy, y
