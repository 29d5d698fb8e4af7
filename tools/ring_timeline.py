#!/usr/bin/env python
"""s_memtime timeline of ONE block (block 0, wave 0) of the ring kernel: where a K tile's time goes on a single resident block.
Needs the bench library (DLLM_BENCH_MODES=1 python -m dreamllm_amd.build): tile codes 269 (four-stage) / 270 (two-stage) stamp the
shader clock before the tile's vmcnt wait, after it, and after the tile barrier.  Per shape: cycles per K tile split into
compute (barrier exit -> next wait), wait (vmcnt) and barrier, median over the tiles of the steady state.

    python tools/ring_timeline.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DREAMLLM_HIP_LIB", os.path.join(ROOT, "dreamllm_amd", "libdreamllm_hip_bench.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamllm_amd import _lib, ops  # noqa: E402

BF = torch.bfloat16


def report(tag, ws, v):
    torch.cuda.synchronize()
    st = ws.view(torch.int64).cpu()
    nt = int(st[1000])
    t0, tp = int(st[0]), int(st[1])
    comp, wait, bar = [], [], []
    prev_exit = tp
    for t in range(nt - 1):
        a, b, c = int(st[2 + 3 * t]), int(st[3 + 3 * t]), int(st[4 + 3 * t])
        comp.append(a - prev_exit)
        wait.append(b - a)
        bar.append(c - b)
        prev_exit = c
    end_loop, end = int(st[2 + 3 * (nt - 1)]), int(st[3 + 3 * (nt - 1)])

    def med(x):
        x = sorted(x[2:-2] if len(x) > 6 else x)
        return x[len(x) // 2] if x else 0
    tot = end - t0
    print(f"{tag} [{v}] nt={nt:3d} total {tot:6d} clk | prologue {tp - t0:5d} | per K tile (median): compute {med(comp):5d} wait {med(wait):5d} "
          f"barrier {med(bar):5d} = {med(comp) + med(wait) + med(bar):5d} | last tile + epilogue {end - prev_exit:5d} | first tiles wait {wait[:4]} barrier {bar[:4]}",
          flush=True)


def linear(M, N, K, variants=(269, 270)):
    x = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    for v in variants:
        ws = torch.zeros(4096, dtype=torch.float32, device="cuda")
        for _ in range(3):
            _lib.check("dllm_gemm_bf16_splitk", ops._p(x), ops._p(w), ops._p(out), None, None, M, N, K, K, K, N, 0, 0, 0, 0, 0, 0, 1.0, 1,
                       ops._p(ws), None, v, ops._stream())
        report(f"lin  [{M:6d},{N:5d},{K:5d}]", ws, v)


def conv(NB, H, C, CO, variants=(269, 270)):
    x = torch.randn(NB, H, H, C, device="cuda").to(BF)
    w = (torch.randn(CO, 9 * C, device="cuda") * 0.02).to(BF)
    out = torch.empty(NB, H, H, CO, device="cuda", dtype=BF)
    for v in variants:
        ws = torch.zeros(4096, dtype=torch.float32, device="cuda")
        for _ in range(3):
            _lib.check("dllm_conv2d_nhwc_bf16_splitk", ops._p(x), ops._p(w), ops._p(out), None, None, None, NB, H, H, C, H, H, CO, 3, 3, 1, 1,
                       0, 0, 0, 0, 1, ops._p(ws), None, v, ops._stream())
        report(f"conv [{NB * H * H:6d},{CO:5d},{9 * C:5d}]", ws, v)


V4 = (269, 277, 278)     # four-stage: shipped / ablation without the in-loop LDS-DMA requests / ablation without the MFMAs
linear(8192, 320, 1280, V4)      # 192 blocks, 20 K tiles
linear(2048, 640, 2560, V4)      # 80 blocks, 40 K tiles
conv(2, 64, 320, 320, V4)        # 192 blocks, 45 K tiles
conv(2, 16, 1280, 1280, V4)      # 40 blocks, 180 K tiles
