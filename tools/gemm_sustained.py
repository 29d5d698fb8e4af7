"""Sustained (power-limited) throughput of the three dominant GEMM layouts per GROUP_M: each point runs ~2.5 s back to back,
so the number is taken at the clock the socket power limit allows (short bursts run 10-15 % faster).
python tools/gemm_sustained.py [group_m list, default 2,4,8,16] [seconds per point] [tile code, default 0 = the launcher's choice; 259 / 280]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

BF = torch.bfloat16
gms = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2,4,8,16").split(",")]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
code = int(sys.argv[3]) if len(sys.argv) > 3 else 0
T = 32768
# (name, N = out features, K = in features) of the decoder layer's linears (packed q|k|v and gate|up) and one lm_head chunk
shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate|up", 22016, 4096), ("down", 4096, 11008)]
for name, N, K in shapes:
    x = torch.randn(T, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    dy = torch.randn(T, N, device="cuda").to(BF)
    fns = {"fwd": lambda: ops.linear_fwd(x, w), "dgrad": lambda: ops.linear_dgrad(dy, w), "wgrad": lambda: ops.linear_wgrad(dy, x)}
    flops = 2.0 * T * N * K
    for kind, fn in fns.items():
        line = f"{name:8s} {kind:6s}"
        for gm in gms:
            with ops.gemm_variant(code, gm):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 0
                while time.perf_counter() - t0 < secs:
                    for _ in range(20):
                        fn()
                    torch.cuda.synchronize()
                    n += 20
                dt = (time.perf_counter() - t0) / n
            line += f"  gm{gm}: {flops / dt / 1e12:5.0f}"
        print(line, flush=True)
    del x, w, dy
