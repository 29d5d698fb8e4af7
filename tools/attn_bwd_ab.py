"""Backward of the bench shape per combination of kernels (bench library: bits 2 / 3 of ATTN_VARIANT switch the ping-pong dV / dK pass of
csrc/attn_bwd_dkv_pp.hip on): DLLM_BENCH_MODES=1 python -m dreamllm_amd.build; python tools/attn_bwd_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DREAMLLM_HIP_LIB", os.path.join(ROOT, "dreamllm_amd", "libdreamllm_hip_bench.so"))
import torch
from dreamllm_amd import ops
BF = torch.bfloat16
B, S, H, D = 16, 2048, 32, 128
q, k, v, do = (torch.randn(B, S, H, D, device="cuda").to(BF) for _ in range(4))
o, lse = ops.attn_fwd(q, k, v, True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
names = {2: "all 8-wave", 3: "pp dQ (shipped)", 11: "pp dQ + pp dK", 7: "pp dQ + pp dV", 15: "pp dQ + pp dK + pp dV"}
for rnd in range(2):
    for var in (2, 3, 11, 7, 15):
        ops.ATTN_VARIANT = var
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(4):
                ops.attn_bwd(do, q, k, v, o, lse, True)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 4)
        print(f"{names[var]:26s} {best:.3f} ms", flush=True)
