"""Ablation of the causal head_dim-128 attention forward at the bench shape: each variant removes ONE cost (values kept live)
so that the remaining time shows what the loop is bound by.  Needs a -DDLLM_BENCH_MODES build:

    DLLM_BENCH_MODES=1 python -m dreamllm_amd.build --force && python tools/attn_ablate.py ; python -m dreamllm_amd.build --force
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import _lib  # noqa: E402

B, S, H, D = 16, 2048, 32, 128
BF = torch.bfloat16
q, k, v = (torch.randn(B, S, H, D, device="cuda").to(BF) for _ in range(3))
o = torch.empty_like(q)
lse = torch.empty(B, H, S, device="cuda")
fn = _lib.lib().dllm_attn_fwd_ablate
fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
names = {0: "full", 1: "no softmax", 2: "no PV mfma", 4: "no QK mfma", 8: "no loads/LDS stores", 16: "no O rescale", 3: "no softmax+PV",
         6: "no MFMA at all", 7: "no softmax, no MFMA (loads+LDS reads+barriers)", 15: "only LDS reads + barriers", 9: "no softmax, no loads"}
flops = 4 * S * S * D * H * B / 2
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for abl, name in names.items():
    args = (q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, S, D, abl, st)
    for _ in range(2):
        assert fn(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn(*args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"abl {abl:2d} {name:48s} {ms:7.3f} ms  ({flops / ms / 1e9:7.1f} TF-equivalent)", flush=True)
