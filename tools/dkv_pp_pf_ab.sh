#!/bin/bash
cd /root/repo
for lib in tools/bin/libdqab_pf3.so tools/bin/libdqab_pf4.so; do
  sed "s|/root/repo/dreamllm_amd/libdreamllm_hip_bench.so|$PWD/$lib|" tools/attn_bwd_ab.py > /tmp/ab_$$.py
  python /tmp/ab_$$.py 2>&1 | grep " ms" | sed "s|^|$(basename $lib .so) |"
done
DREAMLLM_HIP_LIB=$PWD/tools/bin/libdqab_pf4.so python - <<'PY'
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from dreamllm_amd import ops
BF = torch.bfloat16
def rel(a, b): return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()
for (B, H, Sq, Sk, D, causal) in [(1, 2, 777, 777, 128, True), (1, 5, 600, 600, 128, False), (2, 4, 2048, 2048, 128, True), (1, 2, 40, 600, 128, True), (1, 2, 1100, 1100, 128, True)]:
    torch.manual_seed(0)
    q, do = (torch.randn(B, Sq, H, D, device="cuda").to(BF) for _ in range(2))
    k, v = (torch.randn(B, Sk, H, D, device="cuda").to(BF) for _ in range(2))
    ops.ATTN_VARIANT = 0
    o, lse = ops.attn_fwd(q, k, v, causal)
    out = {}
    for var in (2, 15):
        ops.ATTN_VARIANT = var
        out[var] = ops.attn_bwd(do, q, k, v, o, lse, causal)
    print(f"check Sq{Sq} Sk{Sk} causal={causal}: dk {rel(out[15][1], out[2][1]):.2e} dv {rel(out[15][2], out[2][2]):.2e}")
PY
