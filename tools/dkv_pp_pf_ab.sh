#!/bin/bash
# Same-call A/B of two builds of the ping-pong dK / dV passes (csrc/attn_bwd_dkv_pp.hip; here: ring depth 3 vs 4) through
# tools/attn_bwd_ab.py, then dK / dV of the second build against the 8-wave kernels on five shapes.  Builds first:
#   DLLM_BENCH_MODES=1 python -m dreamllm_amd.build; FILE=attn_bwd_dkv_pp.hip VARIANTS="pf3:-DKP_PF=3 pf4:-DKP_PF=4" tools/dq_pp_ab.sh
cd "$(dirname "$0")/.."
ROOT=$PWD
for lib in tools/bin/libdqab_pf3.so tools/bin/libdqab_pf4.so; do
  DREAMLLM_HIP_LIB=$PWD/$lib python tools/attn_bwd_ab.py 2>&1 | grep " ms" | sed "s|^|$(basename $lib .so) |"
done
DLLM_ROOT=$ROOT DREAMLLM_HIP_LIB=$PWD/tools/bin/libdqab_pf4.so python - <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("DLLM_ROOT", "."))
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
