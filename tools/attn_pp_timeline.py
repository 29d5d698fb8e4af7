#!/usr/bin/env python
"""s_memtime timeline of work-group 0 of the ping-pong attention forward (csrc/attn_fwd_pp.hip) at the bench shape: what each of the
four segments of a key tile costs wave 0 (group A) and wave 4 (group B, one interval behind), and how long each barrier holds them.
Needs the bench library:  DLLM_BENCH_MODES=1 python -m dreamllm_amd.build ;  python tools/attn_pp_timeline.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DREAMLLM_HIP_LIB", os.path.join(ROOT, "dreamllm_amd", "libdreamllm_hip_bench.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamllm_amd import _lib, ops  # noqa: E402

BF = torch.bfloat16
B, S, H, D = 16, 2048, 32, 128
q, k, v = (torch.randn(B, S, H, D, device="cuda").to(BF) for _ in range(3))
o = torch.empty_like(q)
lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
stamps = torch.zeros(1024, dtype=torch.int64, device="cuda")
fn = _lib.lib().dllm_attn_fwd_pp_timeline
fn.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
fn.restype = ctypes.c_int
names = ["L_K", "C_QK", "L_V", "C_PV"]
ABLS = [int(a) for a in sys.argv[1:]] or [0]
for abl in ABLS:
  print(f"=== ablation {abl} (1 no in-loop DMA, 2 no fillers beside P V, 4 no softmax in L_V, 8 no LDS fragment reads)")
  stamps.zero_()
  for _ in range(3):
    rc = fn(ops._p(q), ops._p(k), ops._p(v), ops._p(o), ops._p(lse), ops._p(stamps), B, H, S, abl, ops._stream())
    assert rc == 0, rc
  torch.cuda.synchronize()
  st = stamps.cpu().tolist()
  for w, base in ((0, 0), (4, 512)):
      s = [x for x in st[base:base + 512] if x != 0]
      n = (len(s) - 1) // 8
      print(f"wave {w}: {len(s)} stamps, {n} tiles, first->last {s[-1] - s[0]} clk ({(s[-1] - s[0]) / max(n, 1):.0f} per tile)")
      work = {nm: [] for nm in names}
      bar = {nm: [] for nm in names}
      for j in range(1, n - 1):
          # stamps of tile j: [8j] end L_K, [8j+1] start C_QK, [8j+2] end C_QK, [8j+3] start L_V, [8j+4] end L_V, [8j+5] start C_PV,
          # [8j+6] end C_PV, [8j+7] start of the next L_K
          b = 8 * j
          work["L_K"].append(s[b] - s[b - 1])
          bar["L_K"].append(s[b + 1] - s[b])
          work["C_QK"].append(s[b + 2] - s[b + 1])
          bar["C_QK"].append(s[b + 3] - s[b + 2])
          work["L_V"].append(s[b + 4] - s[b + 3])
          bar["L_V"].append(s[b + 5] - s[b + 4])
          work["C_PV"].append(s[b + 6] - s[b + 5])
          bar["C_PV"].append(s[b + 7] - s[b + 6])

      def med(x):
          x = sorted(x)
          return x[len(x) // 2] if x else 0
      tot = 0
      for nm in names:
          print(f"   {nm:5s} work {med(work[nm]):5d}  (min {min(work[nm]):5d} max {max(work[nm]):5d})   barrier wait {med(bar[nm]):5d}")
          tot += med(work[nm]) + med(bar[nm])
      print(f"   sum of medians {tot} clk per tile")
      print("   tiles 1..4 work:", [[work[nm][i] for nm in names] for i in range(min(4, len(work['L_K'])))])
