#!/usr/bin/env python
"""s_memtime timeline of work-group 0 of the ping-pong attention forward (csrc/attn_fwd_pp.hip) at the bench shape: what each of the
two segments of a key tile (X = C_QK + softmax head, Y = C_PV) costs wave 0 (group A) and wave 4 (group B, one segment behind), and
how long each barrier holds them.  Arguments: ablation codes (see the kernel), e.g. `0 1 2 4`.
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
stamps = torch.zeros(2048, dtype=torch.int64, device="cuda")
fn = _lib.lib().dllm_attn_fwd_pp_timeline
fn.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
fn.restype = ctypes.c_int
ABLS = [int(a) for a in sys.argv[1:]] or [0]
for abl in ABLS:
  print(f"=== ablation {abl} (1 no in-loop DMA, 2 no VALU fillers beside P V, 4 no softmax head)")
  stamps.zero_()
  for _ in range(3):
    rc = fn(ops._p(q), ops._p(k), ops._p(v), ops._p(o), ops._p(lse), ops._p(stamps), B, H, S, abl, ops._stream())
    assert rc == 0, rc
  torch.cuda.synchronize()
  st = stamps.cpu().tolist()
  for w, pb in ((0, 1024), (4, 1088)):
      ph = st[pb:pb + 9]
      print(f"wave {w} phases (clk from kernel entry): " + "  ".join(
          f"pass{p}: start {ph[1 + 4 * p] - ph[0]} loop {ph[2 + 4 * p] - ph[0]} loop-end {ph[3 + 4 * p] - ph[0]} stored {ph[4 + 4 * p] - ph[0]}" for p in (0, 1)))
  for w, base in ((0, 0), (4, 512)):
      s = [x for x in st[base:base + 512] if x != 0]
      n = (len(s) - 1) // 4
      print(f"wave {w}: {len(s)} stamps, {n} tiles, first->last {s[-1] - s[0]} clk ({(s[-1] - s[0]) / max(n, 1):.0f} per tile)")
      names = ["X", "Y"]
      work = {nm: [] for nm in names}
      bar = {nm: [] for nm in names}
      for j in range(1, n - 1):
          # stamps of tile j: [4j] end of X(j), [4j+1] start of Y(j), [4j+2] end of Y(j), [4j+3] start of X(j+1)
          b = 4 * j
          work["X"].append(s[b] - s[b - 1])
          bar["X"].append(s[b + 1] - s[b])
          work["Y"].append(s[b + 2] - s[b + 1])
          bar["Y"].append(s[b + 3] - s[b + 2])

      def med(x):
          x = sorted(x)
          return x[len(x) // 2] if x else 0
      tot = 0
      for nm in names:
          print(f"   {nm:5s} work {med(work[nm]):5d}  (min {min(work[nm]):5d} max {max(work[nm]):5d})   barrier wait {med(bar[nm]):5d}")
          tot += med(work[nm]) + med(bar[nm])
      print(f"   sum of medians {tot} clk per tile")
