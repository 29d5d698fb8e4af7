"""Is the dominant GEMM bound by the socket power limit, and is that limit a property of the kernel or of the arithmetic?
Runs the packed gate|up forward shape [32768 x 22016 x 4096] back to back for ~5 s per case and polls rocm-smi beside it:
  ours      gemm_pipe_kernel (tile code 259) through ops.linear_fwd
  blaslt    torch.matmul (hipBLASLt's pick for the shape) -- a yard-stick for tools only, never part of the product
on three kinds of operand data: N(0,1) bf16 (what bench.py uses), small integers {-1,0,1} (few mantissa bits toggle), zeros.
Prints ms per launch (HIP events around 20 launches, median), TFLOP/s, sclk and socket power per sample.
  python tools/power_probe.py [seconds] [tile codes] [kinds, default normal,ints,zeros] [noblaslt|blaslt] [fwd|dgrad|wgrad]     tile codes of ours, default 259; e.g. 259,261 adds the four-wave experiment kernel]"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
CODES = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "259").split(",")]
KINDS = (sys.argv[3] if len(sys.argv) > 3 else "normal,ints,zeros").split(",")
BLASLT = not (len(sys.argv) > 4 and sys.argv[4] == "noblaslt")
LAYOUT = sys.argv[5] if len(sys.argv) > 5 else "fwd"   # fwd: y = x W^T; dgrad: dx = dy W; wgrad: dW = dy^T x (the same three matrices)
M, N, K = 32768, 22016, 4096
BF = torch.bfloat16
FLOP = 2.0 * M * N * K


def data(kind):
    g = torch.Generator(device="cuda").manual_seed(0)
    if kind == "normal":
        return (torch.randn(M, K, device="cuda", generator=g).to(BF), torch.randn(N, K, device="cuda", generator=g).to(BF))
    if kind == "ints":
        return (torch.randint(-1, 2, (M, K), device="cuda", generator=g).to(BF), torch.randint(-1, 2, (N, K), device="cuda", generator=g).to(BF))
    return torch.zeros(M, K, device="cuda", dtype=BF), torch.zeros(N, K, device="cuda", dtype=BF)


def smi():
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = pw = None
    for ln in r.splitlines():
        if "sclk" in ln:
            sclk = ln.split("(")[-1].split("Mhz")[0]
        if "Power (W)" in ln:
            pw = ln.split(":")[-1].strip()
    return sclk, pw


def run(tag, fn):
    stop, samples = [False], []

    def poll():
        while not stop[0]:
            samples.append(smi())
            time.sleep(0.7)

    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=poll)
    th.start()
    t0, times = time.time(), []
    while time.time() - t0 < SECS:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1) / 20)
    stop[0] = True
    th.join()
    times.sort()
    late = times[len(times) // 2:]          # the second half of the run: clocks have settled on the power limit
    ms = sorted(late)[len(late) // 2]
    tail = samples[len(samples) // 2:]
    print(f"{tag:28s} {ms:7.3f} ms  {FLOP / ms / 1e9:7.1f} TF   first launch group {times[0]:6.3f} ms   "
          + "  ".join(f"{s}MHz/{p}W" for s, p in tail[:5]), flush=True)


for kind in KINDS:
    a, w = data(kind)
    wt = w.t()
    if LAYOUT != "fwd":
        g = torch.Generator(device="cuda").manual_seed(1)
        dy = (torch.zeros(M, N, device="cuda", dtype=BF) if kind == "zeros" else
              torch.randint(-1, 2, (M, N), device="cuda", generator=g).to(BF) if kind == "ints" else torch.randn(M, N, device="cuda", generator=g).to(BF))
        dyt = dy.t()
    ours = {"fwd": lambda: ops.linear_fwd(a, w), "dgrad": lambda: ops.linear_dgrad(dy, w), "wgrad": lambda: ops.linear_wgrad(dy, a)}[LAYOUT]
    theirs = {"fwd": lambda: torch.matmul(a, wt), "dgrad": lambda: torch.matmul(dy, w), "wgrad": lambda: torch.matmul(dyt, a)}[LAYOUT]
    for code in CODES:
        with ops.gemm_variant(code):
            run(f"ours {code} {LAYOUT} {kind}", ours)
    if BLASLT:
        run(f"blaslt {LAYOUT} {kind}", theirs)
    time.sleep(2.0)
    del a, w, wt
