"""Sustained clocks / power while one kernel family runs back to back: python tools/clock_probe.py {attn_bwd,attn_fwd,gemm,gemm259,gemm280,gemm261,idle}
Polls rocm-smi once per second from a thread while the main thread keeps the GPU busy for ~6 s."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "attn_bwd"
BF = torch.bfloat16
B, S, H, D = 16, 2048, 32, 128
q, k, v, do = (torch.randn(B, S, H, D, device="cuda").to(BF) for _ in range(4))
o, lse = ops.attn_fwd(q, k, v, True)
a = torch.randn(32768, 4096, device="cuda").to(BF)
w = torch.randn(11008, 4096, device="cuda").to(BF)
w2 = torch.randn(22016, 4096, device="cuda").to(BF) if what.startswith("gemm2") else None
stop = False
samples = []


def poll():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
        keep = [ln.strip() for ln in r.splitlines() if any(s in ln for s in ("sclk", "Power", "junction", "mclk"))]
        samples.append(" | ".join(keep))
        time.sleep(1.0)


th = threading.Thread(target=poll)
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 6.0:
    for _ in range(20):
        if what == "attn_bwd":
            ops.attn_bwd(do, q, k, v, o, lse, True)
        elif what == "attn_fwd":
            ops.attn_fwd(q, k, v, True)
        elif what == "gemm":
            ops.linear_fwd(a, w)
        elif what.startswith("gemm"):     # gemm259 / gemm280 / gemm261: one kernel family (tile code) on the packed gate|up forward shape
            with ops.gemm_variant(int(what[4:])):
                ops.linear_fwd(a, w2)
        else:
            time.sleep(0.01)
    torch.cuda.synchronize()
    n += 20
stop = True
th.join()
print(what, "iters", n, "ms/iter", round((time.time() - t0) / n * 1e3, 3))
for s in samples:
    print("  ", s)
