"""Attention forward / backward timings at the bench shape (B16 x S2048 x H32 x D128 causal) and the UNet shapes, per kernel
variant (ops.ATTN_VARIANT: 1 = 4-wave, 2 = 8-wave pipelined, 3 = ping-pong forward / ping-pong dQ).  Interleaved rounds in one process (guide rule 24)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

BF = torch.bfloat16


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


shapes = [("llm causal d128", 16, 2048, 2048, 32, 128, True), ("llm ragged-ish S=1536", 16, 1536, 1536, 32, 128, True),
          ("unet self 4096 d64", 16, 4096, 4096, 5, 64, False), ("unet self 1024 d64", 16, 1024, 1024, 10, 64, False),
          ("clip 257 d64", 32, 257, 257, 16, 64, False)]
for name, B, Sq, Sk, H, D, causal in shapes:
    q, k, v, do = (torch.randn(B, max(Sq, Sk), H, D, device="cuda").to(BF) for _ in range(4))
    q, do = q[:, :Sq], do[:, :Sq]
    k, v = k[:, :Sk], v[:, :Sk]
    flops = 4 * Sq * Sk * D * H * B / (2 if causal else 1)
    res = {}
    for rnd in range(2):
        for var in (1, 2, 3):
            ops.ATTN_VARIANT = var
            res.setdefault(var, []).append(timed(lambda: ops.attn_fwd(q, k, v, causal)))
    ops.ATTN_VARIANT = 0
    o, lse = ops.attn_fwd(q, k, v, causal)
    resb = {}
    for rnd in range(2):
        for var in (1, 2, 3):
            ops.ATTN_VARIANT = var
            resb.setdefault(var, []).append(timed(lambda: ops.attn_bwd(do, q, k, v, o, lse, causal)))
    ops.ATTN_VARIANT = 0
    line = f"{name:26s} " + "  ".join(f"fwd[v{var}] {min(ts):7.3f} ms {flops / min(ts) / 1e9:6.0f} TF" for var, ts in res.items())
    line += "  " + "  ".join(f"bwd[v{var}] {min(ts):7.3f} ms {2.5 * flops / min(ts) / 1e9:6.0f} TF(alg)" for var, ts in resb.items())
    print(line, flush=True)
