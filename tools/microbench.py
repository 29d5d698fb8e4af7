"""Per-kernel micro-benchmarks on one MI355X (run through gpurun).  Prints one JSON line per kernel/shape with the
achieved TFLOP/s or GB/s and the roofline fraction (peaks from /opt/skills/guides/MI355X_MICROARCH.md: 2.5 PF dense
bf16 MFMA, 8 TB/s HBM3E).  Random (not zero) operands, as the guide requires.

    python tools/microbench.py [--only gemm,attn,norm,...] [--out gpurun_out/microbench.jsonl]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamllm_amd import ops  # noqa: E402

PEAK_TF = 2500.0
PEAK_GBS = 8000.0
BF = torch.bfloat16


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def emit(out, **kw):
    line = json.dumps(kw)
    print(line, flush=True)
    if out:
        with open(out, "a") as f:
            f.write(line + "\n")


def bench_gemm(out):
    T = 32768
    shapes = [
        ("qkv/o fwd", "fwd", T, 4096, 4096), ("gate/up fwd", "fwd", T, 11008, 4096), ("down fwd", "fwd", T, 4096, 11008),
        ("lm_head fwd", "fwd", T, 32008, 4096),
        ("o dgrad", "dgrad", T, 4096, 4096), ("up dgrad", "dgrad", T, 11008, 4096), ("down dgrad", "dgrad", T, 4096, 11008),
        ("o wgrad", "wgrad", T, 4096, 4096), ("up wgrad", "wgrad", T, 11008, 4096), ("down wgrad", "wgrad", T, 4096, 11008),
        ("square 4096", "fwd", 4096, 4096, 4096), ("square 8192", "fwd", 8192, 8192, 8192),
    ]
    for name, kind, M, N, K in shapes:
        x = torch.randn(M, K, device="cuda").to(BF)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
        dy = torch.randn(M, N, device="cuda").to(BF)
        if kind == "fwd":
            fn = lambda: ops.linear_fwd(x, w)
            ref = lambda: torch.nn.functional.linear(x, w)
        elif kind == "dgrad":
            fn = lambda: ops.linear_dgrad(dy, w)
            ref = lambda: dy @ w
        else:
            fn = lambda: ops.linear_wgrad(dy, x)
            ref = lambda: dy.t() @ x
        ms = timeit(fn)
        ms_ref = timeit(ref)
        tf = 2.0 * M * N * K / ms / 1e9
        emit(out, kernel="gemm_bf16", name=name, kind=kind, M=M, N=N, K=K, ms=round(ms, 4), tflops=round(tf, 1),
             frac_mfma_peak=round(tf / PEAK_TF, 4), hipblaslt_ms=round(ms_ref, 4),
             hipblaslt_tflops=round(2.0 * M * N * K / ms_ref / 1e9, 1))


def bench_attn(out):
    for name, B, H, S, D, causal in [("llm causal", 16, 32, 2048, 128, True), ("clip", 32, 16, 257, 64, False),
                                     ("unet self 4096", 32, 5, 4096, 64, False), ("unet self 1024", 32, 10, 1024, 64, False)]:
        q = torch.randn(B, S, H, D, device="cuda").to(BF)
        k = torch.randn(B, S, H, D, device="cuda").to(BF)
        v = torch.randn(B, S, H, D, device="cuda").to(BF)
        ms = timeit(lambda: ops.attn_fwd(q, k, v, causal))
        fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
        tf = fl / ms / 1e9
        emit(out, kernel="attn_fwd", name=name, B=B, H=H, S=S, D=D, causal=causal, ms=round(ms, 4), tflops=round(tf, 1),
             frac_mfma_peak=round(tf / PEAK_TF, 4))
        if hasattr(ops, "attn_bwd") and "dllm_attn_bwd" in dir(__import__("dreamllm_amd")._lib.lib()):
            pass


def bench_attn_bwd(out):
    from dreamllm_amd import _lib
    if getattr(_lib.lib(), "dllm_attn_bwd", None) is None:
        return
    for name, B, H, S, D, causal in [("llm causal", 16, 32, 2048, 128, True), ("unet self 4096", 32, 5, 4096, 64, False)]:
        q = torch.randn(B, S, H, D, device="cuda").to(BF)
        k = torch.randn(B, S, H, D, device="cuda").to(BF)
        v = torch.randn(B, S, H, D, device="cuda").to(BF)
        do = torch.randn(B, S, H, D, device="cuda").to(BF)
        o, lse = ops.attn_fwd(q, k, v, causal)
        ms = timeit(lambda: ops.attn_bwd(do, q, k, v, o, lse, causal))
        fl = 10.0 * B * H * S * S * D * (0.5 if causal else 1.0)
        tf = fl / ms / 1e9
        emit(out, kernel="attn_bwd", name=name, B=B, H=H, S=S, D=D, causal=causal, ms=round(ms, 4),
             tflops_algorithmic=round(tf, 1), frac_mfma_peak=round(tf / PEAK_TF, 4))


def bench_conv(out):
    """3x3 NHWC implicit-GEMM conv at the SD-2.1 UNet training shapes (32 images)."""
    for name, N, H, C, CO in [("unet L0 320->320 @64", 32, 64, 320, 320), ("unet L1 640->640 @32", 32, 32, 640, 640),
                              ("unet L2 1280->1280 @16", 32, 16, 1280, 1280), ("unet up 2560->1280 @16", 32, 16, 2560, 1280),
                              ("unet up 960->320 @64", 32, 64, 960, 320), ("vae 128->128 @512", 4, 512, 128, 128),
                              ("unet L0 batch2", 2, 64, 320, 320), ("unet L3 batch2", 2, 8, 1280, 1280)]:
        x = torch.randn(N, H, H, C, device="cuda").to(BF)
        w = (torch.randn(CO, 9 * C, device="cuda") * 0.02).to(BF)
        b = torch.zeros(CO, device="cuda", dtype=BF)
        ms = timeit(lambda: ops.conv2d_nhwc(x, w, CO, 3, 3, bias=b))
        tf = 2.0 * N * H * H * CO * 9 * C / ms / 1e9
        emit(out, kernel="conv3x3_nhwc", name=name, N=N, H=H, C=C, CO=CO, ms=round(ms, 4), tflops=round(tf, 1),
             frac_mfma_peak=round(tf / PEAK_TF, 4))


def bench_norm(out):
    rows, D = 32768, 4096
    x = torch.randn(rows, D, device="cuda").to(BF)
    r = torch.randn(rows, D, device="cuda").to(BF)
    w = torch.ones(D, device="cuda", dtype=BF)
    dy = torch.randn(rows, D, device="cuda").to(BF)
    ms = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-6))
    gb = rows * D * 4 / ms / 1e6
    emit(out, kernel="rmsnorm_fwd", rows=rows, D=D, ms=round(ms, 4), gbs=round(gb, 1), frac_hbm_peak=round(gb / PEAK_GBS, 4))
    ms = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-6, residual=r))
    gb = rows * D * 8 / ms / 1e6
    emit(out, kernel="add_rmsnorm_fwd", rows=rows, D=D, ms=round(ms, 4), gbs=round(gb, 1),
         frac_hbm_peak=round(gb / PEAK_GBS, 4))
    _, _, rstd = ops.rmsnorm_fwd(x, w, 1e-6)
    ms = timeit(lambda: ops.rmsnorm_bwd(dy, x, w, rstd))
    gb = rows * D * 6 / ms / 1e6
    emit(out, kernel="rmsnorm_bwd", rows=rows, D=D, ms=round(ms, 4), gbs=round(gb, 1), frac_hbm_peak=round(gb / PEAK_GBS, 4))


def bench_elementwise(out):
    M, Fd = 32768, 11008
    gu = torch.randn(M, 2 * Fd, device="cuda").to(BF)
    ms = timeit(lambda: ops.glu_fwd(gu[:, :Fd], gu[:, Fd:], 0))
    gb = M * Fd * 6 / ms / 1e6
    emit(out, kernel="swiglu_fwd", M=M, F=Fd, ms=round(ms, 4), gbs=round(gb, 1), frac_hbm_peak=round(gb / PEAK_GBS, 4))
    B, S, H, D = 16, 2048, 32, 128
    q = torch.randn(B, S, H, D, device="cuda").to(BF)
    cos = torch.randn(2048, 64, device="cuda")
    sin = torch.randn(2048, 64, device="cuda")
    ms = timeit(lambda: ops.rope_(q, cos, sin))
    gb = q.numel() * 4 / ms / 1e6
    emit(out, kernel="rope", B=B, S=S, H=H, D=D, ms=round(ms, 4), gbs=round(gb, 1), frac_hbm_peak=round(gb / PEAK_GBS, 4))
    R, V = 8192, 32008
    logits = torch.randn(R, V, device="cuda")
    labels = torch.randint(0, V, (R,), device="cuda")
    dl = torch.empty(R, V, dtype=BF, device="cuda")
    gs = torch.ones(1, device="cuda")
    ms = timeit(lambda: ops.cross_entropy_rows(logits, labels, dlogits=dl, gscale=gs))
    gb = R * V * 6 / ms / 1e6
    emit(out, kernel="cross_entropy_fwd_bwd", R=R, V=V, ms=round(ms, 4), gbs=round(gb, 1),
         frac_hbm_peak=round(gb / PEAK_GBS, 4))


def bench_gemv(out):
    """Decode-step GEMVs at the 7B layer shapes (weights streamed once: HBM-bound)."""
    for name, N, K in [("o_proj", 4096, 4096), ("down_proj", 4096, 11008), ("lm_head", 32008, 4096)]:
        x = torch.randn(1, K, device="cuda").to(BF)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
        ms = timeit(lambda: ops.gemv(x, w), iters=50, warmup=5)
        gb = N * K * 2 / ms / 1e6
        emit(out, kernel="gemv", name=name, N=N, K=K, ms=round(ms, 4), gbs=round(gb, 1), frac_hbm_peak=round(gb / PEAK_GBS, 4))
    K = 4096
    x = torch.randn(1, K, device="cuda").to(BF)
    nw = torch.ones(K, device="cuda", dtype=BF)
    ws = [(torch.randn(4096, K, device="cuda") * 0.02).to(BF) for _ in range(3)]
    ms = timeit(lambda: ops.gemv_fused(x, ws, norm_w=nw, eps=1e-5), iters=50, warmup=5)
    gb = 3 * 4096 * K * 2 / ms / 1e6
    emit(out, kernel="gemv_fused", name="rmsnorm+qkv", N=12288, K=K, ms=round(ms, 4), gbs=round(gb, 1), frac_hbm_peak=round(gb / PEAK_GBS, 4))
    wg, wu = [(torch.randn(11008, K, device="cuda") * 0.02).to(BF) for _ in range(2)]
    ms = timeit(lambda: ops.gemv_fused(x, (wg, wu), norm_w=nw, eps=1e-5, swiglu=True), iters=50, warmup=5)
    gb = 2 * 11008 * K * 2 / ms / 1e6
    emit(out, kernel="gemv_fused", name="rmsnorm+gate/up+swiglu", N=11008, K=K, ms=round(ms, 4), gbs=round(gb, 1),
         frac_hbm_peak=round(gb / PEAK_GBS, 4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--tile", type=int, default=0, help="force the GEMM block tile (128/256; 0 = automatic)")
    a = ap.parse_args()
    sel = set(a.only.split(",")) if a.only else None
    from dreamllm_amd import _lib
    ops.GEMM_VARIANT = a.tile  # per-call kernel variant
    if a.out and os.path.exists(a.out):
        os.remove(a.out)
    benches = dict(conv=bench_conv, norm=bench_norm, elementwise=bench_elementwise, attn=bench_attn, attn_bwd=bench_attn_bwd, gemm=bench_gemm, gemv=bench_gemv)
    for name, fn in benches.items():
        if sel is None or name in sel:
            try:
                fn(a.out)
            except Exception as ex:  # keep going: this is a diagnostics tool
                emit(a.out, kernel=name, error=repr(ex))


if __name__ == "__main__":
    main()
