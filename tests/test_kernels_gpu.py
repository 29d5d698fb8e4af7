"""`-m gpu` parity tests of the raw HIP kernels (called through the C-ABI) against the fp32 oracle on identical inputs.

Tolerances (relative L2 against the fp32 oracle evaluated on the SAME bf16-representable inputs):
  * bf16-output kernels: 4e-3  (one bf16 rounding is 2^-9/sqrt(3) = 1.1e-3 RMS; GEMM/attention add the rounding of the
    bf16 probabilities / a second rounding in RMSNorm) -- every assert states its own bound;
  * fp32-output kernels (logits, LSE, loss, rstd): 1e-4 or tighter.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF = torch.bfloat16


def bf16r(t):
    return t.to(BF).float()


def _ops():
    from dreamllm_amd import ops
    return ops


def rnd(*shape, scale=1.0, seed=None):
    if seed is not None:
        torch.manual_seed(seed)
    return (torch.randn(*shape) * scale).to(BF)


# ----------------------------------------------------------------------------- hardware-convention probes
def test_probe_tr16_semantics():
    """ds_read_b64_tr_b16: within a 16-lane group, lane i receives column i of the 4x16 row-major block whose 16
    8-byte chunks are addressed by the group's lanes (chunk t = row t>>2, cols (t&3)*4..+3)."""
    import ctypes
    from dreamllm_amd import _lib
    src = torch.arange(256, dtype=torch.int16, device=DEV)
    out = torch.empty(256, dtype=torch.int16, device=DEV)
    _lib.check("dllm_probe_tr16", ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()),
               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = out.cpu().view(64, 4)
    exp = torch.empty(64, 4, dtype=torch.int16)
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        for j in range(4):
            # block of group g starts at element g*64; row j (16 elements per row), column i
            exp[lane, j] = g * 64 + j * 16 + i
    assert torch.equal(got, exp), f"tr16 layout differs:\n{got[:20]}"


def test_probe_mfma16_layout():
    import ctypes
    from dreamllm_amd import _lib
    torch.manual_seed(1)
    a = torch.randn(16, 32).to(BF)
    b = torch.randn(16, 32).to(BF)  # [col][k]
    out = torch.empty(256, dtype=torch.float32, device=DEV)
    ad, bd = a.to(DEV), b.to(DEV)
    _lib.check("dllm_probe_mfma16", ctypes.c_void_p(ad.data_ptr()), ctypes.c_void_p(bd.data_ptr()),
               ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = out.cpu().view(64, 4)
    ref = a.float() @ b.float().t()  # [row][col]
    exp = torch.empty(64, 4)
    for lane in range(64):
        for r in range(4):
            exp[lane, r] = ref[(lane >> 4) * 4 + r, lane & 15]
    assert rel_l2(got, exp) < 1e-5


# ----------------------------------------------------------------------------- norms
@pytest.mark.parametrize("rows,D", [(5, 96), (33, 1024), (64, 4096), (7, 5120), (130, 320)])
def test_rmsnorm_fwd_bwd(rows, D):
    ops = _ops()
    torch.manual_seed(rows + D)
    x = rnd(rows, D)
    w = (1.0 + 0.1 * torch.randn(D)).to(BF)
    dy = rnd(rows, D)
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    h = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6)
    yr = wr * h
    yr.backward(dy.float())
    y, _, rstd = ops.rmsnorm_fwd(x.to(DEV), w.to(DEV), 1e-6)
    assert rel_l2(y, yr) < 4e-3
    dx, dw = ops.rmsnorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), rstd)
    assert rel_l2(dx, xr.grad) < 4e-3
    assert rel_l2(dw, wr.grad) < 6e-3


def test_rmsnorm_golden(golden):
    """Bit-level check against the reference's own bf16 output (two roundings reproduced in the kernel)."""
    ops = _ops()
    g = golden("rmsnorm.pt")
    y, _, _ = ops.rmsnorm_fwd(g["x"].to(BF).to(DEV), g["w"].to(BF).to(DEV), g["eps"])
    ref_bf16 = g["y_bf16"]
    mism = (y.cpu().float() - ref_bf16.float()).abs() > 0
    # identical up to <=1 bf16 ulp on a handful of elements (reduction order of the fp32 mean)
    assert mism.float().mean() < 0.02
    assert rel_l2(y, g["y"]) < 4e-3


def test_add_rmsnorm_fused():
    ops = _ops()
    x, r = rnd(40, 4096, seed=3), rnd(40, 4096)
    w = (1.0 + 0.1 * torch.randn(4096)).to(BF)
    y, h, rstd = ops.rmsnorm_fwd(x.to(DEV), w.to(DEV), 1e-5, residual=r.to(DEV))
    href = (x.float() + r.float()).to(BF)
    assert torch.equal(h.cpu(), href)
    hr = href.float()
    yr = w.float() * (hr * torch.rsqrt(hr.pow(2).mean(-1, keepdim=True) + 1e-5))
    assert rel_l2(y, yr) < 4e-3


@pytest.mark.parametrize("rows,D", [(9, 1024), (300, 320), (64, 1280)])
def test_layernorm_fwd_bwd(rows, D):
    ops = _ops()
    torch.manual_seed(D)
    x = rnd(rows, D)
    w = (1.0 + 0.1 * torch.randn(D)).to(BF)
    b = (0.1 * torch.randn(D)).to(BF)
    dy = rnd(rows, D)
    xr, wr, br = x.float().requires_grad_(True), w.float().requires_grad_(True), b.float().requires_grad_(True)
    yr = F.layer_norm(xr, (D,), wr, br, 1e-5)
    yr.backward(dy.float())
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5)
    assert rel_l2(y, yr) < 4e-3
    dx, dw, db = ops.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd)
    assert rel_l2(dx, xr.grad) < 4e-3
    assert rel_l2(dw, wr.grad) < 6e-3
    assert rel_l2(db, br.grad) < 6e-3


# ----------------------------------------------------------------------------- GEMM
@pytest.fixture(params=[128, 256, 257, 259, 261, 262, 264, 266, 280])
def gemm_tile(request):
    """run the GEMM tests once per block-tile variant (128x128 / 4 waves and 256x256 / 8 waves)"""
    from dreamllm_amd import ops
    with ops.gemm_variant(request.param):
        yield request.param


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (130, 200, 264), (1, 64, 8), (777, 1000, 1032),
                                   (512, 32008, 256)])
def test_gemm_nt_forward(M, N, K, gemm_tile):
    ops = _ops()
    torch.manual_seed(M + N + K)
    x, w = rnd(M, K), rnd(N, K, scale=0.05)
    ref = x.float() @ w.float().t()
    y = ops.linear_fwd(x.to(DEV), w.to(DEV))
    assert y.shape == (M, N)
    assert rel_l2(y, ref) < 4e-3
    y32 = ops.linear_fwd(x.to(DEV), w.to(DEV), out_dtype=torch.float32)
    assert rel_l2(y32, ref) < 1e-5


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 264, 520), (64, 4096, 1024), (1000, 8, 136)])
def test_gemm_dgrad_wgrad(M, N, K, gemm_tile):
    """dx = dy W (A_K, B_N: transpose reads on the weight) and dW = dy^T x (A_M, B_N: transpose reads on both)."""
    ops = _ops()
    torch.manual_seed(M * 3 + N)
    dy, w, x = rnd(M, N), rnd(N, K, scale=0.05), rnd(M, K)
    dx = ops.linear_dgrad(dy.to(DEV), w.to(DEV))
    assert rel_l2(dx, dy.float() @ w.float()) < 4e-3
    dw = ops.linear_wgrad(dy.to(DEV), x.to(DEV))
    assert dw.shape == (N, K)
    assert rel_l2(dw, dy.float().t() @ x.float()) < 4e-3
    dw32 = ops.linear_wgrad(dy.to(DEV), x.to(DEV), out_dtype=torch.float32)
    assert rel_l2(dw32, dy.float().t() @ x.float()) < 1e-5
    acc = dw32.clone()
    ops.linear_wgrad(dy.to(DEV), x.to(DEV), out=acc, accumulate=True)
    assert rel_l2(acc, 2 * (dy.float().t() @ x.float())) < 1e-5


@pytest.mark.parametrize("epi", [None, "gelu", "quick_gelu", "silu"])
def test_gemm_epilogues(epi, gemm_tile):
    ops = _ops()
    torch.manual_seed(5)
    M, N, K = 200, 328, 256
    x, w, b, r = rnd(M, K), rnd(N, K, scale=0.06), rnd(N), rnd(M, N)
    z = x.float() @ w.float().t() + b.float()
    if epi == "gelu":
        z = F.gelu(z)
    elif epi == "quick_gelu":
        z = z * torch.sigmoid(1.702 * z)
    elif epi == "silu":
        z = F.silu(z)
    ref = z + r.float()
    y = ops.linear_fwd(x.to(DEV), w.to(DEV), bias=b.to(DEV), epi=epi, residual=r.to(DEV))
    assert rel_l2(y, ref) < 4e-3


@pytest.mark.parametrize("variant", ["plain", "bias_act_res", "accumulate_alpha"])
def test_gemm_full_tile_epilogue_and_persistent_walk(variant, gemm_tile):
    """5888 x 5888 outputs = 23 x 23 full 256-tiles (529 > 2 x 256 CUs: several rounds of blocks per CU, and the 256-tile
    kernels take the LDS-staged epilogue on every tile), K = 192 = 3 K tiles (odd number of LDS stage flips), with the epilogue
    variants the LLM uses."""
    ops = _ops()
    torch.manual_seed(11)
    M = N = 5888
    K = 192
    x, w = rnd(M, K), rnd(N, K, scale=0.05)
    ref = x.float() @ w.float().t()
    xd, wd = x.to(DEV), w.to(DEV)
    if variant == "plain":
        y = ops.linear_fwd(xd, wd)
    elif variant == "bias_act_res":
        b, r = rnd(N), rnd(M, N)
        ref = F.silu(ref + b.float()) + r.float()
        y = ops.linear_fwd(xd, wd, bias=b.to(DEV), epi="silu", residual=r.to(DEV))
    else:
        c0 = rnd(M, N)
        y = c0.to(DEV).clone()
        ops.gemm(xd, wd, M, N, K, K, K, 0, 0, out=y, accumulate=True, alpha=0.5)
        ref = 0.5 * ref + c0.float()
    assert rel_l2(y, ref) < 4e-3
    # every tile written exactly once: no stale / missing tile anywhere
    err = (y.float().cpu() - ref).abs().view(23, 256, 23, 256).amax(dim=(1, 3))
    assert float(err.max()) < 0.25, err


@pytest.mark.parametrize("layout,M,N,K", [
    ("fwd", 5632, 4096, 16384),    # 22 x 16 = 352 tiles: one whole round + 96 tail tiles (the gate|up weight gradient's remainder)
    ("fwd", 6100, 4096, 20480),    # 24 x 16 = 384 tiles, ragged last row of tiles: 128 tail tiles
    ("dgrad", 5632, 4096, 16384),
    ("wgrad", 4096, 5632, 16384),  # dW[N=4096 x K'=5632] over T = 16384 tokens
])
def test_gemm_streamk_tail(layout, M, N, K):
    """Stream-K tail of the pipelined 256-tile kernel (include/dreamllm_hip.h): whole rounds as usual, the last partial round's K
    loops spread evenly over the CUs through fp32 slabs + a fix-up launch.  Against the fp32 product, against the unsplit launch
    (same kernel, K sum re-associated in fp32 only), run-to-run bit-identical, with the epilogue variants the weight gradients use
    (fp32 output with accumulate; bf16 output with bias + residual)."""
    ops = _ops()
    from dreamllm_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    rn = lambda *s, scale=1.0: (torch.randn(*s, device=DEV, generator=g) * scale).to(BF)
    la, lb = {"fwd": (0, 0), "dgrad": (0, 1), "wgrad": (1, 1)}[layout]

    def run(streamk, **kw):
        ops.STREAMK = streamk
        try:
            if layout == "fwd":
                return ops.linear_fwd(a, b, **kw)
            if layout == "dgrad":
                return ops.linear_dgrad(a, b)
            return ops.linear_wgrad(a, b, **kw)
        finally:
            ops.STREAMK = True

    sc = 1.0 / math.sqrt(K)
    if layout == "fwd":
        a, b = rn(M, K), rn(N, K, scale=sc)
        ref = a.float() @ b.float().t()
    elif layout == "dgrad":      # dx[M, N] = dy[M, K] W[K, N]
        a, b = rn(M, K), rn(K, N, scale=sc)
        ref = a.float() @ b.float()
    else:                        # dW[M, N] = dy[T, M]^T x[T, N] with T = K
        a, b = rn(K, M), rn(K, N, scale=sc)
        ref = a.float().t() @ b.float()
    assert _lib.call("dllm_gemm_streamk_hint", M, N, K, la, lb) == 1, "the shape must take the stream-K path"
    y1 = run(True)
    y0 = run(False)
    assert rel_l2(y1, ref) < 4e-3 and rel_l2(y0, ref) < 4e-3
    assert rel_l2(y1, y0.float()) < 2e-3                       # same products, fp32 re-association + one bf16 rounding
    assert torch.equal(run(True), y1)                          # fixed summation order: deterministic
    # no stale / missing tile anywhere (tail tiles come from the fix-up launch)
    tm, tn = -(-y1.shape[0] // 256), -(-y1.shape[1] // 256)
    e = torch.zeros(tm * 256, tn * 256, device=DEV)
    e[:y1.shape[0], :y1.shape[1]] = (y1.float() - ref).abs()
    assert float(e.view(tm, 256, tn, 256).amax(dim=(1, 3)).max()) < 0.05 * float(ref.abs().max())
    if layout == "wgrad":      # fp32 accumulate epilogue through the fix-up kernel (the fused lm_head + CE accumulates dW like this)
        acc = torch.ones(M, N, dtype=torch.float32, device=DEV)
        run(True, out=acc, accumulate=True, out_dtype=torch.float32)
        assert rel_l2(acc, ref + 1.0) < 1e-5
    if layout == "fwd":        # bias + residual epilogue on the tail tiles
        bias, res = rn(N), rn(M, N)
        yb = run(True, bias=bias, residual=res)
        assert rel_l2(yb, ref + bias.float() + res.float()) < 4e-3


@pytest.mark.parametrize("layout,M,N,K", [("fwd", 8192, 8192, 512), ("fwd", 8448, 8448, 256), ("dgrad", 8192, 8192, 512),
                                          ("wgrad", 8192, 8448, 2048)])
def test_gemm_persistent_xcd_synchronised_walk(layout, M, N, K):
    """Opt-in variant (bit 24 of `variant`, ops.GEMM_PERSIST): one block per CU walks its XCD's tiles, the 32 blocks of an XCD start
    every tile together.  Same tiles, same arithmetic: results must be BIT-IDENTICAL to the one-tile-per-block launch (1089 tiles:
    a tile count that is not a multiple of 8, so the XCDs own different numbers of tiles), and the counter page must be left at
    zero (a second launch behaves the same)."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    rn = lambda *s, scale=1.0: (torch.randn(*s, device=DEV, generator=g) * scale).to(BF)
    if layout == "fwd":
        a, b = rn(M, K), rn(N, K, scale=0.05)
        fn = lambda: ops.linear_fwd(a, b)
    elif layout == "dgrad":
        a, b = rn(M, K), rn(K, N, scale=0.05)
        fn = lambda: ops.linear_dgrad(a, b)
    else:
        a, b = rn(K, M), rn(K, N)
        fn = lambda: ops.linear_wgrad(a, b)
    y0 = fn()
    ops.GEMM_PERSIST = True
    try:
        y1 = fn()
        y2 = fn()
    finally:
        ops.GEMM_PERSIST = False
    assert torch.equal(y1, y0) and torch.equal(y2, y0)
    ws = ops._streamk_workspace(a.device)
    assert int(ws[:1024].view(torch.int32).abs().sum()) == 0


def test_gemm_streamk_hint_only_where_it_pays():
    from dreamllm_amd import _lib
    T, d, F_ = 32768, 4096, 11008
    h = lambda M, N, K, la, lb: _lib.call("dllm_gemm_streamk_hint", M, N, K, la, lb)
    assert h(2 * F_, d, T, 1, 1) == 1                                  # gate|up weight gradient: 1376 tiles = 5 rounds + 96 (measured +8.6 %)
    assert h(d, F_, T, 1, 1) == 0                                      # down weight gradient: remainder 176 tiles, bandwidth-bound tail (-4 %)
    assert h(T, F_, d, 0, 1) == 0                                      # down dgrad: remainder 128 but K = 4096: the half round is 50 us
    assert h(T, d, d, 0, 0) == 0 and h(T, 2 * F_, d, 0, 0) == 0        # whole rounds: nothing to gain
    assert h(3 * d, d, T, 1, 1) == 0 and h(256, 256, 4096, 0, 0) == 0
    assert _lib.call("dllm_gemm_streamk_ws_bytes") == (2 * 256 * 256 * 256 + 1024) * 4   # slabs + the 4 KiB counter page


def test_gemm_rejects_bad_shapes():
    ops = _ops()
    x, w = rnd(8, 12).to(DEV), rnd(16, 12).to(DEV)  # K = 12 not a multiple of 8
    with pytest.raises(ValueError):
        ops.linear_fwd(x, w)
    with pytest.raises(RuntimeError):
        ops.linear_fwd(rnd(8, 16), rnd(16, 16))  # CPU tensors: no fallback


# ----------------------------------------------------------------------------- attention
def attn_ref(q, k, v, causal, seqlen=None):
    """fp32 oracle; q [B,Sq,H,D], k/v [B,Sk,Hkv,D] -> o [B,Sq,H,D], lse [B,H,Sq]."""
    B, Sq, H, D = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    qf, kf, vf = q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2)
    if Hkv != H:
        kf = kf.repeat_interleave(H // Hkv, 1)
        vf = vf.repeat_interleave(H // Hkv, 1)
    s = qf @ kf.transpose(2, 3) / math.sqrt(D)
    if causal:
        m = torch.ones(Sq, Sk, dtype=torch.bool).tril(Sk - Sq)
        s = s.masked_fill(~m, float("-inf"))
    if seqlen is not None:
        for b, L in enumerate(seqlen):
            s[b, :, :, L:] = float("-inf")
    lse = torch.logsumexp(s, -1)
    o = torch.softmax(s, -1) @ vf
    return o.transpose(1, 2), lse


@pytest.fixture(params=[1, 2, 3], ids=["4wave", "8wave", "pingpong"])
def attn_variant(request):
    """every attention test runs on all kernel families (4-wave blocks / 8-wave pipelined 256-row blocks / the round-5 ping-pong
    forward, which pairs with the 8-wave backward)"""
    from dreamllm_amd import ops
    ops.ATTN_VARIANT = request.param
    yield request.param
    ops.ATTN_VARIANT = 0


@pytest.mark.parametrize("B,H,Hkv,Sq,Sk,D,causal", [
    (2, 4, 4, 128, 128, 128, True),
    (1, 2, 2, 300, 300, 128, True),
    (2, 3, 3, 257, 257, 64, False),    # CLIP-ViT
    (2, 5, 5, 192, 64, 64, False),     # UNet cross-attention over the 64 dream tokens
    (1, 4, 2, 160, 160, 128, True),    # GQA
    (1, 2, 2, 64, 200, 64, False),
    (1, 1, 1, 1024, 1024, 64, False),
    (1, 2, 2, 777, 777, 128, True),    # several 256-query blocks, ragged tail
    (2, 2, 1, 513, 513, 64, True),
    (1, 2, 2, 40, 600, 128, True),     # KV cache: few queries at the end of a long key axis
    (1, 5, 5, 600, 600, 128, False),   # 5 heads: the XCD-aware work-group map pads the grid to 8 heads
    (3, 3, 1, 520, 520, 64, True),     # GQA group of 3, 9 query heads / 3 key heads
])
def test_attn_fwd(B, H, Hkv, Sq, Sk, D, causal, attn_variant):
    ops = _ops()
    torch.manual_seed(Sq + Sk + D)
    q, k, v = rnd(B, Sq, H, D), rnd(B, Sk, Hkv, D), rnd(B, Sk, Hkv, D)
    oref, lref = attn_ref(q, k, v, causal)
    o, lse = ops.attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), causal)
    assert rel_l2(o, oref) < 6e-3   # bf16 output rounding + bf16 probabilities in the PV MFMA
    assert rel_l2(lse, lref) < 1e-4


@pytest.mark.parametrize("B,H,Hkv,Sq,Sk,D", [
    (1, 2, 2, 1024, 512, 128),   # the first two 256-row query blocks have no visible key at all
    (2, 3, 3, 700, 130, 64),
    (1, 2, 1, 300, 40, 128),
])
def test_attn_causal_more_queries_than_keys(B, H, Hkv, Sq, Sk, D, attn_variant):
    """Causal with Sq > Sk (bottom-right aligned, coff = Sk - Sq < 0): query rows in front of the first key see nothing and are
    written as zeros (lse 0, zero gradients); the other rows match the fp32 oracle -- forward and backward."""
    ops = _ops()
    torch.manual_seed(Sq + 3 * Sk + D)
    q, k, v, do = rnd(B, Sq, H, D), rnd(B, Sk, Hkv, D), rnd(B, Sk, Hkv, D), rnd(B, Sq, H, D)
    dead = Sq - Sk   # rows [0, dead) have no visible key
    qr, kr, vr = (x.float().requires_grad_(True) for x in (q, k, v))
    oref, lref = attn_ref(qr[:, dead:], kr, vr, True)   # the live rows form an ordinary causal Sk x Sk problem
    oref.backward(do[:, dead:].float())
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o, lse = ops.attn_fwd(qd, kd, vd, True)
    assert torch.count_nonzero(o[:, :dead]) == 0 and torch.count_nonzero(lse[:, :, :dead]) == 0
    assert rel_l2(o[:, dead:], oref) < 6e-3
    assert rel_l2(lse[:, :, dead:], lref) < 3e-4   # short key axes: small |lse|, the absolute error of exp2 / log is unchanged
    dq, dk, dv = ops.attn_bwd(do.to(DEV), qd, kd, vd, o, lse, True)
    assert torch.count_nonzero(dq[:, :dead]) == 0
    assert rel_l2(dq[:, dead:], qr.grad[:, dead:]) < 1.5e-2
    assert rel_l2(dk, kr.grad) < 1.5e-2
    assert rel_l2(dv, vr.grad) < 1.5e-2


@pytest.mark.parametrize("Sq", [64, 600])
def test_attn_fwd_empty_key_axis(Sq, attn_variant):
    """No visible key for anybody: Sk == 0, and a KV cache whose first-valid-key index lies at the end of the key axis."""
    ops = _ops()
    B, H, D = 2, 2, 128
    q = rnd(B, Sq, H, D, seed=3).to(DEV)
    k0 = torch.empty(B, 0, H, D, dtype=BF, device=DEV)
    for causal in (False, True):
        o, lse = ops.attn_fwd(q, k0, k0, causal)
        assert torch.count_nonzero(o) == 0 and torch.count_nonzero(lse) == 0
    Sk = Sq + 64
    k, v = rnd(B, Sk, H, D).to(DEV), rnd(B, Sk, H, D).to(DEV)
    start = torch.tensor([Sk, 0], dtype=torch.int32, device=DEV)   # row 0: every key is padding; row 1: ordinary
    o, lse = ops.attn_fwd(q, k, v, True, seqstart=start)
    assert torch.count_nonzero(o[0]) == 0 and torch.count_nonzero(lse[0]) == 0
    oref, lref = attn_ref(q[1:].cpu(), k[1:].cpu(), v[1:].cpu(), True)
    assert rel_l2(o[1:], oref) < 6e-3 and rel_l2(lse[1:], lref) < 1e-4


def test_attn_huge_key_stride_takes_the_64bit_kernels():
    """The ping-pong kernels address a (batch, head) key axis with 32-bit byte offsets: Sk * k_ss must stay below 2^29 elements.
    A strided view beyond that (every 2^19-th token row of a large buffer) must take the 8-wave kernels, automatically AND when
    the ping-pong family is forced, with unchanged results."""
    ops = _ops()
    B, S, H, D = 1, 1024, 1, 128
    pitch = 1 << 19                               # elements between consecutive tokens: S * pitch = 2^29
    big = torch.zeros(S * pitch + H * D, dtype=BF, device=DEV)   # 1 GiB
    torch.manual_seed(5)
    q, do = rnd(B, S, H, D).to(DEV), rnd(B, S, H, D).to(DEV)
    kc, vc = rnd(B, S, H, D).to(DEV), rnd(B, S, H, D).to(DEV)
    k = torch.as_strided(big, (B, S, H, D), (0, pitch, D, 1))
    k.copy_(kc)
    vbig = torch.zeros_like(big)
    v = torch.as_strided(vbig, (B, S, H, D), (0, pitch, D, 1))
    v.copy_(vc)
    ops.ATTN_VARIANT = 2
    try:
        o2, l2 = ops.attn_fwd(q, kc, vc, True)
        g2 = ops.attn_bwd(do, q, kc, vc, o2, l2, True)
        for var in (0, 3):
            ops.ATTN_VARIANT = var
            o, lse = ops.attn_fwd(q, k, v, True)
            assert torch.equal(o, o2) and torch.equal(lse, l2), var
            dq, dk, dv = ops.attn_bwd(do, q, k, v, o, lse, True)
            assert torch.equal(dq, g2[0]) and torch.equal(dk, g2[1]) and torch.equal(dv, g2[2]), var
    finally:
        ops.ATTN_VARIANT = 0


def test_attn_fwd_strided_qkv_and_padding(attn_variant):
    """q/k/v as strided views of one fused [B,S,3,H,D] buffer; right padding handled through seqlens."""
    ops = _ops()
    torch.manual_seed(11)
    B, S, H, D = 3, 200, 4, 128
    qkv = rnd(B, S, 3, H, D).to(DEV)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    lens = [200, 131, 64]
    seqlens = torch.tensor(lens, dtype=torch.int32, device=DEV)
    o, lse = ops.attn_fwd(q, k, v, True, seqlens=seqlens)
    oref, lref = attn_ref(q.cpu(), k.cpu(), v.cpu(), True, seqlen=lens)
    for b, L in enumerate(lens):
        assert rel_l2(o[b, :L], oref[b, :L]) < 6e-3
        assert rel_l2(lse[b, :, :L], lref[b, :, :L]) < 1e-4
        assert torch.count_nonzero(o[b, L:]) == 0  # pad_input semantics: zeros at padded positions


def test_attn_fwd_outlier_rescale(attn_variant):
    """Force the online-softmax rescale branch: one key dominates late in the sequence."""
    ops = _ops()
    torch.manual_seed(2)
    B, S, H, D = 1, 512, 2, 128
    q, k, v = rnd(B, S, H, D), rnd(B, S, H, D), rnd(B, S, H, D)
    k[0, 400] = q[0, 450] * 4.0
    oref, lref = attn_ref(q, k, v, True)
    o, lse = ops.attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), True)
    assert rel_l2(o, oref) < 6e-3
    assert rel_l2(lse, lref) < 1e-4


@pytest.mark.parametrize("ramp", [0.02, 0.2, 1.0])
def test_attn_fwd_deferred_max_ramp(ramp, attn_variant):
    """The 8-wave kernel advances a row's running max only when it grows by more than 2^6 within a 32-key half (otherwise the
    probabilities are taken against the stale max and are > 1).  Keys whose scores ramp up along the sequence exercise every
    regime: growth below the threshold for many consecutive tiles (stale max, P up to 64), growth above it (rescale), and a
    spike (first-tile style jump).  Full-tensor fp32 reference, as the branch is data dependent (guide rule 26)."""
    ops = _ops()
    torch.manual_seed(7)
    B, S, H, D = 1, 768, 2, 128
    q, k, v = rnd(B, S, H, D), rnd(B, S, H, D), rnd(B, S, H, D)
    u = torch.nn.functional.normalize(torch.randn(D), dim=0)
    q = bf16r(q + 4.0 * u)                                                   # every query has a large component along u
    k = bf16r(k + ramp * 0.05 * torch.arange(S)[None, :, None, None] * u)    # keys drift along u: scores rise with the key index
    k[0, 700] = bf16r(q[0, 730] * 2.0)                                       # and one spike
    for causal in (True, False):
        oref, lref = attn_ref(q, k, v, causal)
        o, lse = ops.attn_fwd(q.to(BF).to(DEV), k.to(BF).to(DEV), v.to(BF).to(DEV), causal)
        assert rel_l2(o, oref) < 6e-3, (causal, ramp)
        assert (lse.cpu() - lref).abs().max() < 2e-3 * max(1.0, lref.abs().max().item()), (causal, ramp)


# ----------------------------------------------------------------------------- elementwise / gather / loss / optimizer
def test_rope_golden(golden):
    ops = _ops()
    g = golden("rope.pt")
    from oracle import llm_ref
    cos, sin = llm_ref.rope_tables(32, 64)
    half = 16
    # reference layout [B,H,S,D] -> ours [B,S,H,D]
    q = g["q"].transpose(1, 2).contiguous().to(BF).to(DEV)
    k = g["k"].transpose(1, 2).contiguous().to(BF).to(DEV)
    qr, kr = llm_ref.apply_rope(g["q"].to(BF).float(), g["k"].to(BF).float(), cos, sin, g["pos"])
    ct, st = cos[:, :half].contiguous().to(DEV), sin[:, :half].contiguous().to(DEV)
    ops.rope_(q, ct, st, g["pos"].to(DEV))
    ops.rope_(k, ct, st, g["pos"].to(DEV))
    assert rel_l2(q.transpose(1, 2), qr) < 4e-3
    assert rel_l2(k.transpose(1, 2), kr) < 4e-3
    # backward = inverse rotation
    ops.rope_(q, ct, st, g["pos"].to(DEV), backward=True)
    assert rel_l2(q.transpose(1, 2), g["q"].to(BF).float()) < 6e-3


@pytest.mark.parametrize("mode", [0, 1])
def test_glu(mode):
    ops = _ops()
    torch.manual_seed(mode)
    M, Fd = 70, 352
    gu = rnd(M, 2 * Fd)
    a, b = gu[:, :Fd], gu[:, Fd:]
    d = rnd(M, Fd)
    ar, br = a.float().requires_grad_(True), b.float().requires_grad_(True)
    ref = (F.silu(ar) if mode == 0 else F.gelu(ar)) * br
    ref.backward(d.float())
    gd = gu.to(DEV)
    out = ops.glu_fwd(gd[:, :Fd], gd[:, Fd:], mode)
    assert rel_l2(out, ref) < 4e-3
    da, db = ops.glu_bwd(d.to(DEV), gd[:, :Fd], gd[:, Fd:], mode)
    assert rel_l2(da, ar.grad) < 4e-3
    assert rel_l2(db, br.grad) < 4e-3


def test_gather_scatter_embedding_bwd():
    ops = _ops()
    torch.manual_seed(0)
    table = rnd(50, 64).to(DEV)
    ids = torch.randint(0, 50, (4, 9), device=DEV)
    out = ops.gather_rows(table, ids)
    assert torch.equal(out, table[ids.view(-1)])
    dst = torch.zeros(30, 64, dtype=BF, device=DEV)
    idx = torch.tensor([3, 7, 29, 0], device=DEV)
    src = rnd(4, 64).to(DEV)
    ops.scatter_rows_(dst, idx, src)
    ref = torch.zeros(30, 64, dtype=BF, device=DEV)
    ref[idx] = src
    assert torch.equal(dst, ref)
    dy = rnd(36, 64).to(DEV)
    dt = ops.embedding_bwd(dy, ids, 50)
    ref = torch.zeros(50, 64, dtype=torch.float32, device=DEV).index_add_(0, ids.view(-1), dy.float())
    assert rel_l2(dt, ref) < 4e-3


def test_embedding_bwd_long_segments_take_the_chunked_path():
    """A batch where one id (the image placeholder) covers a third of the positions and a few others hundreds: the chunked two-call form
    (fp32 partial rows per ops.EMB_BWD_CHUNK positions, then the partial rows of each token; ops.EMB_BWD_CHUNK) against fp32 index_add and against the
    one-work-group-per-token form; run to run identical."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(3)
    T, V, D = 8192, 1000, 256
    ids = torch.randint(0, V, (T,), device=DEV, generator=g)
    ids[torch.rand(T, device=DEV, generator=g) < 0.33] = 7          # ~2700 positions
    ids[torch.rand(T, device=DEV, generator=g) < 0.05] = 11         # ~400
    ids[:300] = 13                                                  # 300+: three chunks of 128, the last one partial
    dy = (torch.randn(T, D, device=DEV, generator=g)).to(BF)
    ref = torch.zeros(V, D, dtype=torch.float32, device=DEV).index_add_(0, ids, dy.float())
    dt = ops.embedding_bwd(dy, ids, V)
    assert rel_l2(dt, ref) < 4e-3
    assert torch.equal(dt, ops.embedding_bwd(dy, ids, V))
    prev = ops.EMB_BWD_CHUNK
    try:
        ops.EMB_BWD_CHUNK = 0
        dt0 = ops.embedding_bwd(dy, ids, V)
    finally:
        ops.EMB_BWD_CHUNK = prev
    assert rel_l2(dt0, ref) < 4e-3
    short = torch.bincount(ids, minlength=V) <= ops.EMB_BWD_CHUNK                  # single-chunk tokens: the same additions in the same order
    assert torch.equal(dt[short], dt0[short])
    assert rel_l2(dt, dt0.float()) < 4e-3


def test_cross_entropy():
    ops = _ops()
    torch.manual_seed(4)
    R, V = 37, 32008
    logits = (torch.randn(R, V) * 2).to(DEV)
    labels = torch.randint(0, V, (R,), device=DEV)
    labels[::5] = -100
    lr = logits.detach().cpu().requires_grad_(True)
    per = F.cross_entropy(lr, labels.cpu(), reduction="none", ignore_index=-100)
    nvalid = (labels != -100).sum().item()
    (per.sum() / nvalid).backward()
    loss_row = ops.cross_entropy_rows(logits, labels)
    assert rel_l2(loss_row, per) < 1e-5
    dl = torch.empty(R, V, dtype=BF, device=DEV)
    gs = torch.tensor([1.0 / nvalid], device=DEV)
    ops.cross_entropy_rows(logits, labels, dlogits=dl, gscale=gs)
    assert rel_l2(dl, lr.grad) < 4e-3


def test_adamw_matches_torch():
    ops = _ops()
    torch.manual_seed(9)
    n = 10000
    p0, g = torch.randn(n), torch.randn(n) * 0.1
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
    p = p0.clone().to(DEV)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    for step in range(1, 4):
        pt.grad = g.clone() * step
        opt.step()
        ops.adamw_(p, (g * step).to(DEV), m, v, 1e-2, 0.9, 0.98, 1e-8, 0.1, step)
    assert rel_l2(p, pt) < 1e-5
    # bf16 params + bf16 state (the reference's dtype choice, projects/dreamllm/train.py:66-71,170)
    pb = p0.to(BF).to(DEV)
    mb, vb = torch.zeros(n, dtype=BF, device=DEV), torch.zeros(n, dtype=BF, device=DEV)
    ops.adamw_(pb, g.to(BF).to(DEV), mb, vb, 1e-2, 0.9, 0.98, 1e-8, 0.1, 1)
    assert torch.isfinite(pb.float()).all()
    # the 8-wide non-temporal kernel (aligned, n % 8 == 0: every weight matrix) against the scalar kernel (same buffers shifted by
    # one element => unaligned => scalar path): the same parameters and moments over three steps, with a device-side clip factor
    n8 = 8 * 4099
    pa, ga = torch.randn(n8 + 8).to(BF).to(DEV), (torch.randn(n8 + 8) * 0.1).to(BF).to(DEV)
    coef = torch.tensor([0.37], device=DEV)
    bufs = {}
    for tag, off in (("vec", 8), ("scalar", 1)):
        mk = lambda src: torch.cat([src.new_zeros(off), src[8:8 + n8]])[off:]       # same values at a different alignment (a VIEW)
        P, G = mk(pa), mk(ga)
        M, V = torch.zeros(n8 + off, dtype=BF, device=DEV)[off:], torch.zeros(n8 + off, dtype=BF, device=DEV)[off:]
        assert (P.data_ptr() % 16 == 0) == (tag == "vec")
        for step in range(1, 4):
            ops.adamw_(P, G, M, V, 1e-2, 0.9, 0.98, 1e-8, 0.1, step, 1.0, coef)
        bufs[tag] = (P.clone(), M.clone(), V.clone())
    for a, b in zip(bufs["vec"], bufs["scalar"]):   # same formula; -ffast-math contracts the two kernels differently: <= 1 bf16 ulp
        assert rel_l2(a, b.float()) < 2e-3 and float((a != b).float().mean()) < 0.05
    pt = pa[8:8 + n8].float().cpu().clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
    for step in range(1, 4):
        pt.grad = ga[8:8 + n8].float().cpu() * 0.37
        opt.step()
    assert rel_l2(bufs["vec"][0], pt) < 6e-3          # bf16 parameters and moments: three roundings per step


def test_multi_tensor_kernels_with_empty_tensors_past_one_launch():
    """with more than 48 tensors and EMPTY tensors among them the batching loop used to revisit tensors past index 48 in
    the next launch -- a double AdamW update, double-counted partial sums and a write past the partials buffer.  120 tensors with an
    empty one every 7: every tensor updated exactly once, the sum of squares counted once, nothing written past `parts`."""
    ops = _ops()
    torch.manual_seed(33)
    sizes = [0 if i % 7 == 3 else 8 * (1 + (i * 37) % 700) for i in range(120)]
    mk = lambda scale: [(torch.randn(n) * scale).to(BF).to(DEV) for n in sizes]
    p1, g = mk(1.0), mk(0.1)
    p2 = [t.clone() for t in p1]
    m1, v1 = [torch.zeros_like(t) for t in p1], [torch.zeros_like(t) for t in p1]
    m2, v2 = [torch.zeros_like(t) for t in p1], [torch.zeros_like(t) for t in p1]
    ops.adamw_multi_(p1, g, m1, v1, 1e-2, 0.9, 0.98, 1e-8, 0.1, 1, 1.0, None)
    for a, b, c, d in zip(p2, g, m2, v2):
        if a.numel():
            ops.adamw_(a, b, c, d, 1e-2, 0.9, 0.98, 1e-8, 0.1, 1, 1.0, None)
    for a, b in zip(p1 + m1 + v1, p2 + m2 + v2):
        if a.numel():
            assert rel_l2(a, b.float()) < 2e-3
    parts = ops.sumsq_multi(g)
    assert parts.numel() == sum((n + 32767) // 32768 for n in sizes)
    ref = sum(float((t.double() ** 2).sum()) for t in g)
    assert abs(float(ops.reduce_sum_f32(parts)) - ref) < 1e-4 * ref


def test_adamw_and_grad_norm_multi_tensor_match_per_tensor_launches():
    """Multi-tensor AdamW / sum of squares (48 tensors per launch, tables as kernel arguments) against the per-tensor launches on
    130 tensors of assorted sizes (empty-ish, one chunk, chunk boundary + 8, several chunks; > 2 launches' worth): parameters and
    both moments over three steps with a device-side clip factor, and the clipped-norm optimizer end to end (HipAdamW with and
    without `multi_tensor`), including tensors the multi path must leave to the scalar kernel (odd size, fp32)."""
    ops = _ops()
    from dreamllm_amd.optim import HipAdamW
    torch.manual_seed(21)
    sizes = [8, 4096, 32768, 32776, 3 * 32768 + 64, 100000 - 100000 % 8, 1 << 20] * 18 + [24, 262144, 8, 65536]
    assert len(sizes) == 130
    mk = lambda scale: [(torch.randn(n) * scale).to(BF).to(DEV) for n in sizes]
    p1, g = mk(1.0), mk(0.1)
    p2 = [t.clone() for t in p1]
    m1, v1 = [torch.zeros_like(t) for t in p1], [torch.zeros_like(t) for t in p1]
    m2, v2 = [torch.zeros_like(t) for t in p1], [torch.zeros_like(t) for t in p1]
    coef = torch.tensor([0.61], device=DEV)
    for step in range(1, 4):
        ops.adamw_multi_(p1, g, m1, v1, 1e-2, 0.9, 0.98, 1e-8, 0.1, step, 1.0, coef)
        for a, b, c, d in zip(p2, g, m2, v2):
            ops.adamw_(a, b, c, d, 1e-2, 0.9, 0.98, 1e-8, 0.1, step, 1.0, coef)
    for a, b in zip(p1 + m1 + v1, p2 + m2 + v2):     # same formula; two kernels, -ffast-math may contract them differently
        assert rel_l2(a, b.float()) < 2e-3 and float((a != b).float().mean()) < 0.05
    parts = ops.sumsq_multi(g)
    ref = sum(float((t.double() ** 2).sum()) for t in g)
    assert abs(float(ops.reduce_sum_f32(parts)) - ref) < 1e-4 * ref
    assert torch.equal(ops.sumsq_multi(g), parts)                                     # deterministic
    # optimizer end to end, mixed bag of tensors
    def run(multi):
        torch.manual_seed(5)
        ps = [torch.nn.Parameter((torch.randn(n) * 0.5).to(BF).to(DEV)) for n in (4096, 1001, 65536, 8, 32768 * 2 + 8)]
        ps.append(torch.nn.Parameter(torch.randn(777, device=DEV)))                    # fp32, odd size: per-tensor kernel
        opt = HipAdamW(ps, lr=1e-2, betas=(0.9, 0.98), weight_decay=0.05, max_grad_norm=1.0, multi_tensor=multi)
        for it in range(3):
            for i, q in enumerate(ps):
                q.grad = (torch.randn(q.shape, generator=torch.Generator().manual_seed(100 * it + i)) * 0.3).to(q.dtype).to(DEV)
            opt.step()
        return [q.detach().clone() for q in ps], float(opt.last_grad_norm)
    a, na = run(True)
    b, nb = run(False)
    assert abs(na - nb) < 1e-4 * nb
    for x, y in zip(a, b):
        assert rel_l2(x, y.float()) < 2e-3


# ----------------------------------------------------------------------------- autograd wrappers
@pytest.mark.parametrize("V", [1000, 1009])  # 1009: vocabulary not a multiple of 8 (DreamLLM-SDXL has 32009)
def test_linear_autograd_and_lm_head_ce(V):
    ops = _ops()
    torch.manual_seed(6)
    T, d = 96, 128
    h = rnd(T, d)
    w = rnd(V, d, scale=0.05)
    labels = torch.randint(0, V, (T,))
    labels[:7] = -100
    hr, wr = h.float().requires_grad_(True), w.float().requires_grad_(True)
    lref = F.cross_entropy(hr @ wr.t(), labels, ignore_index=-100)
    (lref * 3.0).backward()
    hg, wg = h.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    loss, logits = ops.lm_head_ce(hg, wg, labels.to(DEV), return_logits=True)
    (loss * 3.0).backward()
    assert abs(loss.item() - lref.item()) < 1e-4 * abs(lref.item()) + 1e-5
    assert logits.shape == (T, V) and wg.grad.shape == (V, d)
    assert rel_l2(logits, (hr @ wr.t())) < 1e-5
    assert rel_l2(hg.grad, hr.grad) < 6e-3
    assert rel_l2(wg.grad, wr.grad) < 6e-3


# ----------------------------------------------------------------------------- attention backward
@pytest.mark.parametrize("B,H,Hkv,Sq,Sk,D,causal", [
    (2, 4, 4, 128, 128, 128, True),
    (1, 2, 2, 300, 300, 128, True),
    (2, 3, 3, 257, 257, 64, False),
    (2, 5, 5, 192, 64, 64, False),
    (1, 4, 2, 160, 160, 128, True),
    (1, 2, 2, 64, 200, 64, False),
    (1, 2, 1, 200, 200, 64, True),
    (1, 2, 2, 777, 777, 128, True),    # several 256-row blocks, ragged tail
    (2, 2, 1, 513, 513, 64, True),     # GQA over 256-key blocks
    (1, 2, 2, 40, 600, 128, True),     # few queries at the end of a long key axis
    (1, 1, 1, 1024, 1024, 64, False),
    (1, 5, 5, 600, 600, 128, False),   # non-causal d128 through the split dK / dV kernels, head count not a multiple of 8
    (3, 3, 1, 520, 520, 64, True),     # GQA group of 3 in the fused d64 dK/dV kernel
    (1, 2, 2, 2048, 2048, 128, True),  # the headline sequence length: 8 row blocks per head
])
def test_attn_bwd(B, H, Hkv, Sq, Sk, D, causal, attn_variant):
    """dQ/dK/dV against fp32 autograd of the oracle attention.  Bound 1.5e-2: P and dS enter the MFMAs as bf16."""
    ops = _ops()
    torch.manual_seed(Sq * 7 + Sk + D)
    q, k, v, do = rnd(B, Sq, H, D), rnd(B, Sk, Hkv, D), rnd(B, Sk, Hkv, D), rnd(B, Sq, H, D)
    qr, kr, vr = (x.float().requires_grad_(True) for x in (q, k, v))
    oref, _ = attn_ref(qr, kr, vr, causal)
    oref.backward(do.float())
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o, lse = ops.attn_fwd(qd, kd, vd, causal)
    dq, dk, dv = ops.attn_bwd(do.to(DEV), qd, kd, vd, o, lse, causal)
    assert rel_l2(dq, qr.grad) < 1.5e-2
    assert rel_l2(dk, kr.grad) < 1.5e-2
    assert rel_l2(dv, vr.grad) < 1.5e-2


def test_attn_bwd_padding_and_strided(attn_variant):
    ops = _ops()
    torch.manual_seed(21)
    B, S, H, D = 2, 200, 2, 128
    qkv = rnd(B, S, 3, H, D)
    lens = [200, 90]
    do = rnd(B, S, H, D)
    qr, kr, vr = (qkv[:, :, i].float().requires_grad_(True) for i in range(3))
    oref, _ = attn_ref(qr, kr, vr, True, seqlen=lens)
    mask = torch.zeros(B, S, 1, 1)
    for b, L in enumerate(lens):
        mask[b, :L] = 1
    (oref * mask).backward(do.float())
    qd = qkv.to(DEV)
    q, k, v = qd[:, :, 0], qd[:, :, 1], qd[:, :, 2]
    seqlens = torch.tensor(lens, dtype=torch.int32, device=DEV)
    o, lse = ops.attn_fwd(q, k, v, True, seqlens=seqlens)
    dq, dk, dv = ops.attn_bwd(do.to(DEV), q, k, v, o, lse, True, seqlens=seqlens)
    for b, L in enumerate(lens):
        assert rel_l2(dq[b, :L], qr.grad[b, :L]) < 1.5e-2
        assert rel_l2(dk[b, :L], kr.grad[b, :L]) < 1.5e-2
        assert rel_l2(dv[b, :L], vr.grad[b, :L]) < 1.5e-2
        assert torch.count_nonzero(dq[b, L:]) == 0 and torch.count_nonzero(dk[b, L:]) == 0


def test_flash_attn_autograd():
    ops = _ops()
    torch.manual_seed(8)
    B, S, H, D = 1, 130, 2, 64
    q, k, v = rnd(B, S, H, D), rnd(B, S, H, D), rnd(B, S, H, D)
    qr, kr, vr = (x.float().requires_grad_(True) for x in (q, k, v))
    oref, _ = attn_ref(qr, kr, vr, False)
    oref.square().sum().backward()
    qd, kd, vd = (x.to(DEV).requires_grad_(True) for x in (q, k, v))
    o = ops.flash_attn(qd, kd, vd, causal=False)
    o.float().square().sum().backward()
    assert rel_l2(qd.grad, qr.grad) < 2e-2 and rel_l2(kd.grad, kr.grad) < 2e-2 and rel_l2(vd.grad, vr.grad) < 2e-2


@pytest.mark.parametrize("M,N,K", [(128, 1280, 11520), (512, 640, 5760), (104, 328, 2048)])
def test_gemm_splitk_small_grid(M, N, K):
    """Small grids with deep K (UNet at batch 2) take the deterministic split-K path: all three layouts + fused epilogue."""
    from dreamllm_amd import _lib
    ops = _ops()
    assert _lib.call("dllm_gemm_splitk_hint", M, N, K, 0, 0) > 1
    torch.manual_seed(M + K)
    x, w, b, r = rnd(M, K), rnd(N, K, scale=0.02), rnd(N), rnd(M, N)
    ref = F.gelu(x.float() @ w.float().t() + b.float()) + r.float()
    y = ops.linear_fwd(x.to(DEV), w.to(DEV), bias=b.to(DEV), epi="gelu", residual=r.to(DEV))
    assert rel_l2(y, ref) < 4e-3
    y2 = ops.linear_fwd(x.to(DEV), w.to(DEV), bias=b.to(DEV), epi="gelu", residual=r.to(DEV))
    assert torch.equal(y, y2)  # deterministic reduction order
    dy = rnd(K, N)  # wgrad-shaped: reduction over K rows
    xx = rnd(K, M)
    dw = ops.linear_wgrad(dy.to(DEV), xx.to(DEV), out_dtype=torch.float32)
    assert rel_l2(dw, dy.float().t() @ xx.float()) < 1e-5
    dx = ops.linear_dgrad(rnd(M, K).to(DEV) * 0 + x.to(DEV), rnd(K, N, seed=1).to(DEV))
    assert rel_l2(dx, x.float() @ rnd(K, N, seed=1).float()) < 4e-3


@pytest.mark.parametrize("M,N,K", [(128, 1280, 11520), (512, 1280, 5120), (2048, 640, 5760), (104, 328, 2048)])
def test_gemm_splitk_in_kernel_reduction_is_bit_identical(M, N, K):
    """The in-kernel split-K reduction (last K slice of a tile reduces, agent-scope release / acquire around a ticket) against
    the separate reduce kernel: same summation order => bit-identical, 40 launches back to back (slices of one tile run on
    different XCDs: a stale read would show as a mismatch), counters back at zero afterwards."""
    from dreamllm_amd import _lib
    ops = _ops()
    assert _lib.call("dllm_gemm_splitk_hint", M, N, K, 0, 0) > 1
    torch.manual_seed(M + N)
    x, w, b, r = rnd(M, K).to(DEV), rnd(N, K, scale=0.02).to(DEV), rnd(N).to(DEV), rnd(M, N).to(DEV)
    ref = ops.linear_fwd(x, w, bias=b, epi="silu", residual=r)
    ops.SPLITK_FUSED_REDUCE = True
    try:
        for it in range(40):
            y = ops.linear_fwd(x, w, bias=b, epi="silu", residual=r)
            assert torch.equal(y, ref), it
    finally:
        ops.SPLITK_FUSED_REDUCE = False
    torch.cuda.synchronize()
    for buf in ops._SPLITK_COUNTERS.values():
        assert int(buf.abs().sum()) == 0


# ----------------------------------------------------------------------------- ring-buffered 128 x 128 kernel (gemm_ring.hip)
@pytest.mark.parametrize("M,N,K", [(8192, 320, 320), (2048, 640, 640), (512, 1280, 1280), (128, 1280, 2560), (200, 328, 64),
                                   (130, 136, 128), (257, 648, 192), (1000, 100, 256), (128, 128, 448), (8192, 2560, 320)])
def test_gemm_ring_forward_shapes(M, N, K):
    """The small-grid kernel (tile code 264) on the UNet's linear shapes at batch 2 and on ragged edges: K = 64 ... 2560 covers every
    prologue / tail length of the 4-stage ring (1, 2, 3 stages in flight, then steady state), N % 8 != 0 takes the direct epilogue,
    everything else the LDS-staged one; bias + activation + residual; fp32 output; run-to-run bit-identical (a DMA / fragment-read
    race would show as a mismatch between repeats)."""
    ops = _ops()
    torch.manual_seed(M + N + K)
    x, w, b, r = rnd(M, K), rnd(N, K, scale=0.05), rnd(N), rnd(M, N)
    ref0 = x.float() @ w.float().t()
    ref = F.silu(ref0 + b.float()) + r.float()
    xd, wd, bd, rd = x.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV)
    ops.SPLITK = False
    try:
        with ops.gemm_variant(264):
            y = ops.linear_fwd(xd, wd, bias=bd, epi="silu", residual=rd)
            assert rel_l2(y, ref) < 4e-3
            for _ in range(10):
                assert torch.equal(ops.linear_fwd(xd, wd, bias=bd, epi="silu", residual=rd), y)
            y0 = ops.linear_fwd(xd, wd)
            assert rel_l2(y0, ref0) < 4e-3
            y32 = ops.linear_fwd(xd, wd, out_dtype=torch.float32)
            assert rel_l2(y32, ref0) < 1e-5
        with ops.gemm_variant(128):   # same arithmetic per output as the register-staged kernel: identical fp32 sums
            assert rel_l2(ops.linear_fwd(xd, wd, out_dtype=torch.float32), y32.float()) < 1e-6
    finally:
        ops.SPLITK = True
    # every 128 x 128 tile written exactly once
    err = (y0.float().cpu() - ref0).abs()
    assert float(err.max()) < 0.05 * float(ref0.abs().max()) + 0.05


@pytest.mark.parametrize("M,N,K", [(128, 1280, 11520), (512, 1280, 5120), (2048, 640, 5760), (104, 328, 2048), (8192, 320, 2880)])
def test_gemm_ring_splitk(M, N, K):
    """Split-K on the ring kernel (the automatic choice for these shapes since round 4): slabs + deterministic reduce launch."""
    from dreamllm_amd import _lib
    ops = _ops()
    torch.manual_seed(M + K)
    x, w, b, r = rnd(M, K), rnd(N, K, scale=0.02), rnd(N), rnd(M, N)
    ref = F.gelu(x.float() @ w.float().t() + b.float()) + r.float()
    xd, wd, bd, rd = x.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV)
    y = ops.linear_fwd(xd, wd, bias=bd, epi="gelu", residual=rd)
    assert rel_l2(y, ref) < 4e-3
    for _ in range(5):
        assert torch.equal(ops.linear_fwd(xd, wd, bias=bd, epi="gelu", residual=rd), y)
    for sk in (2, 3, 7):   # uneven K-tile ranges, empty last slices
        out = torch.empty(M, N, dtype=BF, device=DEV)
        ws = torch.empty(sk * M * N, dtype=torch.float32, device=DEV)
        _lib.check("dllm_gemm_bf16_splitk", ops._p(xd), ops._p(wd), ops._p(out), ops._p(bd), ops._p(rd), M, N, K, K, K, N, N, 0, 0,
                   ops.EPI["gelu"], 0, 0, 1.0, sk, ops._p(ws), None, 264, ops._stream())
        assert rel_l2(out, ref) < 4e-3, sk


@pytest.mark.parametrize("M,K,F_", [(512, 1280, 5120), (8192, 320, 1280), (100, 320, 1280), (2048, 640, 2560), (128, 64, 64)])
def test_gemm_ring_geglu_epilogue(M, K, F_):
    """diffusers GEGLU fused into the projection (EPI_GEGLU): out = (x Wh^T + bh) * gelu(x Wg^T + bg), against fp32 torch and against
    the two-launch path (GEMM + element-wise GEGLU), with and without bias."""
    ops = _ops()
    torch.manual_seed(M + F_)
    x, w, b = rnd(M, K), rnd(2 * F_, K, scale=0.05), rnd(2 * F_)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    z = x.float() @ w.float().t() + b.float()
    ref = z[:, :F_] * F.gelu(z[:, F_:])
    y = ops.linear_geglu(xd, wd, bd)
    assert y is not None and y.shape == (M, F_)
    assert rel_l2(y, ref) < 4e-3
    two = ops.geglu_packed(ops.linear_fwd(xd, wd, bias=bd))
    assert rel_l2(y, two.float()) < 6e-3          # the two-launch path rounds the projection to bf16 first
    z0 = x.float() @ w.float().t()
    assert rel_l2(ops.linear_geglu(xd, wd, None), z0[:, :F_] * F.gelu(z0[:, F_:])) < 4e-3
    assert torch.equal(ops.linear_geglu(xd, wd, bd), y)
    assert ops.linear_geglu(rnd(8, 72).to(DEV), rnd(128, 72).to(DEV)) is None   # K % 64 != 0: caller falls back


# ----------------------------------------------------------------------------- greedy-decode kernels
@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (1, 11008, 4096), (3, 1000, 11008), (8, 515, 128), (2, 32008, 512), (1, 4096, 11008),
                                   (1, 1001, 2048), (4, 37, 6152), (2, 12, 8)])
@pytest.mark.parametrize("f32", [False, True])
def test_gemv(M, N, K, f32):
    ops = _ops()
    x, w, r = rnd(M, K, seed=M + N), rnd(N, K, scale=0.05), rnd(M, N)
    ref = x.float() @ w.float().t() + r.float()
    y = ops.gemv(x.to(DEV), w.to(DEV), residual=r.to(DEV), out_dtype=torch.float32 if f32 else BF)
    assert y.dtype == (torch.float32 if f32 else BF) and y.shape == (M, N)
    assert rel_l2(y, ref) < (2e-5 if f32 else 4e-3)
    y2 = ops.gemv(x.to(DEV), w.to(DEV), out_dtype=torch.float32)
    assert rel_l2(y2, x.float() @ w.float().t()) < 2e-5


@pytest.mark.parametrize("B,H,Hkv,D,Smax,lens,nsplit", [
    (1, 32, 32, 128, 2048, [1500], 8), (3, 4, 4, 128, 300, [1, 300, 57], 8), (2, 6, 2, 64, 200, [200, 13], 4),
    (2, 4, 4, 128, 64, [5, 64], 1), (1, 2, 1, 64, 40, [3], 16)])
def test_attn_decode(B, H, Hkv, D, Smax, lens, nsplit):
    """One query token against a KV cache whose valid length lives on the device (entries past it hold garbage)."""
    ops = _ops()
    q, kc, vc = rnd(B, H, D, seed=B + H), rnd(B, Smax, Hkv, D), rnd(B, Smax, Hkv, D)
    for b, n in enumerate(lens):
        kc[b, n:] = 1e4  # must never be read into the softmax
        vc[b, n:] = 1e4
    out = ops.attn_decode(q.to(DEV), kc.to(DEV), vc.to(DEV), torch.tensor(lens, dtype=torch.int32, device=DEV), nsplit=nsplit)
    for b, n in enumerate(lens):
        k = kc[b, :n].float().repeat_interleave(H // Hkv, dim=1)  # [n, H, D]
        v = vc[b, :n].float().repeat_interleave(H // Hkv, dim=1)
        p = torch.softmax(torch.einsum("hd,nhd->hn", q[b].float(), k) * D ** -0.5, dim=-1)
        ref = torch.einsum("hn,nhd->hd", p, v)
        assert rel_l2(out[b], ref) < 6e-3, (b, n)


def test_gemv_fused_norm_multi_swiglu_and_rope_append():
    """Fused decode launches against the unfused operator chain they replace: bit-identical (same roundings, same
    accumulation order) for RMSNorm + q/k/v, RMSNorm + gate/up + SwiGLU, RMSNorm + fp32 head; RoPE + cache append."""
    ops = _ops()
    M, K, F_ = 3, 4096, 1376
    x, nw = rnd(M, K, seed=1).to(DEV), (1.0 + 0.1 * torch.randn(K)).to(BF).to(DEV)
    wq, wk, wv = (rnd(n, K, scale=0.03).to(DEV) for n in (512, 256, 256))
    h, _, _ = ops.rmsnorm_fwd(x, nw, 1e-5)
    q, k, v = ops.gemv_fused(x, (wq, wk, wv), norm_w=nw, eps=1e-5)
    for got, w in ((q, wq), (k, wk), (v, wv)):
        assert torch.equal(got, ops.gemv(h, w))
    wg, wu = rnd(F_, K, scale=0.03).to(DEV), rnd(F_, K, scale=0.03).to(DEV)
    act = ops.gemv_fused(x, (wg, wu), norm_w=nw, eps=1e-5, swiglu=True)
    assert torch.equal(act, ops.glu_fwd(ops.gemv(h, wg), ops.gemv(h, wu), 0))
    r = rnd(M, 512).to(DEV)
    y = ops.gemv_fused(x, (wq,), residual=r)[0]  # no norm, residual
    assert torch.equal(y, ops.gemv(x, wq, residual=r))
    lg = ops.gemv_fused(x, (wq,), norm_w=nw, eps=1e-5, out_dtype=torch.float32)[0]
    assert lg.dtype == torch.float32 and torch.equal(lg, ops.gemv(h, wq, out_dtype=torch.float32))
    # rope + append
    B, H, Hkv, D, S = 3, 4, 2, 128, 50
    qq, kk, vv = rnd(B, H, D, seed=2).to(DEV), rnd(B, Hkv, D).to(DEV), rnd(B, Hkv, D).to(DEV)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.arange(S).float()[:, None] * inv[None]
    cos, sin = ang.cos().to(DEV), ang.sin().to(DEV)
    pos = torch.tensor([0, 17, 49], device=DEV)
    kc = torch.zeros(B, S, Hkv, D, dtype=BF, device=DEV)
    vc = torch.zeros_like(kc)
    q_ref, k_ref = qq.clone().view(B, 1, H, D), kk.clone().view(B, 1, Hkv, D)
    ops.rope_(q_ref, cos, sin, pos[:, None])
    ops.rope_(k_ref, cos, sin, pos[:, None])
    ops.rope_append_(qq, kk, vv, kc, vc, cos, sin, pos)
    assert torch.equal(qq, q_ref.view(B, H, D))
    for b in range(B):
        assert torch.equal(kc[b, pos[b]], k_ref[b, 0]) and torch.equal(vc[b, pos[b]], vv[b])
    assert int((kc != 0).any(dim=-1).any(dim=-1).sum()) == B  # exactly one cache row per batch element written


@pytest.mark.parametrize("B,H,Hkv,D,Smax,lens,starts,nsplit", [
    (1, 32, 32, 128, 640, [515], None, 8), (3, 8, 2, 128, 96, [1, 96, 57], None, 8), (2, 6, 6, 64, 64, [64, 13], [3, 0], 4),
    (2, 4, 4, 128, 40, [5, 40], [2, 30], 1)])
def test_attn_decode_rope_fused_matches_rope_append_then_attn_decode(B, H, Hkv, D, Smax, lens, starts, nsplit):
    """RoPE + KV-cache append folded into the decode-attention launch (dllm_attn_decode_rope) against the two launches it replaces,
    on the same inputs: attention output, the appended cache rows (rotated k, v) and every other cache row untouched; grouped-query
    heads, left-padded prompts (kv_start), the new token alone in the cache (len 1), nsplit 1."""
    ops = _ops()
    torch.manual_seed(B * 7 + H)
    q, kn, vn = rnd(B, H, D, seed=B + H).to(DEV), rnd(B, Hkv, D).to(DEV), rnd(B, Hkv, D).to(DEV)
    kc0, vc0 = rnd(B, Smax, Hkv, D).to(DEV), rnd(B, Smax, Hkv, D).to(DEV)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.arange(Smax + 8).float()[:, None] * inv[None]
    cos, sin = ang.cos().to(DEV), ang.sin().to(DEV)
    kv_len = torch.tensor(lens, dtype=torch.int32, device=DEV)
    kv_start = None if starts is None else torch.tensor(starts, dtype=torch.int32, device=DEV)
    # rotary position = the row's own token count - 1 (left padding: slot and position differ)
    pos = torch.tensor([n - 1 - (0 if starts is None else starts[b]) for b, n in enumerate(lens)], dtype=torch.int64, device=DEV)
    kc1, vc1, q1 = kc0.clone(), vc0.clone(), q.clone()
    ops.rope_append_(q1, kn, vn, kc1, vc1, cos, sin, pos, kv_len=kv_len)
    ref = ops.attn_decode(q1, kc1, vc1, kv_len, nsplit=nsplit, kv_start=kv_start)
    kc2, vc2, q2 = kc0.clone(), vc0.clone(), q.clone()
    out = ops.attn_decode_rope(q2, kn, vn, kc2, vc2, kv_len, cos, sin, pos, nsplit=nsplit, kv_start=kv_start)
    assert torch.equal(q2, q)                                   # q is read only
    # the in-launch merge of the split-KV states (no combine launch): same bits as the combine kernel, 30 launches back to back
    # (a stale cross-XCD read would show as a mismatch), counters back at zero
    cnt = torch.zeros(B * H, dtype=torch.int32, device=DEV)
    for it in range(30):
        kc3, vc3 = kc0.clone(), vc0.clone()
        o3 = ops.attn_decode_rope(q2, kn, vn, kc3, vc3, kv_len, cos, sin, pos, nsplit=nsplit, kv_start=kv_start, counters=cnt)
        assert torch.equal(o3, out), it
    assert int(cnt.abs().sum()) == 0 and torch.equal(kc3, kc2) and torch.equal(vc3, vc2)
    assert rel_l2(out, ref.float()) < 2e-3
    assert rel_l2(kc2, kc1.float()) < 1e-3 and torch.equal(vc2, vc1)
    same = (kc2 == kc1).all(dim=-1).all(dim=-1)                 # [B, Smax]: only the appended row may differ (by fp32 contraction order)
    for b, n in enumerate(lens):
        assert bool(same[b, : n - 1].all()) and bool(same[b, n:].all())
        assert rel_l2(kc2[b, n - 1], kc1[b, n - 1].float()) < 4e-3


@pytest.mark.parametrize("V", [1000, 1001])
def test_lm_head_ce_fused_matches_unfused_and_torch(V):
    """Fused lm_head + CE (modeling_dreamllm.py:1452-1470 without the [T,V] logits: chunked GEMM -> CE fwd+bwd -> dgrad / wgrad
    inside the forward) against the unfused unit (same kernels, full logits) and an fp32 torch reference; odd vocabulary
    (DreamLLM-SDXL's 32009 case), ignored rows, more rows than one chunk, upstream gradient != 1."""
    from dreamllm_amd import ops
    torch.manual_seed(V)
    R, d = 2500, 256
    h = bf16r(torch.randn(R, d) * 0.5)
    w = bf16r(torch.randn(V, d) * 0.05)
    lab = torch.randint(0, V, (R,))
    lab[torch.rand(R) < 0.3] = -100
    hr, wr = h.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.cross_entropy(F.linear(hr, wr), lab, ignore_index=-100)
    (ref * 0.37).backward()
    old = ops.LM_HEAD_CE_CHUNK_ROWS
    ops.LM_HEAD_CE_CHUNK_ROWS = 1024
    try:
        res = {}
        for fused in (True, False):
            hd, wd = h.to(BF).to(DEV).requires_grad_(True), w.to(BF).to(DEV).requires_grad_(True)
            out = ops.lm_head_ce(hd, wd, lab.to(DEV), return_logits=not fused)
            loss = out if fused else out[0]
            (loss * 0.37).backward()
            res[fused] = (loss.detach(), hd.grad, wd.grad)
            if not fused:
                assert out[1].shape == (R, V) and rel_l2(out[1], F.linear(h, w)) < 1e-5
    finally:
        ops.LM_HEAD_CE_CHUNK_ROWS = old
    assert abs(res[True][0].item() - ref.item()) < 1e-4 * abs(ref.item())       # fp32 logits, fp32 CE
    assert torch.equal(res[True][0], res[False][0])                               # same kernels, same row order
    assert rel_l2(res[True][1], hr.grad) < 8e-3 and rel_l2(res[True][2], wr.grad) < 8e-3
    # the fused unit rounds its gradients to bf16 BEFORE the upstream scalar (0.37 here; exactly 1.0 in the stage-II loss mix,
    # where the product is exact) is applied, the unfused one folds it in before rounding: two roundings vs one
    assert rel_l2(res[True][1], res[False][1]) < 7e-3
    assert rel_l2(res[True][2], res[False][2]) < 7e-3


@pytest.mark.parametrize("N,Sq,Sk,C", [(2, 256, 256, 64), (1, 1024, 1024, 512), (1, 100, 72, 128)])
def test_wide_head_attention_and_row_softmax(N, Sq, Sk, C):
    """`ops.attention_wide_head` (GEMM -> `dllm_softmax_rows` -> GEMM): the VAE mid-block attention, one head of width 512."""
    from dreamllm_amd import ops
    torch.manual_seed(C)
    q, k, v = bf16r(torch.randn(N, Sq, C)), bf16r(torch.randn(N, Sk, C)), bf16r(torch.randn(N, Sk, C))
    ref = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    out = ops.attention_wide_head(q.to(BF).to(DEV), k.to(BF).to(DEV), v.to(BF).to(DEV))
    assert rel_l2(out, ref) < 6e-3
    x = torch.randn(37, 1003, device=DEV) * 3
    x4 = torch.nn.functional.pad(x, (0, 1))[:, :1004]  # pitch multiple of 4
    assert rel_l2(ops.softmax_rows(x4[:, :1000].contiguous()), torch.softmax(x4[:, :1000].float(), -1).cpu()) < 4e-3


# ----------------------------------------------------------------------------- fused SwiGLU epilogues (round 6)
@pytest.mark.parametrize("M,F_,K", [(256, 256, 64), (512, 512, 256), (1024, 768, 320), (256, 1280, 4096), (8192, 2816, 1024)])
def test_gemm_swiglu_fwd_equals_gemm_plus_glu(M, F_, K):
    """gate|up projection with the SwiGLU in its epilogue (dllm_gemm_swiglu_fwd) == the GEMM followed by dllm_glu_fwd: the product is
    bit for bit what dllm_glu_fwd makes of the packed gate|up tile the same launch stored; that tile equals the plain GEMM's output bit
    for bit wherever the plain GEMM runs on the same kernel (small grids take other families: another summation order over K) and the
    fp32 oracle to bf16 rounding everywhere."""
    ops = _ops()
    torch.manual_seed(M + F_ + K)
    x, w = rnd(M, K).to(DEV), rnd(2 * F_, K, scale=K ** -0.5).to(DEV)
    out = ops.linear_swiglu_fwd(x, w)
    assert out is not None
    gu, act = out
    assert torch.equal(act, ops.glu_fwd(gu[:, :F_], gu[:, F_:], 0))
    ref = x.float().cpu() @ w.float().cpu().t()
    assert rel_l2(gu, ref) < 4e-3
    assert rel_l2(act, F.silu(ref[:, :F_]) * ref[:, F_:]) < 8e-3   # two bf16 roundings (gate|up, then the product)
    with ops.gemm_variant(259):
        gu0 = ops.linear_fwd(x, w)
    if (M // 256) * (2 * F_ // 256) >= 128 or K <= 320:   # grids on which variant 259 keeps the 256 x 256 pipelined kernel
        assert torch.equal(gu, gu0)
    else:
        assert rel_l2(gu, gu0.float()) < 2e-3


@pytest.mark.parametrize("M,F_,D", [(256, 256, 64), (512, 512, 256), (768, 1024, 320), (256, 256, 4096)])
def test_gemm_swiglu_bwd_equals_dgrad_plus_glu_bwd(M, F_, D):
    """the down projection's input gradient with the SwiGLU backward in its epilogue (dllm_gemm_swiglu_bwd) == dgrad GEMM + dllm_glu_bwd,
    bit for bit; and against fp32 autograd."""
    ops = _ops()
    torch.manual_seed(M + F_ + D + 1)
    dy, wd, gu = rnd(M, D).to(DEV), rnd(D, F_, scale=D ** -0.5).to(DEV), rnd(M, 2 * F_).to(DEV)
    dgu = ops.linear_dgrad_swiglu(dy, wd, gu)
    assert dgu is not None
    with ops.gemm_variant(259):
        d_act = ops.linear_dgrad(dy, wd)
    dgu0 = torch.empty_like(gu)
    ops.glu_bwd(d_act, gu[:, :F_], gu[:, F_:], 0, da=dgu0[:, :F_], db=dgu0[:, F_:])
    if D <= 320:      # (deep reductions on tiny grids: the unfused GEMM takes another kernel family, see the forward test)
        assert torch.equal(dgu, dgu0)
    else:
        assert rel_l2(dgu, dgu0.float()) < 4e-3
    gr = gu.float().cpu().requires_grad_(True)
    act = F.silu(gr[:, :F_]) * gr[:, F_:]
    act.backward(dy.float().cpu() @ wd.float().cpu())
    assert rel_l2(dgu, gr.grad) < 8e-3


def test_gemm_swiglu_rejects_what_it_does_not_take():
    ops = _ops()
    x = rnd(200, 64).to(DEV)          # M % 256 != 0: the wrappers decline (the caller runs the unfused launches) ...
    assert ops.linear_swiglu_fwd(x, rnd(256, 64).to(DEV)) is None
    assert ops.linear_dgrad_swiglu(rnd(256, 64).to(DEV), rnd(64, 128).to(DEV), rnd(256, 256).to(DEV)) is None   # F % 256 != 0
    from dreamllm_amd import _lib     # ... and the C entry points refuse instead of touching memory
    import ctypes
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    a, w, g, o = rnd(200, 64).to(DEV), rnd(256, 64).to(DEV), torch.empty(200, 256, dtype=BF, device=DEV), torch.empty(200, 128, dtype=BF, device=DEV)
    assert _lib.call("dllm_gemm_swiglu_fwd", p(a), p(w), p(g), p(o), 200, 128, 64, 64, 64, 256, 128, 0, None) == -1


@pytest.mark.parametrize("code", [261, 280])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (512, 768, 192), (1024, 512, 4096), (2048, 2304, 1024)])
def test_gemm_w4_experiment(M, N, K, code):
    """tile codes 261 / 280: the four-wave 256 x 256 kernels of round 6 (one wave per SIMD, MFMA 32x32x16 / 16x16x32, buffer-form LDS-DMA, LDS stage
    released half a tile early; csrc/gemm_w4.hip) against the fp32 oracle, on one, two, three and many K tiles (prologue / steady state / the two
    peeled tail bodies).  280 adds the products in the order of the 8-wave kernel: the same bits."""
    ops = _ops()
    torch.manual_seed(M + N + K)
    x, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    ref = x.float() @ w.float().t()
    with ops.gemm_variant(code):
        y = ops.linear_fwd(x.to(DEV), w.to(DEV))
        y2 = ops.linear_fwd(x.to(DEV), w.to(DEV))
    assert rel_l2(y, ref) < 4e-3
    assert torch.equal(y, y2)
    with ops.gemm_variant(259):
        y0 = ops.linear_fwd(x.to(DEV), w.to(DEV))
    assert rel_l2(y, y0.float()) < 3e-3
    if code == 280 and M * N >= 256 * 256 * 48:    # (smaller grids: 259 hands the shape to another family)
        assert torch.equal(y, y0)


@pytest.mark.parametrize("layout", ["fwd", "dgrad", "wgrad"])
@pytest.mark.parametrize("ragged", [False, True])
def test_gemm_w4m_equals_the_8_wave_kernel(layout, ragged):
    """tile code 280 = gemm_w4m_kernel (four waves, one per SIMD, LDS stage released half a tile early; csrc/gemm_w4.hip) on the three dense
    layouts of the training step, more than one round of 256 tiles, five K tiles, full and ragged edge tiles: the same bits as the 8-wave
    kernel (tile code 259: same MFMA, same order of the K sum, same epilogues) and the fp32 product by tolerance; the weight gradient also
    with fp32 output + accumulate (the generic epilogue)."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(7 + ragged)
    rn = lambda *s, scale=1.0: (torch.randn(*s, device=DEV, generator=g) * scale).to(BF)
    M, N, K = (4352 + (40 if ragged else 0)), 4096 + (24 if ragged else 0), 320

    def run(code):
        with ops.gemm_variant(code):
            if layout == "fwd":
                return [ops.linear_fwd(a, b), ops.linear_fwd(a, b, bias=bias, epi="silu", residual=res)]
            if layout == "dgrad":
                return [ops.linear_dgrad(a, b)]
            out = acc0.clone()
            ops.linear_wgrad(a, b, out=out, accumulate=True)
            return [ops.linear_wgrad(a, b), out]

    if layout == "fwd":
        a, b, bias, res = rn(M, K), rn(N, K, scale=K ** -0.5), rn(N), rn(M, N)
        refs = [a.float() @ b.float().t()]
        refs.append(F.silu(refs[0] + bias.float()) + res.float())
    elif layout == "dgrad":                       # dx[M, N] = dy[M, K] W[K, N]
        a, b = rn(M, K), rn(K, N, scale=K ** -0.5)
        refs = [a.float() @ b.float()]
    else:                                         # dW[M, N] = dy[K, M]^T x[K, N] over K tokens
        a, b = rn(K, M), rn(K, N, scale=K ** -0.5)
        acc0 = torch.randn(M, N, device=DEV, generator=g)
        refs = [a.float().t() @ b.float()]
        refs.append(acc0 + refs[0])
    got, base = run(280), run(259)
    for y, y0, ref in zip(got, base, refs):
        assert rel_l2(y, ref) < (1e-5 if y.dtype == torch.float32 else 4e-3)
        assert torch.equal(y, y0)


def test_fused_gemm_epilogues_have_the_same_bits_on_both_kernel_families():
    """The three fused launches of the decoder layer (gate|up + SwiGLU, down-projection dgrad + SwiGLU backward, q|k|v + RoPE) on the four-wave
    kernel (gemm_variant 280 -> bits 8-9 of group_m) against the 8-wave kernel (259), more than one round of tiles: identical outputs.  The
    SwiGLU arithmetic is one fixed operation order (common.h: swiglu_fwd_elem / swiglu_bwd_elem) -- left to -ffast-math the call sites differed
    by an ulp."""
    ops = _ops()
    from oracle import llm_ref
    g = torch.Generator(device=DEV).manual_seed(21)
    rn = lambda *s, scale=1.0: (torch.randn(*s, device=DEV, generator=g) * scale).to(BF)
    M, K, F_, D = 4352, 512, 2048, 128
    x, wgu = rn(M, K), rn(2 * F_, K, scale=K ** -0.5)
    dy, wd, gu = rn(M, K), rn(K, F_, scale=K ** -0.5), rn(M, 2 * F_)
    Hq = Hkv = 8
    wqkv = rn((Hq + 2 * Hkv) * D, K, scale=K ** -0.5)
    cos, sin = llm_ref.rope_tables(D, 1024)
    ct, st = cos[:, : D // 2].contiguous().to(DEV), sin[:, : D // 2].contiguous().to(DEV)
    pos = torch.randint(0, 1024, (M,), device=DEV, generator=g)
    outs = {}
    for code in (259, 280):
        with ops.gemm_variant(code):
            a = ops.linear_swiglu_fwd(x, wgu)
            b = ops.linear_dgrad_swiglu(dy, wd, gu)
            c = ops.linear_rope_qkv(x, wqkv, ct, st, pos, Hq + Hkv, D, 256)
        assert a is not None and b is not None and c is not None
        outs[code] = (a[0], a[1], b, c)
    for t259, t280 in zip(outs[259], outs[280]):
        assert torch.equal(t259, t280)
    # and against the unfused pair on the element-wise kernels
    with ops.gemm_variant(259):
        gu0 = ops.linear_fwd(x, wgu)
        dact = ops.linear_dgrad(dy, wd)
    assert torch.equal(outs[280][0], gu0)
    assert torch.equal(outs[280][1], ops.glu_fwd(gu0[:, :F_], gu0[:, F_:], 0))
    dg, du = ops.glu_bwd(dact, gu[:, :F_], gu[:, F_:], 0)
    assert torch.equal(outs[280][2][:, :F_], dg) and torch.equal(outs[280][2][:, F_:], du)


@pytest.mark.parametrize("M,Hq,Hkv,K,with_pos", [(256, 2, 2, 64, False), (512, 4, 2, 256, True), (1024, 6, 2, 320, False), (2048, 32, 32, 4096, True)])
def test_gemm_rope_qkv_equals_gemm_plus_rope(M, Hq, Hkv, K, with_pos):
    """packed q|k|v projection with the rotary embedding in its epilogue (dllm_gemm_rope_qkv, head_dim 128) == GEMM + dllm_rope on the q and
    k heads, bit for bit (where the plain GEMM runs on the same kernel), v columns untouched; against the fp32 oracle by tolerance."""
    ops = _ops()
    from oracle import llm_ref
    D, S = 128, 256
    torch.manual_seed(M + Hq + K)
    N = (Hq + 2 * Hkv) * D
    x, w = rnd(M, K).to(DEV), rnd(N, K, scale=K ** -0.5).to(DEV)
    cos, sin = llm_ref.rope_tables(D, 512)
    ct, st = cos[:, : D // 2].contiguous().to(DEV), sin[:, : D // 2].contiguous().to(DEV)
    pos = torch.randint(0, 512, (M,), device=DEV) if with_pos else None
    y = ops.linear_rope_qkv(x, w, ct, st, pos, Hq + Hkv, D, S)
    assert y is not None
    with ops.gemm_variant(259):
        y0 = ops.linear_fwd(x, w)
    B = M // S
    ops.rope_(y0.view(B, S, Hq + 2 * Hkv, D)[:, :, : Hq + Hkv], ct, st, pos)
    if (M // 256) * (N // 256) >= 128 or K <= 320:
        assert torch.equal(y, y0)
    else:
        assert rel_l2(y, y0.float()) < 2e-3
    # fp32 oracle: rotate the fp32 projection
    ref = (x.float().cpu() @ w.float().cpu().t()).view(M, Hq + 2 * Hkv, D)
    p = (pos.cpu() if with_pos else torch.arange(M) % S)
    c, s_ = cos[p][:, None, :], sin[p][:, None, :]
    rot = torch.cat([-ref[..., D // 2:], ref[..., : D // 2]], -1)
    ref_r = ref.clone()
    ref_r[:, : Hq + Hkv] = (ref * c + rot * s_)[:, : Hq + Hkv]
    assert rel_l2(y.view(M, -1, D), ref_r) < 6e-3


@pytest.mark.parametrize("B,H,Hkv,D,Smax,lens,nsplit", [(1, 32, 32, 128, 640, [514], 8), (2, 8, 2, 128, 300, [37, 300], 4), (3, 4, 4, 64, 128, [1, 64, 128], 8)])
def test_gemv_attn_combine_equals_combine_then_gemv(B, H, Hkv, D, Smax, lens, nsplit):
    """round 6: the o projection fed with the split-KV partials (dllm_gemv_attn_combine after dllm_attn_decode_rope with out = NULL) ==
    attention with its combine launch followed by dllm_gemv_bf16, bit for bit (residual included)."""
    ops = _ops()
    from oracle import llm_ref
    torch.manual_seed(B + H + D + Smax)
    q, kn, vn = rnd(B, H, D).to(DEV), rnd(B, Hkv, D).to(DEV), rnd(B, Hkv, D).to(DEV)
    kc, vc = rnd(B, Smax, Hkv, D).to(DEV), rnd(B, Smax, Hkv, D).to(DEV)
    kv_len = torch.tensor(lens, dtype=torch.int32, device=DEV)
    pos = (kv_len.long() - 1)
    cos, sin = llm_ref.rope_tables(D, Smax + 1)
    ct, st = cos[:, : D // 2].contiguous().to(DEV), sin[:, : D // 2].contiguous().to(DEV)
    w, res = rnd(320, H * D, scale=(H * D) ** -0.5).to(DEV), rnd(B, 320).to(DEV)
    kc2, vc2 = kc.clone(), vc.clone()
    o = ops.attn_decode_rope(q, kn, vn, kc, vc, kv_len, ct, st, pos, nsplit=nsplit)
    y0 = ops.gemv(o.view(B, H * D), w, residual=res)
    ws = ops.attn_decode_rope(q, kn, vn, kc2, vc2, kv_len, ct, st, pos, nsplit=nsplit, partials_only=True)
    y1 = ops.gemv_attn_combine(ws, w, B, H, D, nsplit, residual=res)
    assert y1 is not None and torch.equal(y0, y1)
    assert torch.equal(kc, kc2) and torch.equal(vc, vc2)      # the cache append is unchanged


def test_gemm_huge_leading_dimension_takes_the_64bit_kernels():
    """The LDS-DMA kernels address a tile's rows with 32-bit byte offsets from a per-tile descriptor (round 6): 256 rows x ld x 2 bytes must
    stay below 2 GiB.  A strided view with ld = 2^23 elements (4 GiB per 256 rows) has to take the register-staged kernels -- same results,
    not zeros from an out-of-range descriptor offset."""
    ops = _ops()
    M, K, N, ld = 512, 256, 512, 1 << 23
    torch.manual_seed(9)
    buf = torch.zeros((M - 1) * ld + K, dtype=BF, device=DEV)      # 8.6 GB
    xs = torch.as_strided(buf, (M, K), (ld, 1))
    xc = rnd(M, K).to(DEV)
    xs.copy_(xc)
    w = rnd(N, K, scale=K ** -0.5).to(DEV)
    y = ops.linear_fwd(xs, w)
    assert rel_l2(y, xc.float().cpu() @ w.float().cpu().t()) < 4e-3
    with ops.gemm_variant(259):
        y259 = ops.linear_fwd(xs, w)
    assert torch.equal(y259, ops.linear_fwd(xc, w)) or rel_l2(y259, ops.linear_fwd(xc, w).float()) < 2e-3
    assert ops.linear_swiglu_fwd(xs, rnd(2 * 256, K).to(DEV)) is None     # the fused wrappers decline (the caller runs the unfused launches)
