"""Python operators over the HIP C-ABI (`include/dreamllm_hip.h`): raw launches + torch.autograd Functions.

PyTorch is plumbing here: it owns device memory and streams; every arithmetic op below is a hand-written gfx950
kernel.  There is NO eager fallback -- a CPU tensor or a missing library raises.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Optional

import torch

from . import _lib
from ._lib import DLLM_BF16, DLLM_F32, check

_vp = ctypes.c_void_p


def _p(t: Optional[torch.Tensor]):
    return None if t is None else _vp(t.data_ptr())


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dreamllm_amd ops run only on a ROCm device (gfx950); there is no CPU fallback")


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return DLLM_BF16
    if t.dtype == torch.float32:
        return DLLM_F32
    raise TypeError(f"unsupported dtype {t.dtype}")


def _bf16(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.bfloat16:
            raise TypeError(f"expected bfloat16, got {t.dtype}")


EPI = {None: 0, "none": 0, "gelu": 1, "quick_gelu": 2, "silu": 3, "geglu": 4}

SPLITK = True  # split-K for small grids with deep reductions (see dllm_gemm_splitk_hint)
# Opt-in: the last K slice of a tile reduces inside the GEMM launch (no separate reduce kernel).  Correct and bit-identical to the
# reduce kernel (tests), but SLOWER on MI355X: the slices of a tile run on different XCDs, so the slabs must be written through
# and read around the non-coherent L2s (109 -> 85 denoise steps/s; with release / acquire fences 72).  The reduce kernel stays.
SPLITK_FUSED_REDUCE = os.environ.get("DREAMLLM_SPLITK_FUSED", "0") == "1"
_SPLITK_COUNTERS = {}


def _splitk_counters(device):
    """Arrival counters of the in-kernel split-K reduction: int32[16384] per (device, stream), zero at rest (the kernels leave
    them zero).  Allocated once, outside any stream capture when the first split GEMM runs before the capture (warm-up)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _SPLITK_COUNTERS.get(key)
    if buf is None:
        buf = _SPLITK_COUNTERS[key] = torch.zeros(16384, dtype=torch.int32, device=device)
    return buf

# Kernel variant handed to every GEMM / conv launch (include/dreamllm_hip.h `variant`): 0 = automatic.  Python-side knob for
# tests and microbenchmarks (`with ops.gemm_variant(259): ...`); the C library itself keeps no state.
GEMM_VARIANT = 0
# A/B knob (tools, bench): DREAMLLM_GEMM_NO_RING=1 sets bit 25 of `variant` on every automatic call: the library then never picks the
# ring-buffered 128 x 128 kernel by itself (round 3's kernel selection), so one process can time both selections on one box.
GEMM_NO_RING = (1 << 25) if os.environ.get("DREAMLLM_GEMM_NO_RING", "0") == "1" else 0
# DREAMLLM_RING_4STAGE=1: bit 27, the ring kernel never takes its two-stage (two blocks per CU) form (A/B knob)
GEMM_NO_RING |= (1 << 27) if os.environ.get("DREAMLLM_RING_4STAGE", "0") == "1" else 0


# Attention kernel choice handed to dllm_attn_fwd / dllm_attn_bwd in bits 1-2 of `causal` (include/dreamllm_hip.h): 0 automatic,
# 1 the 4-wave kernels, 2 the 8-wave pipelined 256-row kernels, 3 the ping-pong kernels (forward: csrc/attn_fwd_pp.hip; backward: the
# dQ kernel of csrc/attn_bwd_pp.hip at head_dim 128, 8-wave dK / dV).  Tests run every shape through all three.
ATTN_VARIANT = 0


# A/B knob (tools / bench): DREAMLLM_W4M=0 keeps the 8-wave 256 x 256 kernel where the library would pick the four-wave one (same results)
W4M = os.environ.get("DREAMLLM_W4M", "1") != "0"
NO_W4M_BIT = 1 << 28


class gemm_variant:
    def __init__(self, tile=0, group_m=0):
        self.v = int(tile) | (int(group_m) << 16)

    def __enter__(self):
        global GEMM_VARIANT
        self.prev, GEMM_VARIANT = GEMM_VARIANT, self.v
        return self

    def __exit__(self, *exc):
        global GEMM_VARIANT
        GEMM_VARIANT = self.prev
        return False

# bench.py sets this to a list to time every launch of the MFMA GEMM/conv kernel with HIP events recorded on the stream the
# kernel is launched on (torch's current stream): entries are (start_event, end_event, flops, tag).
GEMM_PROFILE = None


_GEMM_TAG = {(0, 0): "fwd", (0, 1): "dgrad", (1, 1): "wgrad", (1, 0): "tn"}


GEMM_EVENT_POOL = []  # pre-created (and once-recorded, so the HIP event exists) timing events; see prealloc_gemm_events


def prealloc_gemm_events(n):
    """Create n timing events ahead of a profiled region: hipEventCreate costs ~0.2 ms, and 2 per GEMM launch inside the
    first timed step would otherwise show up as half a second of host time in that step."""
    for _ in range(n):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()  # torch creates the underlying event lazily at the first record
        GEMM_EVENT_POOL.append(ev)


def _timing_event():
    return GEMM_EVENT_POOL.pop() if GEMM_EVENT_POOL else torch.cuda.Event(enable_timing=True)


class _GemmTimer:
    __slots__ = ("s", "flops", "tag")

    def __init__(self, flops, tag):
        self.flops, self.tag, self.s = flops, tag, None

    def __enter__(self):
        if GEMM_PROFILE is not None:
            self.s = _timing_event()
            self.s.record()
        return self

    def __exit__(self, *a):
        if self.s is not None:
            e = _timing_event()
            e.record()
            GEMM_PROFILE.append((self.s, e, self.flops, self.tag))
        return False

# --------------------------------------------------------------------------------------------- raw launches


def rmsnorm_fwd(x, w, eps, residual=None):
    """-> (y, h, rstd); h = x + residual (or x itself).  modeling_dreamllm.py:86-91."""
    _need_gpu(x, w, residual)
    _bf16(x, w, residual)
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    rows = x2.shape[0]
    y = torch.empty_like(x2)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    h = x2
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, D)
        if not r2.is_contiguous():
            r2 = r2.contiguous()
        h = torch.empty_like(x2)
    check("dllm_rmsnorm_fwd", _p(x2), _p(r2), _p(w), _p(h) if residual is not None else None, _p(y), _p(rstd), rows, D,
          float(eps), _stream())
    return y.view(x.shape), h.view(x.shape), rstd


def rmsnorm_bwd(dy, h, w, rstd, dh_in=None, need_dw=True):
    _need_gpu(dy, h, w)
    D = h.shape[-1]
    dy2 = dy.reshape(-1, D).contiguous()
    h2 = h.reshape(-1, D)
    rows = h2.shape[0]
    dx = torch.empty_like(h2)
    dhi = dh_in.reshape(-1, D).contiguous() if dh_in is not None else None
    dw = part = None
    if need_dw:
        nparts = _lib.call("dllm_norm_bwd_nparts", rows)
        part = torch.empty(nparts, D, dtype=torch.float32, device=h.device)
        dw = torch.empty(D, dtype=w.dtype, device=h.device)
    check("dllm_rmsnorm_bwd", _p(dy2), _p(h2), _p(w), _p(rstd), _p(dhi), _p(dx), _p(part), _p(dw), _dt(w) if need_dw else 0,
          rows, D, _stream())
    return dx.view(h.shape), dw


def layernorm_fwd(x, w, b, eps, save_stats=True):
    _need_gpu(x, w, b)
    _bf16(x, w, b)
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    rows = x2.shape[0]
    y = torch.empty_like(x2)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check("dllm_layernorm_fwd", _p(x2), _p(w), _p(b), _p(y), _p(mean), _p(rstd), rows, D, float(eps), _stream())
    return y.view(x.shape), mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd, need_dw=True):
    D = x.shape[-1]
    dy2 = dy.reshape(-1, D).contiguous()
    x2 = x.reshape(-1, D).contiguous()
    rows = x2.shape[0]
    dx = torch.empty_like(x2)
    dw = db = pw = pb = None
    if need_dw:
        nparts = _lib.call("dllm_norm_bwd_nparts", rows)
        pw = torch.empty(nparts, D, dtype=torch.float32, device=x.device)
        pb = torch.empty(nparts, D, dtype=torch.float32, device=x.device)
        dw = torch.empty(D, dtype=w.dtype, device=x.device)
        db = torch.empty(D, dtype=w.dtype, device=x.device)
    check("dllm_layernorm_bwd", _p(dy2), _p(x2), _p(w), _p(mean), _p(rstd), _p(dx), _p(pw), _p(pb), _p(dw), _p(db),
          _dt(w), rows, D, _stream())
    return dx.view(x.shape), dw, db


# Stream-K tail (include/dreamllm_hip.h): one 128 MiB fp32 workspace per device, handed to the GEMM when the library says the
# problem's last round of tiles would otherwise leave most of the chip idle.  DREAMLLM_STREAMK=0 switches it off (A/B knob).
STREAMK = os.environ.get("DREAMLLM_STREAMK", "1") != "0"
STREAMK_WS_BIT = 1 << 26   # `variant` bit: the workspace handed over with splitk <= 1 is a stream-K workspace (include/dreamllm_hip.h)
_STREAMK_WS, _STREAMK_HINT = {}, {}


def _streamk_hint(M, N, K, layout_a, layout_b):
    key = (M, N, K, layout_a, layout_b)
    h = _STREAMK_HINT.get(key)
    if h is None:
        h = _STREAMK_HINT[key] = bool(_lib.call("dllm_gemm_streamk_hint", M, N, K, layout_a, layout_b))
    return h


def _streamk_workspace(device):
    """One workspace per (device, stream): two GEMMs in flight on different streams must not share slabs / counters.  Zero-filled
    once: its first 4 KiB are the per-XCD counters of the persistent walk, which every launch leaves at zero.  Under stream capture
    (the UNet inside a hipGraph) the buffer comes from the graph's pool; only the counter page is zeroed there (one 4 KiB fill node
    per replay instead of a 128 MiB one: the slabs need no initialisation)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _STREAMK_WS.get(key)
    if ws is None:
        n = int(_lib.call("dllm_gemm_streamk_ws_bytes")) // 4
        if torch.cuda.is_current_stream_capturing():
            ws = torch.empty(n, dtype=torch.float32, device=device)
            ws[:1024].zero_()
        else:
            ws = torch.zeros(n, dtype=torch.float32, device=device)
        _STREAMK_WS[key] = ws
    return ws


# XCD-synchronised persistent walk of the pipelined 256-tile kernel (csrc/gemm.hip: gemm_pipe_persist_kernel): opt-in per call through
# bit 24 of `variant`; DREAMLLM_GEMM_PERSIST=1 / ops.GEMM_PERSIST turn it on for grids of >= 4 rounds of 256-tiles.
GEMM_PERSIST = os.environ.get("DREAMLLM_GEMM_PERSIST", "0") == "1"


# Experiment knob (tools only): DREAMLLM_SPLITK_WANT=n replaces the library's split-K target for tiny grids (blocks the split aims
# at: 384 in dllm_gemm_splitk_hint) so that denoise-loop A/B runs need no rebuild.  Unset = the library's own hint.
_SPLITK_WANT = int(os.environ.get("DREAMLLM_SPLITK_WANT", "0"))


def _splitk_hint(M, N, K, layout_a=0, layout_b=0):
    if not SPLITK:
        return 1
    if _SPLITK_WANT <= 0:
        return _lib.call("dllm_gemm_splitk_hint", M, N, K, layout_a, layout_b)
    if M <= 0 or N <= 0 or (N & 3):
        return 1
    tiles = -(-M // 128) * -(-N // 128)
    ktiles = -(-K // 64)
    if tiles > 256 or ktiles < 16:
        return 1
    want = (512 // tiles) if tiles >= 128 else -(-_SPLITK_WANT // tiles)
    s = min(want, ktiles // 8, 32)
    return 1 if s < 2 else int(s)


def gemm(a, b, M, N, K, lda, ldb, layout_a, layout_b, *, out=None, out_dtype=torch.bfloat16, bias=None, residual=None,
         ldr=0, epi=None, accumulate=False, alpha=1.0):
    """C[M,N] = A*B; layout_a 0: A[m][k] k-contiguous, 1: stored [K][lda]; layout_b 0: B as [N][ldb], 1: [K][ldb]."""
    _need_gpu(a, b, out, bias, residual)
    _bf16(a, b, bias, residual)
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    sk = 1 if epi == "geglu" else _splitk_hint(M, N, K, layout_a, layout_b)
    ws = torch.empty(sk * M * N, dtype=torch.float32, device=a.device) if sk > 1 else None
    persist = 0
    if sk == 1 and (GEMM_VARIANT & 0xffff) in (0, 259, 280):
        if STREAMK and _streamk_hint(M, N, K, layout_a, layout_b):
            ws = _streamk_workspace(a.device)   # the library spreads the last partial round of 256-tiles over the CUs (stream-K tail)
            persist = STREAMK_WS_BIT            # bit 26: "this workspace holds dllm_gemm_streamk_ws_bytes() bytes"
        if GEMM_PERSIST and -(-M // 256) * -(-N // 256) >= 1024 and K % 64 == 0 and not torch.cuda.is_current_stream_capturing():
            ws = _streamk_workspace(a.device)
            persist = (1 << 24) | STREAMK_WS_BIT
    cnt = _splitk_counters(a.device) if (sk > 1 and SPLITK_FUSED_REDUCE and -(-M // 128) * -(-N // 128) <= 16384) else None
    variant = GEMM_VARIANT | (persist & STREAMK_WS_BIT) | (0 if W4M else NO_W4M_BIT)
    if (variant & 0xffff) == 0 and sk == 1 and (variant >> 16) & 0xff == 0:
        gm = _group_m_for(layout_a, layout_b, M, N, K)
        if gm == 0 and GEMM_TUNE_GROUP_M and 2.0 * M * N * K >= _TUNE_MIN_FLOPS:   # opt-in tool, off in the product path
            gm = _tuned_group_m(a, b, out, bias, residual, M, N, K, lda, ldb, ldr, layout_a, layout_b, epi, accumulate, alpha)
        variant = (gm << 16) | persist | (0 if W4M else NO_W4M_BIT)
    if (variant & 0xffff) == 0:
        variant |= GEMM_NO_RING if epi != "geglu" else (GEMM_NO_RING & (1 << 27))
    with _GemmTimer((4.0 if epi == "geglu" else 2.0) * M * N * K, _GEMM_TAG[(layout_a, layout_b)]):
        check("dllm_gemm_bf16_splitk", _p(a), _p(b), _p(out), _p(bias), _p(residual), M, N, K, lda, ldb, out.stride(0),
              ldr if residual is not None else 0, layout_a, layout_b, EPI[epi], _dt(out), int(accumulate), float(alpha),
              sk, _p(ws), _p(cnt), variant, _stream())
    return out


# ---- GROUP_M of the grouped tile order: a table keyed on (layout, M, N, K) --------------------------------------------------
# The best GROUP_M is not a function of the layout alone: sustained sweeps over the decoder layer's GEMMs (tools/gemm_sustained.py,
# profiles/r02_gemm_groupm_sustained.log) put it at 3 for the packed q|k|v and the down projection forward, 8 for the packed gate|up
# forward, 4 for the weight gradients -- 2...5 % apart from the per-layout defaults of the C side, and spiky in between (L2 /
# Infinity-Cache residency of the panels an XCD's 32 concurrent tiles share).  Round 2 timed the candidates inside the operator on
# first use; that made the kernel choice of a process depend on a noisy 3-launch measurement (ranks of one DDP job could disagree,
# the first step contained host syncs, and the driver's config-5 leg ran 17 % below the builder's).  Round 3: the choice is a
# STATIC table measured offline (gemm_group_m.json beside this file, regenerated by `tools/gemm_sustained.py --write-table`);
# shapes that are not in it take the C side's per-layout default.  `DREAMLLM_GEMM_TUNE=1` keeps the in-process timing as a tool.
GEMM_TUNE_GROUP_M = os.environ.get("DREAMLLM_GEMM_TUNE", "0") == "1"
_TUNE_MIN_FLOPS = 5e11
_TUNE_CANDIDATES = (2, 3, 4, 6, 8)
_GROUP_M_CACHE = {}


def _load_group_m_table():
    import json
    fn = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_group_m.json")
    try:
        with open(fn) as f:
            raw = json.load(f)
    except (OSError, ValueError):
        return {}
    return {tuple(int(v) for v in k.split(",")): int(gm) for k, gm in raw.get("table", {}).items()}


_GROUP_M_TABLE = _load_group_m_table()  # (layout_a, layout_b, M, N, K) -> GROUP_M
# The table was measured at T = 32768 tokens; ragged batches on compact rows (round 6) run the same weights at other token counts.  The token
# count is M of the forward / dgrad layouts and K of the weight-gradient layout: a second key without it serves those calls (>= 8192 tokens).
_GROUP_M_BY_WEIGHT = {}
for (_la, _lb, _M, _N, _K), _gm in _GROUP_M_TABLE.items():
    _GROUP_M_BY_WEIGHT[(_la, _lb, _M, _N) if (_la, _lb) == (1, 1) else (_la, _lb, _N, _K)] = _gm


def _group_m_for(layout_a, layout_b, M, N, K):
    gm = _GROUP_M_TABLE.get((layout_a, layout_b, M, N, K), 0)
    if gm == 0:
        tokens = K if (layout_a, layout_b) == (1, 1) else M
        if tokens >= 8192:
            gm = _GROUP_M_BY_WEIGHT.get((layout_a, layout_b, M, N) if (layout_a, layout_b) == (1, 1) else (layout_a, layout_b, N, K), 0)
    return gm


def _tuned_group_m(a, b, out, bias, residual, M, N, K, lda, ldb, ldr, layout_a, layout_b, epi, accumulate, alpha):
    key = (a.device.index, layout_a, layout_b, M, N, K)
    gm = _GROUP_M_CACHE.get(key)
    if gm is not None:
        return gm
    if accumulate or torch.cuda.is_current_stream_capturing():
        return 0  # default of the C side; not cached (a later idempotent call of the shape may still tune)
    best, best_t = 0, None
    args = (_p(a), _p(b), _p(out), _p(bias), _p(residual), M, N, K, lda, ldb, out.stride(0), ldr if residual is not None else 0,
            layout_a, layout_b, EPI[epi], _dt(out), 0, float(alpha), 1, None, None)
    st = _stream()
    for cand in _TUNE_CANDIDATES:
        check("dllm_gemm_bf16_splitk", *args, cand << 16, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            check("dllm_gemm_bf16_splitk", *args, cand << 16, st)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1)
        if best_t is None or t < best_t:
            best, best_t = cand, t
    _GROUP_M_CACHE[key] = best
    return best


def _as2d(x):
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1 or (x2.shape[0] > 1 and x2.stride(0) % 8 != 0):
        x2 = x2.contiguous()
    return x2


def linear_fwd(x, w, bias=None, epi=None, residual=None, out_dtype=torch.bfloat16):
    """y = epi(x W^T + bias) + residual ; x [..., K], w [N, K] (nn.Linear layout)."""
    x2 = _as2d(x)
    M, K = x2.shape
    N = w.shape[0]
    r2 = None
    if residual is not None:
        r2 = _as2d(residual)
    wc = w if w.is_contiguous() else w.contiguous()
    y = gemm(x2, wc, M, N, K, x2.stride(0), K, 0, 0, bias=bias, residual=r2, ldr=r2.stride(0) if r2 is not None else 0,
             epi=epi, out_dtype=out_dtype)
    return y.view(*x.shape[:-1], N)


def linear_geglu(x, w, bias=None):
    """diffusers GEGLU (`hidden, gate = proj(x).chunk(2, -1); hidden * gelu(gate)`, ff.net.0 of BasicTransformerBlock [ext]) as ONE
    launch: the ring-buffered GEMM pairs the hidden / gate columns of an output inside a wave (csrc/gemm_ring.hip, EPI_GEGLU) and
    writes [..., F] directly -- the [..., 2F] projection is never stored and the element-wise pass is gone.  Forward only (the
    inference / denoising path); w [2F, K], bias [2F].  Shapes the kernel does not take (K % 64, F % 64) return None."""
    x2 = _as2d(x)
    M, K = x2.shape
    F_ = w.shape[0] // 2
    if K % 64 != 0 or F_ % 64 != 0 or w.shape[0] != 2 * F_ or (GEMM_VARIANT & 0xffff) not in (0, 264):
        return None
    wc = w if w.is_contiguous() else w.contiguous()
    y = gemm(x2, wc, M, F_, K, x2.stride(0), K, 0, 0, bias=bias, epi="geglu")
    return y.view(*x.shape[:-1], F_)


# Fused SwiGLU epilogues of the MLP's two large GEMMs (round 6; csrc/gemm.hip gemm_epilogue_swiglu_*): same results as the unfused
# launches, one launch and one HBM round trip of an [M, F] / [M, 2F] tensor fewer each.  DREAMLLM_FUSED_SWIGLU=0 is the A/B knob.
FUSED_SWIGLU = os.environ.get("DREAMLLM_FUSED_SWIGLU", "1") != "0"


_DMA_LIM = (1 << 31) - 4096   # the LDS-DMA kernels address a tile's rows with 32-bit byte offsets (csrc/gemm.hip pipe_offsets_ok)


def _ld_too_wide(ld):
    return 256 * int(ld) * 2 >= _DMA_LIM


def _glu_group_m(layout, M, N, K):
    """GROUP_M (bits 0-7) and the kernel family (bits 8-9: 0 the library's choice, 1 the 8-wave kernel under `gemm_variant(259)`, 2 the four-wave
    kernel under `gemm_variant(280)`) of the fused entry points."""
    fam = {259: 1, 280: 2}.get(GEMM_VARIANT & 0xffff, 0 if W4M else 1)
    gm = (GEMM_VARIANT >> 16) & 0xff or _group_m_for(layout[0], layout[1], M, N, K)
    return (gm & 0xff) | (fam << 8)


def linear_swiglu_fwd(x, wgu):
    """(gu, act) = (x wgu^T, silu(gu[:, :F]) * gu[:, F:]) in ONE launch; x [M, K], wgu [2F, K] = packed [gate rows; up rows].
    Returns None for shapes the fused kernel does not take (the caller runs GEMM + glu_fwd)."""
    if not FUSED_SWIGLU or (GEMM_VARIANT & 0xffff) not in (0, 259, 280):
        return None
    x2 = _as2d(x)
    M, K = x2.shape
    F2 = wgu.shape[0]
    F_ = F2 // 2
    if M % 256 or F_ % 128 or K % 64 or K < 64 or not wgu.is_contiguous() or _ld_too_wide(x2.stride(0)) or (F_ + 256) * K * 2 >= _DMA_LIM:
        return None
    _need_gpu(x2, wgu)
    _bf16(x2, wgu)
    gu = torch.empty(M, F2, dtype=x.dtype, device=x.device)
    act = torch.empty(M, F_, dtype=x.dtype, device=x.device)
    with _GemmTimer(2.0 * M * F2 * K, "fwd_swiglu"):   # (its own line in bench.py's by_kind: the launch carries the SwiGLU's stores)
        check("dllm_gemm_swiglu_fwd", _p(x2), _p(wgu), _p(gu), _p(act), M, F_, K, x2.stride(0), K, F2, F_,
              _glu_group_m((0, 0), M, F2, K), _stream())
    return gu, act


FUSED_ROPE = os.environ.get("DREAMLLM_FUSED_ROPE", "1") != "0"


def linear_rope_qkv(x, wqkv, cos, sin, pos, n_rot_heads, head_dim, S):
    """qkv = x wqkv^T with the rotary embedding applied to the first `n_rot_heads` heads (the q and k heads of the packed q|k|v
    projection) in the GEMM's epilogue (csrc/gemm.hip gemm_epilogue_rope): the same results as `linear_fwd` + `rope_`, one launch.
    x [M, K]; wqkv [N, K]; cos / sin fp32 [P, 64]; pos int64 [M] or None (position = row % S).  None for shapes it does not take."""
    if not FUSED_ROPE or (GEMM_VARIANT & 0xffff) not in (0, 259, 280) or head_dim != 128:
        return None
    x2 = _as2d(x)
    M, K = x2.shape
    N = wqkv.shape[0]
    rope_cols = n_rot_heads * head_dim
    if M % 256 or N % 256 or rope_cols % 256 or rope_cols > N or K % 64 or K < 64 or not wqkv.is_contiguous() or _ld_too_wide(x2.stride(0)) \
            or _ld_too_wide(K):
        return None
    if cos.dtype != torch.float32 or sin.dtype != torch.float32 or cos.shape[-1] != 64 or not cos.is_contiguous() or not sin.is_contiguous():
        return None
    _need_gpu(x2, wqkv, cos, sin, pos)
    _bf16(x2, wqkv)
    if pos is not None:
        pos = pos.reshape(-1)
        if pos.numel() != M:
            return None
        if pos.dtype != torch.int64:
            pos = pos.long()
    out = torch.empty(M, N, dtype=x.dtype, device=x.device)
    with _GemmTimer(2.0 * M * N * K, "fwd_rope"):
        check("dllm_gemm_rope_qkv", _p(x2), _p(wqkv), _p(out), _p(cos), _p(sin), _p(pos), M, N, K, rope_cols, int(S), x2.stride(0), K, N,
              _glu_group_m((0, 0), M, N, K), _stream())
    return out


def linear_dgrad_swiglu(dy, wd, gu, dgu=None):
    """d(gate|up) [M, 2F] of act = silu(gate) * up, given dy [M, D] of the down projection (weight wd [D, F]) and the forward's packed
    gate|up buffer: the input gradient d_act = dy wd never leaves the GEMM.  None for shapes the fused kernel does not take."""
    if not FUSED_SWIGLU or (GEMM_VARIANT & 0xffff) not in (0, 259, 280):
        return None
    d2 = _as2d(dy)
    M, D = d2.shape
    F_ = wd.shape[1]
    if M % 256 or F_ % 256 or D % 64 or D < 64 or not wd.is_contiguous() or gu.shape != (M, 2 * F_) or gu.stride(1) != 1 or gu.stride(0) % 8:
        return None
    if _ld_too_wide(d2.stride(0)) or (64 * F_ + F_) * 2 >= _DMA_LIM:
        return None
    _need_gpu(d2, wd, gu)
    _bf16(d2, wd, gu)
    if dgu is None:
        dgu = torch.empty(M, 2 * F_, dtype=gu.dtype, device=gu.device)
    with _GemmTimer(2.0 * M * F_ * D, "dgrad_swiglu"):
        check("dllm_gemm_swiglu_bwd", _p(d2), _p(wd), _p(gu), _p(dgu), M, F_, D, d2.stride(0), F_, gu.stride(0), dgu.stride(0),
              _glu_group_m((0, 1), M, F_, D), _stream())
    return dgu


def linear_dgrad(dy, w):
    """dx = dy W ; dy [..., N], w [N, K]."""
    d2 = _as2d(dy)
    M, N = d2.shape
    K = w.shape[1]
    wc = w if w.is_contiguous() else w.contiguous()
    dx = gemm(d2, wc, M, K, N, d2.stride(0), K, 0, 1)
    return dx.view(*dy.shape[:-1], K)


def linear_wgrad(dy, x, out=None, accumulate=False, out_dtype=torch.bfloat16):
    """dW[N,K] = dy^T x ; dy [..., N], x [..., K]."""
    d2 = _as2d(dy)
    x2 = _as2d(x)
    T, N = d2.shape
    K = x2.shape[1]
    return gemm(d2, x2, N, K, T, d2.stride(0), x2.stride(0), 1, 1, out=out, accumulate=accumulate, out_dtype=out_dtype)


def colsum(dy, out_dtype=torch.bfloat16):
    """bias gradient: sum over rows (fp32 accumulate).  Uses the norm partial-sum reducer on a [rows, N] view."""
    d2 = _as2d(dy)
    # tiny reduction: torch handles it (plumbing-level reduction over an existing buffer)
    return d2.sum(dim=0, dtype=torch.float32).to(out_dtype)


def attn_fwd(q, k, v, causal, scale=None, seqlens=None, need_lse=True, seqstart=None):
    """q [B,Sq,H,D], k/v [B,Sk,Hkv,D] (any batch/seq/head strides, d contiguous) -> o [B,Sq,H,D], lse [B,H,Sq].
    seqlens / seqstart (int32 [B] on device): the valid span [start, start + len) of each padded row (see the header)."""
    _need_gpu(q, k, v)
    _bf16(q, k, v)
    B, Sq, H, D = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    if k.stride() != v.stride():
        v = v.contiguous()
        k = k.contiguous()
    if q.stride(-1) != 1 or k.stride(-1) != 1:
        raise ValueError("head dim must be contiguous")
    scale = scale if scale is not None else 1.0 / math.sqrt(D)
    o = torch.empty(B, Sq, H, D, dtype=q.dtype, device=q.device)
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device=q.device) if need_lse else None
    check("dllm_attn_fwd", _p(q), _p(k), _p(v), _p(o), _p(lse), _p(seqlens), _p(seqstart), B, H, Hkv, Sq, Sk, D, q.stride(0), q.stride(1),
          q.stride(2), k.stride(0), k.stride(1), k.stride(2), o.stride(0), o.stride(1), o.stride(2), float(scale),
          int(bool(causal)) | (ATTN_VARIANT << 1), _stream())
    return o, lse


def attn_bwd(dout, q, k, v, o, lse, causal, scale=None, seqlens=None, dq=None, dk=None, dv=None, seqstart=None):
    """-> dq [B,Sq,H,D], dk/dv [B,Sk,Hkv,D].  dq/dk/dv may be preallocated (strided) views, e.g. slices of one packed
    dQKV buffer; dk and dv must share strides."""
    B, Sq, H, D = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    scale = scale if scale is not None else 1.0 / math.sqrt(D)
    if k.stride() != v.stride():
        raise ValueError("k and v must share strides")
    dout = dout if (dout.stride(-1) == 1 and dout.stride() == o.stride()) else dout.contiguous()
    if dout.stride() != o.stride():
        o = o.contiguous()
    if dq is None:
        dq = torch.empty(B, Sq, H, D, dtype=q.dtype, device=q.device)
    if dk is None:
        dk = torch.empty(B, Sk, Hkv, D, dtype=q.dtype, device=q.device)
        dv = torch.empty(B, Sk, Hkv, D, dtype=q.dtype, device=q.device)
    if dk.stride() != dv.stride():
        raise ValueError("dk and dv must share strides")
    delta = torch.empty(3, B, H, Sq, dtype=torch.float32, device=q.device)  # workspace planes: delta, -delta, -lse/scale
    check("dllm_attn_bwd", _p(dout), _p(q), _p(k), _p(v), _p(o), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), _p(seqlens),
          _p(seqstart), B, H, Hkv, Sq, Sk, D, q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
          o.stride(0), o.stride(1), o.stride(2), dq.stride(0), dq.stride(1), dq.stride(2), dk.stride(0), dk.stride(1),
          dk.stride(2), float(scale), int(bool(causal)) | (ATTN_VARIANT << 1), _stream())
    return dq, dk, dv


def rope_(x, cos, sin, pos=None, backward=False):
    """In-place rotary embedding on x [B,S,NH,D] view (d contiguous, uniform token stride).  cos/sin fp32 [P, D/2]."""
    _need_gpu(x, cos, sin, pos)
    _bf16(x)
    B, S, NH, D = x.shape
    if x.stride(3) != 1 or x.stride(0) != S * x.stride(1):
        raise ValueError("rope_: x must have a uniform token stride")
    if pos is not None:
        pos = pos.reshape(-1) if pos.numel() == B * S else pos.expand(B, S).contiguous().view(-1)
        if pos.dtype != torch.int64:
            pos = pos.long()
    check("dllm_rope", _p(x), _p(cos), _p(sin), _p(pos), B * S, S, NH, D, x.stride(1), x.stride(2), int(backward), _stream())
    return x


# SwiGLU forward / backward with non-temporal loads and stores (bit 8 of the ABI's `mode`) once the operands are far larger than the
# caches (the LLM's [T, F] streams; default: operands of at least 64 MiB); tools/elementwise_ab.py is the A/B.
# Measured at [32768, 11008] (profiles/r03_elementwise_ab.log): forward 5.55 -> 5.92 TB/s, backward 5.74 -> 6.08 TB/s.
GLU_NT_MIN_BYTES = int(os.environ.get("DREAMLLM_GLU_NT_MIN_BYTES", str(64 << 20)))


def _glu_mode(mode, M, F):
    return (mode | 0x100) if (mode == 0 and 2 * M * F >= GLU_NT_MIN_BYTES) else mode


def glu_fwd(a, b, mode):
    """mode 0: silu(a)*b (SwiGLU), mode 1: gelu(a)*b (GEGLU).  a, b: [..., F] views with unit inner stride."""
    _need_gpu(a, b)
    _bf16(a, b)
    F = a.shape[-1]
    a2, b2 = a.reshape(-1, F), b.reshape(-1, F)
    M = a2.shape[0]
    out = torch.empty(M, F, dtype=a.dtype, device=a.device)
    check("dllm_glu_fwd", _p(a2), _p(b2), _p(out), M, F, a2.stride(0), b2.stride(0), F, _glu_mode(mode, M, F), _stream())
    return out.view(*a.shape[:-1], F)


def glu_bwd(dout, a, b, mode, da=None, db=None, act_out=None):
    """-> (da, db); `act_out` [M, F] (optional) additionally receives the forward product (recomputed in the same pass)."""
    F = a.shape[-1]
    a2, b2 = a.reshape(-1, F), b.reshape(-1, F)
    d2 = dout.reshape(-1, F)
    if d2.stride(-1) != 1:
        d2 = d2.contiguous()
    M = a2.shape[0]
    if da is None:
        da = torch.empty(M, F, dtype=a.dtype, device=a.device)
        db = torch.empty(M, F, dtype=a.dtype, device=a.device)
    check("dllm_glu_bwd", _p(d2), _p(a2), _p(b2), _p(da), _p(db), _p(act_out), M, F, d2.stride(0), a2.stride(0), b2.stride(0),
          da.stride(0), db.stride(0), act_out.stride(0) if act_out is not None else 0, _glu_mode(mode, M, F), _stream())
    return da, db


def gather_rows(table, idx):
    _need_gpu(table, idx)
    _bf16(table)
    idx = idx.reshape(-1).contiguous()
    n, D = idx.numel(), table.shape[1]
    out = torch.empty(n, D, dtype=table.dtype, device=table.device)
    check("dllm_gather_rows", _p(table), _p(idx), _p(out), n, D, table.stride(0), D, _stream())
    return out


def scatter_rows_(dst, idx, src):
    """dst[idx[i]] = src[i] (idx unique)."""
    _need_gpu(dst, idx, src)
    _bf16(dst, src)
    idx = idx.reshape(-1).contiguous()
    src2 = src.reshape(-1, src.shape[-1])
    if not src2.is_contiguous():
        src2 = src2.contiguous()
    check("dllm_scatter_rows", _p(src2), _p(idx), _p(dst), idx.numel(), dst.shape[1], src2.stride(0), dst.stride(0), _stream())
    return dst


# positions per chunk of the embedding gradient's segment sums (0: one work-group per token id, as before round 6)
EMB_BWD_CHUNK = int(os.environ.get("DREAMLLM_EMB_BWD_CHUNK", "128"))


def embedding_bwd(dy, ids, num_rows):
    """Deterministic embedding gradient: sort ids, segment-sum rows in fp32."""
    dy2 = dy.reshape(-1, dy.shape[-1])
    if not dy2.is_contiguous():
        dy2 = dy2.contiguous()
    ids = ids.reshape(-1)
    sorted_ids, order = torch.sort(ids, stable=True)
    uid, counts = torch.unique_consecutive(sorted_ids, return_counts=True)
    seg = torch.zeros(uid.numel() + 1, dtype=torch.int64, device=ids.device)
    seg[1:] = torch.cumsum(counts, 0)
    dtable = torch.zeros(num_rows, dy2.shape[1], dtype=dy.dtype, device=dy.device)
    U, D = uid.numel(), dy2.shape[1]
    order = order.contiguous()
    if EMB_BWD_CHUNK > 0 and dy.dtype == torch.bfloat16 and ids.numel() >= 16 * EMB_BWD_CHUNK:
        # one work-group walks a segment: a token with ~10^4 positions in the batch (the image placeholder) took 2.7 ms alone.  Segments are
        # cut into chunks of EMB_BWD_CHUNK positions, the chunks summed into fp32 partial rows, then each token's partial rows in order
        # (deterministic; tokens with a single chunk -- all but a handful -- add their rows in the same order as before)
        L = EMB_BWD_CHUNK
        nchunk = (counts + (L - 1)) // L
        vfirst = torch.zeros(U + 1, dtype=torch.int64, device=ids.device)
        vfirst[1:] = torch.cumsum(nchunk, 0)
        V = int(vfirst[-1])   # (the sort / unique above already synchronised with the host)
        if V > U:
            u_of_v = torch.repeat_interleave(torch.arange(U, device=ids.device), nchunk, output_size=V)
            c_of_v = torch.arange(V, device=ids.device) - vfirst[u_of_v]
            vseg = torch.empty(V + 1, dtype=torch.int64, device=ids.device)
            vseg[:V] = seg[u_of_v] + c_of_v * L
            vseg[V] = ids.numel()
            partial = torch.empty(V, D, dtype=torch.float32, device=dy.device)
            check("dllm_segment_sum_rows_ex", _p(dy2), _p(order), _p(vseg), None, _p(partial), V, D, dy2.stride(0), D, 0, 1, _stream())
            check("dllm_segment_sum_rows_ex", _p(partial), None, _p(vfirst), _p(uid.contiguous()), _p(dtable), U, D, D, dtable.stride(0), 1, 0,
                  _stream())
            return dtable
    check("dllm_segment_sum_rows", _p(dy2), _p(order), _p(seg), _p(uid.contiguous()), _p(dtable), U, D, dy2.stride(0),
          dtable.stride(0), _stream())
    return dtable


def cross_entropy_rows(logits, labels, dlogits=None, gscale=None):
    """Per-row CE over fp32 logits [R, V]; labels int64 [R] (-100 ignored).  Optionally writes bf16 dlogits."""
    _need_gpu(logits, labels)
    if logits.dtype != torch.float32:
        raise TypeError("cross_entropy_rows expects fp32 logits (reference: logits.float())")
    R, V = logits.shape
    loss_row = torch.empty(R, dtype=torch.float32, device=logits.device)
    check("dllm_cross_entropy", _p(logits), _p(labels), _p(loss_row), _p(dlogits), _p(gscale), R, V, logits.stride(0),
          dlogits.stride(0) if dlogits is not None else 0, _stream())
    return loss_row


def softmax_rows(x):
    """softmax over the last dim of fp32 x [..., C] -> bf16 (block-per-row kernel)."""
    _need_gpu(x)
    if x.dtype != torch.float32:
        raise TypeError("softmax_rows expects fp32 scores")
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    y = torch.empty(x2.shape, dtype=torch.bfloat16, device=x.device)
    check("dllm_softmax_rows", _p(x2), _p(y), x2.shape[0], C, x2.stride(0), y.stride(0), _stream())
    return y.view(x.shape)


def attention_wide_head(q, k, v, scale=None):
    """softmax(q k^T * scale) v for ONE head of any width (multiple of 8): q [N,Sq,C], k/v [N,Sk,C] bf16 -> [N,Sq,C].
    Two MFMA GEMMs around a row-softmax kernel; the [Sq,Sk] scores are materialised in fp32 (64 MB at 4096 tokens), which is
    what the VAE mid-block attention needs once per image and the flash kernels (head_dim 64 / 128) cannot serve."""
    _need_gpu(q, k, v)
    _bf16(q, k, v)
    N, Sq, C = q.shape
    Sk = k.shape[1]
    scale = scale if scale is not None else 1.0 / math.sqrt(C)
    out = torch.empty(N, Sq, C, dtype=q.dtype, device=q.device)
    for n in range(N):
        qn, kn, vn = q[n].contiguous(), k[n].contiguous(), v[n].contiguous()
        s = gemm(qn, kn, Sq, Sk, C, C, C, 0, 0, out_dtype=torch.float32, alpha=scale)
        p = softmax_rows(s)
        gemm(p, vn, Sq, C, Sk, Sk, C, 0, 1, out=out[n])
    return out


def adamw_(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, grad_scale_dev=None):
    _need_gpu(p, g, m, v)
    if g.dtype != p.dtype:
        g = g.to(p.dtype)
    check("dllm_adamw", _p(p), _p(g.contiguous()), _p(m), _p(v), p.numel(), _dt(p), _dt(m), float(lr), float(beta1),
          float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale), _p(grad_scale_dev), _stream())


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _multi_ok(t):
    return t.dtype == torch.bfloat16 and t.is_contiguous() and t.numel() % 8 == 0 and t.data_ptr() % 16 == 0


def adamw_multi_(ps, gs, ms, vs, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, grad_scale_dev=None):
    """One fused AdamW update over a LIST of bf16 tensors (bf16 moments, shared hyper-parameters and step): the pointer table
    travels as a kernel argument, 48 tensors per launch (include/dreamllm_hip.h: dllm_adamw_multi).  Every tensor must satisfy
    `_multi_ok`; same update, element for element, as `adamw_`."""
    if not ps:
        return
    _need_gpu(*ps)
    n = (ctypes.c_int64 * len(ps))(*[t.numel() for t in ps])
    check("dllm_adamw_multi", _ptr_array(ps), _ptr_array(gs), _ptr_array(ms), _ptr_array(vs), n, len(ps), float(lr), float(beta1),
          float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale), _p(grad_scale_dev), _stream())


def sumsq_multi(xs):
    """fp32 per-chunk partial sums of squares of a list of bf16 tensors (`_multi_ok`), in tensor-then-chunk order: a fixed layout, so
    data-parallel replicas derive bit-identical norms.  ceil(len / 48) launches."""
    _need_gpu(*xs)
    n = (ctypes.c_int64 * len(xs))(*[t.numel() for t in xs])
    parts = int(_lib.lib().dllm_sumsq_multi_parts(n, len(xs)))
    out = torch.empty(parts, dtype=torch.float32, device=xs[0].device)
    check("dllm_sumsq_multi", _ptr_array(xs), n, len(xs), _p(out), _stream())
    return out


SUMSQ_PARTS = 256


def sumsq_partials_(x, partials):
    """partials: zeroed fp32 [SUMSQ_PARTS] slice receiving per-block partial sums of x^2 (deterministic)."""
    check("dllm_sumsq", _p(x), x.numel(), _dt(x), _p(partials), _stream())


def reduce_sum_f32(x):
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    check("dllm_reduce_sum_f32", _p(x), x.numel(), _p(out), _stream())
    return out


def add_bcast(a, b):
    """a + b with b broadcast along the leading dims (b.numel() divides a.numel(); trailing layout identical)."""
    _need_gpu(a, b)
    _bf16(a, b)
    a = a.contiguous()
    b = b.contiguous()
    out = torch.empty_like(a)
    check("dllm_add_bcast", _p(a), _p(b), _p(out), a.numel(), b.numel(), _stream())
    return out


def add_rowgroup(a, b):
    """a [G, R, C] + b [G, C] broadcast over R (UNet time-embedding add on NHWC activations)."""
    _need_gpu(a, b)
    _bf16(a, b)
    a = a.contiguous()
    b = b.contiguous()
    G, C = b.shape
    out = torch.empty_like(a)
    check("dllm_add_rowgroup", _p(a), _p(b), _p(out), G, a.numel() // (G * C), C, _stream())
    return out


ACT = {"gelu": 1, "quick_gelu": 2, "silu": 3}


def act_fwd(x, mode):
    _need_gpu(x)
    _bf16(x)
    x = x.contiguous()
    out = torch.empty_like(x)
    check("dllm_act_fwd", _p(x), _p(out), x.numel(), ACT[mode], _stream())
    return out


def act_bwd(dy, x, mode):
    dy = dy.contiguous()
    dx = torch.empty_like(x)
    check("dllm_act_bwd", _p(dy), _p(x), _p(dx), x.numel(), ACT[mode], _stream())
    return dx


# --------------------------------------------------------------------------------------------- autograd Functions
class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.bshape = b.shape
        ctx.same = a.shape == b.shape
        return add_bcast(a, b)

    @staticmethod
    def backward(ctx, d):
        db = d if ctx.same else d.reshape(-1, *ctx.bshape).sum(0, dtype=torch.float32).to(d.dtype)
        return d, db


def add(a, b):
    """bf16 add (b may be a trailing-shape broadcast of a)."""
    return AddFn.apply(a, b)


class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.mode = mode
        return act_fwd(x, mode)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return act_bwd(dy, x, ctx.mode), None


def gelu(x):
    return ActFn.apply(x, "gelu")


def silu(x):
    return ActFn.apply(x, "silu")


def quick_gelu(x):
    return ActFn.apply(x, "quick_gelu")



class RMSNormFn(torch.autograd.Function):
    """DreamLLMRMSNorm.forward (modeling_dreamllm.py:86-91)."""

    @staticmethod
    def forward(ctx, x, weight, eps):
        y, h, rstd = rmsnorm_fwd(x, weight, eps, None)
        ctx.save_for_backward(x, weight, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, rstd = ctx.saved_tensors
        dx, dw = rmsnorm_bwd(dy, x, weight, rstd, dh_in=None, need_dw=ctx.needs_input_grad[1])
        return dx, dw, None


class AddRMSNormFn(torch.autograd.Function):
    """h = x + residual (decoder residual stream, modeling_dreamllm.py:632,638) fused with the RMSNorm that follows."""

    @staticmethod
    def forward(ctx, x, residual, weight, eps):
        y, h, rstd = rmsnorm_fwd(x, weight, eps, residual)
        ctx.save_for_backward(h, weight, rstd)
        return h, y

    @staticmethod
    def backward(ctx, dh, dy):
        h, weight, rstd = ctx.saved_tensors
        dx, dw = rmsnorm_bwd(dy, h, weight, rstd, dh_in=dh, need_dw=ctx.needs_input_grad[2])
        return dx, dx, dw, None


def rmsnorm(x, weight, eps):
    return RMSNormFn.apply(x, weight, eps)


def add_rmsnorm(x, residual, weight, eps):
    """h = x + residual ; y = RMSNorm(h) -> (h, y)."""
    return AddRMSNormFn.apply(x, residual, weight, eps)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        need = x.requires_grad or weight.requires_grad
        y, mean, rstd = layernorm_fwd(x, weight, bias, eps, save_stats=need)
        if need:
            ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, rstd = ctx.saved_tensors
        need_dw = ctx.needs_input_grad[1]
        dx, dw, db = layernorm_bwd(dy, x, weight, mean, rstd, need_dw=need_dw)
        return dx, dw, (db if ctx.needs_input_grad[2] else None), None


def layernorm(x, weight, bias, eps):
    return LayerNormFn.apply(x, weight, bias, eps)


class LinearFn(torch.autograd.Function):
    """y = x W^T (+ bias) (+ residual): nn.Linear call sites of the hot path (q/k/v/o, gate/up/down, lm_head, projectors)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, out_fp32):
        y = linear_fwd(x, weight, bias=bias, residual=residual, out_dtype=torch.float32 if out_fp32 else torch.bfloat16)
        ctx.save_for_backward(x if weight.requires_grad else None, weight if x.requires_grad else None)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.wshape = weight.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = linear_dgrad(dy, weight)
        if ctx.needs_input_grad[1]:
            dw = linear_wgrad(dy, x)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy)
        dres = dy if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres, None


def linear(x, weight, bias=None, residual=None, out_fp32=False):
    return LinearFn.apply(x, weight, bias, residual, out_fp32)


class GLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mode):
        ctx.save_for_backward(a, b)
        ctx.mode = mode
        return glu_fwd(a, b, mode)

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        da, db = glu_bwd(dout, a, b, ctx.mode)
        return da.view(a.shape), db.view(b.shape), None


def swiglu(gate, up):
    return GLUFn.apply(gate, up, 0)


def geglu(gate, value):
    return GLUFn.apply(gate, value, 1)


class RoPEFn(torch.autograd.Function):
    """apply_rotary_pos_emb (modeling_dreamllm.py:184-209) on q and k in place ([B,S,H,D] views of the QKV GEMM output)."""

    @staticmethod
    def forward(ctx, q, k, cos, sin, pos):
        rope_(q, cos, sin, pos)
        rope_(k, cos, sin, pos)
        ctx.mark_dirty(q, k)
        ctx.save_for_backward(cos, sin, pos)
        return q, k

    @staticmethod
    def backward(ctx, dq, dk):
        cos, sin, pos = ctx.saved_tensors
        dq = dq.contiguous() if not _uniform_tok(dq) else dq.clone()
        dk = dk.contiguous() if not _uniform_tok(dk) else dk.clone()
        rope_(dq, cos, sin, pos, backward=True)
        rope_(dk, cos, sin, pos, backward=True)
        return dq, dk, None, None, None


def _uniform_tok(x):
    return x.stride(3) == 1 and x.stride(0) == x.shape[1] * x.stride(1)


def rope(q, k, cos, sin, pos=None):
    return RoPEFn.apply(q, k, cos, sin, pos)


class FlashAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, causal, scale, seqlens, seqstart):
        need = q.requires_grad or k.requires_grad or v.requires_grad
        o, lse = attn_fwd(q, k, v, causal, scale, seqlens, need_lse=need, seqstart=seqstart)
        if need:
            ctx.save_for_backward(q, k, v, o, lse, seqlens, seqstart)
        ctx.causal, ctx.scale = causal, scale
        return o

    @staticmethod
    def backward(ctx, dout):
        q, k, v, o, lse, seqlens, seqstart = ctx.saved_tensors
        dq, dk, dv = attn_bwd(dout, q, k, v, o, lse, ctx.causal, ctx.scale, seqlens, seqstart=seqstart)
        return dq, dk, dv, None, None, None, None


def flash_attn(q, k, v, causal=False, scale=None, seqlens=None, seqstart=None):
    """q [B,Sq,H,D], k/v [B,Sk,Hkv,D] -> [B,Sq,H,D] (flash_attn_func layout, modeling_dreamllm.py:547-549)."""
    return FlashAttnFn.apply(q, k, v, causal, scale, seqlens, seqstart)


class EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, ids):
        ctx.save_for_backward(ids)
        ctx.nrows = weight.shape[0]
        return gather_rows(weight, ids).view(*ids.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        return embedding_bwd(dy, ids, ctx.nrows), None


def embedding(weight, ids):
    return EmbeddingFn.apply(weight, ids)


class ScatterRowsFn(torch.autograd.Function):
    """Multimodal splice (modeling_dreamllm.py:1081-1141): base[idx[i]] = rows[i], as ONE index-scatter kernel."""

    @staticmethod
    def forward(ctx, base, idx, rows):
        out = base.clone() if base.requires_grad or True else base
        scatter_rows_(out.view(-1, out.shape[-1]), idx, rows)
        ctx.save_for_backward(idx)
        ctx.rows_shape = rows.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        d2 = dout.reshape(-1, dout.shape[-1])
        drows = gather_rows(d2, idx).view(ctx.rows_shape) if ctx.needs_input_grad[2] else None
        dbase = None
        if ctx.needs_input_grad[0]:
            dbase = dout.clone()
            zeros = torch.zeros(idx.numel(), dout.shape[-1], dtype=dout.dtype, device=dout.device)
            scatter_rows_(dbase.view(-1, dout.shape[-1]), idx, zeros)
        return dbase, None, drows


def scatter_rows(base, idx, rows):
    return ScatterRowsFn.apply(base, idx, rows)


class GatherRowsFn(torch.autograd.Function):
    """out[i] = x[idx[i]] with unique idx (dream-query hidden-state gather, modeling_dreamllm.py:1399-1418)."""

    @staticmethod
    def forward(ctx, x2d, idx):
        ctx.save_for_backward(idx)
        ctx.shape = x2d.shape
        return gather_rows(x2d, idx)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dx = torch.zeros(ctx.shape, dtype=dout.dtype, device=dout.device)
        scatter_rows_(dx, idx, dout)
        return dx, None


def gather_rows_unique(x2d, idx):
    return GatherRowsFn.apply(x2d, idx)


class PackRowsFn(torch.autograd.Function):
    """Padded token grid -> compact rows (ragged batches, round 6): out[i] = x[idx[i]] for the tv valid tokens; rows tv.. of the compact
    matrix are filler (they repeat one pad position so that the row count is a multiple of the GEMM tile) and carry no gradient."""

    @staticmethod
    def forward(ctx, x2d, idx, tv):
        ctx.save_for_backward(idx)
        ctx.shape, ctx.tv = x2d.shape, tv
        return gather_rows(x2d, idx)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dx = torch.zeros(ctx.shape, dtype=dout.dtype, device=dout.device)
        scatter_rows_(dx, idx[: ctx.tv], dout[: ctx.tv].contiguous())
        return dx, None, None


class UnpackRowsFn(torch.autograd.Function):
    """Compact rows -> padded token grid [n_rows, H]: out[idx[i]] = rows[i] for the tv valid tokens, zeros elsewhere (pad_input
    semantics, modeling_dreamllm.py:545)."""

    @staticmethod
    def forward(ctx, rows, idx, tv, n_rows):
        ctx.save_for_backward(idx)
        ctx.tv, ctx.rows_shape = tv, rows.shape
        out = torch.zeros(n_rows, rows.shape[-1], dtype=rows.dtype, device=rows.device)
        scatter_rows_(out, idx[:tv], rows[:tv].contiguous())
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        d = torch.zeros(ctx.rows_shape, dtype=dout.dtype, device=dout.device)
        d[: ctx.tv] = gather_rows(dout.contiguous(), idx[: ctx.tv])
        return d, None, None, None


LM_HEAD_CE_CHUNK_ROWS = int(os.environ.get("DREAMLLM_CE_CHUNK_ROWS", "4096"))  # rows of hidden states per chunk of the fused lm_head + CE (525 MB of fp32 logits at V = 32008)


def _pad_vocab(weight):
    """The GEMMs of the backward contract over / produce the vocabulary axis: they need 16-byte rows (V % 8 == 0; DreamLLM-SDXL has
    32009) and run on the pipelined LDS-DMA kernel only when the contraction length is a multiple of 64 (32008 = 64 * 500 + 8: the
    dHidden GEMM of every chunk otherwise falls to the register-staged kernel, 0.87 instead of 1.25 PF).  So the unit runs on a
    weight zero-padded to a multiple of 64 rows (one 262 MB copy per call, 0.1 ms); the CE kernel still sees V columns (row pitch
    Vp), so the pad columns enter neither the softmax nor the gradient."""
    V = weight.shape[0]
    Vp = (V + 63) // 64 * 64
    if Vp == V:
        return weight
    wp = torch.zeros(Vp, weight.shape[1], dtype=weight.dtype, device=weight.device)
    wp[:V].copy_(weight)
    return wp


class LMHeadCEFn(torch.autograd.Function):
    """lm_head + fp32 logits + shifted masked CE (modeling_dreamllm.py:1452-1470) as ONE unit that never holds the [T, V]
    logits: the reference materialises them in fp32 (4.2 GB at T = 32768, V = 32008) and reads them ~3 times.

    hidden [R, d] bf16, weight [V, d] bf16, labels int64 [R] (already shifted; -100 ignored) -> mean CE over valid rows.
    Rows are processed in chunks of LM_HEAD_CE_CHUNK_ROWS: logits chunk (fp32, straight from the MFMA accumulators) ->
    `dllm_cross_entropy` (per-row loss AND bf16 dlogits = (softmax - onehot) / n_valid in the same pass, n_valid read from a
    device scalar: no host sync) -> dHidden rows (dgrad GEMM) and dW += dlogits^T hidden (wgrad GEMM accumulating in fp32).
    Loss and both gradients therefore come out of the FORWARD; backward only scales them by the incoming dloss.  Peak extra
    memory is one chunk (0.8 GB) instead of 4.2 GB + 2.1 GB, and a chunk's logits are consumed while still cache-resident.
    Same arithmetic as the unfused form (same GEMM kernel, same CE kernel, fp32 accumulation of dW across chunks)."""

    @staticmethod
    def forward(ctx, hidden, weight, labels, with_grads=True):
        R, d = hidden.shape
        V = weight.shape[0]
        wp = _pad_vocab(weight)
        Vp = wp.shape[0]
        # `needs_input_grad` is True under torch.no_grad() as well (it reflects requires_grad of the inputs); `lm_head_ce` passes
        # `with_grads=False` there so that evaluation with labels does not pay for the dgrad / wgrad GEMMs and the fp32 dW buffer
        need_dh, need_dw = (ctx.needs_input_grad[0] and with_grads), (ctx.needs_input_grad[1] and with_grads)
        nvalid = (labels != -100).sum()
        denom = torch.clamp(nvalid, min=1).to(torch.float32)
        gscale = (1.0 / denom).reshape(1).contiguous()
        loss_rows = torch.empty(R, dtype=torch.float32, device=hidden.device)
        dh = torch.empty(R, d, dtype=hidden.dtype, device=hidden.device) if need_dh else None
        dw32 = None
        C = max(256, int(LM_HEAD_CE_CHUNK_ROWS))
        for i, r0 in enumerate(range(0, R, C)):
            r1 = min(R, r0 + C)
            h_c = hidden[r0:r1]
            logits_c = linear_fwd(h_c, wp, out_dtype=torch.float32)
            dl_c = None
            if need_dh or need_dw:
                dl_c = (torch.zeros if Vp != V else torch.empty)(r1 - r0, Vp, dtype=torch.bfloat16, device=hidden.device)
            loss_rows[r0:r1] = cross_entropy_rows(logits_c[:, :V], labels[r0:r1], dlogits=None if dl_c is None else dl_c[:, :V],
                                                  gscale=gscale if dl_c is not None else None)
            del logits_c
            if need_dh:
                gemm(dl_c, wp, r1 - r0, d, Vp, dl_c.stride(0), d, 0, 1, out=dh[r0:r1])
            if need_dw:
                if dw32 is None:
                    dw32 = torch.empty(Vp, d, dtype=torch.float32, device=hidden.device)
                linear_wgrad(dl_c, h_c, out=dw32, accumulate=i > 0, out_dtype=torch.float32)
        if need_dw and dw32 is None:   # no rows at all (a batch without a single labelled token): the gradient is zero, not absent
            dw32 = torch.zeros(Vp, d, dtype=torch.float32, device=hidden.device)
        dw = dw32[:V].to(weight.dtype) if dw32 is not None else None
        ctx.save_for_backward(dh, dw)
        return loss_rows.sum() / denom

    @staticmethod
    def backward(ctx, dloss):
        dh, dw = ctx.saved_tensors
        g = dloss.to(torch.float32)
        # dloss is a scalar (1 / loss_scale, times the lm weight): one multiply per gradient element by the fp32 scalar (a value
        # such as 1/3 from gradient accumulation is not rounded to bf16 first), OUT of place so that a second backward
        # (retain_graph) scales the saved gradients once, not twice
        if dh is not None:
            dh = dh.to(torch.float32).mul_(g).to(dh.dtype)   # (bf16 tensor * 0-dim fp32 tensor would round g to bf16 first)
        if dw is not None:
            dw = dw.to(torch.float32).mul_(g).to(dw.dtype)
        return dh, dw, None, None


class LMHeadCELogitsFn(torch.autograd.Function):
    """The unfused form: also returns the full fp32 logits [R, V] (callers that want them next to the loss, e.g. an
    evaluation loop with `prediction_loss_only=False`).  Backward recomputes the softmax from the saved logits."""

    @staticmethod
    def forward(ctx, hidden, weight, labels):
        V = weight.shape[0]
        wp = _pad_vocab(weight)
        logits_p = linear_fwd(hidden, wp, out_dtype=torch.float32)
        logits = logits_p[:, :V]
        loss_row = cross_entropy_rows(logits, labels)
        nvalid = (labels != -100).sum()
        denom = torch.clamp(nvalid, min=1).to(torch.float32)
        ctx.save_for_backward(hidden, wp, logits_p, labels, denom)
        ctx.V = V
        ctx.mark_non_differentiable(logits)
        return loss_row.sum() / denom, logits

    @staticmethod
    def backward(ctx, dloss, _dlogits_unused):
        hidden, wp, logits_p, labels, denom = ctx.saved_tensors
        V, Vp = ctx.V, wp.shape[0]
        gscale = (dloss.to(torch.float32) / denom).reshape(1).contiguous()
        if Vp != V:
            dlogits_p = torch.zeros(logits_p.shape, dtype=torch.bfloat16, device=logits_p.device)
        else:
            dlogits_p = torch.empty(logits_p.shape, dtype=torch.bfloat16, device=logits_p.device)
        cross_entropy_rows(logits_p[:, :V], labels, dlogits=dlogits_p[:, :V], gscale=gscale)
        dh = linear_dgrad(dlogits_p, wp) if ctx.needs_input_grad[0] else None
        dw = linear_wgrad(dlogits_p, hidden) if ctx.needs_input_grad[1] else None
        if dw is not None and Vp != V:
            dw = dw[:V]
        return dh, dw, None


def lm_head_ce(hidden2d, weight, labels1d, return_logits=False, rows=None):
    """-> loss (fused, no [T,V] tensor) or (loss, fp32 logits) with `return_logits=True` (unfused).
    `rows` (int64, unique, ascending: the rows whose label is not -100, from the data pipeline): the fused unit then runs on those
    rows only -- ignored rows (image / dream patch slots, padding: a third of an interleaved document) contribute neither to the
    loss nor to any gradient, so their logits are never computed.  Same loss, same gradients (zero rows of dHidden)."""
    if return_logits:
        return LMHeadCELogitsFn.apply(hidden2d, weight, labels1d)
    if rows is not None:
        hidden2d = gather_rows_unique(hidden2d, rows)   # backward: scatter into a zero [T, d]
        labels1d = labels1d[rows]
    return LMHeadCEFn.apply(hidden2d, weight, labels1d, torch.is_grad_enabled())


# --------------------------------------------------------------------------------------------- UNet operators (NHWC)
def conv2d_nhwc(x, w2d, CO, KH, KW, stride=1, pad=1, OH=None, OW=None, bias=None, residual=None, image_bias=None, up2=False,
                even_only=False, epi=None, out_dtype=torch.bfloat16):
    """x [N,H,W,C] bf16 (C % 8 == 0), w2d [CO, KH*KW*C] k-contiguous -> [N,OH,OW,CO].  Implicit-GEMM MFMA kernel."""
    _need_gpu(x, w2d, bias, residual, image_bias)
    _bf16(x, w2d, bias, residual, image_bias)
    x = x if x.is_contiguous() else x.contiguous()
    N, H, W, C = x.shape
    if OH is None:
        Hl, Wl = (2 * H, 2 * W) if up2 else (H, W)
        OH = (Hl + 2 * pad - KH) // stride + 1
        OW = (Wl + 2 * pad - KW) // stride + 1
    out = torch.empty(N, OH, OW, CO, dtype=out_dtype, device=x.device)
    if residual is not None and not residual.is_contiguous():
        residual = residual.contiguous()
    Mg = N * OH * OW
    sk = _splitk_hint(Mg, CO, KH * KW * C, 2, 0)
    ws = torch.empty(sk * Mg * CO, dtype=torch.float32, device=x.device) if sk > 1 else None
    cnt = _splitk_counters(x.device) if (sk > 1 and SPLITK_FUSED_REDUCE and -(-Mg // 128) * -(-CO // 128) <= 16384) else None
    skbit = 0
    if sk == 1 and STREAMK and GEMM_VARIANT == 0 and C % 64 == 0 and _streamk_hint(Mg, CO, KH * KW * C, 2, 0):
        ws = _streamk_workspace(x.device)   # small grid, deep K: every tile's K loop is spread over the CUs (stream-K)
        skbit = STREAMK_WS_BIT
    with _GemmTimer(2.0 * N * OH * OW * CO * KH * KW * C, "conv"):
        check("dllm_conv2d_nhwc_bf16_splitk", _p(x), _p(w2d), _p(out), _p(bias), _p(residual), _p(image_bias), N, H, W, C, OH,
              OW, CO, KH, KW, stride, pad, int(up2), int(even_only), EPI[epi], _dt(out), sk, _p(ws), _p(cnt),
              GEMM_VARIANT | skbit | (GEMM_NO_RING if (GEMM_VARIANT & 0xffff) == 0 else 0), _stream())
    return out


def sumpool2(x):
    """[N,2H,2W,C] -> [N,H,W,C]: backward of nearest-2x upsampling."""
    N, H2, W2, C = x.shape
    x = x.contiguous()
    out = torch.empty(N, H2 // 2, W2 // 2, C, dtype=x.dtype, device=x.device)
    check("dllm_sumpool2_nhwc", _p(x), _p(out), N, H2 // 2, W2 // 2, C, _stream())
    return out


class ConvFn(torch.autograd.Function):
    """Conv2d on NHWC with FROZEN weights: forward + input gradient only (the UNet/VAE are frozen in every dreamllm recipe,
    modeling_plugins.py:405-407).  mode: "same" (stride 1), "down" (stride 2), "up" (nearest-2x upsample fused in front)."""

    @staticmethod
    def forward(ctx, x, w_fwd, w_bwd, bias, residual, image_bias, CO, K, mode):
        pad = 1 if K == 3 else 0
        if mode == "down_asym":  # VAE encoder Downsample2D: F.pad(x, (0,1,0,1)) then stride-2 conv with padding 0
            y = conv2d_nhwc(x, w_fwd, CO, K, K, stride=2, pad=0, OH=x.shape[1] // 2, OW=x.shape[2] // 2, bias=bias)
        else:
            y = conv2d_nhwc(x, w_fwd, CO, K, K, stride=2 if mode == "down" else 1, pad=pad, bias=bias, residual=residual,
                            image_bias=image_bias, up2=(mode == "up"))
        ctx.save_for_backward(w_bwd)
        ctx.meta = (x.shape, K, mode, residual is not None, image_bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (w_bwd,) = ctx.saved_tensors
        xshape, K, mode, has_res, has_ib = ctx.meta
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            N, H, W, C = xshape
            pad = 1 if K == 3 else 0
            if dy.shape[-1] % 8 != 0:  # conv_out: 4 output channels -> pad the gradient's channels for 16-byte rows
                dyp = torch.zeros(*dy.shape[:-1], (dy.shape[-1] + 7) // 8 * 8, dtype=dy.dtype, device=dy.device)
                dyp[..., : dy.shape[-1]] = dy
            else:
                dyp = dy
            Cw = w_bwd.shape[0]
            if mode == "same":
                dx = conv2d_nhwc(dyp, w_bwd, Cw, K, K, 1, pad)
            elif mode == "down":
                dx = conv2d_nhwc(dyp, w_bwd, Cw, K, K, 1, pad, OH=H, OW=W, even_only=True)
            else:  # up: gradient w.r.t. the upsampled image, then fold the 2x2 replicas
                dx = sumpool2(conv2d_nhwc(dyp, w_bwd, Cw, K, K, 1, pad))
            if Cw != C:  # channel-padded input (conv_in)
                dx = torch.nn.functional.pad(dx, (0, C - Cw))
        dres = dy if (has_res and ctx.needs_input_grad[4]) else None
        dib = None
        if has_ib and ctx.needs_input_grad[5]:
            dib = dy.reshape(dy.shape[0], -1, dy.shape[-1]).sum(1, dtype=torch.float32).to(dy.dtype)
        return dx, None, None, None, dres, dib, None, None, None


# GroupNorm for tiny batches (the denoising loop): 4 blocks per (image, group) meeting in a persistent zero-initialised sync buffer
# (include/dreamllm_hip.h: dllm_groupnorm_fwd_split); one buffer per (device, stream).  DREAMLLM_GN_SPLIT=0 disables it.
GN_SPLIT = os.environ.get("DREAMLLM_GN_SPLIT", "1") != "0"
_GN_SYNC = {}


def _gn_sync(device):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    b = _GN_SYNC.get(key)
    if b is None:
        b = _GN_SYNC[key] = torch.zeros(128 * 32, dtype=torch.int32, device=device)
    return b


def gn_split_fallbacks(device):
    """How often a block of the split GroupNorm gave up on its partners and computed its slice alone (word 31 of every slot of every
    sync buffer of `device`): 0 unless the GPU is shared with another process."""
    return sum(int(b.view(-1, 32)[:, 31].sum()) for (dev, _), b in _GN_SYNC.items() if dev == device)


def groupnorm_fwd(x, gamma, beta, G, eps, act):
    _need_gpu(x, gamma, beta)
    _bf16(x, gamma, beta)
    x = x if x.is_contiguous() else x.contiguous()
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    if (GN_SPLIT and N * G <= 128 and HW >= 1024 and HW % 4 == 0 and N * HW * C <= (1 << 23) and (C // G) % 2 == 0
            and C // G <= 512):
        # (the sync buffer of a capturing stream is the one allocated by the warm-up calls on that stream, before capture)
        mean = torch.empty(N, G, dtype=torch.float32, device=x.device)
        rstd = torch.empty(N, G, dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        check("dllm_groupnorm_fwd_split", _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _p(_gn_sync(x.device)), N, HW, C, G,
              float(eps), int(act), _stream())
        return y, mean, rstd
    ws = _lib.lib().dllm_groupnorm_ws_floats(N, HW, C)
    if ws < 0:
        raise ValueError("groupnorm: unsupported shape")
    part = torch.empty(ws, dtype=torch.float32, device=x.device)
    mean = torch.empty(N, G, dtype=torch.float32, device=x.device)
    rstd = torch.empty(N, G, dtype=torch.float32, device=x.device)
    ab = torch.empty(N, C, 2, dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    check("dllm_groupnorm_fwd", _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _p(ab), _p(part), N, HW, C, G, float(eps),
          int(act), _stream())
    return y, mean, rstd


def groupnorm_bwd(dy, x, gamma, beta, mean, rstd, G, act):
    dy = dy.contiguous()
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    ws = _lib.lib().dllm_groupnorm_ws_floats(N, HW, C)
    part = torch.empty(ws, dtype=torch.float32, device=x.device)
    c1 = torch.empty(N, G, dtype=torch.float32, device=x.device)
    c2 = torch.empty(N, G, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    check("dllm_groupnorm_bwd", _p(dy), _p(x), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dx), _p(c1), _p(c2), _p(part), N, HW,
          C, G, int(act), _stream())
    return dx


class GroupNormFn(torch.autograd.Function):
    """GroupNorm(+SiLU) on NHWC, frozen affine: forward + input gradient."""

    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, act):
        x = x.contiguous()
        y, mean, rstd = groupnorm_fwd(x, gamma, beta, G, eps, act)
        if x.requires_grad:
            ctx.save_for_backward(x, gamma, beta, mean, rstd)
            ctx.meta = (G, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        G, act = ctx.meta
        return groupnorm_bwd(dy, x, gamma, beta, mean, rstd, G, act), None, None, None, None, None


def groupnorm(x, gamma, beta, G, eps, act=False):
    return GroupNormFn.apply(x, gamma, beta, G, eps, act)


class MSELossFn(torch.autograd.Function):
    """F.mse_loss(pred.float(), target.float(), reduction='mean') (modeling_plugins.py:559) with pred bf16, target fp32."""

    @staticmethod
    def forward(ctx, pred, target):
        pred = pred.contiguous()
        target = target.contiguous()
        n = pred.numel()
        nparts = max(1, min(1024, (n + 2047) // 2048))           # >= 8 elements per thread; summed in index order below (deterministic)
        parts = torch.empty(nparts, dtype=torch.float32, device=pred.device)
        check("dllm_mse_sum", _p(pred), _p(target), n, _p(parts), nparts, _stream())
        acc = reduce_sum_f32(parts)
        ctx.save_for_backward(pred, target)
        return (acc / max(n, 1)).reshape(())

    @staticmethod
    def backward(ctx, dloss):
        pred, target = ctx.saved_tensors
        gs = dloss.to(torch.float32).reshape(1).contiguous()
        dp = torch.empty_like(pred)
        check("dllm_mse_bwd", _p(pred), _p(target), pred.numel(), _p(gs), _p(dp), _stream())
        return dp, None


def mse_loss(pred, target):
    _need_gpu(pred, target)
    _bf16(pred)
    if target.dtype != torch.float32:
        target = target.float()
    return MSELossFn.apply(pred, target)


def mse_loss_per_sample(pred, target):
    """per-sample mean squared error (min-SNR weighting path, modeling_plugins.py:569-571)."""
    return torch.stack([mse_loss(pred[i], target[i]) for i in range(pred.shape[0])])


def cfg_ddim_step_(pred, latents, next_in, guidance, a_t, a_prev, v_prediction=False):
    """Fused CFG combine + DDIM(eta=0) update (+ next UNet input).  pred bf16 [2B,P,4] NHWC, latents fp32 [B,P,4] in place,
    next_in bf16 [2B,P,8] or None."""
    _need_gpu(pred, latents, next_in)
    n_half = latents.numel() // 4
    check("dllm_cfg_ddim_step", _p(pred), _p(latents), _p(next_in), n_half, 0, float(guidance), float(a_t) ** 0.5,
          float(1 - a_t) ** 0.5, float(a_prev) ** 0.5, float(1 - a_prev) ** 0.5, int(v_prediction), _stream())
    return latents


class PackedAttnFn(torch.autograd.Function):
    """Self-attention on a packed [B,S,3,H,D] QKV GEMM output; the backward writes dQ/dK/dV straight into one packed
    buffer (no zero-filled slice gradients)."""

    @staticmethod
    def forward(ctx, qkv, causal, scale):
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        o, lse = attn_fwd(q, k, v, causal, scale, None, need_lse=qkv.requires_grad)
        if qkv.requires_grad:
            ctx.save_for_backward(qkv, o, lse)
        ctx.meta = (causal, scale)
        return o

    @staticmethod
    def backward(ctx, dout):
        qkv, o, lse = ctx.saved_tensors
        causal, scale = ctx.meta
        dqkv = torch.empty_like(qkv)
        attn_bwd(dout, qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], o, lse, causal, scale, None, dq=dqkv[:, :, 0],
                 dk=dqkv[:, :, 1], dv=dqkv[:, :, 2])
        return dqkv, None, None


def packed_self_attn(qkv, causal=False, scale=None):
    return PackedAttnFn.apply(qkv, causal, scale)


class PackedKVAttnFn(torch.autograd.Function):
    """Cross-attention: q [B,Sq,H,D] and a packed [B,Sk,2,H,D] K/V GEMM output."""

    @staticmethod
    def forward(ctx, q, kv, scale):
        need = q.requires_grad or kv.requires_grad
        o, lse = attn_fwd(q, kv[:, :, 0], kv[:, :, 1], False, scale, None, need_lse=need)
        if need:
            ctx.save_for_backward(q, kv, o, lse)
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, dout):
        q, kv, o, lse = ctx.saved_tensors
        dkv = torch.empty_like(kv)
        dq, _, _ = attn_bwd(dout, q, kv[:, :, 0], kv[:, :, 1], o, lse, False, ctx.scale, None, dk=dkv[:, :, 0], dv=dkv[:, :, 1])
        return dq, dkv, None


def packed_cross_attn(q, kv, scale=None):
    return PackedKVAttnFn.apply(q, kv, scale)


class PackedGLUFn(torch.autograd.Function):
    """GLU over ONE GEMM output h = [first | second] (last dim 2F).  mode 0: silu(first) * second (SwiGLU with
    [gate | up]); mode 1: first * gelu(second) (diffusers GEGLU: hidden, gate = chunk(2)).  The backward writes both
    halves of dh in place."""

    @staticmethod
    def forward(ctx, h, mode):
        F_ = h.shape[-1] // 2
        h2 = h.reshape(-1, 2 * F_)
        a, b = (h2[:, :F_], h2[:, F_:]) if mode == 0 else (h2[:, F_:], h2[:, :F_])
        ctx.save_for_backward(h)
        ctx.mode = mode
        return glu_fwd(a, b, mode).view(*h.shape[:-1], F_)

    @staticmethod
    def backward(ctx, dout):
        (h,) = ctx.saved_tensors
        mode = ctx.mode
        F_ = h.shape[-1] // 2
        h2 = h.reshape(-1, 2 * F_)
        dh = torch.empty_like(h2)
        if mode == 0:
            glu_bwd(dout, h2[:, :F_], h2[:, F_:], 0, da=dh[:, :F_], db=dh[:, F_:])
        else:
            glu_bwd(dout, h2[:, F_:], h2[:, :F_], 1, da=dh[:, F_:], db=dh[:, :F_])
        return dh.view(h.shape), None


def geglu_packed(h):
    return PackedGLUFn.apply(h, 1)


def swiglu_packed(h):
    return PackedGLUFn.apply(h, 0)


# --------------------------------------------------------------------------------------------- greedy-decode operators
def gemv(x, w, residual=None, out_dtype=torch.bfloat16, out=None):
    """y[M,N] = x[M,K] w[N,K]^T (+ residual), M <= 8: the HBM-bound decode form of nn.Linear (one wave per output row)."""
    _need_gpu(x, w, residual)
    _bf16(x, w, residual)
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    M, K = x2.shape
    N = w.shape[0]
    wc = w if w.is_contiguous() else w.contiguous()
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, N)
        if r2.stride(-1) != 1:
            r2 = r2.contiguous()
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=x.device)
    check("dllm_gemv_bf16", _p(x2), _p(wc), _p(out), _p(r2), M, N, K, x2.stride(0), K, out.stride(0),
          r2.stride(0) if r2 is not None else 0, _dt(out), _stream())
    return out


def attn_decode(q, kcache, vcache, kv_len, scale=None, nsplit=8, out=None, kv_start=None):
    """q [B,H,D]; kcache/vcache [B,Smax,Hkv,D] (same strides); kv_len int32 [B] ON DEVICE (valid cache length, read by the
    kernel: the launch does not depend on it); kv_start int32 [B] on device or None (first valid slot of a left-padded
    prompt) -> [B,H,D]."""
    _need_gpu(q, kcache, vcache, kv_len)
    _bf16(q, kcache, vcache)
    B, H, D = q.shape
    Hkv = kcache.shape[2]
    if kcache.stride() != vcache.stride() or kcache.stride(3) != 1 or q.stride(2) != 1:
        raise ValueError("attn_decode: k/v caches must share strides; head dim contiguous")
    if kv_len.dtype != torch.int32:
        raise TypeError("attn_decode: kv_len must be int32")
    if out is None:
        out = torch.empty(B, H, D, dtype=q.dtype, device=q.device)
    ws = torch.empty(_lib.lib().dllm_attn_decode_ws_floats(B, H, D, nsplit), dtype=torch.float32, device=q.device)
    check("dllm_attn_decode", _p(q), _p(kcache), _p(vcache), _p(kv_len), _p(kv_start), _p(out), _p(ws), B, H, Hkv, D, q.stride(0), q.stride(1),
          kcache.stride(0), kcache.stride(1), kcache.stride(2), out.stride(0), out.stride(1),
          float(scale if scale is not None else D ** -0.5), nsplit, _stream())
    return out


def attn_decode_rope(q, k_new, v_new, kcache, vcache, kv_len, cos, sin, pos, scale=None, nsplit=8, out=None, kv_start=None,
                     counters=None, partials_only=False):
    """`rope_append_` + `attn_decode` in one launch pair: q [B,H,D] UN-rotated (left untouched), k_new / v_new [B,Hkv,D] the step's
    key / value (k un-rotated); the kernel rotates q in registers, writes rotate(k_new) and v_new to cache slot kv_len[b] - 1 and
    attends to them in the same launch.  pos int64 [B] and kv_len int32 [B] on device.  `counters` (int32 [B*H], zero at rest): the
    last split to finish merges the partial states inside the launch instead of a combine launch.  `partials_only`: no merge at all --
    returns the split-KV partial states (fp32 [B*H, nsplit, D+2]) for `gemv_attn_combine`, which merges them while it stages its x."""
    _need_gpu(q, k_new, v_new, kcache, vcache, kv_len, cos, sin, pos)
    _bf16(q, k_new, v_new, kcache, vcache)
    B, H, D = q.shape
    Hkv = kcache.shape[2]
    if kcache.stride() != vcache.stride() or kcache.stride(3) != 1 or q.stride(2) != 1:
        raise ValueError("attn_decode_rope: k/v caches must share strides; head dim contiguous")
    if not (k_new.is_contiguous() and v_new.is_contiguous()) or k_new.shape != (B, Hkv, D) or v_new.shape != (B, Hkv, D):
        raise ValueError("attn_decode_rope: contiguous k_new / v_new of shape [B, Hkv, D] required")
    if kv_len.dtype != torch.int32 or pos.dtype != torch.int64:
        raise TypeError("attn_decode_rope: kv_len must be int32, pos int64")
    if out is None and not partials_only:
        out = torch.empty(B, H, D, dtype=q.dtype, device=q.device)
    ws = torch.empty(_lib.lib().dllm_attn_decode_ws_floats(B, H, D, nsplit), dtype=torch.float32, device=q.device)
    check("dllm_attn_decode_rope", _p(q), _p(k_new), _p(v_new), _p(kcache), _p(vcache), _p(cos), _p(sin), _p(pos.reshape(-1)), _p(kv_len),
          _p(kv_start), _p(None if partials_only else out), _p(ws), _p(None if partials_only else counters), B, H, Hkv, D, q.stride(0), q.stride(1),
          k_new.stride(0), kcache.stride(0), kcache.stride(1), kcache.stride(2), H * D if partials_only else out.stride(0),
          D if partials_only else out.stride(1), float(scale if scale is not None else D ** -0.5), nsplit, _stream())
    return ws if partials_only else out


def gemv_attn_combine(ws, w, B, H, D, nsplit, residual=None, out_dtype=torch.bfloat16):
    """o projection of a token step straight from the split-KV partials of `attn_decode_rope(..., partials_only=True)`:
    y [B, N] = merge(ws) w^T (+ residual), the merge being attn_decode's combine (same bits) done while each block stages x.
    None when the shape is not covered (B > 4, H * D too large for the LDS staging): the caller combines and calls `gemv`."""
    if B > 4 or B * H * D * 2 > 60 * 1024 or D not in (64, 128):
        return None
    _need_gpu(ws, w, residual)
    _bf16(w, residual)
    N = w.shape[0]
    y = torch.empty(B, N, dtype=out_dtype, device=w.device)
    check("dllm_gemv_attn_combine", _p(ws), _p(w), _p(y), _p(residual), B, H, D, nsplit, N, w.stride(0), y.stride(0),
          residual.stride(0) if residual is not None else 0, _dt(y), _stream())
    return y


# --------------------------------------------------------------------------------------------- torch.compile coexistence
_DYNAMO_OPAQUE = False


def make_dynamo_opaque():
    """Call once before `torch.compile(model)` (the reference's inference scripts compile the model:
    projects/dreamllm/inference.py:70, omni/eval/vqa/vqa_inference.py:297; SURVEY.md §8-b1 threading note).

    Every kernel launch of this package goes through ctypes, which Dynamo cannot trace.  This marks each operator entry point
    of the module (and the autograd Functions behind them) as opaque to Dynamo: it breaks the graph around the call and runs
    it eagerly on the HIP kernels instead of failing or falling back.  Not applied by default because the wrapper costs a
    few microseconds per call in plain eager mode."""
    global _DYNAMO_OPAQUE
    if _DYNAMO_OPAQUE:
        return
    import types
    g = globals()
    for name, fn in list(g.items()):
        if isinstance(fn, types.FunctionType) and fn.__module__ == __name__ and not name.startswith("_") \
                and name != "make_dynamo_opaque":
            g[name] = torch.compiler.disable(fn, recursive=True)
    # autograd Functions that modules call through `.apply` directly
    fns = [c for c in g.values() if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c.__module__ == __name__]
    from . import modeling_dreamllm
    fns.append(modeling_dreamllm._DecoderLayerFn)
    for c in fns:
        c.apply = staticmethod(torch.compiler.disable(c.apply, recursive=True))
    _DYNAMO_OPAQUE = True


def gemv_fused(x, weights, norm_w=None, eps=0.0, residual=None, swiglu=False, out_dtype=torch.bfloat16):
    """Decode-step Linear(s) in one launch: h = RMSNorm(x; norm_w, eps) if norm_w is given else x; then
    `swiglu=False`: [h W_i^T for W_i in weights] (1..3 matrices sharing K; residual is added to the first),
    `swiglu=True`: weights = (gate, up) -> silu(h gate^T) * (h up^T).  x [M<=8, K]."""
    _need_gpu(x, norm_w, residual, *weights)
    _bf16(x, norm_w, residual, *weights)
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    M, K = x2.shape
    ws = [w if w.is_contiguous() else w.contiguous() for w in weights]
    if any(w.shape[1] != K for w in ws) or not 1 <= len(ws) <= 3:
        raise ValueError("gemv_fused: 1..3 weight matrices with a common K")
    Ns = [w.shape[0] for w in ws]
    if swiglu:
        if len(ws) != 2 or Ns[0] != Ns[1]:
            raise ValueError("gemv_fused(swiglu): weights = (gate, up) of equal shape")
        outs = [torch.empty(M, Ns[0], dtype=torch.bfloat16, device=x.device)]
    else:
        outs = [torch.empty(M, n, dtype=out_dtype, device=x.device) for n in Ns]
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, Ns[0])
        if r2.stride(-1) != 1:
            r2 = r2.contiguous()
    wp = [_p(w) for w in ws] + [None] * (3 - len(ws))
    yp = [_p(o) for o in outs] + [None] * (3 - len(outs))
    nn_ = Ns + [0] * (3 - len(Ns))
    ldy = [o.stride(0) for o in outs] + [0] * (3 - len(outs))
    check("dllm_gemv_fused", _p(x2), _p(norm_w), float(eps), wp[0], wp[1], wp[2], yp[0], yp[1], yp[2], _p(r2), M, nn_[0],
          nn_[1], nn_[2], K, x2.stride(0), K, ldy[0], ldy[1], ldy[2], r2.stride(0) if r2 is not None else 0,
          int(swiglu), _dt(outs[0]), _stream())
    return outs[0] if swiglu else outs


def rope_append_(q, k, v, kcache, vcache, cos, sin, pos, kv_len=None):
    """q [B,H,D] rotated in place with rotary position pos[b]; k [B,Hkv,D] rotated into kcache[b, slot]; v copied into
    vcache[b, slot]; slot = kv_len[b] - 1 (int32 [B] on device) or pos[b] when kv_len is None (pos: int64 [B] on device).
    caches [B,Smax,Hkv,D] with identical strides."""
    _need_gpu(q, k, v, kcache, vcache, cos, sin, pos)
    _bf16(q, k, v, kcache, vcache)
    B, H, D = q.shape
    Hkv = k.shape[1]
    if not (q.is_contiguous() and k.is_contiguous() and v.is_contiguous()) or kcache.stride() != vcache.stride():
        raise ValueError("rope_append_: contiguous q/k/v and equal cache strides required")
    if pos.dtype != torch.int64:
        raise TypeError("rope_append_: pos must be int64")
    check("dllm_rope_append", _p(q), _p(k), _p(v), _p(kcache), _p(vcache), _p(cos), _p(sin), _p(pos.reshape(-1)), _p(kv_len), B, H, Hkv, D,
          q.stride(0), k.stride(0), kcache.stride(0), kcache.stride(1), kcache.stride(2), _stream())
    return q


class HipLayerNorm(torch.nn.LayerNorm):
    """nn.LayerNorm shell over the HIP operator (keeps torch.nn's parameter names for state_dict compatibility)."""

    def forward(self, x):
        return layernorm(x, self.weight, self.bias, self.eps)
