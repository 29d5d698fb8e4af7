"""ctypes loader for the C-ABI HIP library (`include/dreamllm_hip.h`).

The library is the product's only compute path: if it is missing, `lib()` raises -- there is no CPU / eager fallback.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  -- before the library is dlopen'ed: it must bind to the HIP runtime the PyTorch-ROCm wheel loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DREAMLLM_HIP_LIB") or os.path.join(_HERE, "libdreamllm_hip.so")

c_void_p, c_int, c_i64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

DLLM_BF16, DLLM_F32 = 0, 1

_ERRORS = {-1: "bad shape", -2: "unsupported dtype", -3: "kernel launch failed", -4: "misaligned pointer/stride"}

# name -> argtypes (return type is always int)
SIGNATURES = {
    "dllm_norm_bwd_nparts": [c_i64],
    "dllm_rmsnorm_fwd": [c_void_p] * 6 + [c_i64, c_int, c_float, c_void_p],
    "dllm_rmsnorm_bwd": [c_void_p] * 8 + [c_int, c_i64, c_int, c_void_p],
    "dllm_layernorm_fwd": [c_void_p] * 6 + [c_i64, c_int, c_float, c_void_p],
    "dllm_layernorm_bwd": [c_void_p] * 10 + [c_int, c_i64, c_int, c_void_p],
    "dllm_gemm_bf16": [c_void_p] * 5 + [c_i64] * 7 + [c_int] * 5 + [c_float, c_void_p],
    "dllm_gemm_splitk_hint": [c_i64, c_i64, c_i64, c_int, c_int],
    "dllm_gemm_swiglu_fwd": [c_void_p] * 4 + [c_i64] * 7 + [c_int, c_void_p],
    "dllm_gemm_swiglu_bwd": [c_void_p] * 4 + [c_i64] * 7 + [c_int, c_void_p],
    "dllm_gemm_rope_qkv": [c_void_p] * 6 + [c_i64] * 4 + [c_int] + [c_i64] * 3 + [c_int, c_void_p],
    "dllm_gemm_streamk_hint": [c_i64, c_i64, c_i64, c_int, c_int],
    "dllm_gemm_bf16_splitk": [c_void_p] * 5 + [c_i64] * 7 + [c_int] * 5 + [c_float, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "dllm_conv2d_nhwc_bf16_splitk": [c_void_p] * 6 + [c_int] * 15 + [c_int, c_void_p, c_void_p, c_int, c_void_p],
    "dllm_conv2d_nhwc_bf16": [c_void_p] * 6 + [c_int] * 15 + [c_void_p],
    "dllm_groupnorm_fwd": [c_void_p] * 8 + [c_int] * 4 + [c_float, c_int, c_void_p],
    "dllm_groupnorm_fwd_split": [c_void_p] * 7 + [c_int] * 4 + [c_float, c_int, c_void_p],
    "dllm_groupnorm_bwd": [c_void_p] * 10 + [c_int] * 5 + [c_void_p],
    "dllm_sumpool2_nhwc": [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p],
    "dllm_cfg_ddim_step": [c_void_p] * 3 + [c_i64, c_i64] + [c_float] * 5 + [c_int, c_void_p],
    "dllm_gemv_bf16": [c_void_p] * 4 + [c_int] + [c_i64] * 6 + [c_int, c_void_p],
    "dllm_gemv_fused": [c_void_p, c_void_p, c_float] + [c_void_p] * 7 + [c_int] + [c_i64] * 10 + [c_int, c_int, c_void_p],
    "dllm_rope_append": [c_void_p] * 9 + [c_int] * 4 + [c_i64] * 5 + [c_void_p],
    "dllm_attn_decode": [c_void_p] * 7 + [c_int] * 4 + [c_i64] * 7 + [c_float, c_int, c_void_p],
    "dllm_attn_decode_rope": [c_void_p] * 13 + [c_int] * 4 + [c_i64] * 8 + [c_float, c_int, c_void_p],
    "dllm_gemv_attn_combine": [c_void_p] * 4 + [c_int] * 4 + [c_i64] * 4 + [c_int, c_void_p],
    "dllm_attn_fwd": [c_void_p] * 7 + [c_int] * 6 + [c_i64] * 9 + [c_float, c_int, c_void_p],
    "dllm_attn_bwd": [c_void_p] * 12 + [c_int] * 6 + [c_i64] * 15 + [c_float, c_int, c_void_p],
    "dllm_rope": [c_void_p] * 4 + [c_i64, c_int, c_int, c_int, c_i64, c_i64, c_int, c_void_p],
    "dllm_glu_fwd": [c_void_p] * 3 + [c_i64, c_int, c_i64, c_i64, c_i64, c_int, c_void_p],
    "dllm_glu_bwd": [c_void_p] * 6 + [c_i64, c_int] + [c_i64] * 6 + [c_int, c_void_p],
    "dllm_gather_rows": [c_void_p] * 3 + [c_i64, c_int, c_i64, c_i64, c_void_p],
    "dllm_scatter_rows": [c_void_p] * 3 + [c_i64, c_int, c_i64, c_i64, c_void_p],
    "dllm_segment_sum_rows": [c_void_p] * 5 + [c_i64, c_int, c_i64, c_i64, c_void_p],
    "dllm_segment_sum_rows_ex": [c_void_p] * 5 + [c_i64, c_int, c_i64, c_i64, c_int, c_int, c_void_p],
    "dllm_cross_entropy": [c_void_p] * 5 + [c_i64, c_int, c_i64, c_i64, c_void_p],
    "dllm_softmax_rows": [c_void_p, c_void_p, c_i64, c_int, c_i64, c_i64, c_void_p],
    "dllm_adamw": [c_void_p] * 4 + [c_i64, c_int, c_int] + [c_float] * 5 + [c_int, c_float, c_void_p, c_void_p],
    "dllm_sumsq": [c_void_p, c_i64, c_int, c_void_p, c_void_p],
    "dllm_adamw_multi": [c_void_p] * 5 + [c_int] + [c_float] * 5 + [c_int, c_float, c_void_p, c_void_p],
    "dllm_sumsq_multi": [c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "dllm_reduce_sum_f32": [c_void_p, c_i64, c_void_p, c_void_p],
    "dllm_mse_sum": [c_void_p, c_void_p, c_i64, c_void_p, c_int, c_void_p],
    "dllm_mse_bwd": [c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p],
    "dllm_add_bcast": [c_void_p] * 3 + [c_i64, c_i64, c_void_p],
    "dllm_add_rowgroup": [c_void_p] * 3 + [c_i64, c_i64, c_int, c_void_p],
    "dllm_act_fwd": [c_void_p, c_void_p, c_i64, c_int, c_void_p],
    "dllm_act_bwd": [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p],
    "dllm_probe_tr16": [c_void_p, c_void_p, c_void_p],
    "dllm_probe_mfma16": [c_void_p, c_void_p, c_void_p, c_void_p],
}

# functions whose return type is not int
RESTYPES = {"dllm_groupnorm_ws_floats": (c_i64, [c_int, c_int, c_int]),
            "dllm_gemm_streamk_ws_bytes": (c_i64, []),
            "dllm_sumsq_multi_parts": (c_i64, [c_void_p, c_int]),
            "dllm_attn_decode_ws_floats": (c_i64, [c_int, c_int, c_int, c_int])}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -m dreamllm_amd.build` (hipcc, gfx950). "
                "dreamllm_amd has no CPU fallback."
            )
        _lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            try:
                fn = getattr(_lib, name)
            except AttributeError:
                continue  # reported by tests/test_abi.py; calling it raises below
            fn.argtypes = argtypes
            fn.restype = c_int
        for name, (rt, argtypes) in RESTYPES.items():
            fn = getattr(_lib, name, None)
            if fn is not None:
                fn.argtypes = argtypes
                fn.restype = rt
    return _lib


def call(name: str, *args) -> int:
    fn = getattr(lib(), name, None)
    if fn is None:
        raise HipLibraryMissing(f"symbol {name} missing from {LIB_PATH}")
    rc = fn(*args)
    return rc


def check(name: str, *args) -> None:
    """Call and turn negative error codes into the exceptions the reference would raise (ValueError for shapes)."""
    rc = call(name, *args)
    if rc == 0:
        return
    msg = f"{name} failed: {_ERRORS.get(rc, rc)}"
    if rc in (-1, -4):
        raise ValueError(msg)
    if rc == -2:
        raise TypeError(msg)
    raise RuntimeError(msg)
