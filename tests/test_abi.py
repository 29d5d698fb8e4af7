"""CPU: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/dreamllm_hip.h declares
(and nothing the Python layer calls is missing).  No compute calls here."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "dreamllm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t)\s+(dllm_\w+)\s*\(", txt)))


def test_library_builds_and_exports_header_symbols():
    from dreamllm_amd import build
    lib = ctypes.CDLL(build.build())
    syms = _header_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_shipped_library_exports_nothing_but_the_header():
    """Bench-mode entry points (timelines, ablations, A/B kernels: DLLM_BENCH_MODES builds -> libdreamllm_hip_bench.so) must not leak into
    the product library: its dynamic `dllm_*` symbols are exactly the header's."""
    import subprocess
    from dreamllm_amd import build
    out = subprocess.run(["nm", "-D", "--defined-only", build.build()], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("dllm_")})
    assert exported == _header_symbols(), sorted(set(exported) ^ set(_header_symbols()))


def test_python_signatures_cover_the_header():
    from dreamllm_amd import _lib
    declared = set(_lib.SIGNATURES) | set(_lib.RESTYPES)
    assert set(_header_symbols()) == declared, set(_header_symbols()) ^ declared


def test_no_cpu_fallback():
    import pytest
    import torch
    from dreamllm_amd import ops
    x = torch.randn(4, 64).bfloat16()
    with pytest.raises(RuntimeError):
        ops.rmsnorm_fwd(x, torch.ones(64).bfloat16(), 1e-6)
    with pytest.raises(RuntimeError):
        ops.linear_fwd(x, torch.randn(8, 64).bfloat16())


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "dreamllm_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn


def test_ctypes_signatures_match_header_prototypes():
    """Every prototype in include/dreamllm_hip.h against the ctypes argtypes the Python layer registers: same number of
    arguments and the same class per argument (pointer / int / int64 / float).  A mismatch here is silent UB at call time."""
    from dreamllm_amd import _lib
    txt = open(os.path.join(ROOT, "include", "dreamllm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    protos = re.findall(r"\b(?:int|int64_t)\s+(dllm_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S)
    assert len(protos) >= 30

    def cls_of_c(arg):
        arg = " ".join(arg.split())
        if "*" in arg:
            return "ptr"
        ty = arg.rsplit(" ", 1)[0] if " " in arg else arg
        ty = ty.replace("const ", "").strip()
        return {"int": "int", "int64_t": "i64", "float": "float"}[ty]

    def cls_of_ctypes(t):
        if t in (ctypes.c_void_p,):
            return "ptr"
        if t is ctypes.c_int:
            return "int"
        if t in (ctypes.c_int64, ctypes.c_longlong, ctypes.c_long):
            return "i64"
        if t is ctypes.c_float:
            return "float"
        raise AssertionError(t)

    sigs = dict(_lib.SIGNATURES)
    sigs.update({k: v[1] for k, v in _lib.RESTYPES.items()})
    checked = 0
    for name, args in protos:
        args = args.strip()
        c_args = [] if args in ("", "void") else [cls_of_c(a) for a in args.split(",")]
        py_args = [cls_of_ctypes(t) for t in sigs[name]]
        assert c_args == py_args, (name, c_args, py_args)
        checked += 1
    assert checked == len(set(n for n, _ in protos))


def test_scheduling_hints_are_pure_host_functions():
    """The split-K / stream-K plans are host arithmetic of the library (no device needed): pin them at the shapes the bench runs, so
    a change of the plan shows up here and not as a silent throughput change.  T = 16 x 2048 tokens, d = 4096, F = 11008."""
    from dreamllm_amd import _lib
    T, d, F_ = 32768, 4096, 11008
    sk = lambda M, N, K, la, lb: _lib.call("dllm_gemm_streamk_hint", M, N, K, la, lb)
    # stream-K tail: only the packed gate|up weight gradient (1376 tiles = 5 rounds + 96)
    assert sk(2 * F_, d, T, 1, 1) == 1
    for shape in [(d, F_, T, 1, 1), (3 * d, d, T, 1, 1), (d, d, T, 1, 1), (T, F_, d, 0, 1), (T, 2 * F_, d, 0, 0), (T, d, d, 0, 0),
                  (T, d, F_, 0, 0), (T, 3 * d, d, 0, 0), (T, d, 2 * F_, 0, 1)]:
        assert sk(*shape) == 0, shape
    # small grid, deep K (layout 2 = implicit-GEMM conv): the UNet's 1280-channel 3x3 convs at 16 x 16, batch 16 (80 tiles); at batch 32 (160
    # tiles: more than half a round) and for the 640-channel convs at 32 x 32, batch 16 (192 tiles) the whole-tile launch is faster (round 6)
    assert sk(16 * 256, 1280, 9 * 1280, 2, 0) == 1 and sk(32 * 256, 1280, 9 * 2560, 2, 0) == 0 and sk(16 * 1024, 640, 9 * 1920, 2, 0) == 0
    assert sk(2 * 256, 1280, 9 * 1280, 2, 0) == 0           # batch 2: 10 tiles -> the split-K kernels
    assert sk(16 * 4096, 320, 9 * 320, 2, 0) == 0           # 512 tiles: whole rounds
    # split-K of tiny grids (the denoise loop at batch 2)
    h = lambda M, N, K, la=0, lb=0: _lib.call("dllm_gemm_splitk_hint", M, N, K, la, lb)
    # (round 4: the ring-buffered kernel's cost model -- far fewer slices than the register-staged kernel wanted: 10 / 22 / 2 before)
    assert h(T, d, d) == 1 and h(512, 1280, 11520) == 6 and h(128, 1280, 11520) == 15 and h(8192, 320, 2880) == 1
    assert h(2048, 640, 5760) == 3 and h(512, 1280, 1280) == 1 and h(8192, 320, 1280) == 1
    # the ring model applies to the layouts the ring kernel runs (k-contiguous / gathered A, k-contiguous B) only; the input- and
    # weight-gradient layouts of a small grid run on the register-staged kernel and take its (many-slices) rule
    assert h(512, 1280, 11520, 2, 0) == 6 and h(512, 1280, 11520, 0, 1) == 10 and h(512, 1280, 11520, 1, 1) == 10
    assert _lib.call("dllm_gemm_streamk_ws_bytes") == (2 * 256 * 256 * 256 + 1024) * 4


def test_gemm_kernel_family_knobs_are_plain_arguments():
    """Round 6: the four-wave GEMM kernel is chosen per call -- `variant` tile code 280 / bit 28 of `variant`, bits 8-9 of the fused entry
    points' `group_m` -- and the Python side derives those bits from `gemm_variant` / DREAMLLM_W4M alone (no library state).  Host logic only."""
    from dreamllm_amd import ops
    key = (0, 0, 32768, 22016, 4096)
    gm = ops._group_m_for(*key)
    assert gm == 8
    prev = ops.W4M
    try:
        ops.W4M = True
        assert ops._glu_group_m((0, 0), *key[2:]) == gm                      # family 0: the library's choice
        with ops.gemm_variant(259):
            assert ops._glu_group_m((0, 0), *key[2:]) == (gm | (1 << 8))     # the 8-wave kernel
        with ops.gemm_variant(280, 3):
            assert ops._glu_group_m((0, 0), *key[2:]) == (3 | (2 << 8))      # the four-wave kernel, GROUP_M from the variant
        ops.W4M = False
        assert ops._glu_group_m((0, 0), *key[2:]) == (gm | (1 << 8))         # DREAMLLM_W4M=0: the 8-wave kernel everywhere
        assert ops.NO_W4M_BIT == 1 << 28
    finally:
        ops.W4M = prev
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "dreamllm_hip.h")).read()
    assert "280 the four-wave kernel" in hdr and "bit 28 = never choose the four-wave kernel" in hdr and "bits 8-9 the kernel family" in hdr
