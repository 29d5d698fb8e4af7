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
