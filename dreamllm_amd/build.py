"""Build the HIP C-ABI library `libdreamllm_hip.so` for gfx950 in-tree (no torch dependency, plain hipcc).

`python -m dreamllm_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles gfx950 without a GPU.
Objects are cached per source under csrc/build/ and rebuilt when the source or a header is newer.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BENCH = os.environ.get("DLLM_BENCH_MODES") == "1"
BUILD = os.path.join(CSRC, "build_bench" if BENCH else "build")
LIB = os.path.join(HERE, "libdreamllm_hip_bench.so" if BENCH else "libdreamllm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only",
         "-Wno-unused-result", "-Wno-pass-failed"]
# DLLM_BENCH_MODES=1 python -m dreamllm_amd.build : a SEPARATE library (libdreamllm_hip_bench.so) that additionally contains the
# wrong-result diagnostic modes used by tools/ (GEMM no-prefetch / no-store, attention ablations); tools select it with
# DREAMLLM_HIP_LIB=.../libdreamllm_hip_bench.so.  The shipped library never contains them.
if BENCH:
    FLAGS.append("-DDLLM_BENCH_MODES")
    FLAGS += os.environ.get("DLLM_EXTRA_FLAGS", "").split()  # experiment switches, bench library only


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    return max([os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h")] + [0.0])


def _compile(src: str, hdr_mtime: float, verbose: bool) -> str:
    obj = os.path.join(BUILD, src[:-4] + ".o")
    spath = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(spath), hdr_mtime):
        return obj
    cmd = [HIPCC, *FLAGS, "-c", spath, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def build(verbose: bool = False, force: bool = False) -> str:
    os.makedirs(BUILD, exist_ok=True)
    srcs = _sources()
    hdr = _headers_mtime()
    if force:
        for f in os.listdir(BUILD):
            os.remove(os.path.join(BUILD, f))
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, hdr, verbose), srcs))
    if (not os.path.exists(LIB)) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
