#!/bin/bash
# A/B of s_setprio placements in the pipelined 256 x 256 GEMM K loop (csrc/gemm.hip, GEMM_PRIO = 0..3): builds one library per
# variant under tools/bin/ (here, on the build host), `tools/gemm_prio_ab.sh run` times them on the GPU box in interleaved rounds.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wno-unused-result -Wno-pass-failed"
if [ "$1" != "run" ]; then
  mkdir -p $ROOT/tools/bin
  python -m dreamllm_amd.build > /dev/null
  for k in ${VARIANTS:-0 1 2 3}; do
    /opt/rocm/bin/hipcc $FLAGS -DGEMM_PRIO=$k -c $ROOT/dreamllm_amd/csrc/gemm.hip -o $ROOT/tools/bin/gemm_prio_$k.o &
  done
  wait
  for k in ${VARIANTS:-0 1 2 3}; do
    OBJS=$(ls $ROOT/dreamllm_amd/csrc/build/*.o | grep -v "/gemm.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $ROOT/tools/bin/gemm_prio_$k.o -o $ROOT/tools/bin/libdllm_prio_$k.so
  done
  ls -la $ROOT/tools/bin/*.so
else
  for round in 1 2; do
    for k in ${VARIANTS:-0 1 2 3}; do
      echo "== round $round GEMM_PRIO=$k"
      DREAMLLM_HIP_LIB=$ROOT/tools/bin/libdllm_prio_$k.so python $ROOT/tools/gemm_sustained.py 0 1.2 2>&1 | grep -v amdgpu.ids
    done
  done
fi
