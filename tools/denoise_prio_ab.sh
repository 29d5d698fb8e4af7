#!/bin/bash
# A/B of s_setprio placements in the ring kernel (RING_PRIO 1 / 2) and of the GEMM request-group priority on the conv layouts
# (GEMM_PRIO_CONV) over the SD-2.1 denoise loop: one library per variant under tools/bin/, `run` = interleaved rounds on the GPU box.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wno-unused-result -Wno-pass-failed"
if [ "$1" != "run" ]; then
  mkdir -p $ROOT/tools/bin
  python -m dreamllm_amd.build > /dev/null
  /opt/rocm/bin/hipcc $FLAGS -DGEMM_PRIO_CONV -c $ROOT/dreamllm_amd/csrc/gemm.hip -o $ROOT/tools/bin/gemm_conv.o &
  /opt/rocm/bin/hipcc $FLAGS -DRING_PRIO=1 -c $ROOT/dreamllm_amd/csrc/gemm_ring.hip -o $ROOT/tools/bin/ring_1.o &
  /opt/rocm/bin/hipcc $FLAGS -DRING_PRIO=2 -c $ROOT/dreamllm_amd/csrc/gemm_ring.hip -o $ROOT/tools/bin/ring_2.o &
  wait
  B=$ROOT/dreamllm_amd/csrc/build
  OTHER=$(ls $B/*.o | grep -v "/gemm.o" | grep -v "/gemm_ring.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHER $B/gemm.o $B/gemm_ring.o -o $ROOT/tools/bin/libdllm_dn_base.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHER $ROOT/tools/bin/gemm_conv.o $B/gemm_ring.o -o $ROOT/tools/bin/libdllm_dn_conv.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHER $B/gemm.o $ROOT/tools/bin/ring_1.o -o $ROOT/tools/bin/libdllm_dn_ring1.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHER $B/gemm.o $ROOT/tools/bin/ring_2.o -o $ROOT/tools/bin/libdllm_dn_ring2.so
  ls -la $ROOT/tools/bin/*.so
else
  for round in 1 2; do
    for v in base conv ring1 ring2; do
      echo "== round $round $v"
      DREAMLLM_HIP_LIB=$ROOT/tools/bin/libdllm_dn_$v.so python $ROOT/tools/bench_configs.py --only 3 2>&1 | grep -E "steps/s|value" | head -4
    done
  done
fi
