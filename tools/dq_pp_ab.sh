#!/bin/bash
# A/B of build-time variants of one kernel file (default: the ping-pong dQ kernel csrc/attn_bwd_pp.hip; FILE=attn_fwd_pp.hip for the
# forward): each VARIANT ("name:-DMACRO,-DMACRO") is compiled into its own copy of the bench library (only that file is recompiled, the
# other objects are those of the last bench build), then `tools/dq_pp_ab.sh run` runs RUNCMD (default: the dQ timeline tool) against
# every copy, interleaved, two rounds.
# usage:  DLLM_BENCH_MODES=1 python -m dreamllm_amd.build; VARIANTS="base: nol:-DBP_NO_LAUNDER" tools/dq_pp_ab.sh ; gpurun -- 'tools/dq_pp_ab.sh run'
cd "$(dirname "$0")/.."
OBJ=dreamllm_amd/csrc/build_bench
[ -d $OBJ ] || OBJ=dreamllm_amd/csrc/build
if [ "$1" != "run" ]; then
  mkdir -p tools/bin
  rm -f tools/bin/libdqab_*.so
  for v in ${VARIANTS:-base:}; do
    name=${v%%:*}; flags=${v#*:}; flags=${flags//,/ }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wno-unused-result -Wno-pass-failed \
      -DDLLM_BENCH_MODES $flags -c dreamllm_amd/csrc/${FILE:-attn_bwd_pp.hip} -o /tmp/dqab_$name.o || exit 1
    f=${FILE:-attn_bwd_pp.hip}; objs=$(ls $OBJ/*.o | grep -v "/${f%.hip}.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/dqab_$name.o -o tools/bin/libdqab_$name.so || exit 1
    echo "built $name ($flags)"
  done
  exit 0
fi
for round in 1 2; do
  for lib in tools/bin/libdqab_*.so; do
    DREAMLLM_HIP_LIB=$PWD/$lib python ${RUNCMD:-tools/attn_bwd_pp_timeline.py} 2>&1 | grep "alone\|backward variant\|fwd " | sed "s|^|$(basename $lib .so) |"
  done
done
