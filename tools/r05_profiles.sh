#!/bin/bash
# Round-5 profile campaign (run on the GPU box through gpurun): rocprofv3 kernel traces of the training step and the denoise loop,
# PMC passes on the attention forward and on the dominant GEMM shape.  Summaries land in gpurun_out/ (copied to profiles/ afterwards).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_trace() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o x -- "$@" > $OUT/r05_rocprofv3_${name}_run.log 2>&1
  DB=$(find /tmp/prof_$name -name "*.db" | head -1)
  python $ROOT/tools/rocpd_stats.py $DB > $OUT/r05_${name}_kernel_stats.csv
  python $ROOT/tools/rocpd_gaps.py $DB 30 > $OUT/r05_${name}_launch_table.txt 2>/dev/null
}
run_trace bench python $ROOT/bench.py --steps 2 --warmup 1 --no-configs --no-cpu-baseline --no-denoise --no-ragged
run_trace denoise python $ROOT/tools/bench_configs.py --only 3 --denoise-batches 1
run_trace denoise_b8 python $ROOT/tools/bench_configs.py --only 3 --denoise-batches 8
# PMC: attention forward (automatic choice = ping-pong kernel) and the 8-wave kernel beside it
$ROOT/tools/pmc_attn_fwd.sh 0 gpurun_out/r05_pmc_attention_fwd_pp.txt > /dev/null 2>&1
$ROOT/tools/pmc_attn_fwd.sh 2 gpurun_out/r05_pmc_attention_fwd_8wave.txt > /dev/null 2>&1
# PMC: FETCH_SIZE / WRITE_SIZE of the dominant GEMM shape, separate passes
cd /tmp
: > $OUT/r05_pmc_gemm_fetch_write.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_gemm_$C
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_gemm_$C -o x -- python $ROOT/tools/gemm_once.py > /tmp/pmc_gemm.log 2>&1
  DB=$(find /tmp/pmc_gemm_$C -name "*.db" | head -1)
  echo "## counter: $C" >> $OUT/r05_pmc_gemm_fetch_write.txt
  python $ROOT/tools/rocpd_pmc.py $DB gemm >> $OUT/r05_pmc_gemm_fetch_write.txt 2>&1
done
head -12 $OUT/r05_bench_kernel_stats.csv; head -8 $OUT/r05_denoise_kernel_stats.csv; cat $OUT/r05_pmc_gemm_fetch_write.txt; head -20 $OUT/r05_pmc_attention_fwd_pp.txt
