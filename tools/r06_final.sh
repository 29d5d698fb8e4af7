#!/bin/bash
# Closing validation + profile campaign of round 6 on one GPU box (through gpurun): full GPU test suite + smoke(), the bench line, rocprofv3
# kernel traces of the training step and both denoise legs, PMC passes (separate runs, --kernel-trace only beside --pmc) on the dominant GEMM
# shape (FETCH_SIZE / WRITE_SIZE, SQ set) and on the ring kernel in the denoise loop.  Summaries land in gpurun_out/r06/ (copied to profiles/).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
if [ "$1" != "profiles-only" ]; then
  python -m pytest tests -q -m gpu 2>&1 | tail -12 > $OUT/r06_gpu_tests.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $OUT/r06_gpu_tests.log
  cp $ROOT/gpurun_out/parity_report.json $OUT/r06_parity_report.json 2>/dev/null
  python bench.py --steps 10 --warmup 2 > $OUT/r06_bench_final.log 2> $OUT/r06_bench_final.err
  python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_attn_bench.log
fi
cd /tmp && export TMPDIR=/tmp
run_trace() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o x -- "$@" > $OUT/r06_rocprofv3_${name}_run.log 2>&1
  DB=$(find /tmp/prof_$name -name "*.db" | head -1)
  python $ROOT/tools/rocpd_stats.py $DB > $OUT/r06_${name}_kernel_stats.csv
  python $ROOT/tools/rocpd_gaps.py $DB 30 > $OUT/r06_${name}_launch_table.txt 2>/dev/null
}
run_trace bench python $ROOT/bench.py --steps 4 --warmup 1 --no-configs --no-cpu-baseline --no-denoise --no-ragged --no-grad-ckpt-leg
run_trace denoise python $ROOT/tools/bench_configs.py --only 3 --denoise-batches 1
run_trace denoise_b8 python $ROOT/tools/bench_configs.py --only 3 --denoise-batches 8
# PMC: the dominant GEMM shape (packed gate|up forward and its two gradients): fabric traffic and the SQ picture, one counter set per pass
: > $OUT/r06_pmc_gemm_fetch_write.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_gemm_$C
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_gemm_$C -o x -- python $ROOT/tools/gemm_once.py > /tmp/pmc_gemm.log 2>&1
  DB=$(find /tmp/pmc_gemm_$C -name "*.db" | head -1)
  echo "## counter: $C" >> $OUT/r06_pmc_gemm_fetch_write.txt
  python $ROOT/tools/rocpd_pmc.py $DB gemm >> $OUT/r06_pmc_gemm_fetch_write.txt 2>&1
done
: > $OUT/r06_pmc_gemm_sq.txt
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmc_gemm_sq$i
  rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmc_gemm_sq$i -o x -- python $ROOT/tools/gemm_once.py > /tmp/pmc_gemm.log 2>&1
  DB=$(find /tmp/pmc_gemm_sq$i -name "*.db" | head -1)
  echo "## counters: $SET" >> $OUT/r06_pmc_gemm_sq.txt
  python $ROOT/tools/rocpd_pmc.py $DB gemm >> $OUT/r06_pmc_gemm_sq.txt 2>&1
done
# PMC: the ring kernel inside the denoise loop (the r04 ring evidence was a round old)
: > $OUT/r06_pmc_ring.txt
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmc_ring$i
  rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmc_ring$i -o x -- python $ROOT/tools/ring_once.py > /tmp/pmc_ring.log 2>&1
  DB=$(find /tmp/pmc_ring$i -name "*.db" | head -1)
  echo "## counters: $SET" >> $OUT/r06_pmc_ring.txt
  python $ROOT/tools/rocpd_pmc.py $DB ring --by-grid >> $OUT/r06_pmc_ring.txt 2>&1
done
tail -4 $OUT/r06_gpu_tests.log 2>/dev/null; tail -c 1500 $OUT/r06_bench_final.log 2>/dev/null; head -14 $OUT/r06_bench_kernel_stats.csv; head -8 $OUT/r06_denoise_kernel_stats.csv; cat $OUT/r06_pmc_gemm_fetch_write.txt | head -30
