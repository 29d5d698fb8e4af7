#!/bin/bash
# Closing validation of round 5 on one GPU box: full GPU test suite + smoke, the bench line, the kernel trace of the train step, the
# attention benches / timelines.  Outputs under gpurun_out/ (copied to profiles/ afterwards).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -12 > $OUT/r05_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $OUT/r05_gpu_tests.log
cp $OUT/parity_report.json $OUT/r05_parity_report.json 2>/dev/null
python bench.py --steps 10 --warmup 2 > $OUT/r05_bench_final.log 2> $OUT/r05_bench_final.err
python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/r05_attn_bench.log
python tools/attn_bwd_pp_timeline.py 2>&1 | grep -v amdgpu.ids > $OUT/r05_attn_bwd_pp_timeline.log
python tools/attn_bwd_ab.py 2>&1 | grep " ms" > $OUT/r05_attn_bwd_ab.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o x -- python $ROOT/bench.py --steps 2 --warmup 1 --no-configs --no-cpu-baseline --no-denoise --no-ragged > $OUT/r05_rocprofv3_bench_run.log 2>&1
DB=$(find /tmp/prof_bench -name "*.db" | head -1)
python $ROOT/tools/rocpd_stats.py $DB > $OUT/r05_bench_kernel_stats.csv
python $ROOT/tools/rocpd_gaps.py $DB 30 > $OUT/r05_bench_launch_table.txt 2>/dev/null
tail -6 $OUT/r05_gpu_tests.log; tail -c 1200 $OUT/r05_bench_final.log; head -8 $OUT/r05_bench_kernel_stats.csv; cat $OUT/r05_attn_bench.log | head -3; cat $OUT/r05_attn_bwd_ab.log
