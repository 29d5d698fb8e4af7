#!/bin/bash
# Memory-side PMC passes (one counter per run) over the attention forward / backward kernels at the bench shape:
#   tools/pmc_attn_mem.sh <out.txt>      (FETCH_SIZE / WRITE_SIZE are in KiB... see MI355X_MICROARCH.md, HBM / rocprofv3 section)
OUT=${1:-gpurun_out/pmc_attn_mem.txt}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out"
: > "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum; do
  D=/tmp/pmc_attn_mem_$C
  rm -rf $D
  timeout 120 rocprofv3 --kernel-trace --pmc $C -d $D -o x -- python $ROOT/tools/attn_once.py > /tmp/pmc_run.log 2>&1
  DB=$(find $D -name "*.db" 2>/dev/null | head -1)
  echo "## counter: $C" >> "$ROOT/$OUT"
  if [ -n "$DB" ]; then python $ROOT/tools/rocpd_pmc.py $DB attn >> "$ROOT/$OUT" 2>&1; else tail -3 /tmp/pmc_run.log >> "$ROOT/$OUT"; fi
done
cat "$ROOT/$OUT"
