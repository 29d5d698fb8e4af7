#!/bin/bash
# SQ-side PMC passes (four counter sets, separate runs) over every attention kernel of the bench shape (forward + backward):
#   tools/pmc_attn_sq.sh <out.txt>
OUT=${1:-gpurun_out/pmc_attn_sq.txt}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out"
: > "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  D=/tmp/pmc_attn_sq_$i
  rm -rf $D
  timeout 120 rocprofv3 --kernel-trace --pmc $SET -d $D -o x -- python $ROOT/tools/attn_once.py > /tmp/pmc_run.log 2>&1
  DB=$(find $D -name "*.db" | head -1)
  echo "## counters: $SET" >> "$ROOT/$OUT"
  python $ROOT/tools/rocpd_pmc.py $DB attn >> "$ROOT/$OUT" 2>&1
done
cat "$ROOT/$OUT"
