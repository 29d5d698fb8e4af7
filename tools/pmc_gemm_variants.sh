#!/bin/bash
ROOT=$1
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/r06_pmc_gemm_w4.txt
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pmc_w4_$i
  rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmc_w4_$i -o x -- python $ROOT/tools/gemm_variant_once.py > /tmp/pmc_w4.log 2>&1
  DB=$(find /tmp/pmc_w4_$i -name "*.db" | head -1)
  echo "## counters: $SET" >> $OUT/r06_pmc_gemm_w4.txt
  python $ROOT/tools/rocpd_pmc.py $DB gemm >> $OUT/r06_pmc_gemm_w4.txt 2>&1
done
cat $OUT/r06_pmc_gemm_w4.txt
