#!/bin/bash
# Scratch (spill) instructions between the first and the last MFMA of every kernel of one .hip file: spills in the cold set-up code of
# the attention kernels are harmless, one inside the key-tile loop costs a scratch round trip per tile.
# usage: tools/loop_spills.sh dreamllm_amd/csrc/attn_bwd_pp.hip [-DDLLM_BENCH_MODES]
src=$1; shift
out=$(mktemp /tmp/spills.XXXXXX.s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only "$@" -S --cuda-device-only "$src" -o "$out" 2>/dev/null || { echo "compile failed"; exit 1; }
awk '
/^_Z[A-Za-z0-9_]*:/ { if (name != "") report(); name=$1; first=0; last=0; n=0; delete sl }
/v_mfma/ { if (!first) first=NR; last=NR }
/scratch_(load|store)/ { sl[++n]=NR }
function report(   i,c) { c=0; for (i=1;i<=n;i++) if (sl[i]>first && sl[i]<last) c++; printf "%-100s scratch ops: %3d total, %3d between first and last MFMA\n", substr(name,1,100), n, c }
END { if (name != "") report() }' "$out"
rm -f "$out"
