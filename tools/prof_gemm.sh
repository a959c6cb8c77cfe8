#!/bin/bash
set -u
R=gpurun_out/r02gemm
mkdir -p $R
root="$PWD"
export TMPDIR=/tmp
cd /tmp
i=0
while IFS= read -r set; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $root/$R/pass$i -- python3 $root/tools/experiments/bench_dense.py > $root/$R/pass$i.log 2>&1 || { tail -3 $root/$R/pass$i.log; }
done <<'SETS'
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
SETS
cd $root
python tools/summarize_pmc.py $R gemm > $R/gemm_pmc.md
cat $R/pass1.log | tail -8
grep -v "^$" $R/gemm_pmc.md | head -150
