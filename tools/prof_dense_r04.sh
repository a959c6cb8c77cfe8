#!/bin/bash
# SQ counter passes over the dense kernels at the c4 shapes (round 4: the LDS-staged tn product with and without the mask)
set -u
R=gpurun_out/${1:-r04dense}
mkdir -p $R
root="$PWD"
export TMPDIR=/tmp
cd /tmp
i=0
while IFS= read -r set; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $root/$R/pass$i -- python3 $root/tools/experiments/run_dense_once.py > $root/$R/pass$i.log 2>&1 || { tail -3 $root/$R/pass$i.log; }
done <<'SETS'
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE
SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
SETS
cd $root
python tools/summarize_pmc.py $R gemm > $R/dense_pmc.md   # (TGCN_TALL_ONE_PER_CU=0 in the environment: the unmasked kernels two / three per CU)
find $R -name "*_agent_info.csv" -delete
grep -v "^$" $R/dense_pmc.md | head -120
