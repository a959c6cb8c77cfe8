#!/bin/bash
# PMC passes over one command (run on the GPU box).  Counters go in their own rocprofv3 runs, next to
# --kernel-trace only (gpurun refuses --pmc together with the hip/hsa/memory trace domains).
#   tools/prof_pmc.sh <out_dir> <python script + args ...>
# Pass list: TCC traffic (FETCH_SIZE alone: it takes 3 of the 4 TCC slots), WRITE_SIZE + L2 hit/miss,
# EA requests by destination, SQ instruction mix, SQ wait/active cycles.
set -u
out="$1"; shift
root="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
i=0
while IFS= read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  d="$out/pass$i"
  echo "== pass $i: $set" | tee -a "$out/passes.txt"
  # shellcheck disable=SC2086
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$d" -- python3 "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed (see pass$i.log)"; tail -5 "$out/pass$i.log"; }
done <<'SETS'
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
SETS
cd "$root"
