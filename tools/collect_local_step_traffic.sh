#!/bin/bash
# Counter traffic of the per-rank local operators of the c4 (or c5: second argument) partition at 2 / 4 / 8 ranks, measured on ONE GPU (rank 0's operators):
# kernel trace + FETCH_SIZE and WRITE_SIZE passes of tools/prof_local_step.py (separate --pmc passes, no trace domains beside
# --kernel-trace).  Then, here:  python tools/local_step_traffic.py gpurun_out/<tag>
set -u
tag=${1:-r05_local_traffic}
cfg=${2:-c4}
R=$PWD/gpurun_out/$tag
mkdir -p $R
root=$PWD
export TMPDIR=/tmp
cd /tmp
for W in 2 4 8; do
  CMD="$root/tools/prof_local_step.py $W 0 12 $cfg"
  rocprofv3 --kernel-trace --output-format csv -d $R/w$W/stats -- python3 $CMD > $R/w$W.stats.log 2>&1 || { tail -5 $R/w$W.stats.log; exit 1; }
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/w$W/fetch -- python3 $CMD > $R/w$W.fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/w$W/write -- python3 $CMD > $R/w$W.write.log 2>&1 || exit 1
  grep LOCAL_STEPS $R/w$W.stats.log
done
cd $root
find $R -name "*_agent_info.csv" -delete
du -sh $R
