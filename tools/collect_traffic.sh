#!/bin/bash
# Counter traffic of one bench configuration (run on the GPU box): rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of
# `bench.py --config <cfg>`; then, here:  TGCN_TRAFFIC_KEY=<cfg>_n1 python profiles/summarize.py <tag> <dir>/stats <dir>/fetch <dir>/write
#   usage: tools/collect_traffic.sh <cfg> <tag> [steps]
set -u
cfg=$1; tag=$2; steps=${3:-10}
R=gpurun_out/$tag
mkdir -p $R
root="$PWD"
export TMPDIR=/tmp
BENCH="$root/bench.py --config $cfg --steps $steps --warmup 3 --no-cpu-baseline --no-epoch --no-hbm-activity --mode reference"
echo "python3 bench.py --config $cfg --steps $steps --warmup 3 --no-cpu-baseline --no-epoch --no-hbm-activity --mode reference" > $R/command.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$R/stats -- python3 $BENCH > $root/$R/stats.log 2>&1 || { tail -5 $root/$R/stats.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $root/$R/fetch -- python3 $BENCH > $root/$R/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $root/$R/write -- python3 $BENCH > $root/$R/write.log 2>&1 || exit 1
cd $root
find $R -name "*_agent_info.csv" -delete
grep '"metric"' $R/stats.log | cut -c1-200
du -sh $R
