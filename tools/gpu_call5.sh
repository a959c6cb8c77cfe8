#!/bin/bash
set -u
R=gpurun_out/r02e
mkdir -p $R
root="$PWD"
export TMPDIR=/tmp
cd /tmp
for v in 1 4; do
TGCN_SWEEP_VEC=$v rocprofv3 --kernel-trace --stats --output-format csv -d $root/$R/ceil_v$v -- python3 $root/tools/exp_sweep_ceiling.py 200 > $root/$R/ceil_v$v.log 2>&1 || { tail -5 $root/$R/ceil_v$v.log; exit 1; }
tail -3 $root/$R/ceil_v$v.log | cut -c1-300
f=$(find $root/$R/ceil_v$v -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-200
done
