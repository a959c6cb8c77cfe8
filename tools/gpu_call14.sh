#!/bin/bash
set -u
R=gpurun_out/r02n
mkdir -p $R
for u in 4 2 8; do
TGCN_ADAM_U=$u timeout -k 10 300 python tools/bench_spmm_adam.py 2>&1 | tail -1 >> $R/spmm_adam.log || exit 1
done
cat $R/spmm_adam.log
