#!/bin/bash
set -u
R=gpurun_out/r02j
mkdir -p $R
for u in 4 8 16; do for F in 64 32 128; do
  TGCN_SPMM_NARROW_U=$u timeout -k 10 200 python tools/sweep_spmm.py one c4 $F 2>&1 | tail -1 | cut -c1-140 >> $R/narrow_u.log || exit 1
done; done
cat $R/narrow_u.log
