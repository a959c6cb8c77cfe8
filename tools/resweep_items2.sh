#!/bin/bash
set -u
R=gpurun_out/r02s
mkdir -p $R
for cb in 4096 8192 16384 32768; do for w in 320 384 448; do
  TGCN_COL_BLOCK=$cb TGCN_ITEM_WEIGHT=$w timeout -k 10 120 python tools/sweep_spmm.py one c4 200 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['knobs'], 'F', d['F'], 'ms', d['ms_median'], d['ms_min'], 'items', d['items'])" >> $R/resweep2.log || exit 1
done; done
for mp in 16 24 32 48; do
  TGCN_MIN_PIECE=$mp timeout -k 10 120 python tools/sweep_spmm.py one c4 200 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['knobs'], 'F', d['F'], 'ms', d['ms_median'], d['ms_min'], 'items', d['items'])" >> $R/resweep2.log || exit 1
done
cat $R/resweep2.log
