#!/bin/bash
set -u
R=gpurun_out/r02r
mkdir -p $R
for w in 256 384 512 768 1024; do for mp in 32 64; do for F in 64 200; do
  TGCN_ITEM_WEIGHT=$w TGCN_MIN_PIECE=$mp timeout -k 10 120 python tools/sweep_spmm.py one c4 $F 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['knobs'], 'F', d['F'], 'ms', d['ms_median'], 'items', d['items'], 'segments', d['segments'])" >> $R/resweep.log || exit 1
done; done; done
cat $R/resweep.log
