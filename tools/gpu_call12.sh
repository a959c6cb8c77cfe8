#!/bin/bash
set -u
R=gpurun_out/r02l
mkdir -p $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "hot or capturable or threads or graphed or c4 or c3 or reproducible or split or non_finite" > $R/tests.log 2>&1
rc=$?
tail -5 $R/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for ov in 0 1; do for F in 200 64; do
  TGCN_HOT_OVERLAP=$ov timeout -k 10 200 python tools/sweep_spmm.py one c4 $F 2>&1 | tail -1 | cut -c1-140 >> $R/overlap.log || exit 1
done; done
cat $R/overlap.log
