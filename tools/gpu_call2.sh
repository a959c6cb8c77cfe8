#!/bin/bash
set -u
R=gpurun_out/r02b
mkdir -p $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "c5 or c4 or non_finite" > $R/tests.log 2>&1
rc=$?
tail -5 $R/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python tools/exp_mid_rows.py --F 200 > $R/exp_mid_200.log 2>&1 || { tail -5 $R/exp_mid_200.log; exit 1; }
cat $R/exp_mid_200.log
for cb in 32768 131072 200000 524288; do
  for ord in 3 1; do
    TGCN_COL_BLOCK=$cb TGCN_ITEM_ORDER=$ord timeout -k 10 200 python tools/sweep_spmm.py one c4 200 2>&1 | tail -1 | cut -c1-400 >> $R/sweep_colblock.log || exit 1
  done
done
cat $R/sweep_colblock.log
exit $rc
