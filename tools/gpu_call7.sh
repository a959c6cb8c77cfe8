#!/bin/bash
set -u
R=gpurun_out/r02g
mkdir -p $R
root="$PWD"
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider > $R/gpu_tests.log 2>&1
rc=$?
tail -8 $R/gpu_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
for nb in 0 1; do for F in 64 32 128; do
  TGCN_SPMM_NARROW_BUF=$nb timeout -k 10 200 python tools/sweep_spmm.py one c4 $F 2>&1 | tail -1 | cut -c1-200 >> $R/narrow.log || exit 1
done; done
cat $R/narrow.log
exit $rc
