#!/bin/bash
# A/B of the narrow SpMM on c4 (old kernel / buffer-addressed kernel) + the narrow-path tests
set -u
R=gpurun_out/r02p
mkdir -p $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "widths or long_rows or non_finite or hot or empty_rows or stress or c2 or reproducible" > $R/tests.log 2>&1
rc=$?
tail -3 $R/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for nb in 0 1; do for F in 64 32 128; do
  TGCN_SPMM_NARROW_BUF=$nb timeout -k 10 200 python tools/sweep_spmm.py one c4 $F 2>&1 | tail -1 | cut -c1-140 >> $R/narrow.log || exit 1
done; done
cat $R/narrow.log
