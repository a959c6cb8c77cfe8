#!/bin/bash
# rows of a block handed to the sub-groups in order of falling degree (TGCN_ROW_SORT) on c4: parity, then launch times
set -u
R=gpurun_out/r02sort
mkdir -p $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "widths or long_rows or non_finite or hot or empty_rows or stress or c2 or reproducible" > $R/tests.log 2>&1
rc=$?
tail -3 $R/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for srt in 0 1 0 1; do for F in 64 32 128; do
  echo "row_sort=$srt F=$F" >> $R/sort.log
  TGCN_ROW_SORT=$srt timeout -k 10 200 python tools/sweep_spmm.py one c4 $F 2>&1 | tail -1 | cut -c1-120 >> $R/sort.log || exit 1
done; done
cat $R/sort.log
