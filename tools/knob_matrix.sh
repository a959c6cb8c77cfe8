#!/bin/bash
# The single-process GPU parity tests under the settings of the knobs the library still reads (HISTORY.md 4.8).
set -u
R=gpurun_out/knobs
mkdir -p $R
rc_all=0
only=" $* "           # optional: names of the runs to do (default: all) -- a full matrix exceeds one 20-minute GPU call
run() {
  name="$1"; shift
  if [ "$only" != "  " ] && [[ "$only" != *" $name "* ]]; then return; fi
  env "$@" timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "not c5 and not hot_block and not c4" > $R/$name.log 2>&1
  rc=$?
  echo "$name rc=$rc $(tail -1 $R/$name.log)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  [ $rc -ne 0 ] && rc_all=1
}
run no_hot TGCN_HOT_ROWS=0
run gemm_split TGCN_GEMM_SPLIT=1
run big_items TGCN_ITEM_WEIGHT=2048 TGCN_MIN_PIECE=128 TGCN_COL_BLOCK=0
run small_items TGCN_ITEM_WEIGHT=128 TGCN_MIN_PIECE=8 TGCN_COL_BLOCK=1024
run many_per_cu TGCN_TALL_ONE_PER_CU=0
run old_tn TGCN_TN_STAGED_OFF=1
exit $rc_all
