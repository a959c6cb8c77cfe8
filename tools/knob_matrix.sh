#!/bin/bash
# The single-process GPU parity tests under the alternative code paths the A/B knobs select.
set -u
R=gpurun_out/r02knobs
mkdir -p $R
rc_all=0
run() {
  name="$1"; shift
  env "$@" timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "not c5 and not sweep_block and not hot_block" > $R/$name.log 2>&1
  rc=$?
  echo "$name rc=$rc $(tail -1 $R/$name.log)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  [ $rc -ne 0 ] && rc_all=1
}
run sweep_on TGCN_SWEEP=1
run narrow_old TGCN_SPMM_NARROW_BUF=0
run no_hot TGCN_HOT_ROWS=0
run no_narrow TGCN_SPMM_NARROW=0
run plain_loads TGCN_SPMM_VARIANT=8:0
run gemm_split TGCN_GEMM_SPLIT=1
run ce_kpl4 TGCN_CE_KPL=4
run no_row_sort TGCN_ROW_SORT=0
run big_items TGCN_ITEM_WEIGHT=2048 TGCN_MIN_PIECE=128 TGCN_COL_BLOCK=0
exit $rc_all
