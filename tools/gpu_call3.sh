#!/bin/bash
set -u
R=gpurun_out/r02c
mkdir -p $R
root="$PWD"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "sweep or hot or c4 or c3 or c2 or widths or long_rows or split or reproducible or stress" > $R/tests.log 2>&1
rc=$?
tail -15 $R/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for sw in 0 1; do for F in 200 64; do
  TGCN_SWEEP=$sw timeout -k 10 200 python tools/sweep_spmm.py one c4 $F 2>&1 | tail -1 | cut -c1-500 >> $R/sweep_onoff.log || exit 1
done; done
cat $R/sweep_onoff.log
timeout -k 10 600 tools/prof_pmc.sh "$root/$R/pmc_f200" "$root/tools/sweep_spmm.py" one c4 200 || exit 1
python tools/summarize_pmc.py $R/pmc_f200 > $R/pmc_f200.md 2>&1
grep -A3 "k_spmm_sweep\|k_spmm_gather" $R/pmc_f200.md | head -40
