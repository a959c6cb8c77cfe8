#!/bin/bash
set -u
R=gpurun_out/r02h
mkdir -p $R
root="$PWD"
export TGCN_SWEEP=1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "sweep" > $R/tests.log 2>&1
rc=$?
tail -3 $R/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in 4 1; do for F in 200 64; do
  TGCN_SWEEP_VEC=$v timeout -k 10 200 python tools/sweep_spmm.py one c4 $F 2>&1 | tail -1 | cut -c1-200 >> $R/sweep_vec.log || exit 1
done; done
cat $R/sweep_vec.log
for v in 4 1; do
TGCN_SWEEP_VEC=$v timeout -k 10 600 tools/prof_pmc.sh "$root/$R/pmc_f200_v$v" "$root/tools/sweep_spmm.py" one c4 200 > /dev/null || exit 1
python tools/summarize_pmc.py $R/pmc_f200_v$v > $R/pmc_f200_v$v.md 2>&1
grep -A28 "k_spmm_sweep" $R/pmc_f200_v$v.md | grep "k_spmm\|FETCH\|TCC_HIT\|TCC_MISS\|WAVE_CYC\|BUSY_CYC"
done
