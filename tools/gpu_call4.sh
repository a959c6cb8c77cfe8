#!/bin/bash
set -u
R=gpurun_out/r02d
mkdir -p $R
root="$PWD"
(cd /tmp && hipcc --offload-arch=gfx950 -O2 $root/tools/xcc_probe.hip -o /tmp/xcc_probe && timeout -k 5 60 /tmp/xcc_probe 1 32768 > $root/$R/xcc_probe.log 2>&1; timeout -k 5 60 /tmp/xcc_probe 4 32768 >> $root/$R/xcc_probe.log 2>&1; timeout -k 5 60 /tmp/xcc_probe 1 104448 >> $root/$R/xcc_probe.log 2>&1)
grep "workgroups on" $R/xcc_probe.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "sweep" > $R/tests.log 2>&1
rc=$?
tail -5 $R/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in 1 2 4; do for F in 200 64; do
  TGCN_SWEEP_VEC=$v timeout -k 10 200 python tools/sweep_spmm.py one c4 $F 2>&1 | tail -1 | cut -c1-300 >> $R/sweep_vec.log || exit 1
done; done
cat $R/sweep_vec.log
timeout -k 10 600 tools/prof_pmc.sh "$root/$R/pmc_f200" "$root/tools/sweep_spmm.py" one c4 200 || exit 1
python tools/summarize_pmc.py $R/pmc_f200 > $R/pmc_f200.md 2>&1
grep -A28 "k_spmm_sweep" $R/pmc_f200.md | grep "k_spmm\|FETCH\|TCC_HIT\|TCC_MISS\|WAIT\|INSTS_V\|INSTS_S\|INSTS_L\|WAVE_CYC"
