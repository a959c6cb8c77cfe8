#!/bin/bash
# round-2 GPU call 1: full GPU test-suite, then the HBM-activity probe and PMC passes of both SpMM widths
set -u
R=gpurun_out/r02a
mkdir -p $R
root="$PWD"
(rocprofv3 -L > $R/counters.txt 2>&1 || true)
(ls -la /sys/class/drm/ > $R/sysfs.txt 2>&1; for f in /sys/class/drm/card*/device/mem_busy_percent; do echo "$f: $(cat $f 2>&1)"; done >> $R/sysfs.txt 2>&1 || true)
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $R/gpu_tests.log 2>&1
rc=$?
tail -25 $R/gpu_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest killed (rc=$rc): stopping"; exit $rc; fi
timeout -k 10 300 python tools/hbm_activity.py --out $R/hbm_activity.json > $R/hbm_activity.log 2>&1
rc2=$?
tail -12 $R/hbm_activity.log
if [ $rc2 -eq 124 ] || [ $rc2 -eq 137 ]; then exit $rc2; fi
timeout -k 10 400 tools/prof_pmc.sh "$root/$R/pmc_f64" "$root/tools/sweep_spmm.py" one c4 64 || exit 1
timeout -k 10 400 tools/prof_pmc.sh "$root/$R/pmc_f200" "$root/tools/sweep_spmm.py" one c4 200 || exit 1
python tools/summarize_pmc.py $R/pmc_f64 > $R/pmc_f64.md 2>&1
python tools/summarize_pmc.py $R/pmc_f200 > $R/pmc_f200.md 2>&1
# keep the merge small: drop the raw per-dispatch csv of everything but the counter files
find $R -name "*_agent_info.csv" -delete 2>/dev/null
du -sh $R
exit $rc
