#!/bin/bash
# Evidence of a round (run on the GPU box; results land under gpurun_out/<tag>): rocprofv3 kernel stats and PMC
# traffic of the bench command, the kernel table of an epoch, counters of the narrow kernel and of the SpMM with the
# optimizer in its epilogue, the kernel list of config c3 (no vendor GEMMs), the L2-resident ceilings, HBM activity,
# and the bench record itself.   usage: tools/collect_evidence.sh <tag> [quick]
set -u
R=gpurun_out/${1:-evidence}
QUICK=${2:-}
mkdir -p $R
root="$PWD"
export TMPDIR=/tmp
BENCH="$root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-epoch --no-hbm-activity --mode reference"
echo "python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-epoch --no-hbm-activity --mode reference" > $R/command.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$R/stats -- python3 $BENCH > $root/$R/stats.log 2>&1 || { tail -5 $root/$R/stats.log; exit 1; }
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $root/$R/fetch -- python3 $BENCH > $root/$R/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $root/$R/write -- python3 $BENCH > $root/$R/write.log 2>&1 || exit 1
echo "pmc done"
if [ -z "$QUICK" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$R/epoch -- python3 $root/tools/profile_epoch.py best > $root/$R/epoch.log 2>&1 || exit 1
echo "epoch done"
# config c3 (219 classes): the kernel list of its fused epoch must hold no rocBLAS (Cijk_*) kernel
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$R/c3 -- python3 $root/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-activity > $root/$R/c3.log 2>&1 || { tail -5 $root/$R/c3.log; exit 1; }
echo "c3 done"
cd $root
timeout -k 10 600 tools/prof_pmc.sh "$root/$R/pmc_f64" "$root/tools/sweep_spmm.py" one c4 64 > /dev/null || exit 1
python tools/summarize_pmc.py $R/pmc_f64 > $R/pmc_f64.md
timeout -k 10 900 tools/prof_pmc.sh "$root/$R/pmc_adam" "$root/tools/bench_spmm_adam.py" > /dev/null || exit 1
python tools/summarize_pmc.py $R/pmc_adam > $R/pmc_adam.md
timeout -k 10 300 python tools/ceiling_spmm.py 200 > $R/ceiling_200.log 2>&1 || exit 1
timeout -k 10 300 python tools/ceiling_spmm.py 64 > $R/ceiling_64.log 2>&1 || exit 1
grep case $R/ceiling_200.log $R/ceiling_64.log
timeout -k 10 300 python tools/hbm_activity.py --out $R/hbm_activity.json > $R/hbm_activity.log 2>&1 || exit 1
tail -3 $R/hbm_activity.log | cut -c1-200
fi
cd $root
timeout -k 10 600 python bench.py > $R/bench_c4_n1.json 2> $R/bench.err || exit 1
cut -c1-300 $R/bench_c4_n1.json
find $R -name "*_agent_info.csv" -delete
du -sh $R
