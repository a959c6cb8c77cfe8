#!/bin/bash
# round-2 final evidence: kernel stats + PMC traffic of the bench command, epoch kernel table, narrow-kernel PMC,
# L2-resident ceilings, HBM activity, the bench record itself
set -u
R=gpurun_out/r02z
mkdir -p $R
root="$PWD"
export TMPDIR=/tmp
BENCH="$root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-epoch --no-hbm-activity"
echo "python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-epoch --no-hbm-activity" > $R/command.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$R/stats -- python3 $BENCH > $root/$R/stats.log 2>&1 || { tail -5 $root/$R/stats.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $root/$R/fetch -- python3 $BENCH > $root/$R/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $root/$R/write -- python3 $BENCH > $root/$R/write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$R/epoch -- python3 $root/tools/profile_epoch.py > $root/$R/epoch.log 2>&1 || exit 1
cd $root
python profiles/summarize.py r02 $R/stats $R/fetch $R/write > $R/summary.txt 2>&1; tail -3 $R/summary.txt | cut -c1-400
python profiles/summarize.py epoch r02 $R/epoch 5 > $R/epoch_summary.txt 2>&1; head -20 $R/epoch_summary.txt
timeout -k 10 600 tools/prof_pmc.sh "$root/$R/pmc_f64" "$root/tools/sweep_spmm.py" one c4 64 > /dev/null || exit 1
python tools/summarize_pmc.py $R/pmc_f64 > $R/pmc_f64.md
timeout -k 10 300 python tools/ceiling_spmm.py 200 > $R/ceiling_200.log 2>&1 || exit 1
timeout -k 10 300 python tools/ceiling_spmm.py 64 > $R/ceiling_64.log 2>&1 || exit 1
cat $R/ceiling_200.log $R/ceiling_64.log
timeout -k 10 300 python tools/hbm_activity.py --out $R/hbm_activity.json > $R/hbm_activity.log 2>&1 || exit 1
tail -3 $R/hbm_activity.log | cut -c1-300
timeout -k 10 600 python bench.py > $R/bench_c4_n1.json 2> $R/bench.err || exit 1
cut -c1-400 $R/bench_c4_n1.json
find $R -name "*_agent_info.csv" -delete; find $R -name "*kernel_trace.csv" -size +3M -delete; du -sh $R
