#!/bin/bash
# Round-3 closing runs on the GPU box: the c3 kernel list (no vendor GEMM), the bench records of c4 / c2 / c3 / c5 on
# one GPU, and a two-rank rehearsal of the c4 bench over gloo on one card.
set -u
R=gpurun_out/r03f
mkdir -p $R
root="$PWD"
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "collapse or c3" > $R/tests.log 2>&1; tail -2 $R/tests.log
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$R/c3 -- python3 $root/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-activity > $root/$R/c3.log 2>&1 || { tail -5 $root/$R/c3.log; }
cd $root
grep -c Cijk $R/c3/*/*_kernel_stats.csv
grep '"metric"' $R/c3.log > $R/bench_c3_n1.json
timeout -k 10 600 python bench.py > $R/bench_c4_n1.json 2> $R/bench_c4.err || tail -5 $R/bench_c4.err
cut -c1-200 $R/bench_c4_n1.json
timeout -k 10 300 python bench.py --config c2 > $R/bench_c2_n1.json 2> $R/bench_c2.err || tail -5 $R/bench_c2.err
cut -c1-200 $R/bench_c2_n1.json
timeout -k 10 900 python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline > $R/bench_c5_n1.json 2> $R/bench_c5.err || tail -5 $R/bench_c5.err
cut -c1-200 $R/bench_c5_n1.json
TGCN_BENCH_BACKEND=gloo TGCN_BENCH_DEVICE=0 timeout -k 10 900 python bench.py --gpus 2 --steps 3 --warmup 1 > $R/bench_c4_gloo2.json 2> $R/bench_c4_gloo2.err || tail -8 $R/bench_c4_gloo2.err
cut -c1-300 $R/bench_c4_gloo2.json
find $R -name "*_agent_info.csv" -delete
