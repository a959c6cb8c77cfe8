#!/bin/bash
set -u
R=gpurun_out/r02i
mkdir -p $R
( time timeout -k 10 900 python bench.py > $R/bench_c4.json 2> $R/bench_c4.err ) 2> $R/bench_c4.time
rc=$?
tail -3 $R/bench_c4.err; cat $R/bench_c4.time; cut -c1-3000 $R/bench_c4.json
exit $rc
