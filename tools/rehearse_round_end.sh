#!/bin/bash
# what the driver runs at round end: the GPU test-suite (with -x), smoke(), the default bench
set -u
R=gpurun_out/rehearsal
mkdir -p $R
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $R/gpu_tests.log 2>&1
rc=$?
tail -4 $R/gpu_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1 || { tail -5 $R/smoke.log; exit 1; }
tail -1 $R/smoke.log
timeout -k 10 600 python bench.py > $R/bench.json 2> $R/bench.err || { tail -5 $R/bench.err; exit 1; }
python -c "
import json; d=json.load(open('$R/bench.json')); r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','epoch_ms','epoch_ms_fused','epoch_ms_fused_w1_update_in_backward')})
print({k:r.get(k) for k in ('frac','frac_basis','frac_algorithmic','frac_fabric','frac_hbm','frac_compulsory','l2_resident_ceiling_ms','launch_ms')})
print(d['cpu_baseline']['value'], d['cpu_baseline']['csr']['value'])"
