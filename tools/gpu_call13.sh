#!/bin/bash
set -u
R=gpurun_out/r02m
mkdir -p $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "w1_update or fused or graphed or capturable or training" > $R/tests.log 2>&1
rc=$?
tail -15 $R/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --no-cpu-baseline --no-hbm-activity > $R/bench.json 2> $R/bench.err
rc=$?
tail -2 $R/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02m/bench.json'))
print({k:v for k,v in d.items() if k.startswith('epoch') or k in ('value','ms_per_step')})
PY
exit $rc
