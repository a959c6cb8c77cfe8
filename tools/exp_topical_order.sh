#!/bin/bash
# Times and fabric fetches of the topical-order experiment (tools/exp_topical_order.py): run on the GPU box.
#   bash tools/exp_topical_order.sh <tag> [c3|c4]
set -u
R=$PWD/gpurun_out/${1:-topical}
cfg=${2:-c3}
mkdir -p $R
root=$PWD
export TMPDIR=/tmp
python tools/exp_topical_order.py $cfg > $R/times_$cfg.log 2>&1 || { tail -5 $R/times_$cfg.log; exit 1; }
grep "F=" $R/times_$cfg.log
cd /tmp
for v in by_topic shuffled reordered_mid no_topics; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/fetch_${cfg}_$v -- python3 $root/tools/exp_topical_order.py $cfg --one $v 200 > $R/fetch_${cfg}_$v.log 2>&1 || { tail -3 $R/fetch_${cfg}_$v.log; exit 1; }
  python3 - <<PY
import csv, glob
f = glob.glob("$R/fetch_${cfg}_$v/**/*counter_collection.csv", recursive=True)[0]
per = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "k_spmm" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("k_spmm_", 1)[1].split("<")[0]
        per.setdefault(k, []).append(float(r["Counter_Value"]))
tot = sum(sum(v) / len(v) for v in per.values())
print("$cfg $v F=200: fabric fetches per launch = 2 x FETCH_SIZE = %.2f GB (" % (2 * tot * 1024 / 1e9)
      + ", ".join("%s %.2f" % (k, 2 * sum(v) / len(v) * 1024 / 1e9) for k, v in per.items()) + ")")
PY
done | tee $R/fetch_$cfg.log
cd $root
find $R -name "*_agent_info.csv" -delete
