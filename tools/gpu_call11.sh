#!/bin/bash
set -u
R=gpurun_out/r02k
mkdir -p $R
export TGCN_BENCH_BACKEND=gloo TGCN_BENCH_DEVICE=0
for n in 2 4; do
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 3 --warmup 1 --config c2 > $R/bench_c2_n$n.json 2> $R/bench_c2_n$n.err
rc=$?
echo "n=$n rc=$rc"; tail -3 $R/bench_c2_n$n.err | cut -c1-600; cut -c1-1500 $R/bench_c2_n$n.json
if [ $rc -ne 0 ]; then exit $rc; fi
done
timeout -k 10 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29530 bench.py --gpus 2 --steps 3 --warmup 1 > $R/bench_c4_n2.json 2> $R/bench_c4_n2.err
rc=$?
echo "c4 n=2 rc=$rc"; tail -3 $R/bench_c4_n2.err | cut -c1-600; cut -c1-1800 $R/bench_c4_n2.json
exit $rc
