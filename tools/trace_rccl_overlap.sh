#!/bin/bash
# Kernel trace of rank 0 of a 2-rank RCCL run on ONE GPU (HISTORY 6.7): rank 1 runs beside it unprofiled.  No launcher:
# rocprofv3 gets `python3 bench.py` itself, the rank's environment is exported here.
#   bash tools/trace_rccl_overlap.sh <chunks>                                   c4, RCCL collectives, A_r in <chunks> row chunks
#   CFG=c5s EXCHANGE=pipeline STAGES=4 bash tools/trace_rccl_overlap.sh 1       no hub structure: the pipelined exchange
set -u
K=${1:-1}
CFG=${CFG:-c4}
EXCHANGE=${EXCHANGE:-collective}
R=$PWD/gpurun_out/rccl_trace_${CFG}_${EXCHANGE}_k$K
mkdir -p $R
export TMPDIR=/tmp WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29590+K)) TGCN_BENCH_DEVICE=0
export TGCN_EXCHANGE=$EXCHANGE TGCN_RS_CHUNKS=$K
if [ -n "${STAGES:-}" ]; then export TGCN_PIPE_STAGES=$STAGES; fi
ARGS="--gpus 2 --config $CFG --steps 4 --warmup 1 --no-epoch --no-cpu-baseline --no-hbm-activity"
RANK=1 LOCAL_RANK=1 timeout -k 10 400 python3 bench.py $ARGS > $R/rank1.out 2> $R/rank1.err &
peer=$!
export RANK=0 LOCAL_RANK=0
root=$PWD
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/trace -- python3 $root/bench.py $ARGS > $R/rank0.json 2> $R/rank0.err
rc=$?
cd $root
wait $peer
echo "rank 0 rc=$rc, rank 1 rc=$?"
csv=$(find $R/trace -name '*kernel_trace.csv' | head -1)
python3 tools/rccl_overlap.py "$csv" 70 > $R/overlap.md
head -60 $R/overlap.md
