#!/bin/bash
# The driver's N > 1 launch line over RCCL on a ONE-GPU box: every rank on device 0, each posing as its own host
# (pytextgcn_amd.sharded.let_rccl_ranks_share_a_device), so RCCL joins them over loopback sockets.  The rates say
# nothing about xGMI; what the records show is that the whole N > 1 bench runs over RCCL itself.
#   bash tools/rehearse_rccl_shared_gpu.sh <config> <ranks> [<ranks> ...]
set -u
cfg=${1:-c4}; shift
R=gpurun_out/rccl_shared
mkdir -p $R
export TGCN_BENCH_DEVICE=0
port=29570
for n in "$@"; do
  port=$((port+1))
  echo "== $cfg, $n RCCL ranks on one GPU, torch.distributed.run"
  timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
      --master-port $port bench.py --gpus $n --config $cfg --steps 5 --warmup 1 \
      > $R/bench_${cfg}_rccl${n}.json 2> $R/bench_${cfg}_rccl${n}.err || { tail -20 $R/bench_${cfg}_rccl${n}.err; exit 1; }
  python - <<PY
import json
d = json.load(open("$R/bench_${cfg}_rccl${n}.json"))
print({k: d.get(k) for k in ("n_gpus", "value", "ms_per_step", "sharded_epoch_ms", "setup_s")})
print(d["rccl"]["backend"], d["rccl"]["ranks"], "distinct devices", d["rccl"]["distinct_devices"], "high priority", d["rccl"]["high_priority_stream"])
print(d.get("exchange_selection"))
print("parity vs the single-device plan:", d.get("distributed_parity"))
PY
done
