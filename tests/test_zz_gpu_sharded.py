"""The sharded product path with the HIP engine on the GPU(s) of the test box: gloo ranks sharing
cuda:0 (everything but RCCL itself), a single-rank RCCL group (the nccl code path) and, when the box
has two or more GPUs, a two-rank RCCL group with one rank per GPU.

The file name sorts last on purpose: these are multi-process tests, and a failure here under
`pytest -x` must not keep the single-process graph-builder / Text2Graph GPU tests from running."""
import glob
import os
import sys
import tempfile

import pytest
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _sharded_worker as worker  # noqa: E402
from test_sharded import free_port  # noqa: E402

pytestmark = pytest.mark.gpu


def run(world, kinds, backend, device="cuda:0"):
    with tempfile.TemporaryDirectory() as d:
        err = os.path.join(d, "err")
        try:
            mp.spawn(worker.main, args=(world, free_port(), kinds, err, backend, device),
                     nprocs=world, join=True)
        except Exception:
            msgs = [open(f).read() for f in sorted(glob.glob(err + ".*"))]
            pytest.fail("rank failure:\n" + "\n".join(msgs))


def test_two_gloo_ranks_on_one_gpu_hub_partition(cuda):
    run(2, ["wordoc_big", "wordoc_allhubs"], "gloo")


def test_single_rank_rccl_group(cuda):
    run(1, ["wordoc_big"], "nccl")


def test_four_gloo_ranks_pairwise_exchange(cuda, monkeypatch):
    """World 4 with TGCN_EXCHANGE=p2p (batched send/recv + all-to-all) and the HIP engine."""
    monkeypatch.setenv("TGCN_EXCHANGE", "p2p")
    run(4, ["wordoc_big"], "gloo")


def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("exchange", ["collective", "p2p", "halo"])
def test_two_rccl_ranks_one_per_gpu(cuda, monkeypatch, exchange):
    """RCCL with more than one rank: all-gather / reduce-scatter / all-reduce, the pairwise form and the halo form
    (all_to_all_single with split sizes) over xGMI -- and, inside the worker, every form against every other
    (check_exchange_forms).  Needs two GPUs; the one-GPU test box skips it."""
    if _n_gpus() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    monkeypatch.setenv("TGCN_EXCHANGE", exchange)
    run(2, ["wordoc_big", "wordoc_allhubs"], "nccl", device="cuda:{rank}")


def test_bench_two_ranks_as_a_plain_command(cuda):
    """`python3 bench.py --gpus 2 --config c2` with no launcher around it: two gloo ranks sharing cuda:0 rehearse the
    whole N > 1 bench (self-launch, graph broadcast, partition, exchange-form trial steps, timed region, sharded epoch)
    and rank 0's single JSON line comes back through the parent."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TGCN_BENCH_BACKEND="gloo", TGCN_BENCH_DEVICE="0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "c2",
                          "--steps", "3", "--warmup", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         env=env, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["unit"] == "edges/s"
    assert rec["exchange_selection"]["chosen"] in rec["exchange_selection"]["ms_per_step"]
