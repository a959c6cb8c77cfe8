"""The sharded product path with the HIP engine on the one GPU of the test box: two gloo ranks
sharing cuda:0 (everything but RCCL itself), and a single-rank RCCL group (the nccl code path)."""
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


def run(world, kinds, backend):
    with tempfile.TemporaryDirectory() as d:
        err = os.path.join(d, "err")
        try:
            mp.spawn(worker.main, args=(world, free_port(), kinds, err, backend, "cuda:0"),
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
