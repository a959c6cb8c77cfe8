"""The N > 1 path on CPU: world_size 2 and 3 over gloo.  What runs here is the product's partition,
exchange (all-gather / reduce-scatter / all-reduce) and autograd logic of pytextgcn_amd.sharded;
the local operators are supplied by a test-only oracle engine (tests/_sharded_worker.py)."""
import glob
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _sharded_worker as worker  # noqa: E402

from pytextgcn_amd import sharded, synth  # noqa: E402


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run(world, kinds):
    with tempfile.TemporaryDirectory() as d:
        err = os.path.join(d, "err")
        try:
            mp.spawn(worker.main, args=(world, free_port(), kinds, err), nprocs=world, join=True)
        except Exception:
            msgs = [open(f).read() for f in sorted(glob.glob(err + ".*"))]
            pytest.fail("rank failure:\n" + "\n".join(msgs))


def test_partition_is_balanced_and_rank_local():
    g = synth.word_doc_graph(3000, 40000, seed=4)
    hubs = torch.arange(3000) < g.n_vocab
    p = sharded.Partition(g.edge_index, 3000, 4, hubs)
    assert p.hp == (g.n_vocab + 3) // 4 and p.rp == (3000 - g.n_vocab + 3) // 4
    # non-zeros of the local operators A_r + B_r: M[t, s] is computed by owner(s) when it joins a hub
    # row to a regular column, else by owner(t)
    s_, t_ = g.edge_index[0], g.edge_index[1]
    by = torch.where(hubs[t_] & ~hubs[s_], p.owner[s_], p.owner[t_])
    per_rank = torch.bincount(by, minlength=4).float()
    assert per_rank.max() <= 1.05 * per_rank.min(), per_rank
    ids = torch.cat([p.owned(q) for q in range(4)])
    assert ids[ids >= 0].unique().numel() == 3000
    # doc-doc edges across ranks cannot be represented with hubs = words only
    a, b = torch.full((9,), 2990), torch.arange(2991, 3000)
    bad = torch.cat([g.edge_index, torch.stack([torch.cat([a, b]), torch.cat([b, a])])], 1)
    with pytest.raises(ValueError, match="regular"):
        sharded.Partition(bad, 3000, 4, hubs)
    sharded.Partition(bad, 3000, 4, None)                    # all-gather partition accepts it


def test_world2_hub_partition_and_allgather_partition():
    run(2, ["wordoc", "wordoc_allhubs"])


def test_world3_asymmetric_graph_and_uneven_shards():
    run(3, ["asym", "wordoc"])


def test_world2_without_added_loops_and_without_normalisation():
    """GCNConv(add_self_loops=False) and GCNConv(normalize=False) through the chunked operator construction."""
    run(2, ["asym_keep_loops", "asym_raw"])


def test_world4_hub_partition():
    run(4, ["wordoc"])


def test_world2_accurate_mode_shares_one_operator_pair_between_m_and_its_transpose():
    """degree_sum="accurate" (opt-in): a symmetric graph's operator stays bitwise symmetric, so the partition builds
    one (A_r, B_r) pair; the package default (reference order) builds two."""
    run(2, ["wordoc@accurate", "asym@accurate"])


def test_world3_pairwise_exchange(monkeypatch):
    """TGCN_EXCHANGE=p2p: the same hub exchange as batched send/recv + all-to-all with a local sum."""
    monkeypatch.setenv("TGCN_EXCHANGE", "p2p")            # inherited by the spawned ranks
    run(3, ["wordoc", "asym"])


def test_world3_halo_exchange_with_chunked_reduce(monkeypatch):
    """TGCN_EXCHANGE=halo + TGCN_RS_CHUNKS=2 through the whole model-level check: only referenced hub rows are
    gathered, only touched partial rows are reduced, A_r runs in two row chunks."""
    monkeypatch.setenv("TGCN_EXCHANGE", "halo")
    monkeypatch.setenv("TGCN_RS_CHUNKS", "2")
    run(3, ["wordoc", "asym"])


def test_world4_graph_without_hub_structure_true_halo(monkeypatch):
    """hubs=None (every node's rows may be needed anywhere, config c5's situation): the halo form sends the rows the
    receiving rank's operator references instead of all-gathering the whole operand."""
    monkeypatch.setenv("TGCN_EXCHANGE", "halo")
    run(4, ["powerlaw_allhubs"])


def test_world4_pipelined_exchange_is_the_default_without_hubs(monkeypatch):
    """hubs=None without TGCN_EXCHANGE: the pipelined exchange (own-column block at once, three stage blocks accumulated as
    their rows land) through the exchange-form and the model-level checks (forward, backward, three Adam steps)."""
    monkeypatch.delenv("TGCN_EXCHANGE", raising=False)
    monkeypatch.setenv("TGCN_PIPE_STAGES", "3")
    run(4, ["powerlaw_allhubs", "asym"])


def test_world8_pipelined_exchange_one_peer_per_stage(monkeypatch):
    """The same with the per-peer scheme at the target rank count: stage k is the whole contribution of rank r - k - 1."""
    monkeypatch.delenv("TGCN_EXCHANGE", raising=False)
    monkeypatch.setenv("TGCN_PIPE_SCHEME", "peer")
    run(8, ["powerlaw_allhubs"])


def test_world8_every_exchange_form(monkeypatch):
    """The rank count of the target node (8 GPUs, BASELINE.json configs c4 / c5), over gloo with the oracle engine: the
    hub partition with uneven shards and padding rows, all three exchange forms with one and three row chunks
    (check_exchange_forms), and the model-level check under the halo form."""
    monkeypatch.setenv("TGCN_EXCHANGE", "halo")
    run(8, ["wordoc", "powerlaw_allhubs"])
