"""The sharded product path with the HIP engine on the GPU(s) of the test box: gloo ranks sharing
cuda:0, a single-rank RCCL group, RCCL groups of 2 / 3 / 4 ranks that SHARE cuda:0 (each rank poses as its own host, so
RCCL connects them over loopback sockets) and, when the box has two or more GPUs, a two-rank RCCL group with one rank
per GPU.

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
    """One RCCL rank: every collective degenerates; the hub-less graph runs the pipelined exchange with stages that carry no
    rows (all_to_all_single with all-zero splits, stage blocks without entries)."""
    run(1, ["wordoc_big", "wordoc_allhubs"], "nccl")


def test_four_gloo_ranks_pairwise_exchange(cuda, monkeypatch):
    """World 4 with TGCN_EXCHANGE=p2p (batched send/recv + all-to-all) and the HIP engine."""
    monkeypatch.setenv("TGCN_EXCHANGE", "p2p")
    run(4, ["wordoc_big"], "gloo")


def _n_gpus():
    import torch
    return torch.cuda.device_count()


_RCCL_SHARED = {}


# what RCCL prints when it REFUSES ranks that share a device or cannot use the loopback interface: the only outcomes of the
# probe that mean "this box cannot do it" (skip).  Anything else that goes wrong in the probe -- a timeout, a signal, an
# abort, a wrong result -- is a failure of RCCL on the box and FAILS the tests that depend on it.
_RCCL_REFUSALS = ("duplicate gpu detected", "no socket interface found", "no interface found", "bootstrap : no socket",
                  "failed to find a usable interface")


def _rccl_ranks_can_share_the_gpu():
    """Two RCCL ranks on one GPU need loopback sockets and an RCCL that honours NCCL_HOSTID: probed
    ONCE per session (tests/_rccl_one_gpu_probe.py: init + all-reduce / all-gather / reduce-scatter / send-recv, bounded).
    A box whose RCCL refuses the arrangement (the signatures above) skips the shared-GPU RCCL tests -- they test the
    product's call sequences under RCCL, not the box's networking; a probe that hangs, dies or computes a wrong value
    fails them, with its output attached."""
    if "ok" not in _RCCL_SHARED:
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env.update(PROBE_BUDGET_S="90", MASTER_PORT=str(free_port()))
        try:
            res = subprocess.run([sys.executable, os.path.join(here, "_rccl_one_gpu_probe.py")],
                                 env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=150)
            text = res.stdout.decode("utf-8", "replace")
            _RCCL_SHARED["ok"] = res.returncode == 0
            _RCCL_SHARED["refused"] = res.returncode != 0 and res.returncode != 124 and \
                any(sig in text.lower() for sig in _RCCL_REFUSALS)
            _RCCL_SHARED["why"] = f"exit {res.returncode}: " + text[-1200:]
        except subprocess.TimeoutExpired as e:
            _RCCL_SHARED["ok"], _RCCL_SHARED["refused"] = False, False
            _RCCL_SHARED["why"] = "the probe did not finish in 150 s: " + (e.stdout or b"").decode("utf-8", "replace")[-1200:]
    if _RCCL_SHARED["ok"]:
        return
    if _RCCL_SHARED["refused"]:
        pytest.skip("RCCL refuses ranks that share the GPU on this box: " + _RCCL_SHARED["why"][-400:])
    pytest.fail("the RCCL probe (two ranks on one GPU over loopback) hung, died or computed a wrong value -- not a "
                "refusal signature:\n" + _RCCL_SHARED["why"])


@pytest.mark.parametrize("exchange", ["collective", "p2p", "halo"])
def test_two_rccl_ranks_one_per_gpu(cuda, monkeypatch, exchange):
    """RCCL with more than one rank: all-gather / reduce-scatter / all-reduce, the pairwise form and the halo form
    (all_to_all_single with split sizes) over xGMI -- and, inside the worker, every form against every other
    (check_exchange_forms).  Needs two GPUs; the one-GPU test box skips it."""
    if _n_gpus() < 2:
        pytest.skip("needs >= 2 GPUs (the one-GPU form of this test is test_rccl_ranks_sharing_one_gpu)")
    monkeypatch.setenv("TGCN_EXCHANGE", exchange)
    run(2, ["wordoc_big", "wordoc_allhubs"], "nccl", device="cuda:{rank}")


@pytest.mark.parametrize("prefix", ["0", "4096"])
def test_two_rccl_ranks_one_per_gpu_pipelined_exchange(cuda, monkeypatch, prefix):
    """The pipelined exchange of a hub-less graph with one rank per GPU (xGMI / PCIe P2P between two devices): packed
    all-to-all stages, and with a prefix the unpacked ranges (batched send / recv of one source buffer).  Needs two GPUs;
    the one-GPU test box skips it (its one-GPU form is test_rccl_ranks_sharing_one_gpu_pipelined_exchange)."""
    if _n_gpus() < 2:
        pytest.skip("needs >= 2 GPUs")
    monkeypatch.delenv("TGCN_EXCHANGE", raising=False)
    monkeypatch.setenv("TGCN_PIPE_STAGES", "3")
    monkeypatch.setenv("TGCN_PIPE_PREFIX", prefix)
    run(2, ["powerlaw_big_allhubs"], "nccl", device="cuda:{rank}")


@pytest.mark.parametrize("world,exchange,chunks", [(2, "collective", "1"), (2, "p2p", "1"), (2, "halo", "1"),
                                                   (2, "collective", "4"), (4, "collective", "2"), (3, "halo", "1")])
def test_rccl_ranks_sharing_one_gpu(cuda, monkeypatch, world, exchange, chunks):
    """RCCL itself with world size > 1 on a ONE-GPU box: every rank on cuda:0 with its own NCCL_HOSTID, so the
    communicator joins them over loopback sockets (sharded.let_rccl_ranks_share_a_device).  What gloo cannot show: the
    collectives are ENQUEUED on RCCL's (high-priority) stream and overlap the local SpMMs, the send / recv pairs are
    batched groups, reduce_scatter_tensor / all_gather_into_tensor are the native calls.  Same checks as every other
    backend: the partition against the oracle, every exchange form against every other, the sharded network."""
    _rccl_ranks_can_share_the_gpu()
    monkeypatch.setenv("TGCN_EXCHANGE", exchange)
    monkeypatch.setenv("TGCN_RS_CHUNKS", chunks)
    run(world, ["wordoc_big", "wordoc_allhubs"] if world == 2 else ["wordoc_big"], "nccl")


def test_two_gloo_ranks_on_one_gpu_pipelined_exchange_without_hubs(cuda, monkeypatch):
    """hubs=None with the HIP engine: the pipelined exchange (the default there) -- own-column block at once, the stage blocks
    added by `tgcn_spmm_acc` as their rows land -- against the oracle, against the halo form, through the sharded network
    and the W1 update in the backward pass."""
    monkeypatch.delenv("TGCN_EXCHANGE", raising=False)
    monkeypatch.setenv("TGCN_PIPE_STAGES", "3")
    monkeypatch.setenv("TGCN_PIPE_PREFIX", "2048")           # three unpacked ranges of the prefix + one packed stage
    run(2, ["powerlaw_big_allhubs", "wordoc_allhubs"], "gloo")


@pytest.mark.parametrize("world,stages,scheme,prefix", [(2, "2", "slices", "0"), (4, "3", "slices", "0"), (3, "", "peer", "0"),
                                                        (2, "3", "slices", "4096"), (3, "2", "slices", "auto")])
def test_rccl_ranks_sharing_one_gpu_pipelined_exchange(cuda, monkeypatch, world, stages, scheme, prefix):
    """The same over RCCL (ranks sharing cuda:0 over loopback sockets): K all_to_all_single calls with split sizes are
    posted on the communicator's stream before the own-column block is launched, and every stage block waits for its own.
    With a prefix the first K stages send contiguous ranges of the operand unpacked (batched send / recv) and one packed
    stage follows."""
    _rccl_ranks_can_share_the_gpu()
    monkeypatch.delenv("TGCN_EXCHANGE", raising=False)
    monkeypatch.setenv("TGCN_PIPE_STAGES", stages)
    monkeypatch.setenv("TGCN_PIPE_SCHEME", scheme)
    monkeypatch.setenv("TGCN_PIPE_PREFIX", prefix)
    run(world, ["powerlaw_big_allhubs"], "nccl")


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
    par = rec["distributed_parity"]               # the timed pair against the single-device plan, inside the same run
    assert par["ok"] is True and par["max_rel_err_forward"] < 1e-5 and par["max_rel_err_transposed"] < 1e-5, par


def _bench_two_rccl_ranks(cmd_prefix, extra_env=None):
    import json
    import subprocess
    _rccl_ranks_can_share_the_gpu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TGCN_BENCH_DEVICE="0", **(extra_env or {}))        # backend: the default, "nccl" = RCCL
    res = subprocess.run(cmd_prefix + [os.path.join(root, "bench.py"), "--gpus", "2", "--config", "c2", "--steps", "3",
                                       "--warmup", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env,
                         timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["unit"] == "edges/s"
    assert rec["rccl"]["backend"] == "nccl (RCCL)" and rec["rccl"]["ranks"] == 2
    assert rec["rccl"]["distinct_devices"] == 1                  # a rehearsal, and the record says so
    assert rec["rccl"]["high_priority_stream"] is True
    assert rec["distributed_parity"]["ok"] is True, rec["distributed_parity"]
    return rec


def test_bench_two_rccl_ranks_sharing_one_gpu_as_a_plain_command(cuda):
    """`python3 bench.py --gpus 2` over RCCL (not gloo): the parent's budget / fallback machinery, every exchange form
    in the trial steps (the parent's watchdog covers the pairwise ones), the timed region and the sharded epoch."""
    rec = _bench_two_rccl_ranks([sys.executable])
    assert "fallback" not in rec, rec.get("fallback")
    assert rec["exchange_selection"]["chosen"] in rec["exchange_selection"]["ms_per_step"]


def test_bench_two_rccl_ranks_sharing_one_gpu_under_torch_distributed_run(cuda):
    """The driver's own launch line for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`), over RCCL."""
    rec = _bench_two_rccl_ranks([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                 "--master-addr", "127.0.0.1", "--master-port", str(free_port())])
    sel = rec["exchange_selection"]
    # nobody watches this launch: RCCL's own collectives and the all_to_all_single forms are tried, the pairwise form is not
    assert sel["forms_tried"] == ["collective", "halo"] and sel["watched_by_launcher"] is False
    assert sel["chosen"].split("/")[0] in ("collective", "halo") and not any(k.startswith("p2p") for k in sel["ms_per_step"])


def test_bench_two_rccl_ranks_without_hub_structure_try_the_pipelined_exchange(cuda):
    """BASELINE config c5's situation at a rehearsal size (`--config c5s`: 1 M nodes / 25 M edges, power law, h = 256) under the
    driver's launch line: no hub structure, so the trial steps time the whole-operand all-gather, the halo form and the
    pipelined exchange, the record says which won, and `scaling_model` reports the pipeline's phases and what of the
    exchange stayed exposed."""
    import json
    import subprocess
    _rccl_ranks_can_share_the_gpu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TGCN_BENCH_DEVICE="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "2",
                          "--config", "c5s", "--steps", "3", "--warmup", "1", "--no-epoch"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    sel = rec["exchange_selection"]
    assert "pipeline" in sel["forms_tried"] and any(k.startswith("pipeline-slices/") for k in sel["ms_per_step"]), sel
    assert all(v is not None for v in sel["ms_per_step"].values()), sel          # every form ran
    assert rec["distributed_parity"]["ok"] is True, rec["distributed_parity"]
    if sel["chosen"].startswith("pipeline"):
        sm = rec["scaling_model"]
        assert sm["form"].startswith("pipeline-") and sm["per_spmm_ms"]["compute_side_own_plus_stage_blocks"] > 0


def test_bench_rank_failing_alone_in_a_secondary_measurement_costs_neither_a_hang_nor_the_headline(cuda):
    """a rank that raises inside a guarded secondary section has left its peers inside a collective.  It must
    not enter another collective (mis-paired calls) and must not be swallowed: the run ends at once (no 180 s timeout),
    rank 0 hands over the headline it holds -- marked `aborted`, with the secondaries that had completed -- and, started
    as a plain command, the launcher's fresh-child fallback delivers a clean record."""
    import json
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TGCN_BENCH_BACKEND="gloo", TGCN_BENCH_DEVICE="0", TGCN_BENCH_TEST_FAIL="secondary")
    args = [os.path.join(root, "bench.py"), "--gpus", "2", "--config", "c2", "--steps", "3", "--warmup", "1"]
    # (1) under the driver's own launch line: a non-zero exit, quickly, and ONE line that still carries the headline
    t0 = time.time()
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(free_port())] + args, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, env=env, timeout=600, cwd=root)
    assert res.returncode != 0 and time.time() - t0 < 170, (res.returncode, time.time() - t0)
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, (lines, res.stderr.decode()[-2000:])
    rec = json.loads(lines[0])
    assert rec["value"] > 0 and rec["n_gpus"] == 2 and "aborted" in rec
    assert rec["distributed_parity"]["ok"] is True            # what had completed before the failure is in the record
    # (2) as a plain command: the launcher sees the failure and runs the plain configuration in a fresh child
    res = subprocess.run([sys.executable] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["value"] > 0 and "exited with code" in rec["fallback"] and "aborted" not in rec
