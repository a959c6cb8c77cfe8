"""Host-side logic that needs no GPU: the Data container, the synthetic graph generator, the
module surface mirrored from the reference, and the loud failure of the product path on CPU."""
import pickle

import pytest
import torch

import pytextgcn_amd as pkg
from pytextgcn_amd import synth
from pytextgcn_amd.data import Data


def test_data_container_surface():
    d = Data(x=torch.eye(3), edge_index=torch.zeros(2, 4, dtype=torch.long), edge_attr=torch.ones(4),
             y=torch.arange(3), train_mask=torch.tensor([True, False, True]), n_vocab=2)
    assert d.num_nodes == 3 and d.num_edges == 4 and d.n_vocab == 2
    assert "train_mask" in d and set(d.keys) >= {"x", "edge_index", "edge_attr", "y", "n_vocab"}
    d2 = pickle.loads(pickle.dumps(d))                      # text2graph.py:195-217 save/load
    assert torch.equal(d2.edge_index, d.edge_index)
    assert d.to("cpu") is d


def test_word_doc_graph_shape_contract():
    N, E = 3000, 40000
    g = synth.word_doc_graph(N, E, seed=44, n_classes=7)
    ei = g.edge_index
    V = g.n_vocab
    assert ei.shape == (2, E) and ei.dtype == torch.int64 and ei.stride() == (1, 2)   # coo.T view
    assert g.edge_attr.shape == (E,) and g.edge_attr.dtype == torch.float32
    assert (g.edge_attr > 0).all()
    assert (ei[0] != ei[1]).all()
    key = ei[0] * N + ei[1]
    assert key.unique().numel() == E                        # no duplicates
    rev = ei[1] * N + ei[0]
    assert torch.equal(key.sort().values, rev.sort().values)            # symmetric structure
    # both directions carry the same weight
    o1, o2 = key.argsort(), rev.argsort()
    assert torch.equal(g.edge_attr[o1], g.edge_attr[o2])
    ww = (ei[0] < V) & (ei[1] < V)
    n_ww = int(ww.sum())
    assert ww[:n_ww].all() and not ww[n_ww:].any()          # word-word block first
    assert torch.equal(ei[0, :n_ww:2], ei[1, 1:n_ww:2])     # interleaved (i,j),(j,i)
    half = (E - n_ww) // 2
    assert (ei[0, n_ww:n_ww + half] >= V).all() and (ei[1, n_ww:n_ww + half] < V).all()
    assert (ei[0, n_ww + half:] < V).all() and (ei[1, n_ww + half:] >= V).all()
    assert not ((ei[0] >= V) & (ei[1] >= V)).any()          # no doc-doc edges
    assert g.x.is_sparse and g.x.shape == (N, N)
    assert (g.y[:V] == 0).all() and g.y.max() < 7
    m = g.train_mask.long() + g.val_mask.long() + g.test_mask.long()
    assert (m[:V] == 0).all() and (m[V:] == 1).all()
    g2 = synth.word_doc_graph(N, E, seed=44, n_classes=7)
    assert torch.equal(g2.edge_index, ei) and torch.equal(g2.edge_attr, g.edge_attr)


def test_power_law_graph():
    g = synth.power_law_graph(2000, 30000, seed=1)
    ei = g.edge_index
    assert ei.shape == (2, 30000) and (ei[0] != ei[1]).all()
    assert (ei[0] * 2000 + ei[1]).unique().numel() == 30000
    deg = torch.bincount(ei[1], minlength=2000)
    assert deg.max() > 20 * deg.float().mean()              # heavy tail


def test_model_surface_matches_reference():
    m = pkg.models.GCN(50, 4, n_hidden_gcn=16, dropout=0.3)
    assert sorted(m.state_dict()) == ["layers.0.bias", "layers.0.weight", "layers.1.bias",
                                      "layers.1.weight"]
    assert m.layers[0].weight.shape == (50, 16) and m.layers[1].weight.shape == (16, 4)
    assert (m.layers[0].bias == 0).all()
    a = (6.0 / (50 + 16)) ** 0.5
    assert m.layers[0].weight.abs().max() <= a
    assert isinstance(m.activation, torch.nn.ReLU) and m.dropout == 0.3
    m3 = pkg.GCN(50, 4, n_gcn=3)
    assert [tuple(l.weight.shape) for l in m3.layers] == [(50, 64), (64, 64), (64, 4)]
    m4 = pickle.loads(pickle.dumps(m))                      # th.save(gcn, ...) whole-module pickle
    assert torch.equal(m4.layers[1].weight, m.layers[1].weight)


def test_product_path_has_no_cpu_fallback():
    g = synth.word_doc_graph(100, 600, seed=1, n_classes=3)
    m = pkg.GCN(100, 3, n_hidden_gcn=8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(g)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.colsum(torch.ones(4, 4))


def test_fused_training_helpers_refuse_cpu_tensors():
    from pytextgcn_amd import dense, functional, graphbuilder, optim
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        functional.masked_cross_entropy(torch.zeros(4, 3), torch.zeros(4, dtype=torch.long),
                                        torch.ones(4, dtype=torch.bool))
    p = torch.nn.Parameter(torch.zeros(8))
    p.grad = torch.ones(8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        optim.Adam([p]).step()
    with pytest.raises(ValueError):
        optim.Adam([p], lr=-1.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):                 # no vendor / CPU matmul behind the layers
        dense.xw(torch.zeros(4, 8), torch.zeros(8, 2))
    if not torch.cuda.is_available():
        import numpy as np
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            graphbuilder.compute_word_word_edges(np.zeros((1, 2), dtype=np.int32), 3, 1, 2, 2)


def test_sharded_module_imports_without_a_process_group():
    from pytextgcn_amd import sharded
    g = synth.word_doc_graph(400, 3000, seed=2)
    p = sharded.Partition(g.edge_index, 400, 3, torch.arange(400) < g.n_vocab)
    assert p.n_local == p.hp + p.rp and p.hp * 3 >= g.n_vocab


def test_reference_import_paths_exist():
    # flat_amazon.py:13-14, test_cfunc.py:7 with `textgcn` -> `pytextgcn_amd`
    from pytextgcn_amd import Text2GraphTransformer  # noqa: F401
    from pytextgcn_amd.lib.models import GCN
    from pytextgcn_amd.lib import sliding_window_tester, test_sym_matrix, compute_word_word_edges  # noqa: F401
    assert GCN is pkg.GCN and test_sym_matrix() == 1


def test_committed_bench_line_follows_the_contract():
    """profiles/r01b_bench_c4_n1.json is a bench.py line measured on an MI355X; its shape is the
    driver's contract (task statement) plus the `roofline` and `cpu_baseline` objects."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                        "r01b_bench_c4_n1.json")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "edges/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] is not None
    assert abs(d["value"] - 2 * 50_000_000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "edges/s" and c["sample"]


def test_committed_round2_bench_line_carries_a_believable_roofline():
    """profiles/r02_bench_c4_n1.json (final round-2 library, one MI355X): the contract keys, a roofline fraction
    in [0, 1] that is the measured HBM bytes over time over peak, the other three byte counts beside it, and both
    CPU baseline rows of BASELINE.md section 3."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_bench_c4_n1.json")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "epoch_ms", "epoch_ms_fused"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["unit"] == "edges/s" and d["vs_baseline"] is None and "model" not in d["config"]
    assert abs(d["value"] - 2 * 50_000_000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.0 < r["frac"] <= 1.0
    launch_s = r["launch_ms"] * 1e-3
    assert abs(r["frac"] - r["traffic"] / launch_s / 1e9 / r["peak"]) < 1e-6        # frac = measured HBM bytes / time / peak
    assert abs(r["frac_algorithmic"] - r["achieved"] / r["peak"]) < 1e-9 and r["frac_algorithmic"] > 1.0
    assert r["frac_compulsory"] < r["frac"] < r["frac_traffic"] <= 1.0              # compulsory < HBM < fabric
    assert r["hbm_activity"]["bytes_per_step"] > 0 and "mem_busy_percent" in r["traffic_basis"]
    assert 0.0 < r["l2_resident_ceiling_ms"] < r["launch_ms"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "edges/s" and "cpu_model" in c
    assert c["csr"]["value"] > c["value"] > 0 and c["csr"]["unit"] == "edges/s"      # CPU-ref and CPU-csr rows
    assert d["epoch_ms_fused"] < d["epoch_ms"]


def test_fused_dropout_hash_has_no_measurable_structure():
    """The keep decision of the fused dropout is a stateless hash of (seed, row, col) (csrc/dense.hip: drop_row_key,
    drop_col_term, drop_elem), restated here in numpy: keep rate, per-column rates and the correlation between
    adjacent columns, columns two apart and adjacent rows stay within 4 sigma over 4 M elements, for three seeds and
    three rates.  (A one-multiply finaliser fails this at columns two apart, z = +5 ... +6: HISTORY.md 4.5.)  The
    GPU side reads the real mask back in test_fused_dropout_gemms_share_one_mask."""
    import numpy as np
    m32 = np.uint64(0xFFFFFFFF)

    def u32(x):
        return x & m32

    def row_key(s_lo, s_hi, row):
        h = u32(row ^ np.uint64(s_lo))
        h = u32(h * np.uint64(0xCC9E2D51))
        h = u32((h << np.uint64(15)) | (h >> np.uint64(17)))
        h = u32(h * np.uint64(0x1B873593))
        h = h ^ u32((row >> np.uint64(32)) + np.uint64(s_hi))
        h = h ^ (h >> np.uint64(16))
        return u32(h * np.uint64(0x85EBCA6B))

    def elem(key, col):
        h = u32(key + u32(col * np.uint64(0x9E3779B1)))
        h = h ^ (h >> np.uint64(15))
        h = u32(h * np.uint64(0x2C1B3C6D))
        h = h ^ (h >> np.uint64(12))
        h = u32(h * np.uint64(0x297A2D39))
        return h ^ (h >> np.uint64(15))

    rows = np.arange(20000, dtype=np.uint64)[:, None]
    cols = np.arange(200, dtype=np.uint64)[None, :]
    worst = 0.0
    for seed in (0x0123456789ABCDEF, 1, 2 ** 63 + 12345):
        h = elem(row_key(seed & 0xFFFFFFFF, seed >> 32, rows), cols)
        for p in (0.5, 0.7, 0.1):
            keep = (h >= np.uint64(int(p * 2 ** 32))).astype(np.float64)
            q = 1.0 - p
            var = q * (1.0 - q)
            z = [(keep.mean() - q) / np.sqrt(var / keep.size)]
            for a, b in ((keep[:, 1:], keep[:, :-1]), (keep[:, 2:], keep[:, :-2]), (keep[1:], keep[:-1])):
                z.append(((a - q) * (b - q)).mean() / var * np.sqrt(a.size))
            z.append(np.abs((keep.mean(0) - q) / np.sqrt(var / keep.shape[0])).max() - 1.0)   # max over 200 columns: ~3 expected
            worst = max(worst, max(abs(v) for v in z))
    assert worst < 4.0, worst


def test_mask_count_cache_is_keyed_by_object_not_by_address():
    """A dead temporary's storage (and id) can be handed to the next mask: the cached row count must not
    follow it."""
    from pytextgcn_amd.functional import _mask_count
    for k in range(1, 40):
        m = torch.zeros(64, dtype=torch.bool)
        m[:k] = True
        assert _mask_count(m) == k
        del m
    keep = torch.ones(10, dtype=torch.bool)
    assert _mask_count(keep) == 10 and _mask_count(keep) == 10
    keep[0] = False                                   # in-place edit bumps the version
    assert _mask_count(keep) == 9


def test_bench_launches_its_own_ranks_for_n_gt_1():
    """`python3 bench.py --gpus 2` as a plain command (no torch.distributed.run around it): bench.py starts the
    ranks itself as a child process, relays exactly one JSON line and the return code.  `--launch-check` stops after
    the rendezvous, so this runs without a GPU."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec == {"launch_check": True, "n_gpus": 2}


@pytest.mark.parametrize("fail", ["exit", "hang"])
def test_bench_falls_back_to_the_plain_configuration_when_the_first_attempt_dies_or_hangs(fail):
    """The N > 1 measurement must not be lost to one bad attempt (the driver's scaling run is one shot): a child whose
    last rank exits with an error -- or hangs past the wall budget while its peer waits in a collective -- is ended by
    its process group, a fresh child runs the plain configuration (TGCN_EXCHANGE=collective TGCN_RS_CHUNKS=1
    --no-epoch) and its record is relayed with the reason in `"fallback"`.  Rendezvous only (--launch-check): no GPU."""
    import json
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TGCN_BENCH_TEST_FAIL=fail, TGCN_BENCH_BUDGET_S="20")
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["launch_check"] is True and rec["n_gpus"] == 2
    assert ("budget" in rec["fallback"]) if fail == "hang" else ("exited with code" in rec["fallback"])
    assert time.time() - t0 < 200
    # without the fallback the failure is the caller's to see
    env["TGCN_BENCH_NO_FALLBACK"] = "1"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert res.returncode != 0 and not res.stdout.decode().strip()


def test_bench_knows_every_baseline_configuration():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.CONFIGS["c4"][:3] == (2_000_000, 50_000_000, 200)
    assert bench.CONFIGS["c5"][:3] == (8_000_000, 200_000_000, 256)
    assert bench.CONFIGS["c2"][:3] == (100_000, 2_000_000, 200)


def test_power_law_graph_carries_labels_and_masks():
    g = synth.power_law_graph(2000, 30000, seed=3, n_classes=5, features="sparse_identity")
    assert g.n_vocab == 0 and g.y.shape == (2000,) and int(g.y.max()) < 5
    assert int(g.train_mask.sum() + g.val_mask.sum() + g.test_mask.sum()) == 2000
    assert g.x.is_sparse and g.x.shape == (2000, 2000)


def test_build_refuses_spills_in_the_asm_ring_kernels():
    """pytextgcn_amd/build.py parses hipcc's kernel-resource-usage remarks for dense.hip: a kernel that feeds its
    operand through an inline-asm load ring (hand-counted s_waitcnt) must not use scratch or spill VGPRs."""
    from pytextgcn_amd import build

    def remark(name, scratch, vspill):
        return (f"dense.hip:1:1: remark: Function Name: {name} [-Rpass-analysis=kernel-resource-usage]\n"
                f"remark:     VGPRs: 200 [-R]\nremark:     ScratchSize [bytes/lane]: {scratch} [-R]\n"
                f"remark:     SGPRs Spill: 3 [-R]\nremark:     VGPRs Spill: {vspill} [-R]\n")
    ring = "_ZN4tgcn12_GLOBAL__N_111k_gemm_tallILi7ELb1ELb1ELi8ELb1ELb1EEEvPKf"
    loop = "_ZN4tgcn12_GLOBAL__N_111k_gemm_tallILi8ELb1ELb0ELi0ELb1ELb1EEEvPKf"          # NQ = 0: ordinary loads
    split = "_ZN4tgcn12_GLOBAL__N_117k_gemm_tall_splitILi2ELb0ELi13ELb1ELb0EEEvPKf"
    build.check_asm_ring_kernels(remark(ring, 0, 0) + remark(loop, 64, 9) + remark(split, 0, 0))
    with pytest.raises(RuntimeError, match="must not spill"):
        build.check_asm_ring_kernels(remark(ring, 152, 37) + remark(split, 0, 0))
    with pytest.raises(RuntimeError, match="must not spill"):
        build.check_asm_ring_kernels(remark(ring, 0, 0) + remark(split, 0, 2))
    with pytest.raises(RuntimeError, match="no kernel-resource-usage remarks"):
        build.check_asm_ring_kernels(remark(loop, 0, 0))
    # a remark block without the two fields (another compiler release's wording) is a clear error, not an AttributeError
    with pytest.raises(RuntimeError, match="carries no"):
        build.check_asm_ring_kernels(remark(ring, 0, 0).replace("ScratchSize [bytes/lane]", "Scratch bytes per lane"))


def test_build_refuses_scratch_in_any_kernel_that_is_not_allow_listed():
    """Every .hip source is compiled with -Rpass-analysis=kernel-resource-usage and `check_no_spills` fails the build on a
    kernel with scratch memory or spilled VGPRs (round 5 shipped k_gemm_tn_staged<4, 8, true, true> with 11 spilled VGPRs and
    nothing said so); rocPRIM's own kernels are allow-listed with a reason; SGPRs parked in VGPR lanes (0 bytes of scratch)
    are counted, not refused.  The report of the build that produced the library says "0 kernels with scratch"."""
    from pytextgcn_amd import build

    def remark(name, scratch, vspill, sspill=0):
        return (f"x.hip:1:1: remark: Function Name: {name} [-Rpass-analysis=kernel-resource-usage]\n"
                f"remark:     VGPRs: 200 [-R]\nremark:     ScratchSize [bytes/lane]: {scratch} [-R]\n"
                f"remark:     SGPRs Spill: {sspill} [-R]\nremark:     VGPRs Spill: {vspill} [-R]\n")
    own = "_ZN4tgcn12_GLOBAL__N_116k_gemm_tn_stagedILi4ELi8ELb1ELb1EEEvPKf"
    lib = "_ZN7rocprim17ROCPRIM_400200_NS6detail17trampoline_kernelINS1_34wrapped_radix_sort_onesweep"
    ok = build.check_no_spills("x.hip", remark(own, 0, 0, 22) + remark(lib, 80, 0) + remark("k_plain", 0, 0))
    assert ok == {"kernels": 3, "with_scratch": 0, "allow_listed_with_scratch": 1, "sgprs_parked_in_vgpr_lanes": 1}
    with pytest.raises(RuntimeError, match="use scratch memory"):
        build.check_no_spills("x.hip", remark(own, 48, 11))
    with pytest.raises(RuntimeError, match="use scratch memory"):
        build.check_no_spills("x.hip", remark("k_plain", 16, 0))
    with pytest.raises(RuntimeError, match="carries no"):
        build.check_no_spills("x.hip", remark(own, 0, 0).replace("VGPRs Spill", "Vector registers spilled"))
    rep = build.build_report()
    if rep:                                                            # this checkout's own build
        assert "0 kernels of csrc/ with scratch" in rep["summary"], rep["summary"]
        assert set(rep["sources"]) == {s for s in build.SOURCES if s.endswith(".hip")}
        assert all(v["with_scratch"] == 0 for v in rep["sources"].values())
        assert "no instruction touches" in rep["sources"]["dense.hip"]["pipe_kernel_isa_check"] \
            or "TGCN_NT_PIPE=0" in rep["sources"]["dense.hip"]["pipe_kernel_isa_check"]


def test_build_does_not_ship_an_unverified_pipe_kernel(tmp_path, monkeypatch):
    """when the disassembly scan of k_gemm_pipe cannot run (no llvm-objdump, no gfx950 bundle) the build must not
    ship the kernel unchecked and must not stay silent: it warns, compiles dense.hip again with -DTGCN_NT_PIPE=0 and records
    that in the build report.  The compiler is faked here (canned remarks, empty objects) and every path points into a
    scratch directory: this tests the build's LOGIC, not hipcc."""
    import subprocess
    import warnings
    from pytextgcn_amd import build

    def remark(name):
        return (f"dense.hip:1:1: remark: Function Name: {name} [-Rpass-analysis=kernel-resource-usage]\n"
                "remark:     VGPRs: 200 [-R]\nremark:     ScratchSize [bytes/lane]: 0 [-R]\n"
                "remark:     SGPRs Spill: 0 [-R]\nremark:     VGPRs Spill: 0 [-R]\n")
    canned = remark("_ZN4tgcn12_GLOBAL__N_111k_gemm_pipeILi7ELi8ELb1ELb1ELb1EEEvPKf") + \
        remark("_ZN4tgcn12_GLOBAL__N_111k_gemm_tallILi7ELb1ELb1ELi8ELb1ELb1EEEvPKf")
    commands = []

    class FakeProc:
        returncode = 0

        def __init__(self, cmd, **kw):
            commands.append(cmd)
            open(cmd[cmd.index("-o") + 1], "wb").close()

        def communicate(self):
            return canned, None

    def fake_run(cmd, **kw):
        commands.append(cmd)
        open(cmd[cmd.index("-o") + 1], "wb").close()
        return subprocess.CompletedProcess(cmd, 0, stdout=canned)
    monkeypatch.setattr(build, "LIB_DIR", str(tmp_path))
    monkeypatch.setattr(build, "OBJ_DIR", str(tmp_path / "obj"))
    monkeypatch.setattr(build, "LIB_PATH", str(tmp_path / "libtgcn.so"))
    monkeypatch.setattr(build, "REPORT_PATH", str(tmp_path / "build_report.json"))
    monkeypatch.setattr(build, "SOURCES", ["dense.hip"])
    monkeypatch.setattr(build.subprocess, "Popen", FakeProc)
    monkeypatch.setattr(build.subprocess, "run", fake_run)
    monkeypatch.setattr(build.shutil, "which", lambda name: "/bin/sh")           # (any existing path: it is never run)
    monkeypatch.setattr(build, "check_pipe_kernel_isa", lambda obj: "k_gemm_pipe ISA check skipped: llvm-objdump not found")
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        build.build(force=True)
    assert any("TGCN_NT_PIPE=0" in str(w.message) and "skipped" in str(w.message) for w in seen)
    compiles = [c for c in commands if "-c" in c]
    assert len(compiles) == 2 and "-DTGCN_NT_PIPE=0" not in compiles[0] and "-DTGCN_NT_PIPE=0" in compiles[1]
    note = build.build_report()["sources"]["dense.hip"]["pipe_kernel_isa_check"]
    assert "skipped" in note and note.endswith("built with -DTGCN_NT_PIPE=0")
    # ... and a scan that runs and passes leaves one compile and no warning
    commands.clear()
    monkeypatch.setattr(build, "check_pipe_kernel_isa", lambda obj: "k_gemm_pipe ISA check: 4 instantiations, no instruction touches a ring register in flight")
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        build.build(force=True)
    assert not seen and len([c for c in commands if "-c" in c]) == 1


def test_build_refuses_a_touched_ring_register_of_the_block_pipelined_kernel():
    """k_gemm_pipe keeps inline-asm loads in flight across a whole k-loop; the build scans its disassembly
    (pytextgcn_amd/build.py: scan_pipe_isa / check_pipe_kernel_isa) and fails if ANY instruction names a destination
    register of such a load before the `s_waitcnt vmcnt(0)` that covers it -- a copy slipped in by the register allocator
    would read stale data without any diagnostic (the spill remarks cannot see it)."""
    import os
    from pytextgcn_amd import build
    head = "0000000000001000 <_ZN4tgcn12_GLOBAL__N_111k_gemm_pipeILi7ELi8ELb1ELb1ELb1EEEvPKf>:\n"
    good = head + """\tglobal_load_dwordx4 v[10:13], v[2:3], off
\tglobal_load_dwordx4 v[14:17], v[2:3], off offset:32
\tv_mfma_f32_32x32x2_f32 v[100:115], v20, v21, v[100:115]
\ts_waitcnt vmcnt(0)
\tv_mfma_f32_32x32x2_f32 v[100:115], v10, v21, v[100:115]
\tglobal_store_dword v[4:5], v100, off
"""
    assert build.scan_pipe_isa(good) == (1, [])
    copied = good.replace("\tv_mfma_f32_32x32x2_f32 v[100:115], v20, v21, v[100:115]\n", "\tv_mov_b32_e32 v30, v12\n", 1)
    seen, bad = build.scan_pipe_isa(copied)
    assert seen == 1 and len(bad) == 1 and "v_mov_b32_e32 v30, v12" in bad[0]
    used_as_address = good.replace("v[14:17], v[2:3], off offset:32", "v[14:17], v[10:11], off")
    assert len(build.scan_pipe_isa(used_as_address)[1]) == 1
    other = good.replace("k_gemm_pipe", "k_gemm_tall")                 # other kernels are not this check's business
    assert build.scan_pipe_isa(other.replace("v20, v21", "v12, v21")) == (0, [])
    obj = os.path.join(build.OBJ_DIR, "dense.o")
    if os.path.exists(obj):                                            # the object of this checkout's own build
        note = build.check_pipe_kernel_isa(obj)
        assert "skipped" in note or "no instruction touches" in note, note


def test_counter_traffic_on_file_belongs_to_the_current_kernels():
    """bench.py quotes the L2 <-> fabric bytes of one tgcn_spmm launch from profiles/traffic.json, a constant collected
    with rocprofv3 counters.  It is only valid for the kernels it was collected on: the record carries a fingerprint
    of csrc/spmm.hip + plan.hip + common.h, and this test fails when those sources have changed since (re-run
    tools/collect_evidence.sh on the GPU box and commit the new profile)."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = json.load(open(os.path.join(root, "profiles", "traffic.json")))["c4_n1"]
    assert rec.get("kernel_sha16") == bench.spmm_kernel_sha16(), \
        "profiles/traffic.json[c4_n1] predates the last change of the SpMM kernels: collect the counters again"
    b, src, fresh = bench.fabric_traffic("c4", 1)
    assert fresh and b == rec["bytes_per_launch"] and 1e9 < b < 1e11


def test_committed_round3_bench_line_names_the_basis_of_its_roofline_fraction():
    """profiles/r03_bench_c4_n1.json: `roofline.frac` is the counter traffic of profiles/traffic.json (collected on the
    same kernel sources) over the live launch time over the peak, `frac_basis` says so in one word, and the HBM,
    algorithmic and compulsory fractions stand beside it; no kernel is named that did not run."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.loads(open(os.path.join(root, "profiles", "r03_bench_c4_n1.json")).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None and "model" not in d["config"]
    assert abs(d["value"] - 2 * 50_000_000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["frac_basis"] == "fabric" and r["traffic_fabric_fresh"] is True
    assert abs(r["frac"] - r["traffic"] / (r["launch_ms"] * 1e-3) / 1e9 / r["peak"]) < 1e-6 and 0 < r["frac"] <= 1.0
    assert r["frac_compulsory"] < r["frac_hbm"] < r["frac_fabric"] <= 1.0 < r["frac_algorithmic"]
    assert "sweep" not in r["kernel"] and "k_spmm_gather" in r["kernel"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["csr"]["value"] > c["value"] > 0


def _check_frozen_roofline_contract(name, exact_committed_traffic):
    """A committed bench line of the headline configuration (one MI355X).  The roofline object as frozen in round 4:
    `achieved` IS the GB/s `frac` is a fraction of (`frac == achieved / peak`, `achieved == traffic / launch time`), on
    the basis `frac_basis` names -- the counter traffic of profiles/traffic.json, collected on these kernel sources --;
    the algorithmic / fabric / HBM / compulsory figures stand side by side, each consistent with its byte count; nothing
    is capped (`non_physical` would say so); traffic.json alone reproduces the fraction (it carries the launch time of
    the counter run); the CPU-ref baseline ran on the FULL operator; the record says who took part."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.loads(open(os.path.join(root, "profiles", name)).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "rccl", "setup_s"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None and "model" not in d["config"]
    assert abs(d["value"] - 2 * 50_000_000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    per_s = 1.0 / (r["launch_ms"] * 1e-3) / 1e9
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["frac_basis"] == "fabric" and r["traffic_fabric_fresh"] is True and r["non_physical"] is False
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["traffic"] * per_s) < 1e-6 * r["achieved"]
    assert r["achieved"] == r["achieved_fabric"] and r["frac"] == r["frac_fabric"]
    assert abs(r["achieved_algorithmic"] - r["algorithmic_bytes_per_launch"] * per_s) < 1e-6 * r["achieved_algorithmic"]
    assert abs(r["frac_algorithmic"] - r["achieved_algorithmic"] / r["peak"]) < 1e-12
    assert abs(r["frac_hbm"] - r["achieved_hbm"] / r["peak"]) < 1e-12
    assert r["frac_compulsory"] < r["frac_hbm"] < r["frac_fabric"] <= 1.0 < r["frac_algorithmic"]
    assert abs(r["frac_fabric_of_gather_ceiling"] - r["achieved_fabric"] / r["fabric_gather_ceiling_GBps"]) < 1e-12
    t = json.load(open(os.path.join(root, "profiles", "traffic.json")))["c4_n1"]
    # `traffic` is measured INSIDE the run (two child runs under rocprofv3 --pmc, bench.live_fabric_traffic) when that
    # works, else it is the committed figure; either way the committed figure is in the record, and a live one agrees
    # with it (same kernels, same graph: the counters repeat to a fraction of a per cent)
    if exact_committed_traffic:          # the record of the CURRENT round: traffic.json as it stands was on file when it ran
        assert r["traffic_fabric_committed"]["bytes_per_launch"] == t["bytes_per_launch"]
    else:                                # an earlier round's record: traffic.json has been re-collected since (same kernels)
        assert abs(r["traffic_fabric_committed"]["bytes_per_launch"] - t["bytes_per_launch"]) < 1e-3 * t["bytes_per_launch"]
    assert r["traffic_fabric_committed"]["collected_on_these_kernel_sources"] is True
    if r["traffic_fabric_live_note"] and r["traffic_fabric_live_note"].startswith("live:"):
        assert abs(r["traffic"] - t["bytes_per_launch"]) < 0.02 * t["bytes_per_launch"]
        assert r["traffic_basis"].startswith("live:")
    else:
        assert abs(t["bytes_per_launch"] - r["traffic"]) < 1e-3 * r["traffic"]
    # the counter run's own launch time (rocprof kernel sum ~ HIP events of the bench inside that run) gives the same rate
    assert abs(t["launch_ms_rocprof_kernel_sum"] - t["launch_ms_bench_hip_events"]) < 0.02 * t["launch_ms_bench_hip_events"]
    assert abs(t["fabric_GBps_at_rocprof_launch_time"] - r["achieved"]) < 0.03 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["full_operator"] is True and "FULL operator" in c["sample"] and c["cores"] >= 1
    assert c["csr"]["value"] > c["value"] > 0
    assert d["rccl"]["ranks"] == 1 and d["rccl"]["distinct_devices"] == 1 and d["rccl"]["devices"][0]["pci_bus_id"]
    assert d["epoch_ms_fused_w1_update_in_backward_with_activation_reuse"] < d["epoch_ms_fused"] < d["epoch_ms"]
    return d


def test_committed_round4_bench_line_holds_the_frozen_roofline_contract():
    _check_frozen_roofline_contract("r04_bench_c4_n1.json", exact_committed_traffic=False)


def test_committed_round5_bench_line_is_in_the_reference_mode_with_the_accurate_mode_beside_it():
    """profiles/r05_bench_c4_n1.json (final round-5 library): the frozen roofline contract, and what round 5 added -- the
    headline is the reference-order normalisation mode (the package default: PyG's own fp32 arithmetic, weights bit for
    bit the oracle's, M^T stored), `config` says so, and the accurate mode's figure of the same run stands beside it and
    shows that the modes cost the same."""
    d = _check_frozen_roofline_contract("r05_bench_c4_n1.json", exact_committed_traffic=False)
    _check_reference_mode_headline(d)


def _check_reference_mode_headline(d):
    assert d["config"]["normalisation_mode"] == "reference" and "reference" in d["config"]["workload"]
    assert d["plan"]["mode"] == "reference" and d["plan"]["stores_transpose"] is True
    acc = d["accurate_mode"]
    assert acc["mode"] == "accurate" and acc["stores_transpose"] is False and d["value_accurate_mode"] == acc["value"]
    assert acc["plan_device_bytes"] < 0.6 * d["plan"]["device_bytes"]                  # one stored block instead of two
    assert abs(d["value"] - acc["value"]) < 0.02 * d["value"]                           # the modes cost the same
    assert d["rccl"]["hsa_env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_committed_round6_bench_line():
    """profiles/r06_bench_c4_n1.json (final round-6 library; `traffic.json[c4_n1]` re-collected on it): the frozen roofline
    contract on the counter figure as it stands, the reference-mode headline with the accurate mode beside it, and the epoch
    reported four ways (the other switches moved behind `--epoch-matrix`)."""
    d = _check_frozen_roofline_contract("r06_bench_c4_n1.json", exact_committed_traffic=True)
    _check_reference_mode_headline(d)
    assert d["epoch_ms_flat_loop"] < d["epoch_ms_fused_w1_update_in_backward_with_activation_reuse"] < d["epoch_ms_fused"]
    assert d["epoch_matrix"] is None and "epoch_ms_fused_with_collapsed_eval" not in d


def test_padded_row_buffers_are_recognised_only_as_they_were_handed_out():
    """plan.alloc_padded / padded_base (odd layer widths): the [n, F] view is the leading part of a zero-padded
    [n, F4] buffer; the buffer is given back for exactly that view and for nothing that merely looks like it."""
    from pytextgcn_amd import plan
    v = plan.alloc_padded(10, 7, "cpu")
    assert v.shape == (10, 7) and v.stride() == (8, 1)
    base = plan.padded_base(v, 8)
    assert base is not None and base.shape == (10, 8) and base.data_ptr() == v.data_ptr()
    assert float(base[:, 7].abs().sum()) == 0.0
    v.fill_(3.0)                                             # writes through the view never touch the pad column
    assert float(base[:, 7].abs().sum()) == 0.0 and float(base[:, :7].min()) == 3.0
    assert plan.padded_base(v[1:], 8) is None                # another address
    assert plan.padded_base(v[:, :6], 8) is None             # a narrower cut: column 6 is data, not padding
    assert plan.padded_base(torch.zeros(10, 7), 8) is None   # somebody else's tensor
    w = plan.alloc_padded(5, 8, "cpu")                       # a multiple of 4: a plain contiguous tensor
    assert w.is_contiguous() and plan.padded_base(w, 8) is None
    del base, v
    import gc
    gc.collect()
    assert all(r[0]() is not None for r in plan._PADDED.values())   # dead buffers leave the registry


def test_topical_generator_leaves_the_benchmark_graphs_alone_and_has_the_locality_it_claims():
    """synth.word_doc_graph(..., n_topics=T): the corpus with topical locality of round 5's document-order experiment (profiles/r05_exp_topical_order.log).  With
    n_topics = 0 (every BASELINE configuration) not one extra random number is drawn -- the seed-44 graphs keep the bytes
    they had in round 4 (fingerprint of a small instance) --; with topics the same graph comes out in both document orders
    (edge multiset up to the re-labelling, labels following their documents), documents of a topic are adjacent in the
    by_topic order and adjacent documents share several times more words than in the shuffled one."""
    import hashlib
    from pytextgcn_amd import synth
    g = synth.word_doc_graph(60000, 1200000, seed=44, n_classes=9)
    h = hashlib.sha256(g.edge_index.contiguous().numpy().tobytes() + g.edge_attr.numpy().tobytes() + g.y.numpy().tobytes())
    assert h.hexdigest()[:16] == "6d6b325ca1a94237"
    kw = dict(n_nodes=20000, n_edges=400000, seed=44, n_classes=10, n_topics=10, vocab_frac=0.03, doc_word_share=0.9,
              features="none")
    gt, gs = synth.word_doc_graph(**kw), synth.word_doc_graph(**kw, doc_order="shuffled")
    V, D = gt.n_vocab, 20000 - gt.n_vocab
    assert gt.edge_index.shape == gs.edge_index.shape == (2, 400000)
    assert float((gt.y[V + 1:] == gt.y[V:-1]).float().mean()) > 0.99 > 0.2 > float((gs.y[V + 1:] == gs.y[V:-1]).float().mean())
    assert torch.equal(torch.bincount(gt.y[V:], minlength=10), torch.bincount(gs.y[V:], minlength=10))
    # the same graph up to the numbering of the documents: word-side degrees, weights and the word-word block agree
    for a, b in ((gt, gs),):
        wa, wb = a.edge_index[1] < V, b.edge_index[1] < V
        assert torch.equal(torch.bincount(a.edge_index[1][wa], minlength=V), torch.bincount(b.edge_index[1][wb], minlength=V))
        assert torch.equal(a.edge_attr.sort().values, b.edge_attr.sort().values)

    def adjacent_jaccard(g):
        m = (g.edge_index[0] >= V) & (g.edge_index[1] < V)
        d, w = g.edge_index[0][m] - V, g.edge_index[1][m]
        order = torch.argsort(d, stable=True)
        d, w = d[order], w[order]
        ptr = torch.zeros(D + 1, dtype=torch.long)
        ptr[1:] = torch.bincount(d, minlength=D).cumsum(0)
        tot, n = 0.0, 0
        for i in range(0, D - 1, 37):
            a, b = set(w[ptr[i]:ptr[i + 1]].tolist()), set(w[ptr[i + 1]:ptr[i + 2]].tolist())
            tot, n = tot + len(a & b) / max(1, len(a | b)), n + 1
        return tot / n
    assert adjacent_jaccard(gt) > 2.0 * adjacent_jaccard(gs)
    with pytest.raises(ValueError):
        synth.word_doc_graph(**dict(kw, doc_order="sorted"))
