"""The drop-in scripts run end to end (GPU): `examples/flat_synthetic.py` is the literal replay of the reference's training
script (flat_amazon.py:60-134: corpus -> Text2GraphTransformer -> Data -> GCN(graph) -> CE on train_mask -> Adam(amsgrad) ->
eval with val_loss and host-side metrics -> test metrics) on this package, `examples/flat_synthetic_multigpu.py` the same
loop on the 1-D partition.  Each is started as a child process, as a user would start it."""
import os
import re
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPOCH = re.compile(r"\[\s*(\d+)\] loss:\s*([-\d.]+), val_loss:\s*([-\d.]+), (?:training accuracy:\s*([-\d.]+), val_f1|val accuracy):\s*([-\d.]+)")


def _run(cmd, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT, timeout=timeout)
    out = res.stdout.decode()
    assert res.returncode == 0, (out[-1500:], res.stderr.decode()[-3000:])
    return out


def _epochs(out):
    rows = [m.groups() for m in map(EPOCH.search, out.splitlines()) if m]
    assert len(rows) >= 3, out[-1500:]
    return rows


def test_flat_script_trains_in_both_forms_and_they_agree(cuda):
    """`flat_synthetic.py --docs 2000 --epochs 30`, plain (torch's CE / Adam / dropout around the HIP operators: what the
    import swap alone gives) and `--fused` (train.FlatLoop): both end, the loss falls, the training accuracy rises, a
    validation loss is reported every epoch (flat_amazon.py:110), and the two forms reach the same test accuracy +- 1 %."""
    accs = {}
    for form in ([], ["--fused"]):
        out = _run([sys.executable, os.path.join(ROOT, "examples", "flat_synthetic.py"), "--docs", "2000", "--epochs", "30"] + form)
        rows = _epochs(out)
        first, last = rows[0], rows[-1]
        assert int(last[0]) == 30
        assert float(last[1]) < 0.5 * float(first[1]), (first, last)            # training loss
        assert float(last[2]) < float(first[2]), (first, last)                  # validation loss
        assert float(last[3]) > float(first[3]) + 0.2 and float(last[3]) > 0.9, (first, last)   # training accuracy
        m = re.search(r"Test Accuracy:\s*([\d.]+)\s+F1-Macro:\s*([\d.]+)", out)
        assert m, out[-800:]
        accs["fused" if form else "plain"] = float(m.group(1))
        assert accs["fused" if form else "plain"] > 0.9
    assert abs(accs["plain"] - accs["fused"]) <= 0.01 + 1e-9, accs


def test_flat_script_with_the_dbpedia_settings(cuda):
    """`flat_synthetic.py --preset dbpedia`: flat_dbpedia.py's hyper-parameters (window 5, documents cut to 15 tokens, max_df 0.4,
    hidden width 32, dropout 0.5, validation documents appended behind the training documents) and 19 classes -- a class count
    that is no multiple of 4, as DBpedia's 219 -- in both forms of the loop."""
    accs = {}
    for form in ([], ["--fused"]):
        out = _run([sys.executable, os.path.join(ROOT, "examples", "flat_synthetic.py"), "--docs", "3000", "--epochs", "60",
                    "--preset", "dbpedia"] + form)
        rows = _epochs(out)
        first, last = rows[0], rows[-1]
        # (15-token documents, 19 classes: the network memorises the training rows -- the validation loss is reported every
        # epoch, flat_dbpedia.py:110, but need not fall)
        assert int(last[0]) == 60 and float(last[1]) < 0.2 * float(first[1]) and float(last[2]) > 0, (first, last)
        assert float(last[3]) > 0.9 > float(first[3]), (first, last)
        m = re.search(r"Test Accuracy:\s*([\d.]+)\s+F1-Macro:\s*([\d.]+)", out)
        assert m, out[-800:]
        accs["fused" if form else "plain"] = float(m.group(1))
    assert min(accs.values()) > 0.25 and abs(accs["plain"] - accs["fused"]) <= 0.05, accs      # (chance: 1 / 19)


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("extra", [[], ["--loop-object", "--hidden", "200"]])
def test_multi_gpu_script_under_gloo_world_2(cuda, extra):
    """`flat_synthetic_multigpu.py` as two gloo ranks sharing the test box's GPU (the launch line of its docstring): the plain
    loop and the loop object (W1's update inside the backward SpMM, activation reuse, rows=...)."""
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(_port()), os.path.join(ROOT, "examples", "flat_synthetic_multigpu.py"),
                "--docs", "2000", "--epochs", "30", "--backend", "gloo", "--device", "0"] + extra, timeout=900)
    rows = _epochs(out)
    first, last = rows[0], rows[-1]
    assert int(last[0]) == 30 and float(last[1]) < 0.5 * float(first[1]) and float(last[2]) < float(first[2]), (first, last)
    assert float(last[4]) > float(first[4]) and float(last[4]) > 0.9, (first, last)          # validation accuracy
    m = re.search(r"2 rank\(s\): 30 epochs in [\d.]+ s; test accuracy ([\d.]+)", out)
    assert m and float(m.group(1)) > 0.9, out[-800:]


def test_bench_single_gpu_record_and_its_epoch_matrix(cuda):
    """`python bench.py --config c2 --epoch-matrix` on the test box's GPU (no rocprofv3 child runs, no CPU leg): the contract
    keys, the roofline object, the FOUR epoch figures a record carries and -- with the flag -- the switches one by one."""
    import json
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c2", "--steps", "5", "--warmup", "2",
                "--no-cpu-baseline", "--no-live-traffic", "--no-hbm-activity", "--epoch-matrix"], timeout=600)
    lines = [ln for ln in out.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out[-1500:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["unit"] == "edges/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert abs(d["value"] - 2 * 2_000_000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    four = [d[k] for k in ("epoch_ms", "epoch_ms_fused", "epoch_ms_fused_w1_update_in_backward_with_activation_reuse",
                           "epoch_ms_flat_loop")]
    assert all(isinstance(v, float) and v > 0 for v in four) and four[1] < four[0], four
    m = d["epoch_matrix"]
    assert set(m) == {"fused_with_activation_reuse", "fused_with_collapsed_eval", "fused_w1_update_in_backward",
                      "fused_w1_reuse_needed_rows_only", "fused_w1_reuse_split_bf16_gemms"}
    assert all(isinstance(v, float) and v > 0 for v in m.values()), m
    assert d["secondary_errors"] is None
