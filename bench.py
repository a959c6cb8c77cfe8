#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's configuration.

metric   edges/sec of the forward + backward normalised SpMM pair at width F = h = 200
         (= 2*E / (t_fwd + t_bwd)), on the synthetic 2 M-node / 50 M-edge
         PMI / TF-IDF word-document graph (config c4, seed 44), inputs resident in HBM.
step     one pass of the hot path: out = M @ X + b (GCNConv propagate + bias, layer-1 width) and
         dXW = M^T @ dOut (its autograd), both through libtgcn.so.
N > 1    the SAME graph 1-D row-partitioned over N GPUs ("strong" scaling): per SpMM one RCCL
         all-gather of the row-sharded operand, then the local row block (pytextgcn_amd/sharded.py).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel:
the CSR gather SpMM, HBM-bound) and `cpu_baseline` (the oracle's reference formulation timed on
this box's host cores over a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (n_nodes, n_edges, F, n_classes)
    "c2": (100_000, 2_000_000, 200, 64),
    "c4": (2_000_000, 50_000_000, 200, 64),
    "c3": (1_000_000, 24_000_000, 200, 219),     # DBpedia-shaped: V = 30 k, 219 classes (l3)
    "c5": (8_000_000, 200_000_000, 256, 64),     # generic power law (no word / document structure), h = 256
    "c5s": (1_000_000, 25_000_000, 256, 64),     # c5 at an eighth (NOT a BASELINE config: rehearsals of the hub-less N > 1 path)
}
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
FABRIC_GATHER_CEILING_GBPS = 8600.0   # MI355X_MICROARCH.md 'Indexed rows': random rows of an Infinity-Cache-resident table


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-epoch", action="store_true", help="skip the epoch-time measurement")
    p.add_argument("--epoch-matrix", action="store_true",
                   help="also time the epoch under the switches one by one (`epoch_matrix` in the record); the default record "
                        "carries the four figures the README quotes")
    p.add_argument("--no-verify", action="store_true",
                   help="N > 1: skip the comparison of the distributed SpMM pair with the single-device plan")
    p.add_argument("--cpu-sample-frac", type=float, default=1.0 / 32)
    p.add_argument("--cpu-ref", choices=["auto", "full", "sample"], default="auto",
                   help="CPU-ref baseline on the full operator (needs ~100 GB of host memory at c4), on a row sample, "
                        "or full when MemAvailable allows it (default)")
    p.add_argument("--no-hbm-activity", action="store_true", help="skip the live memory-controller measurement")
    p.add_argument("--no-live-traffic", action="store_true",
                   help="N = 1: do not measure the counter traffic in child runs under rocprofv3 (use profiles/traffic.json)")
    p.add_argument("--mode", choices=["both", "reference", "accurate"], default="both",
                   help="normalisation mode of the operator (pytextgcn_amd.plan): the headline `value` is ALWAYS the "
                        "reference-order mode's unless --mode accurate; `both` (default) additionally times the accurate "
                        "mode in the same run at N = 1 (`value_accurate_mode`)")
    p.add_argument("--launch-check", action="store_true",
                   help="rendezvous only: every rank joins the group, rank 0 prints {launch_check, n_gpus}; no GPU work")
    return p.parse_args()


def _run_ranks(cmd, env, budget_s):
    """One child `torch.distributed.run` in its own session (so that the whole tree -- launcher and ranks -- can be ended
    by its process-group id, the exact group started here).  Returns (return code or None on timeout, stdout text)."""
    import signal
    import subprocess
    import tempfile
    with tempfile.TemporaryFile() as out:
        proc = subprocess.Popen(cmd, stdout=out, env=env, start_new_session=True)      # stderr passes straight through
        try:
            rc = proc.wait(timeout=budget_s)
        except subprocess.TimeoutExpired:
            rc = None
        if rc is None or rc != 0:
            # over budget, or a rank died while its peers may still sit in a collective: end the group we started
            for sig in (signal.SIGTERM, signal.SIGKILL):
                try:
                    os.killpg(proc.pid, sig)
                except (ProcessLookupError, PermissionError):
                    break
                try:
                    proc.wait(timeout=10)
                    break
                except subprocess.TimeoutExpired:
                    continue
        out.seek(0)
        return rc, out.read().decode("utf-8", "replace")


def _last_record(text):
    record = None
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                record = json.loads(ln)
            except ValueError:
                pass
    return record


def launch_ranks(args) -> int:
    """`python3 bench.py --gpus N` (N > 1) without a rendezvous environment: start one rank per GPU as a CHILD
    `python -m torch.distributed.run` (never an exec, and before this process has made any GPU call -- it never
    makes one), relay the one JSON line rank 0 prints and hand back the child's return code.

    The measurement must not be lost to a form of the exchange that hangs or dies on this node: the child gets a wall
    budget (TGCN_BENCH_BUDGET_S, default 420 s).  If it exceeds it, exits non-zero or prints no record, its process group
    is ended and a FRESH child runs the plainest configuration -- RCCL's own collectives, A_r in one piece, no trial
    steps, no epoch, no parity check (TGCN_EXCHANGE=collective TGCN_RS_CHUNKS=1 --no-epoch --no-verify; budget TGCN_BENCH_FALLBACK_BUDGET_S, default
    300 s) -- and its record is relayed with a `"fallback"` field saying why.  Under torch.distributed.run (WORLD_SIZE
    set) this function is not reached."""
    import socket

    def free_port():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        return port

    def command(extra):
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] \
            + sys.argv[1:] + extra
    env = dict(os.environ)          # (HSA_ENABLE_IPC_MODE_LEGACY: decided by every rank itself, sharded.prepare_hsa_env)
    env.setdefault("OMP_NUM_THREADS", "4")
    budget = float(os.environ.get("TGCN_BENCH_BUDGET_S", "420"))
    t0 = time.time()
    # the ranks may try every form of the exchange: this process watches them and has a plain run to fall back on
    rc, text = _run_ranks(command([]), dict(env, TGCN_BENCH_WATCHDOG="1"), budget)
    record = _last_record(text)
    if rc == 0 and record is not None:
        print(json.dumps(record), flush=True)
        return 0
    reason = (f"first attempt exceeded its budget of {budget:.0f} s" if rc is None else
              f"first attempt exited with code {rc}" if rc != 0 else "first attempt printed no record")
    print(f"bench.py: {reason} after {time.time() - t0:.0f} s; running the plain configuration "
          "(TGCN_EXCHANGE=collective TGCN_RS_CHUNKS=1 --no-epoch --no-verify)", file=sys.stderr, flush=True)
    if os.environ.get("TGCN_BENCH_NO_FALLBACK") == "1":
        return rc if rc else 1
    env2 = dict(env, TGCN_EXCHANGE="collective", TGCN_RS_CHUNKS="1")
    env2.pop("TGCN_BENCH_TEST_FAIL", None)               # (the test hook below applies to the first attempt only)
    plain = [f for f in ("--no-epoch", "--no-verify") if f not in sys.argv]     # nothing but the measurement itself
    rc2, text2 = _run_ranks(command(plain), env2,
                            float(os.environ.get("TGCN_BENCH_FALLBACK_BUDGET_S", "300")))
    record = _last_record(text2)
    if rc2 == 0 and record is not None:
        record["fallback"] = reason
        print(json.dumps(record), flush=True)
        return 0
    print(f"bench.py: the fallback run failed too (return code {rc2})", file=sys.stderr, flush=True)
    return rc2 if rc2 else 1


def launch_check(world, rank):
    """--launch-check: the rendezvous and the record relay without any GPU work (CPU test of the N > 1 launch)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    # test hook of the fallback path (tests/test_host.py): the last rank of the FIRST attempt dies or hangs
    fail = os.environ.get("TGCN_BENCH_TEST_FAIL")
    if fail and rank == world - 1:
        if fail == "exit":
            os._exit(1)
        time.sleep(3600)
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    ok = t.item() == world * (world - 1) / 2
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": bool(ok), "n_gpus": world}), flush=True)
    return 0 if ok else 1


def init_group(dist, backend, dev, **kw):
    """The process group of the bench: pytextgcn_amd.sharded.init_process_group (RCCL on a HIGH-PRIORITY stream, so that
    the exchange is scheduled while the SpMM grids fill the CUs) with a bound on every collective -- one that does not
    complete within TGCN_BENCH_COLLECTIVE_TIMEOUT_S (default 180 s; a rank died, a form of the exchange hung) ends the
    process instead of waiting for ever, so that the parent (launch_ranks) can run its fallback."""
    from pytextgcn_amd.sharded import init_process_group
    init_process_group(backend, dev, timeout_s=float(os.environ.get("TGCN_BENCH_COLLECTIVE_TIMEOUT_S", "180")), **kw)


def rccl_info(dist, backend, world, rank, local_rank, dev):
    """Who took part: per rank the device index, its name, PCI bus id and UUID -- so that a reader of the record can
    tell N GPUs from N ranks on fewer (rehearsals put every gloo rank on one card)."""
    pr = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "local_rank": local_rank, "device": dev.index, "name": pr.name,
            "pci_bus_id": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0),
                                              getattr(pr, "pci_device_id", 0)),
            "uuid": str(getattr(pr, "uuid", ""))}
    info = [mine]
    if dist is not None and world > 1:
        info = [None] * world
        dist.all_gather_object(info, mine)
    hi = None
    if dist is not None and backend == "nccl":
        try:
            hi = bool(dist.distributed_c10d._get_default_group()._get_backend(dev).options.is_high_priority_stream)
        except Exception:                   # noqa: BLE001 - introspection only
            hi = None
    return {"backend": {"nccl": "nccl (RCCL)"}.get(backend, backend) if dist is not None else None, "ranks": world,
            "distinct_devices": len({(d["pci_bus_id"], d["uuid"]) for d in info}),
            "high_priority_stream": hi, "devices": info}


def distributed_parity(sg, g, N, F, x_local, gout_local, bias, dev, dist, mode):
    """The timed distributed SpMM pair against the SINGLE-DEVICE plan of the same graph, inside the same run: every rank
    builds the whole-graph plan (the graph is on every rank), gathers the operands, and compares ITS rows of `M @ X + b`
    and `M^T @ G` -- so a scaling record carries the evidence that all N ranks computed what one GPU computes (the sums
    associate differently over ranks: max-norm relative error, tolerance 1e-5, not bits).  Collective; never raises: a
    failure that hits every rank alike is reported in the record (decisions are all-reduced); a DEVICE error, or a failure
    on one rank alone (its peers are inside a collective then), is raised to the caller, whose guard (`secondary` in
    main) ends the run instead of pairing later collectives with the wrong calls."""
    from pytextgcn_amd.plan import GraphPlan

    def agree(ok):
        f = torch.tensor([1.0 if ok else 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(f, op=dist.ReduceOp.MIN, group=sg.group)
        return f.item() > 0
    t0 = time.perf_counter()
    need = 4 * N * F * 4 + 24 * (g.edge_index.size(1) + N) * 2        # two operands, two results, a plan with M^T
    free = torch.cuda.mem_get_info(dev)[0]
    if not agree(free > 1.5 * need):
        return {"skipped": f"needs about {need / 1e9:.1f} GB free per device"}
    out = {"tolerance": 1e-5, "normalisation_mode": mode,
           "against": "single-device GraphPlan of the same edge list and mode, built in this run on every rank"}
    # the one step that can fail on one rank alone without a collective around it: agree on it before going on
    plan, err = None, None
    try:
        plan = GraphPlan(g.edge_index, g.edge_attr, N, degree_sum=mode)
    except Exception as e:                 # noqa: BLE001 - classified by the caller when it is a device error
        if is_device_error(e):
            raise
        err = f"{type(e).__name__}: {e}"[:300]
    if not agree(err is None):
        out["error"] = err or "the single-device plan could not be built on another rank"
        return out
    errs = []
    for name, loc, b, tr in (("forward", x_local, bias, False), ("transposed", gout_local, None, True)):
        full = sg.gather_rows(loc)                                   # collective
        ref = sg.scatter_rows(plan.spmm(full, b, transpose=tr))
        mine = sg.spmm(loc, b, transpose=tr)                         # collective
        scale = float(ref.abs().max().item()) or 1.0
        e = float((mine[sg.real] - ref[sg.real]).abs().max().item()) / scale if bool(sg.real.any()) else 0.0
        t = torch.tensor([e], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=sg.group)
        out[f"max_rel_err_{name}"] = t.item()
        errs.append(t.item())
        del full, ref, mine
    out["ok"] = bool(max(errs) < 1e-5)
    out["rows_checked"] = N
    # the SAME step on ONE device, timed here on this rank's GPU with the whole-graph plan that was just checked against
    # (every rank times its own card at the same time; the slowest is reported): the denominator of `scaling_model`
    xs = torch.randn(N, F, device=dev)
    ys = torch.empty(N, F, device=dev)
    for _ in range(2):
        plan.spmm(xs, bias, out=ys), plan.spmm(xs, None, transpose=True, out=ys)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        plan.spmm(xs, bias, out=ys), plan.spmm(xs, None, transpose=True, out=ys)
    ev[1].record()
    torch.cuda.synchronize()
    t1 = torch.tensor([ev[0].elapsed_time(ev[1]) / 5], device=dev, dtype=torch.float64)
    dist.all_reduce(t1, op=dist.ReduceOp.MAX, group=sg.group)
    out["single_device_ms_per_step"] = t1.item()
    del xs, ys
    out["seconds"] = round(time.perf_counter() - t0, 2)
    return out


XGMI_LINK_GBPS = 153.0      # per link and direction (task statement / MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU)


def scaling_model(world, ms_per_step, parity, diag, sg, F):
    """What the link arithmetic allows, next to what was measured -- so that a scaling line can be read without opening
    DESIGN.md.  All times are per bench step (forward + transposed SpMM at width F) on this node:
      compute    the rank's local operators alone (A_r and B_r, both directions; max over ranks);
      exchange   the two collectives of each SpMM alone (all-gather + reduce-scatter of the hub block, measured here);
      hidden / exposed   the step time if the exchange overlapped the compute completely / not at all."""
    if not diag or not parity or "single_device_ms_per_step" not in parity:
        return None

    def ms(key):
        v = diag.get(key)
        return v.get("max_ms") if isinstance(v, dict) else None
    a, b = ms("local_A_hub_rows_x_own_regulars") or 0.0, ms("local_B_own_rows")
    ag, rs = ms("all_gather_into_tensor"), ms("reduce_scatter_tensor")
    if b is None or ag is None or rs is None:
        return None
    t1 = parity["single_device_ms_per_step"]
    has_rs = sg.rp > 0
    compute = 2.0 * (a + b)
    exchange = 2.0 * (ag + (rs if has_rs else 0.0))
    link_bytes = sg.hp * F * 4                       # what one rank sends to ONE peer in one collective (its own link)
    link_floor = 2.0 * (2 if has_rs else 1) * link_bytes / (XGMI_LINK_GBPS * 1e9) * 1e3
    pipe = diag.get("pipeline")
    if pipe is not None:
        # no hub structure, pipelined exchange: the step is the compute side (own-column block + accumulated stage blocks)
        # plus whatever of the exchange side does not hide under it; the halo rows of a rank arrive over W - 1 links
        comp = (pipe["compute_side_own_plus_stage_blocks"] or {}).get("max_ms")
        exch = (pipe["exchange_side_packs_plus_stages"] or {}).get("max_ms")
        whole = ms("whole_spmm_overlapped")
        rows_in = sum(pipe["stage_rows_received"])
        link_ms = rows_in * F * 4 / ((sg.world - 1) * XGMI_LINK_GBPS * 1e9) * 1e3
        last_block = None
        if comp is not None and pipe["stage_block_entries"]:
            total = pipe["own_block_entries"] + sum(pipe["stage_block_entries"])
            last_block = comp * pipe["stage_block_entries"][-1] / max(1, total)
        return {
            "single_device_ms_per_step": t1, "ms_per_step_measured": ms_per_step, "speedup_measured": t1 / ms_per_step,
            "form": f"pipeline-{pipe['scheme']}/{pipe['stages']}",
            "per_spmm_ms": {"compute_side_own_plus_stage_blocks": comp, "exchange_side_packs_plus_stages_alone": exch,
                            "whole_overlapped": whole, "B_r_as_one_operator": b, "whole_operand_all_gather_alone": ag,
                            "halo_gather_alone": ms("halo_gather_referenced_rows_only"),
                            "exchange_at_link_peak": link_ms, "last_stage_block_estimate": last_block},
            "exposed_exchange_ms_per_spmm": (whole - comp) if (whole is not None and comp is not None) else None,
            "compute_scaling": t1 / (2.0 * comp) if comp else None,
            "ms_per_step_if_exchange_fully_hidden": 2.0 * max(comp or 0.0, exch or 0.0),
            "ms_per_step_if_exchange_fully_exposed": 2.0 * ((comp or 0.0) + (exch or 0.0)),
            "ms_per_step_halo_form_exposed": 2.0 * ((b or 0.0) + (ms("halo_gather_referenced_rows_only") or 0.0)),
            "link_GBps_assumed": XGMI_LINK_GBPS,
            "note": "no hub structure: nothing but the rank's own-column block can run before operand rows arrive, so the "
                    "halo rows travel in stages and each stage's column block is added (tgcn_spmm_acc) while the next "
                    "stage is in flight; the exposed exchange is the first stage's head start minus the own block, plus "
                    "whatever the links cannot deliver under the blocks (exposed_exchange_ms_per_spmm)"}
    return {
        "per_spmm_ms": {"A_r": a, "B_r": b, "all_gather_alone": ag, "reduce_scatter_alone": rs if has_rs else None,
                        "whole_overlapped": ms("whole_spmm_overlapped")},
        "single_device_ms_per_step": t1, "ms_per_step_measured": ms_per_step, "speedup_measured": t1 / ms_per_step,
        "compute_ms_per_step": compute, "compute_scaling": t1 / compute if compute else None,
        "exchange_ms_per_step_measured_alone": exchange,
        "ms_per_step_if_exchange_fully_hidden": max(compute, exchange),
        "ms_per_step_if_exchange_fully_exposed": compute + exchange,
        "speedup_if_exchange_fully_hidden": t1 / max(compute, exchange),
        "speedup_if_exchange_fully_exposed": t1 / (compute + exchange),
        "collectives_per_step": 2 * (2 if has_rs else 1),
        "bytes_per_link_and_collective": link_bytes,
        "link_GBps_assumed": XGMI_LINK_GBPS,
        "exchange_ms_per_step_at_link_peak": link_floor,
        "speedup_if_exchange_at_link_peak_and_exposed": t1 / (compute + link_floor),
        "note": "a 1-D row partition with the word block replicated exchanges 2 x |hubs| x F floats per SpMM whatever the "
                "number of ranks, while the compute shrinks with 1 / W: the SpMM-pair metric is exchange-bound at 8 ranks "
                "unless the collectives hide behind the local operators completely (north_star's >= 6x needs that); the "
                "training step additionally has the narrow exchange (exchange_floats_per_hub_row_and_step)"}


_DEVICE_ERROR_WORDS = ("hip error", "hiperror", "hsa_status", "hsa error", "memory access fault", "illegal memory access",
                       "device-side assert", "unspecified launch failure", "hardware exception", "gpu hang")


def is_device_error(e) -> bool:
    """A failure of the device / its runtime (the HIP context may be poisoned), as opposed to a Python-level error:
    torch's AcceleratorError, or a RuntimeError whose text names HIP / HSA / a fault.  An out-of-memory answer of the
    allocator is NOT one (the context is intact; torch.OutOfMemoryError)."""
    if isinstance(e, getattr(torch, "OutOfMemoryError", ())):
        return False
    if isinstance(e, getattr(torch, "AcceleratorError", ())):
        return True
    text = str(e).lower()
    return isinstance(e, RuntimeError) and any(w in text for w in _DEVICE_ERROR_WORDS)


_T0 = time.time()


def note(msg, rank=0):
    """A progress line on STDERR (stdout carries the one JSON line): long configurations (c5: minutes of host-side work in
    the CPU legs and the counter passes) must not look hung to whoever watches the run."""
    if rank == 0:
        print(f"bench.py [{time.time() - _T0:6.1f} s] {msg}", file=sys.stderr, flush=True)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def mem_available_gb():
    try:
        with open("/proc/meminfo") as f:
            for ln in f:
                if ln.startswith("MemAvailable:"):
                    return int(ln.split()[1]) / 1e6           # kB -> GB
    except (OSError, ValueError, IndexError):
        pass
    return None


def cpu_baseline(plan, F, frac, E, seed=44, warm=3, reps=5, full="auto", ref_budget_s=150.0):
    """The two CPU rows of BASELINE.md section 3, on this box's host cores, same graph / seed / fp32, median of the timed
    repetitions:
      CPU-ref (`value`)  the reference formulation (PyG-1.6.3 index_select -> scale -> scatter_add,
                         oracle/gcn_oracle.py `propagate`), forward and transposed.  It materialises two nnz x F
                         temporaries (41.6 GB each at c4), so: when the host has the memory (MemAvailable >= 200 GB, or
                         `full=True`) it runs on the FULL operator -- the same workload the GPU is timed on -- with 1
                         warm-up pair and up to 3 timed pairs inside `ref_budget_s` seconds (at least one); otherwise
                         on the sub-operator whose TARGET rows are a random `frac` of the nodes (3 + 5 repetitions),
                         and `sample` says which and why;
      CPU-csr (`csr`)    the C / OpenMP CSR restatement (oracle/csr_spmm.c, `oracle_csr_spmm_f32`) of M @ X and
                         M^T @ G on the FULL operator, `warm` + `reps` repetitions."""
    from oracle import csr_oracle, gcn_oracle as O
    N = plan.num_nodes
    gen = torch.Generator().manual_seed(seed)
    rp, col, val = (t.cpu() for t in plan.export_csr())
    deg = (rp[1:] - rp[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(N), deg)
    avail = mem_available_gb()
    need_gb = 2.2 * float(plan.nnz) * F * 4 / 1e9 + 20.0          # two nnz x F temporaries + operands, with margin
    run_full = (full is True) or (full == "auto" and avail is not None and avail >= max(200.0, 1.5 * need_gb))
    if run_full:
        tgt, src, w = rows, col.long(), val
        why = (f"the FULL operator (MemAvailable {avail:.0f} GB >= the {need_gb:.0f} GB the two nnz x F temporaries need)"
               if avail is not None else "the FULL operator")
        warm_ref, reps_ref = 1, 3
    else:
        pick = torch.rand(N, generator=gen) < frac
        sel = pick[rows]
        tgt, src, w = rows[sel], col[sel].long(), val[sel]
        del sel
        why = (f"the rows of a random {frac:.4f} of the nodes of the same graph (MemAvailable "
               f"{'unknown' if avail is None else '%.0f GB' % avail} < the {max(200.0, 1.5 * need_gb):.0f} GB bound for the full "
               f"operator: it materialises two nnz x F temporaries)")
        warm_ref, reps_ref = warm, reps
    del rows
    n_edges = int((tgt != src).sum())                       # self-loops are not graph edges
    x = torch.randn(N, F, generator=gen)
    fwd = torch.stack([src, tgt])                           # out[tgt] += w * x[src]
    bwd = torch.stack([tgt, src])                           # dxw[src] += w * g[tgt]
    times = []
    t_leg = time.perf_counter()
    for rep in range(warm_ref + reps_ref):
        t0 = time.perf_counter()
        O.propagate(fwd, x, w, N)
        O.propagate(bwd, x, w, N)
        if time.perf_counter() - t0 > 20.0:
            note(f"CPU-ref repetition {rep + 1} of at most {warm_ref + reps_ref}")
        if rep >= warm_ref:
            times.append(time.perf_counter() - t0)
            # the leg is BOUNDED in both forms: the full operator by `ref_budget_s`, the sample by a third of it (at c5 one
            # pair over the 1 / 32 sample is ~90 s of host work -- the [N, F] result alone is 8 GB -- so it is timed once)
            if time.perf_counter() - t_leg > (ref_budget_s if run_full else ref_budget_s / 3.0):
                break
        elif not run_full and time.perf_counter() - t_leg > ref_budget_s / 3.0:
            warm_ref = rep + 1                              # a warm-up pass that long: the next pass is the timed one
    t_ref = sorted(times)[len(times) // 2]
    n_ref_reps = len(times)
    n_sel = int(w.numel())
    del fwd, bwd, tgt, src, w
    # CPU-csr at full size (the operator is symmetric here; otherwise the transposed CSR is exported)
    rp64 = rp.long()
    if plan.symmetric:
        rpt, colt, valt = rp64, col, val
    else:
        rpt, colt, valt = (t.cpu() for t in plan.export_csr(transpose=True))
        rpt = rpt.long()
    lib = csr_oracle.lib()
    y = torch.empty(N, F)

    def csr(rp_, col_, val_):
        lib.oracle_csr_spmm_f32(N, rp_.data_ptr(), col_.data_ptr(), val_.data_ptr(), x.data_ptr(), x.stride(0), F,
                                None, y.data_ptr(), y.stride(0))
    note(f"CPU-ref done ({t_ref:.2f} s per pair); CPU-csr on the full operator")
    times = []
    for rep in range(warm + reps):
        t0 = time.perf_counter()
        csr(rp64, col, val)
        csr(rpt, colt, valt)
        if rep >= warm:
            times.append(time.perf_counter() - t0)
        if time.perf_counter() - t0 > 20.0:
            note(f"CPU-csr repetition {rep + 1} of {warm + reps}")
    t_csr = sorted(times)[len(times) // 2]
    threads = torch.get_num_threads()
    return {"value": 2.0 * n_edges / t_ref, "unit": "edges/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(), "full_operator": bool(run_full),
            "mem_available_GB": avail,
            "sample": f"CPU-ref: reference formulation (gather->scale->index_add, fwd+bwd) on {why}: {n_edges} edges, "
                      f"{n_sel} non-zeros, F={F}, {warm_ref} warm-up + {n_ref_reps} timed reps, median {t_ref:.2f} s per "
                      f"fwd+bwd pair, torch threads {threads}",
            "csr": {"value": 2.0 * E / t_csr, "unit": "edges/s", "kind": "port",
                    "cores": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)),
                    "sample": f"CPU-csr: oracle_csr_spmm_f32 (C, OpenMP) M@X and M^T@G on the FULL operator, "
                              f"{int(plan.nnz)} non-zeros, F={F}, {warm} warm-up + {reps} timed reps, median "
                              f"{t_csr:.3f} s per fwd+bwd pair"}}


def device_copy_gbps(dev, n_bytes=1 << 30, reps=10):
    """Secondary roofline denominator: what this box's HBM delivers to a plain
    device-to-device copy, read + written bytes per second."""
    a = torch.empty(n_bytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        b.copy_(a)
    ev[1].record()
    torch.cuda.synchronize()
    return 2.0 * n_bytes * reps / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9


SPMM_KERNEL_SOURCES = ("pytextgcn_amd/csrc/spmm.hip", "pytextgcn_amd/csrc/plan.hip", "pytextgcn_amd/csrc/common.h")


def spmm_kernel_sha16():
    """Fingerprint of the sources that decide what one tgcn_spmm launch moves: the counter figures of
    profiles/traffic.json are valid for the library they were collected on and no other."""
    import hashlib
    h = hashlib.sha256()
    for rel in SPMM_KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def fabric_traffic(config, n_gpus):
    """L2 <-> fabric bytes per SpMM launch from the committed rocprofv3 PMC passes (profiles/traffic.json:
    2*FETCH_SIZE + WRITE_SIZE, collected and corrected as MI355X_MICROARCH.md 'HBM' prescribes).  Returns
    (bytes, provenance, fresh) or (None, None, False).  NOT measured by this run: a constant from the named profile;
    `fresh` says whether it was collected on THESE kernel sources (its `kernel_sha16` against spmm_kernel_sha16())."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f).get(f"{config}_n{n_gpus}", {})
        if "bytes_per_launch" in rec:
            fresh = rec.get("kernel_sha16") == spmm_kernel_sha16()
            src = f"profiles/traffic.json[{config}_n{n_gpus}] ({rec.get('round', '?')})"
            if n_gpus > 1:        # N > 1: the counters of ONE rank's local operators, taken on one GPU (tools/prof_local_step.py)
                src += ": rank 0's local operators of this partition measured on one GPU; the exchange's bytes are not in it"
            return rec["bytes_per_launch"], src, fresh
    except (OSError, ValueError):
        pass
    return None, None, False


def live_fabric_traffic(config, mode, timeout_s=90):
    """The counter traffic of THIS library on THIS box, measured inside the run: two short child runs of this very bench
    command (3 steps, no epoch, no CPU legs) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, only
    --kernel-trace beside them, the program itself after `--`), summarised exactly as profiles/summarize.py summarises the
    committed passes: per tgcn_spmm launch = sum over its kernels of the mean (2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024.
    Returns (bytes, note) or (None, reason); never raises, bounded by `timeout_s` per pass.  N = 1 only, rank 0, outside
    every timed region."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 is not on PATH"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    means = {}
    t0 = time.perf_counter()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="tgcn_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
               "--no-epoch", "--no-hbm-activity", "--no-live-traffic", "--mode", mode]
        try:
            res = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                 timeout=timeout_s)
            if res.returncode != 0:
                return None, f"rocprofv3 --pmc {ctr} exited with {res.returncode}: {res.stderr.decode()[-200:]}"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, f"rocprofv3 --pmc {ctr} left no counter file"
            per = {}
            with open(files[0]) as f:
                for r in csv.DictReader(f):
                    if r["Counter_Name"] == ctr and "k_spmm" in r["Kernel_Name"]:
                        name = r["Kernel_Name"].split("k_spmm_", 1)[1].split("<")[0].split("(")[0]
                        per.setdefault(name, []).append(float(r["Counter_Value"]))
            if not per:
                return None, f"no k_spmm_* dispatch in the {ctr} pass"
            means[ctr] = {k: sum(v) / len(v) for k, v in per.items()}
        except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
            return None, f"{type(e).__name__} in the {ctr} pass: {e}"[:240]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    kernels = sorted(set(means["FETCH_SIZE"]) | set(means["WRITE_SIZE"]))
    total = sum(2.0 * means["FETCH_SIZE"].get(k, 0.0) + means["WRITE_SIZE"].get(k, 0.0) for k in kernels) * 1024.0
    note = ("live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over two child runs of this bench command on this "
            f"box (3 steps each, {time.perf_counter() - t0:.0f} s); per tgcn_spmm launch = sum over its kernels "
            f"({', '.join('k_spmm_' + k for k in kernels)}) of the mean (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024")
    return total, note


def profile_constant(key):
    """A number recorded under profiles/ (e.g. the L2-resident ceiling of tools/ceiling_spmm.py)."""
    try:
        with open(os.path.join(ROOT, "profiles", "constants.json")) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def hbm_activity(step_fn, dev, seconds=1.5):
    """What this run's SpMM actually pulls from HBM, measured live: the memory controllers' busy percentage
    (sysfs `mem_busy_percent`, sampled at 100 Hz while `step_fn` loops) calibrated against a 1 GiB device copy
    whose HBM bytes are known (tools/hbm_activity.py; round-2 profile profiles/r02_hbm_activity.md shows the
    percentage is linear in bytes/s for copy / read-only / write-only streams).  Outside the timed region.
    Returns None when the box does not expose the counter."""
    try:
        from tools import hbm_activity as H
        sysdir = H.find_sysfs(dev.index or 0)
        if sysdir is None:
            return None
        n = 1 << 28
        a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
        b = torch.empty_like(a)
        quiet = open(os.devnull, "w")
        old = sys.stdout
        sys.stdout = quiet                        # H.measure prints its records
        try:
            idle = H.measure("idle", lambda: None, sysdir, 0.5)
            cal = H.measure("copy", lambda: b.copy_(a), sysdir, seconds, bytes_known=2 * n * 4)
            del a, b
            run = H.measure("step", step_fn, sysdir, seconds)
        finally:
            sys.stdout = old
            quiet.close()
        base = idle["mem_busy_percent_median"] or 0.0
        if not cal["mem_busy_percent_median"] or run["mem_busy_percent_median"] is None:
            return None
        gbps_per_pct = cal["known_hbm_GBps"] / max(cal["mem_busy_percent_median"] - base, 1e-9)
        gbps = (run["mem_busy_percent_median"] - base) * gbps_per_pct
        return {"hbm_GBps": gbps, "ms_per_step": run["ms_per_launch"],
                "bytes_per_step": gbps * 1e9 * run["ms_per_launch"] * 1e-3,
                "mem_busy_percent": run["mem_busy_percent_median"],
                "calibration": {"copy_GBps": cal["known_hbm_GBps"], "copy_mem_busy_percent": cal["mem_busy_percent_median"]}}
    except Exception as e:                        # noqa: BLE001 - a diagnostic must never sink the bench line
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def epoch_time_ms(g, F, n_classes, fused, reps=5, reuse=False, collapse=False, fuse_w1=False, split_gemms=False,
                  needed_rows=False):
    """One epoch as flat_amazon.py:99-117 defines it: train step (fwd, CE on train_mask, zero_grad,
    bwd, Adam(amsgrad) step) + eval forward + validation loss + metric transfer to the host.
    fused=False: the reference's loop body with its own operators (torch CrossEntropyLoss on
    mask-indexed rows, torch.optim.Adam) around pytextgcn_amd.GCN;  fused=True: the same steps with
    pytextgcn_amd.functional.masked_cross_entropy, pytextgcn_amd.optim.Adam and the dropout between the
    layers fused into the layer-2 GEMMs (pytextgcn_amd.enable_fused_dropout).
    "Metric transfer" (BASELINE.md section 2): the arg-max of the masked logits is taken on the
    device and the PREDICTIONS go to the host (fused loop: the rows of both masks gathered through row lists
    taken once, as the narrowest integer type that holds a class id, into pinned memory, with the two loss values: one
    synchronisation per epoch); the reference ships the masked logits themselves
    (flat_amazon.py:111-112) and runs numpy / sklearn on them, which is host work outside this path."""
    import pytextgcn_amd as pkg
    from pytextgcn_amd.functional import masked_cross_entropy
    N = g.y.numel()
    pkg.enable_activation_reuse(reuse)
    pkg.enable_linear_collapse(collapse)
    pkg.enable_fused_dropout(fused)          # dropout fused into the next layer's GEMMs (tgcn_gemm_*_dropout)
    from pytextgcn_amd import dense as _dense
    _dense.enable_split_gemms(split_gemms)   # opt-in numerical mode of the layer-2 products (fp32-accurate, not bit-equal)
    model = pkg.GCN(N, n_classes, n_hidden_gcn=F, dropout=0.5).to(g.y.device).float()
    Opt = pkg.optim.Adam if fused else torch.optim.Adam
    opt = Opt(model.parameters(), lr=0.05, amsgrad=True)
    if fuse_w1:
        # dW1 = M^T dH1 is consumed row by row by Adam inside the backward SpMM (tgcn_spmm_adam): same bits,
        # no N x h gradient, no separate optimizer pass over W1
        opt.fuse_into_backward(model.layers[0].weight)
    crit = torch.nn.CrossEntropyLoss(reduction="mean")
    times = []
    host = None
    # needed_rows (opt-in, `GCN.forward(g, rows=...)`): the last layer computes only the logits rows the loop reads -- the
    # training rows in the training step (flat_amazon.py:101), the validation and training rows in evaluation (:109-114)
    rows_train = g.train_mask if needed_rows else None
    rows_eval = (g.val_mask | g.train_mask) if needed_rows else None
    for rep in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.train()
        if fused:
            loss = masked_cross_entropy(model(g, rows=rows_train) if needed_rows else model(g), g.y, g.train_mask)
        else:
            out = model(g)[g.train_mask]
            loss = crit(out, g.y[g.train_mask])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        model.eval()
        with torch.no_grad():
            logits = model(g, rows=rows_eval) if needed_rows else model(g)
            if fused:
                # validation loss and the arg-max of every row in one pass (tgcn_masked_ce_pred); the class ids
                # cross PCIe as int32 into pinned buffers, all transfers of the epoch behind ONE synchronisation
                val_loss, pred = masked_cross_entropy(logits, g.y, g.val_mask, return_pred=True)
                if host is None:
                    # the masks are static (text2graph.py:180-191): their row lists are taken once, so that the
                    # selection inside the epoch is a gather (boolean indexing synchronises to size its result)
                    n_val = int(g.val_mask.sum().item())
                    rows_sel = torch.cat([g.val_mask.nonzero().flatten(), g.train_mask.nonzero().flatten()])
                    # a class id crosses PCIe in the narrowest integer that holds it (64 classes: one byte per row --
                    # 1.6 MB instead of 6.5 MB at c4, ~0.09 ms of the epoch's tail; numpy / sklearn take any integer labels)
                    pred_dtype = torch.uint8 if n_classes <= 256 else torch.int16 if n_classes <= 32767 else torch.int32
                    host = (n_val, rows_sel, torch.empty(rows_sel.numel(), dtype=pred_dtype).pin_memory(),
                            torch.empty(2, dtype=torch.float32).pin_memory())
                host[2].copy_(pred.index_select(0, host[1]).to(host[2].dtype), non_blocking=True)
                host[3].copy_(torch.stack([loss.detach(), val_loss]), non_blocking=True)
                torch.cuda.current_stream().synchronize()
                pred_val, pred_train = host[2][:host[0]].numpy(), host[2][host[0]:].numpy()
                host[3][0].item()
            else:
                crit(logits[g.val_mask], g.y[g.val_mask])
                pred_val = logits[g.val_mask].argmax(1).cpu().numpy()
                pred_train = logits[g.train_mask].argmax(1).cpu().numpy()
                loss.item()
        torch.cuda.synchronize()
        if rep:
            times.append((time.perf_counter() - t0) * 1e3)
    del model, opt
    pkg.enable_activation_reuse(False)
    pkg.enable_linear_collapse(False)
    pkg.enable_fused_dropout(False)
    _dense.enable_split_gemms(False)
    return sorted(times)[len(times) // 2]


def flat_loop_epoch_ms(g, F, n_classes, reps=5):
    """The same epoch through the package's own loop object, `pytextgcn_amd.train.FlatLoop` (every switch that leaves the
    numbers alone + only the rows that are read): what a user gets from three lines, timed like `epoch_time_ms`."""
    import pytextgcn_amd as pkg
    from pytextgcn_amd.train import FlatLoop
    N = g.y.numel()
    model = pkg.GCN(N, n_classes, n_hidden_gcn=F, dropout=0.5).to(g.y.device).float()
    times = []
    with FlatLoop(model, g, lr=0.05) as loop:
        for rep in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop.epoch()
            torch.cuda.synchronize()
            if rep:
                times.append((time.perf_counter() - t0) * 1e3)
    del model, loop
    return sorted(times)[len(times) // 2]


SHARDED_EPOCH_LOSS = {}       # variant -> global training loss after the timed epochs (same weights, same masks: must agree)


def sharded_epoch_ms(sg, N, F, n_classes, dev, dist, reps=3, reuse=False, fuse_w1=False, narrow=False, rows=False):
    """The epoch of flat_amazon.py:99-117 on the row-partitioned model: every rank owns its rows of
    W1 / H1 / logits and of the Adam state; fused loss and optimizer kernels; small dense gradients
    summed with one all-reduce; predictions of the owned rows go to the host."""
    import pytextgcn_amd as pkg
    from pytextgcn_amd.sharded import ShardedGCN, sharded_cross_entropy
    gen = torch.Generator(device=dev).manual_seed(7)
    y_full = torch.randint(0, n_classes, (N,), device=dev, generator=gen)
    u = torch.rand(N, device=dev, generator=gen)
    is_doc = ~sg.part.hub_mask if not bool(sg.part.hub_mask.all()) else torch.ones_like(sg.part.hub_mask)
    y_l = sg.scatter_rows(y_full)
    train_l = sg.scatter_rows(is_doc & (u < 0.8))
    val_l = sg.scatter_rows(is_doc & (u >= 0.8) & (u < 0.9))
    del y_full, u
    # rows: the caller names the rows it reads (training rows in the step, validation + training rows in evaluation); no
    # hub row is among them, so the last propagate step loses one collective each way (ShardedGraph.rows_view)
    rows_train = train_l if rows else None
    rows_eval = (train_l | val_l) if rows else None
    if rows:
        sg.prepare_rows(rows_train), sg.prepare_rows(rows_eval)     # collective, once: forward(rows=...) only looks them up
    pkg.enable_activation_reuse(reuse)
    # narrow: hub rows cross the links at the class width where the activation-free network allows (pytextgcn_amd/narrow.py).
    # Every variant starts from the same weights and draws the same KEYED dropout masks (a function of a node's place in
    # the partition and a seed common to the group), so the losses the variants reach are comparable: SHARDED_EPOCH_LOSS
    torch.manual_seed(4321)
    model = ShardedGCN(sg, N, n_classes, n_hidden_gcn=F, dropout=0.5, narrow_exchange=narrow,
                       keyed_dropout=True if sg.rp > 0 else None).to(dev)
    with torch.no_grad():
        model.weights[0].uniform_(-0.0017, 0.0017)            # glorot bound of an N x h matrix
    opt = pkg.optim.Adam(model.parameters(), lr=0.05, amsgrad=True)
    if fuse_w1:
        # the regular rows of the rank's W1 shard are updated inside the backward SpMM, the hub slice after its
        # reduce-scatter (ShardedGraph.spmm_adam_w1): no [n_local, h] gradient, no optimizer pass over those rows
        opt.fuse_into_backward(model.weights[0])
    times = []
    for rep in range(reps + 1):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.train()
        loss = sharded_cross_entropy(sg, model(rows=rows_train), y_l, train_l)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        model.sync_grads()
        opt.step()
        model.eval()
        with torch.no_grad():
            logits = model(rows=rows_eval)
            _, pred = sharded_cross_entropy(sg, logits, y_l, val_l, return_pred=True)
            pred_val = pred[val_l].cpu().numpy()
            pred_train = pred[train_l].cpu().numpy()
        loss.item()
        dist.barrier()
        torch.cuda.synchronize()
        if rep:
            times.append((time.perf_counter() - t0) * 1e3)
    pkg.enable_activation_reuse(False)
    del model, opt
    t = torch.tensor([sorted(times)[len(times) // 2]], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    total = loss.detach().double().reshape(1).clone()         # the ranks' shares of the last training loss
    dist.all_reduce(total)
    SHARDED_EPOCH_LOSS["plain" + ("+reuse" if reuse else "") + ("+w1" if fuse_w1 else "") + ("+narrow" if narrow else "")
                       + ("+rows" if rows else "")] = total.item()
    return t.item()


def exchange_diagnostics(sg, F, dev, dist, reps=10):
    """N > 1 only, outside the timed region: where one distributed SpMM spends its time on THIS node.
    Every phase is timed on its own (barrier + sync on both sides, max over ranks): the two local
    operators, the two RCCL collectives, the halo form's gather of the referenced rows, and the overlapped whole
    in the form the bench chose.  A phase that raises is reported as an error string; nothing here feeds `value`."""
    A, B = sg.ops[0]
    W, hp, rp = sg.world, sg.hp, sg.rp
    gen = torch.Generator(device=dev).manual_seed(99)
    x = torch.randn(sg.n_local, F, device=dev, generator=gen)
    xbuf = torch.empty(W * hp, F, device=dev)
    partial = torch.randn(W * hp, F, device=dev, generator=gen)
    rs_out = torch.empty(hp, F, device=dev)

    def phase(fn):
        try:
            for _ in range(2):
                fn()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            dt = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], device=dev, dtype=torch.float64)
            lo = dt.clone()
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            return {"max_ms": round(dt.item(), 4), "min_ms": round(lo.item(), 4)}
        except Exception as e:                       # noqa: BLE001
            # one rank: a failed phase is a line in the record.  Several ranks: the phase holds collectives, the peers
            # may be inside one -- the caller's guard (`secondary`) ends the run; a device error is never swallowed
            if W > 1 or is_device_error(e):
                raise
            return {"error": f"{type(e).__name__}: {e}"[:200]}

    out = {"bytes_each_way_per_collective": int((W - 1) * hp * F * 4)}
    if A is not None:
        out["local_A_hub_rows_x_own_regulars"] = phase(lambda: A.spmm(x[hp:]))
    out["local_B_own_rows"] = phase(lambda: B.spmm(xbuf, None, x2=x[hp:] if rp > 0 else None))
    out["all_gather_into_tensor"] = phase(lambda: dist.all_gather_into_tensor(xbuf, x[:hp].contiguous()))
    out["reduce_scatter_tensor"] = phase(lambda: dist.reduce_scatter_tensor(rs_out, partial))
    d0 = sg.dirs[0]
    direct = sg._stream_ordered(x)

    def halo_gather():
        pack = sg._rows_gather(x[:hp], d0.send_slots)
        recv, work = sg._all_to_all_v(pack, d0.need_counts_l, d0.send_counts_l, direct)
        work.wait()
        sg._rows_scatter(xbuf, d0.need_cols, recv)
    out["halo_gather_referenced_rows_only"] = phase(halo_gather)
    out["rows_received_per_spmm"] = sg.exchange_rows()
    if rp == 0 and sg.exchange == "pipeline":
        # the pipelined exchange of a graph without hub structure, phase by phase: the compute side alone (own-column block
        # + the stage blocks added in sequence, operands standing in), the exchange side alone (packs + the K all-to-all
        # stages, nothing computed), and -- below, `whole_spmm_overlapped` -- the two together as `sg.spmm` runs them
        pipe = sg._pipeline(d0)
        bufs = [torch.randn(sum(st.recv_counts), F, device=dev, generator=gen) for st in pipe.stages]

        def compute_side():
            yy = pipe.own.spmm(x, None)
            for st, b_ in zip(pipe.stages, bufs):
                if st.op is not None:
                    st.op.spmm(b_, out=yy, accumulate=True)

        def exchange_side():
            posted = [sg._post_stage(pipe, k, x, "diag") for k in range(len(pipe.stages))]
            for _, works in posted:
                for w_ in works:
                    w_.wait()
        out["pipeline"] = {"stages": len(pipe.stages), "scheme": pipe.scheme + ("+prefix" if pipe.prefix else ""),
                           "unpacked_prefix_rows_per_rank": pipe.prefix, "own_block_entries": pipe.own_nnz,
                           "stage_block_entries": [st.nnz for st in pipe.stages],
                           "stage_rows_received": [(sg.world - 1) * (st.span[1] - st.span[0]) if st.span is not None
                                                   else sum(st.recv_counts) for st in pipe.stages],
                           "own_block": phase(lambda: pipe.own.spmm(x, None)),
                           "compute_side_own_plus_stage_blocks": phase(compute_side),
                           "exchange_side_packs_plus_stages": phase(exchange_side)}
    out["whole_spmm_overlapped"] = phase(lambda: sg.spmm(x, None))

    # HOST side of one distributed SpMM: the time the Python thread needs to ENQUEUE it (ctypes launches, the async
    # collectives, the closures) with the device free to run behind -- no synchronisation inside the loop, median of 50.
    # If this approaches the device time of the step the device starves (at 8 ranks the step is ~0.6-0.8 ms).
    def host_enqueue(fn, n=50):
        for _ in range(3):
            fn()
        dist.barrier()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
            if len(ts) % 10 == 0:
                torch.cuda.synchronize()          # keep the queue from growing without bound (outside the timed calls)
        torch.cuda.synchronize()
        med = torch.tensor([sorted(ts)[len(ts) // 2]], device=dev, dtype=torch.float64)
        dist.all_reduce(med, op=dist.ReduceOp.MAX)
        return round(med.item(), 4)
    try:
        host_ms = host_enqueue(lambda: sg.spmm(x, None))
        dev_ms = out["whole_spmm_overlapped"].get("max_ms")
        out["host_enqueue_ms_per_spmm"] = host_ms
        out["host_enqueue_over_device_time"] = round(host_ms / dev_ms, 3) if dev_ms else None
        out["host_enqueue_local_launches_only_ms"] = host_enqueue(
            lambda: sg.local_step(d0, x, None, sg._gather_buffer(x, False), rs_out)) if A is not None else None
    except Exception as e:                           # noqa: BLE001 - same rule as `phase`
        if W > 1 or is_device_error(e):
            raise
        out["host_enqueue_ms_per_spmm"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # a plain `python3 bench.py --gpus N`: build once here (hipcc, no GPU call), then one child rank per GPU
        from pytextgcn_amd import build as _build
        if not args.launch_check and not os.path.exists(_build.LIB_PATH):
            _build.build()
        sys.exit(launch_ranks(args))
    if args.launch_check:
        sys.exit(launch_check(int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))))
    # both launch paths (a child of launch_ranks, or the driver's own `python -m torch.distributed.run ... bench.py`) come
    # through here before the first HIP call: ONE place decides the HSA / IPC environment of the rank
    from pytextgcn_amd.sharded import prepare_hsa_env
    hsa_env = prepare_hsa_env()
    # stdout carries exactly ONE line (the JSON record): everything else that writes to file descriptor 1
    # -- RCCL prints a version banner there when a communicator is created -- is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world          # under torch.distributed.run the world size is authoritative
    # a checkout without binaries (the .so is git-ignored): local rank 0 compiles it once (hipcc is on the
    # GPU box, os.replace makes the file appear atomically), the other ranks wait for the file
    from pytextgcn_amd import build as _build
    if not os.path.exists(_build.LIB_PATH):
        if local_rank == 0:
            _build.build()
        else:
            for _ in range(1800):
                if os.path.exists(_build.LIB_PATH):
                    break
                time.sleep(0.5)
    # rehearsal knobs (one-GPU box): all ranks on one card over gloo exercises the N>1 code path
    backend = os.environ.get("TGCN_BENCH_BACKEND", "nccl")
    if "TGCN_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["TGCN_BENCH_DEVICE"])
        if backend == "nccl" and world > 1:
            # RCCL ranks on ONE card (rehearsal): each poses as its own host, RCCL joins them over loopback sockets
            from pytextgcn_amd.sharded import let_rccl_ranks_share_a_device
            let_rccl_ranks_share_a_device(rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        init_group(dist, backend, dev)

    from pytextgcn_amd import synth
    from pytextgcn_amd.plan import GraphPlan

    import pytextgcn_amd as _pkg
    _pkg.set_degree_sum("accurate" if args.mode == "accurate" else "reference")     # what the epoch's GCN / ShardedGCN build
    N, E, F, C = CONFIGS[args.config]
    setup_s = {}                       # wall time of the one-off construction steps (not part of the metric)
    t_setup = time.perf_counter()
    gen_kw = dict(vocab_frac=0.03, doc_word_share=0.9) if args.config == "c3" else {}

    def make_graph(features):
        if args.config in ("c5", "c5s"):  # no word / document structure: every node is an ordinary node
            return synth.power_law_graph(N, E, seed=44, device=dev, n_classes=C, features=features)
        return synth.word_doc_graph(N, E, seed=44, device=dev, n_classes=C, features=features, **gen_kw)
    if world == 1:
        g = make_graph("sparse_identity")
    else:
        # rank 0 generates the graph and broadcasts it, so every rank partitions identical bytes
        from pytextgcn_amd.data import Data
        if rank == 0:
            g0 = make_graph("none")
            coo, attr = g0.edge_index.t().contiguous(), g0.edge_attr.contiguous()
            meta = torch.tensor([g0.n_vocab], device=dev)
            del g0
        else:
            coo = torch.empty(E, 2, dtype=torch.int64, device=dev)
            attr = torch.empty(E, dtype=torch.float32, device=dev)
            meta = torch.zeros(1, dtype=torch.int64, device=dev)
        for t in (coo, attr, meta):
            dist.broadcast(t, src=0)
        g = Data(x=None, edge_index=coo.t(), edge_attr=attr, n_vocab=int(meta.item()))
    torch.cuda.synchronize()
    setup_s["graph_generation" + ("_and_broadcast" if world > 1 else "")] = round(time.perf_counter() - t_setup, 3)
    note(f"graph {args.config} generated", rank)
    t_setup = time.perf_counter()
    gen = torch.Generator(device=dev).manual_seed(1234)
    bias = torch.randn(F, device=dev, generator=gen)

    force_sharded = os.environ.get("TGCN_BENCH_FORCE_SHARDED") == "1"   # rehearse the N>1 path at N=1
    # The normalisation mode of the operator (pytextgcn_amd/plan.py).  The HEADLINE is the reference-order mode -- the
    # reference's own arithmetic (PyG-1.6.3 gcn_norm as textgcn/lib/models.py:11-20 runs it: sequential fp32 degree sums,
    # (dis[src] * w) * dis[dst]), bit for bit the oracle's weights, M^T stored beside M -- because that is the mode whose
    # results meet BASELINE.json's 1e-5 against the reference formulation on this very graph.  The accurate mode (float64
    # degree sums, bitwise symmetric operator, no stored M^T) is timed beside it at N = 1.
    headline_mode = "accurate" if args.mode == "accurate" else "reference"
    if world == 1 and not force_sharded:
        plan = GraphPlan(g.edge_index, g.edge_attr, N, degree_sum=headline_mode)
        x = torch.randn(N, F, device=dev, generator=gen)
        gout = torch.randn(N, F, device=dev, generator=gen)
        y = torch.empty(N, F, device=dev)
        dxw = torch.empty(N, F, device=dev)

        def make_step(pl):
            def step(ev=None):
                if ev is not None:
                    ev[0].record()
                pl.spmm(x, bias, out=y)
                if ev is not None:
                    ev[1].record()
                pl.spmm(gout, None, transpose=True, out=dxw)
                if ev is not None:
                    ev[2].record()
            return step
        step = make_step(plan)
        bytes_fwd = plan.algorithmic_bytes(F, bias=True)
        bytes_bwd = plan.algorithmic_bytes(F, bias=False, transpose=True)
        parallelism = "single"
    else:
        from pytextgcn_amd.sharded import ShardedGraph
        if dist is None:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            init_group(dist, "nccl", dev, rank=0, world_size=1)
        # word nodes: replicated operand block; a graph without them (c5) has no hub structure -> every node's
        # rows may be needed anywhere: the halo exchange sends the referenced ones
        hubs = torch.arange(N, device=dev) < g.n_vocab if g.n_vocab > 0 else None
        sg = ShardedGraph(g.edge_index, g.edge_attr, N, group=dist.group.WORLD, hubs=hubs, degree_sum=headline_mode)
        x = torch.randn(sg.n_local, F, device=dev, generator=gen)
        gout = torch.randn(sg.n_local, F, device=dev, generator=gen)

        def step(ev=None):
            if ev is not None:
                ev[0].record()
            sg.spmm(x, bias)
            if ev is not None:
                ev[1].record()
            sg.spmm(gout, None, transpose=True)
            if ev is not None:
                ev[2].record()
        plan = sg.plan
        ops_f = [op for op in sg.ops[0] if op is not None]
        ops_b = [op for op in sg.ops[-1] if op is not None]
        bytes_fwd = sum(op.algorithmic_bytes(F) for op in ops_f) + 4 * F
        bytes_bwd = sum(op.algorithmic_bytes(F) for op in ops_b)
        parallelism = (f"row{world}: " + ("hubs(words) replicated, hub rows reduce-scattered" if sg.rp > 0 else
                                          "no hub structure: operand rows exchanged") +
                       f" (exchange={sg.exchange}, A_r row chunks={sg.rs_chunks})")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    setup_s["plan" if parallelism == "single" else "partition_normalisation_local_operators_index_lists"] = \
        round(time.perf_counter() - t_setup, 3)
    t_setup = time.perf_counter()

    # N > 1: the exchange of the distributed SpMM has two forms (RCCL collectives / pairwise transfers + all-to-all,
    # pytextgcn_amd.sharded); which one is faster depends on what RCCL makes of the xGMI mesh, so a few untimed
    # steps of each decide (max over ranks, the same answer on every rank).  TGCN_EXCHANGE pins the form.
    exchange_selection = None
    if world > 1 and "TGCN_EXCHANGE" not in os.environ:
        # Which forms are tried.  Started by launch_ranks() (a parent with a wall budget and a fallback run watches):
        # all of them.  Started directly under `python -m torch.distributed.run` there is nobody to fall back on if the
        # pairwise form (batched send / recv) hangs on this node, and a lost run is worse than a slower exchange: that
        # form is left out unless TGCN_BENCH_TRY_ALL_FORMS=1 asks for it.
        watched = os.environ.get("TGCN_BENCH_WATCHDOG") == "1" or os.environ.get("TGCN_BENCH_TRY_ALL_FORMS") == "1"
        # Unwatched: RCCL's own collectives and the forms built on all_to_all_single with split sizes (an ordinary RCCL
        # collective, bounded by the group's timeout like the others): "halo", and for a graph without hub structure the
        # pipelined exchange built on it.  The pairwise form (batched send / recv) stays with the watched launches.
        forms = list(sg.EXCHANGES) if (watched or backend != "nccl") else ["collective", "halo"]
        # candidates (form, A_r row chunks, pipeline stages, pipeline scheme); "collective" with A_r in one piece first:
        # it is the configuration the fallback of launch_ranks() runs, too
        candidates = []
        if sg.rp > 0:
            chunkings = [1, 2, 4] if ("TGCN_RS_CHUNKS" not in os.environ) else [sg.rs_chunks]
            candidates = [(mode, K, 0, None) for K in chunkings for mode in forms]
        else:
            # no hub structure (c5): the whole-operand all-gather, the halo form, and the pipelined exchange with the
            # stage count derived from the bytes, half of it, and one peer per stage
            candidates = [(mode, 1, 0, None) for mode in forms if mode != "p2p" or watched]
            auto_K = sg.pipe_stages
            stage_counts = [auto_K] if "TGCN_PIPE_STAGES" in os.environ else sorted({auto_K, max(1, auto_K // 2), 1}, reverse=True)
            candidates += [("pipeline", 1, K, "slices") for K in stage_counts]
            if "TGCN_PIPE_PREFIX" not in os.environ and (watched or backend != "nccl"):
                # the degree-ordered slots nearly every peer reads travel unpacked, in K contiguous ranges (batched
                # send / recv, like the pairwise form: with the watched launches)
                candidates += [("pipeline", 1, K, "slices+prefix") for K in stage_counts if K > 1]
            if "TGCN_PIPE_SCHEME" not in os.environ and world > 2:
                candidates.append(("pipeline", 1, 0, "peer"))
            forms = forms + ["pipeline"]
        trial = {}
        env_prefix = os.environ.get("TGCN_PIPE_PREFIX", "0")         # pinned by the caller: "auto" or rows per rank
        env_prefix = "auto" if env_prefix == "auto" else (int(env_prefix) if env_prefix.isdigit() else 0)
        for mode, K, stages, scheme in candidates:
            if sg.rp > 0:
                sg.set_rs_chunks(K)
            if mode == "pipeline":                               # ("auto": one all-reduce -- every rank is here)
                sg.set_pipeline(stages, scheme.split("+")[0], prefix="auto" if scheme.endswith("+prefix") else env_prefix)
            sg.exchange = mode
            label = f"{mode}/{K}" if mode != "pipeline" else f"pipeline-{scheme}/{world - 1 if scheme == 'peer' else sg.pipe_stages}"
            # A form this backend / build refuses raises on every rank alike, before anything is enqueued, and is
            # skipped.  A failure on ONE rank in the middle of a step leaves its peers inside a collective: they
            # are released by the group's timeout (init_group), the process ends, and launch_ranks() runs the
            # plain configuration in a fresh child -- nothing here tries to outwit a broken exchange.
            ok = 1.0
            ms = float("inf")
            try:
                step()
            except RuntimeError as e:
                print(f"bench.py: exchange form {label} failed on rank {rank}: {e}"[:300], file=sys.stderr, flush=True)
                ok = 0.0
            flag = torch.tensor([ok], device=dev, dtype=torch.float64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # every rank takes the same branch
            if flag.item() > 0:
                barrier()
                t0 = time.perf_counter()
                for _ in range(3):
                    step()
                barrier()
                ms = (time.perf_counter() - t0) / 3 * 1e3
            t = torch.tensor([ms], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            trial[label] = (t.item(), (mode, K, stages, scheme))
        setting = {k: v[1] for k, v in trial.items()}
        trial = {k: v[0] for k, v in trial.items()}
        best = min(trial, key=trial.get)
        if trial[best] == float("inf"):
            raise RuntimeError(f"no form of the exchange works on this node: {trial}")
        trial = {k: (v if v != float("inf") else None) for k, v in trial.items()}
        mode, K, stages, scheme = setting[best]
        if sg.rp > 0:
            sg.set_rs_chunks(K)
        if mode == "pipeline":
            sg.set_pipeline(stages, scheme.split("+")[0], prefix="auto" if scheme.endswith("+prefix") else env_prefix)
        sg.exchange = mode
        sg.drop_unused_chunks()                    # the operators of the configurations that lost are dead weight
        sg.drop_unused_pipelines()
        exchange_selection = {"ms_per_step": trial, "chosen": best, "rows_received_per_spmm": sg.exchange_rows(),
                              "forms_tried": forms, "watched_by_launcher": os.environ.get("TGCN_BENCH_WATCHDOG") == "1"}
        parallelism = (f"row{world}: " + ("hubs(words) replicated, hub rows reduce-scattered" if sg.rp > 0 else
                                          "no hub structure: operand rows exchanged") +
                       f" (exchange form / A_r row chunks = {best}, the fastest of {trial})")
    if exchange_selection is not None:
        torch.cuda.synchronize()
        setup_s["exchange_trial_steps"] = round(time.perf_counter() - t_setup, 3)
    def timed(step_fn):
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides (max over ranks);
        per-launch kernel time of the SpMM op from HIP events recorded on the launch stream inside the timed region."""
        for _ in range(args.warmup):
            step_fn()
        events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step_fn(events[k])
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return (dt, sum(e[0].elapsed_time(e[1]) for e in events) / args.steps,
                sum(e[1].elapsed_time(e[2]) for e in events) / args.steps)
    note("operators built; timing the headline", rank)
    elapsed, ms_fwd, ms_bwd = timed(step)
    note(f"headline timed: {elapsed / args.steps * 1e3:.3f} ms per step", rank)
    launch_ms = 0.5 * (ms_fwd + ms_bwd)
    # the accurate mode beside it: same graph, same operands, same K / W, same bracket (N = 1)
    other_mode = None
    if parallelism == "single" and args.mode == "both":
        plan_acc = GraphPlan(g.edge_index, g.edge_attr, N, degree_sum="accurate")
        el2, f2, b2 = timed(make_step(plan_acc))
        other_mode = {"mode": "accurate", "value": 2.0 * E / (el2 / args.steps), "ms_per_step": el2 / args.steps * 1e3,
                      "launch_ms_fwd": f2, "launch_ms_bwd": b2, "stores_transpose": not plan_acc.symmetric,
                      "plan_device_bytes": plan_acc.stats()["device_bytes"]}
        plan_acc.close()
        del plan_acc
    launch_bytes = 0.5 * (bytes_fwd + bytes_bwd)
    achieved = launch_bytes / (launch_ms * 1e-3) / 1e9

    # ---- the headline, complete at this point; everything below is secondary ------------------------------------------
    secondary_errors = {}
    completed = {}                     # results of the secondary measurements that finished: an aborted record keeps them
    workload = (f"{args.config}: synthetic {'power-law graph' if args.config in ('c5', 'c5s') else 'PMI/TF-IDF word-doc graph'}, N={N}, "
                f"E={E}, nnz={E + N} (with self loops), F={F}, seed 44; step = M@X+b and M^T@G; "
                + ("normalisation mode 'reference' (PyG-1.6.3 gcn_norm's own fp32 arithmetic, weights bit for bit the "
                   "oracle's; M^T stored beside M)" if headline_mode == "reference" else
                   "normalisation mode 'accurate' (float64 degree sums, bitwise symmetric operator)"))
    headline = {
        "metric": "edges/sec (fwd+bwd SpMM), 2M-node/50M-edge graph h=200"
                  if args.config == "c4" else f"edges/sec (fwd+bwd SpMM), {args.config}",
        "value": 2.0 * E / (elapsed / args.steps), "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "normalisation_mode": headline_mode, "parallelism": parallelism}}
    state = {"written": False}

    def write_record(rec):
        if rank == 0 and not state["written"]:
            state["written"] = True
            os.write(json_fd, (json.dumps(rec) + "\n").encode())

    def abort(where, err):
        """A failure that must not be swallowed (see `secondary`): rank 0 writes the headline it holds, marked as an aborted
        run, and every rank that gets here exits non-zero WITHOUT tearing the group down (its peers may sit in a collective;
        torch.distributed.run ends them when this rank is gone)."""
        print(f"bench.py: rank {rank} aborts in {where}: {err}", file=sys.stderr, flush=True)
        write_record(dict(headline, **completed, aborted={"in": where, "error": err, "rank": rank},
                          secondary_errors=dict(secondary_errors, **{where: err})))
        sys.stderr.flush()
        os._exit(1)
    if rank == 0 and world > 1:
        # A peer that aborts makes torch.distributed.run send SIGTERM to the rest.  Rank 0 is then most likely blocked
        # inside a collective (C++ code: a Python-level signal handler would only run once the call returns, i.e. at its
        # timeout, long after the launcher's SIGKILL).  So the signal is routed to a wake-up descriptor and a helper THREAD
        # -- which runs while the main thread waits inside the collective with the GIL released -- hands the headline over.
        import signal
        import threading
        rd, wr = os.pipe()
        os.set_blocking(wr, False)
        signal.signal(signal.SIGTERM, lambda signum, frame: None)       # keep the default action (die at once) from firing
        signal.set_wakeup_fd(wr, warn_on_full_buffer=False)

        def on_term():
            while True:
                b = os.read(rd, 1)
                if b and b[0] == signal.SIGTERM:
                    write_record(dict(headline, **completed, aborted={"in": "SIGTERM (a peer rank ended)", "rank": 0},
                                      secondary_errors=secondary_errors or None))
                    os._exit(1)
        threading.Thread(target=on_term, daemon=True, name="bench-sigterm").start()

    # what the SpMM pulls from HBM, measured live on this box (outside the timed region; N = 1 only: at N > 1
    # the step contains collectives every rank must enter)
    hbm = hbm_activity(step, dev) if (world == 1 and not force_sharded and not args.no_hbm_activity) else None
    note("memory-controller activity measured; secondary measurements follow", rank)

    def secondary(name, fn):
        """One guarded secondary measurement; the headline is complete before the first of them.
          * N = 1: a Python-level failure is named in `secondary_errors` and the run goes on.  A DEVICE error (the message
            names HIP / HSA / a memory fault) poisons the context and is never swallowed: `abort()` writes the headline,
            marked aborted, and the process exits non-zero.
          * N > 1: the guarded code contains collectives, so a rank that raises may have left its peers inside one, and any
            collective it enters next (were it only to agree on the outcome) could pair with a different call of theirs.
            Every failure therefore ends the run the same way: `abort()` -- rank 0 still hands over the headline (itself,
            or from its SIGTERM handler when a peer went first), torch.distributed.run ends the other ranks, and the
            launcher (launch_ranks) runs the plain configuration in a fresh child."""
        import pytextgcn_amd as pkg
        from pytextgcn_amd import dense as _dense
        note(f"secondary: {name}", rank)
        try:
            if os.environ.get("TGCN_BENCH_TEST_FAIL") == "secondary" and world > 1 and rank == world - 1 \
                    and name == "exchange_diagnostics":
                raise RuntimeError("test hook: the last rank fails alone inside a guarded section")   # tests/test_zz_gpu_sharded.py
            completed[name] = fn()
            return completed[name]
        except Exception as e:                       # noqa: BLE001 - classified here
            err = f"{type(e).__name__}: {e}"[:300]
            print(f"bench.py: {name} failed on rank {rank}: {err}", file=sys.stderr, flush=True)
            if world > 1 or is_device_error(e):
                abort(name, err)
            secondary_errors[name] = err
            return None
        finally:                                     # whatever happened, the next measurement starts from the defaults
            pkg.enable_activation_reuse(False)
            pkg.enable_linear_collapse(False)
            pkg.enable_fused_dropout(False)
            _dense.enable_split_gemms(False)

    parity = None
    if parallelism != "single" and not args.no_verify:
        parity = secondary("distributed_parity",                              # collective: every rank takes part
                           lambda: distributed_parity(sg, g, N, F, x, gout, bias, dev, dist, headline_mode))

    # The epoch of flat_amazon.py:99-117, four ways (the figures README / DESIGN quote):
    #   epoch_ms                  the import swap alone: torch's loss / optimizer / dropout around the HIP operators (N = 1)
    #   epoch_ms_fused            the package's fused loss, optimizer and dropout kernels
    #   epoch_ms_fused_w1_update_in_backward_with_activation_reuse    + the two bitwise-neutral switches
    #   epoch_ms_flat_loop        all five switches behind ONE object (train.FlatLoop / sharded.FlatLoop: + only the logits rows
    #                             that are read; on the partition + the narrow exchange where the widths allow)
    # `--epoch-matrix` adds the switches one by one (`epoch_matrix`).
    epoch_ms = epoch_ms_fused = epoch_ms_w1_reuse = epoch_ms_flat_loop = None
    epoch_matrix = None
    diagnostics = secondary("exchange_diagnostics", lambda: exchange_diagnostics(sg, F, dev, dist)) \
        if (world > 1 or force_sharded) else None
    if (world > 1 or force_sharded) and not args.no_epoch:
        del x, gout
        narrow_ok = sg.rp > 0 and C % 4 == 0 and F % 4 == 0
        epoch_ms_fused = secondary("epoch_ms_fused", lambda: sharded_epoch_ms(sg, N, F, C, dev, dist))
        epoch_ms_w1_reuse = secondary("epoch_ms_fused_w1_update_in_backward_with_activation_reuse",
                                      lambda: sharded_epoch_ms(sg, N, F, C, dev, dist, reuse=True, fuse_w1=True))
        if sg.rp > 0:
            epoch_ms_flat_loop = secondary(
                "epoch_ms_flat_loop",
                lambda: sharded_epoch_ms(sg, N, F, C, dev, dist, reuse=True, fuse_w1=True, narrow=narrow_ok, rows=True))
        if args.epoch_matrix:
            epoch_matrix = {
                "fused_with_activation_reuse": secondary(
                    "epoch_matrix.fused_with_activation_reuse", lambda: sharded_epoch_ms(sg, N, F, C, dev, dist, reuse=True)),
                "fused_w1_update_in_backward": secondary(
                    "epoch_matrix.fused_w1_update_in_backward", lambda: sharded_epoch_ms(sg, N, F, C, dev, dist, fuse_w1=True))}
            if sg.rp > 0:
                epoch_matrix["fused_w1_reuse_needed_rows_only"] = secondary(
                    "epoch_matrix.fused_w1_reuse_needed_rows_only",
                    lambda: sharded_epoch_ms(sg, N, F, C, dev, dist, reuse=True, fuse_w1=True, rows=True))
            if narrow_ok:
                epoch_matrix["fused_w1_reuse_narrow_exchange"] = secondary(
                    "epoch_matrix.fused_w1_reuse_narrow_exchange",
                    lambda: sharded_epoch_ms(sg, N, F, C, dev, dist, reuse=True, fuse_w1=True, narrow=True))
    if world == 1 and not args.no_epoch and not force_sharded:
        del x, gout
        epoch_ms = secondary("epoch_ms", lambda: epoch_time_ms(g, F, C, fused=False))
        epoch_ms_fused = secondary("epoch_ms_fused", lambda: epoch_time_ms(g, F, C, fused=True))
        # both switches are bitwise neutral
        epoch_ms_w1_reuse = secondary("epoch_ms_fused_w1_update_in_backward_with_activation_reuse",
                                      lambda: epoch_time_ms(g, F, C, fused=True, fuse_w1=True, reuse=True))
        epoch_ms_flat_loop = secondary("epoch_ms_flat_loop", lambda: flat_loop_epoch_ms(g, F, C))
        if args.epoch_matrix:
            epoch_matrix = {
                "fused_with_activation_reuse": secondary(
                    "epoch_matrix.fused_with_activation_reuse", lambda: epoch_time_ms(g, F, C, fused=True, reuse=True)),
                # pytextgcn_amd.enable_linear_collapse(): the eval forward of the activation-free network (models.py:22) as
                # two propagations at the class width
                "fused_with_collapsed_eval": secondary(
                    "epoch_matrix.fused_with_collapsed_eval", lambda: epoch_time_ms(g, F, C, fused=True, collapse=True)),
                "fused_w1_update_in_backward": secondary(
                    "epoch_matrix.fused_w1_update_in_backward", lambda: epoch_time_ms(g, F, C, fused=True, fuse_w1=True)),
                # `gcn(g, rows=...)` written out by hand (train.FlatLoop does the same)
                "fused_w1_reuse_needed_rows_only": secondary(
                    "epoch_matrix.fused_w1_reuse_needed_rows_only",
                    lambda: epoch_time_ms(g, F, C, fused=True, fuse_w1=True, reuse=True, needed_rows=True)),
                # the layer-2 products in the opt-in split-bf16 mode (fp32-accurate, NOT bit-equal to the fp32 FMA chain)
                "fused_w1_reuse_split_bf16_gemms": secondary(
                    "epoch_matrix.fused_w1_reuse_split_bf16_gemms",
                    lambda: epoch_time_ms(g, F, C, fused=True, fuse_w1=True, reuse=True, split_gemms=True))}

    model_of_scaling = scaling_model(world, elapsed / args.steps * 1e3, parity, diagnostics, sg, F) \
        if parallelism != "single" else None
    rccl = rccl_info(dist, backend, world, rank, local_rank, dev)        # collective: every rank takes part
    rccl["hsa_env"] = hsa_env
    copy_gbps = device_copy_gbps(dev) if rank == 0 else None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        # ---- roofline of the dominant kernel family (one tgcn_spmm launch), four ways of counting bytes ----
        #   algorithmic  SURVEY 8(d): every gathered row charged to HBM (no cache reuse assumed).  > peak on
        #                this path: the word block is served by L2 / Infinity Cache and the dense hot block
        #                reads X once -- it measures work done, not memory traffic;
        #   hbm          bytes the memory controllers actually moved (live, `hbm_activity`);
        #   fabric       L2 <-> fabric bytes from the committed rocprofv3 PMC passes (Infinity-Cache hits
        #                included: an upper bound on HBM bytes; the link the kernel is in fact bound by)  -> `frac`;
        #   compulsory   every input and output touched exactly once (lower bound).
        per_s = 1.0 / (launch_ms * 1e-3) / 1e9
        n_out = N if parallelism == "single" else sg.n_local
        compulsory = 8 * plan.nnz + 4 * n_out + 8 * n_out * F if parallelism == "single" else None
        fabric, fabric_src, fabric_fresh = fabric_traffic(args.config, world)
        committed = {"bytes_per_launch": fabric, "source": fabric_src, "collected_on_these_kernel_sources": fabric_fresh}
        live_note = None
        if parallelism == "single" and not args.no_live_traffic and not args.no_hbm_activity:
            # the counters of THIS run's library on THIS box (two short child runs under rocprofv3), so that `traffic` is
            # not a constant the builder committed; the committed figure stays in the record beside it
            note("counter passes (two child runs under rocprofv3 --pmc)")
            live, live_note = live_fabric_traffic(args.config, headline_mode, timeout_s=300 if args.config == "c5" else 90)
            if live is not None:
                fabric, fabric_src, fabric_fresh = live, live_note, True
        hbm_bytes = None
        if hbm and "bytes_per_step" in hbm:
            hbm_bytes = hbm["bytes_per_step"] / 2.0            # a step is two launches (forward, transposed)
        # THE CONTRACT OF THIS OBJECT (frozen in round 4; tests/test_host.py holds it):
        #   achieved   GB/s that `frac` is a fraction of:  traffic / launch time;   frac = achieved / peak, raw (never
        #              capped: a value above 1 is reported as it is, with "non_physical": true)
        #   traffic    bytes per launch on the basis `frac_basis` names:
        #                "fabric"  counter traffic (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this bench command,
        #                          committed under profiles/, collected on these very kernel sources): what crosses the
        #                          L2 <-> Infinity Cache / HBM link -- the link this kernel is bound by.  The counters
        #                          count Infinity-Cache hits too, so it is an UPPER bound on HBM bytes;
        #                "hbm"     the live memory-controller figure, when no fresh counter profile is on file;
        #                "fabric-stale" / "algorithmic"  last resorts (older kernels' counters / no measurement at all)
        #   achieved_algorithmic / _fabric / _hbm  and  frac_algorithmic / _fabric / _hbm / _compulsory  always stand
        #   side by side, each = its byte count / launch time (/ peak).
        ach_alg = achieved
        ach_fabric = None if fabric is None else fabric * per_s
        ach_hbm = None if hbm_bytes is None else hbm_bytes * per_s
        frac_fabric = None if ach_fabric is None else ach_fabric / HBM_PEAK_GBPS
        frac_hbm = None if ach_hbm is None else ach_hbm / HBM_PEAK_GBPS
        if fabric is not None and fabric_fresh:
            ach, frac_basis, traffic, traffic_basis = ach_fabric, "fabric", fabric, fabric_src
        elif hbm_bytes is not None:
            ach, frac_basis, traffic = ach_hbm, "hbm", hbm_bytes
            traffic_basis = "live: sysfs mem_busy_percent while the step loops, calibrated on a 1 GiB device copy"
        elif fabric is not None:
            ach, frac_basis, traffic, traffic_basis = ach_fabric, "fabric-stale", fabric, fabric_src + " (older kernels)"
        else:
            ach, frac_basis, traffic, traffic_basis = ach_alg, "algorithmic", launch_bytes, "SURVEY 8(d) gather model, no measurement"
        frac = ach / HBM_PEAK_GBPS
        roofline = {
            "bound": "hbm",
            "bound_detail": "memory-bound gather; the binding link is L2 <-> fabric (Infinity Cache + HBM behind it): "
                            "`frac` on basis 'fabric' is counter traffic over the 8 TB/s HBM peak -- an upper bound on "
                            "the HBM fraction, which is `frac_hbm` (memory controllers, live)",
            "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": frac, "frac_basis": frac_basis, "non_physical": bool(frac > 1.0),
            "traffic": traffic, "traffic_basis": traffic_basis,
            "hbm_activity": hbm,
            "achieved_algorithmic": ach_alg, "achieved_fabric": ach_fabric, "achieved_hbm": ach_hbm,
            "frac_algorithmic": ach_alg / HBM_PEAK_GBPS,
            "frac_hbm": frac_hbm,
            "frac_fabric": frac_fabric,
            # the same counter traffic against what the guide measures for Infinity-Cache-resident row gathers
            # (MI355X_MICROARCH.md 'Indexed rows': 8.6 TB/s chip-wide for a 38 MB table): the delivery ceiling of the link
            "fabric_gather_ceiling_GBps": FABRIC_GATHER_CEILING_GBPS,
            "frac_fabric_of_gather_ceiling": None if ach_fabric is None else ach_fabric / FABRIC_GATHER_CEILING_GBPS,
            "traffic_fabric": fabric, "traffic_fabric_source": fabric_src, "traffic_fabric_fresh": fabric_fresh,
            "traffic_fabric_committed": committed, "traffic_fabric_live_note": live_note,
            "compulsory_bytes_per_launch": compulsory,
            "frac_compulsory": None if compulsory is None else compulsory * per_s / HBM_PEAK_GBPS,
            "l2_resident_ceiling_ms": profile_constant(f"{args.config}_F{F}_l2_resident_ceiling_ms"),
            "kernel": ("one tgcn_spmm launch: k_spmm_gather" + (" + k_spmm_hot" if plan.stats().get("hot_rows") else "")
                       + " + k_spmm_fix") if parallelism == "single"
                      else "one distributed SpMM on this rank: local SpMM launches + the exchange",
            # secondary denominator: this box's device-to-device copy rate (read + write)
            "device_copy_GBps": copy_gbps, "frac_of_device_copy": ach_alg / copy_gbps,
            # the measured HBM rate of the launch against what a plain copy reaches on THIS box
            "frac_hbm_of_device_copy": None if hbm_bytes is None else hbm_bytes * per_s / copy_gbps,
            "launch_ms": launch_ms, "launch_ms_fwd": ms_fwd, "launch_ms_bwd": ms_bwd,
            "algorithmic_bytes_per_launch": launch_bytes}
        out = dict(headline)
        out.update({
            # the same step with the operator normalised in the accurate mode (opt-in: float64 degree sums, one stored
            # block), timed in this run with the same K / W; NOT the headline
            "value_accurate_mode": other_mode["value"] if other_mode else None,
            "accurate_mode": other_mode,
            "plan": {"mode": headline_mode, "stores_transpose": not getattr(plan, "symmetric", False),
                     "device_bytes": plan.stats().get("device_bytes") if hasattr(plan, "stats") else None},
            "roofline": roofline,
            # the epoch of flat_amazon.py:99-117 (NOT part of the metric), four ways -- see where they are measured above
            "epoch_ms": epoch_ms,
            "epoch_ms_fused": epoch_ms_fused,
            "epoch_ms_fused_w1_update_in_backward_with_activation_reuse": epoch_ms_w1_reuse,
            "epoch_ms_flat_loop": epoch_ms_flat_loop,
            "epoch_matrix": epoch_matrix,                     # --epoch-matrix: the switches one by one
            # N > 1: the global training loss every variant of the sharded epoch reached after its timed epochs -- same
            # initial weights, same keyed dropout masks, so they must agree to fp32 rounding carried through Adam
            "sharded_epoch_final_loss": (dict(SHARDED_EPOCH_LOSS, max_rel_spread=(
                (max(SHARDED_EPOCH_LOSS.values()) - min(SHARDED_EPOCH_LOSS.values())) / abs(min(SHARDED_EPOCH_LOSS.values()))))
                if SHARDED_EPOCH_LOSS else None),
            "exchange_floats_per_hub_row_and_step": None if parallelism == "single" else {
                "plain": {"train": 4 * F + 4 * C, "eval": 2 * F + 2 * C},
                "narrow": {"train": 2 * F + 6 * C, "eval": F + 3 * C},
                # forward(rows=...): no hub row of the logits is read -- layer 2 loses one collective each way
                "plain_needed_rows": {"train": 4 * F + 2 * C, "eval": 2 * F + C},
                "narrow_needed_rows": {"train": 2 * F + 4 * C, "eval": F + 2 * C},
                "hub_rows": int(sg.world * sg.hp), "bytes_per_float": 4,
                "note": "each collective moves (W - 1) / W of the [W * hp] hub block per rank; an all-reduce counts as "
                        "reduce-scatter + all-gather; one epoch = one training step + one eval forward"},
            "scaling_model": model_of_scaling,
            # N > 1: phase-by-phase timing of one distributed SpMM on this node (not part of the metric)
            "exchange_diagnostics": diagnostics,
            "exchange_selection": exchange_selection,
            "distributed_parity": parity,
            "secondary_errors": secondary_errors or None,
            "rccl": rccl,
            "setup_s": setup_s,
        })
        if world == 1 and not args.no_cpu_baseline and not force_sharded:
            note("CPU baselines (reference formulation, then the C CSR oracle)")
            out["cpu_baseline"] = cpu_baseline(plan, F, args.cpu_sample_frac, E,
                                               full={"auto": "auto", "full": True, "sample": False}[args.cpu_ref])
        write_record(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
