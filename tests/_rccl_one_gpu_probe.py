"""Can two RCCL ranks share ONE GPU on this box?  RCCL refuses two ranks with the same (host hash, bus id); giving
each process its own NCCL_HOSTID makes them look like two hosts, so the pair talks through the socket transport
(loopback) instead of P2P / shared memory.  Slow, but it is real RCCL with world size 2: the call sequences of
pytextgcn_amd/sharded.py that gloo never takes (reduce_scatter_tensor, all_gather_into_tensor, batched send / recv).

    python tests/_rccl_one_gpu_probe.py          # parent: starts two ranks, bounded by a timeout

Test infrastructure (tests/test_zz_gpu_sharded.py runs it once per session).
"""
import os
import subprocess
import sys
import time


def rank_main(rank: int, world: int) -> None:
    os.environ["NCCL_HOSTID"] = f"tgcn-probe-{rank}"
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (what pytextgcn_amd.sharded.prepare_hsa_env decides)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=None,
                            timeout=__import__("datetime").timedelta(seconds=60))
    t0 = time.time()
    x = torch.full((1 << 20,), float(rank + 1), device=dev)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    assert float(x[0]) == 3.0, float(x[0])
    g = torch.empty(world * 1000, device=dev)
    dist.all_gather_into_tensor(g, torch.full((1000,), float(rank), device=dev))
    torch.cuda.synchronize()
    assert g[:1000].eq(0).all() and g[1000:].eq(1).all()
    r = torch.empty(1000, device=dev)
    dist.reduce_scatter_tensor(r, torch.arange(2000, device=dev, dtype=torch.float32))
    torch.cuda.synchronize()
    assert torch.equal(r, 2 * torch.arange(rank * 1000, rank * 1000 + 1000, device=dev, dtype=torch.float32))
    a = torch.full((4096,), float(rank), device=dev)
    b = torch.empty(4096, device=dev)
    ops = [dist.P2POp(dist.isend, a, 1 - rank), dist.P2POp(dist.irecv, b, 1 - rank)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    assert b.eq(1 - rank).all()
    # bandwidth of the loopback pair, so a rehearsal's numbers can be read for what they are
    big = torch.zeros(16 << 20, device=dev)
    torch.cuda.synchronize()
    t1 = time.time()
    for _ in range(3):
        dist.all_reduce(big)
    torch.cuda.synchronize()
    dt = (time.time() - t1) / 3
    print(f"rank {rank}: ok; init+checks {t1 - t0:.2f} s; 64 MB all-reduce {dt * 1e3:.1f} ms", flush=True)
    dist.destroy_process_group()


def main() -> int:
    if len(sys.argv) == 3:
        rank_main(int(sys.argv[1]), int(sys.argv[2]))
        return 0
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29541"))
    procs = [subprocess.Popen([sys.executable, __file__, str(r), "2"], env=env) for r in range(2)]
    deadline = time.time() + float(os.environ.get("PROBE_BUDGET_S", "150"))
    rc = 0
    for p in procs:
        try:
            rc |= p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            rc |= 124
    for p in procs:
        if p.poll() is None:
            p.kill()
            p.wait()
    print("probe exit", rc, flush=True)
    return rc


if __name__ == "__main__":
    sys.exit(main())
