#!/usr/bin/env python3
"""A/B of "X tiles staged in LDS" on config c4, F = 200 (tools/experiments/lds_tiles/gather_lds.hip, an experiment outside
libtgcn.so): the same simplified gather kernel with the R most-gathered operand rows in LDS against R = 0, in the two
workgroup shapes round 1 had tried, next to the product's tgcn_spmm on the same operator.  Results are checked against
the R = 0 run (bit for bit: the arithmetic is the same) before they are timed.
    python tools/experiments/lds_tiles/run.py > profiles/r04_exp_lds_tiles.log"""
import ctypes
import json
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
from pytextgcn_amd import synth  # noqa: E402
from pytextgcn_amd.plan import GraphPlan  # noqa: E402

so = os.path.join(HERE, "libexp_gather_lds.so")
src = os.path.join(HERE, "gather_lds.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", src, "-o", so], check=True)
lib = ctypes.CDLL(so)
P, I, I64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib.exp_gather.argtypes = [I, I, I, P, I, P, P, P, I64, I, P, I, P, I64, P, P]

dev = torch.device("cuda:0")
N, E, F, T = 2_000_000, 50_000_000, 200, 384
g = synth.word_doc_graph(N, E, seed=44, device=dev, features="none")
os.environ["TGCN_HOT_ROWS"] = "0"                     # the product kernel without its dense hot block: like for like
plan = GraphPlan(g.edge_index, g.edge_attr, N)
rp, col, val = plan.export_csr()
x = torch.randn(N, F, device=dev)

# work items on the host, as plan.hip packs them (row order; long rows in pieces of <= T entries)
rp_h = rp.cpu().tolist()
items, n_carry, wsum, r0 = [], 0, 0, 0
for r in range(N):
    d = rp_h[r + 1] - rp_h[r]
    if d > T:
        if r > r0:
            items.append((r0, r, rp_h[r0], rp_h[r]))
        nseg = (d + T - 1) // T
        seglen = (d + nseg - 1) // nseg
        for s in range(nseg):
            b = min(rp_h[r] + s * seglen, rp_h[r + 1])
            items.append((r, -n_carry - 1, b, min(b + seglen, rp_h[r + 1])))
            n_carry += 1
        r0, wsum = r + 1, 0
    else:
        wsum += d + 1
        if wsum >= T:
            items.append((r0, r + 1, rp_h[r0], rp_h[r + 1]))
            r0, wsum = r + 1, 0
if r0 < N:
    items.append((r0, N, rp_h[r0], rp_h[N]))
items_d = torch.tensor(items, dtype=torch.int32, device=dev)
n_items = items_d.size(0)
counts = torch.bincount(col.long(), minlength=N)
order = torch.argsort(counts, descending=True)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


print(json.dumps({"case": "product tgcn_spmm without the dense hot block (TGCN_HOT_ROWS=0)", "ms": round(timed(lambda: plan.spmm(x)), 3),
                  "items": n_items, "carry_rows": n_carry}), flush=True)
ref = None
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for waves, persist, grid, Rs in ((4, 0, (n_items + 3) // 4, (0, 8, 24)), (16, 1, 256, (0, 40, 80, 160)), (16, 1, 512, (0, 80))):
    for R in Rs:
        hot = order[:R].to(torch.int32).contiguous()
        slot = torch.full((N,), -1, dtype=torch.int64, device=dev)
        slot[hot.long()] = torch.arange(R, device=dev)
        sl = slot[col.long()]
        cmod = torch.where(sl >= 0, sl - (1 << 31), col.long()).to(torch.int32)      # sign bit set, slot in the low bits
        cv = torch.stack([cmod, val.view(torch.int32)], dim=1).contiguous()
        share = float(counts[hot.long()].sum()) / float(col.numel()) if R else 0.0
        y = torch.zeros(N, F, device=dev)
        carry = torch.zeros(max(n_carry, 1), F, device=dev)

        def run():
            st = lib.exp_gather(waves, persist, grid, items_d.data_ptr(), n_items, rp.data_ptr(), cv.data_ptr(), x.data_ptr(), F, F,
                                hot.data_ptr() if R else None, R, y.data_ptr(), F, carry.data_ptr(), stream)
            assert st == 0, st
        run()
        torch.cuda.synchronize()
        if ref is None:
            ref = (y.clone(), carry.clone())
            # the simplified kernel against the product (block rows only; pieces of long rows are summed differently)
            chk = plan.spmm(x)
            short = (rp[1:] - rp[:-1]) <= T
            err = float((y[short] - chk[short]).abs().max() / chk[short].abs().max())
            assert err < 1e-5, err
        same = bool(torch.equal(y, ref[0]) and torch.equal(carry, ref[1]))
        print(json.dumps({"case": f"{waves}-wave workgroups, {'persistent grid ' + str(grid) if persist else 'one item per wave'}",
                          "rows_in_lds": R, "lds_KB": R * F * 4 // 1024, "share_of_gathers_served_from_lds": round(share, 3),
                          "ms": round(timed(run), 3), "bitwise_equal_to_R0": same}), flush=True)
