#!/usr/bin/env python3
"""The flat_amazon.py loop on several GPUs of one node: one process per GPU, RCCL underneath.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/flat_synthetic_multigpu.py [--docs 20000] [--epochs 30] [--backend nccl]

Rank 0 builds the graph (Text2GraphTransformer) and broadcasts it; every rank then owns a block of
rows of the operator, of W1 / H1 / logits and of the Adam state (pytextgcn_amd.sharded).  The word
nodes are the replicated "hubs"; document features never leave their owner.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch as th
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytextgcn_amd import Text2GraphTransformer, optim, synth  # noqa: E402
from pytextgcn_amd.sharded import (ShardedGCN, ShardedGraph, init_process_group, prepare_hsa_env,  # noqa: E402
                                   sharded_cross_entropy)

prepare_hsa_env()                                  # before the first HIP call of this rank (dmabuf IPC for RCCL)

p = argparse.ArgumentParser()
p.add_argument("--docs", type=int, default=20000)
p.add_argument("--epochs", type=int, default=30)
p.add_argument("--backend", default="nccl")
p.add_argument("--device", type=int, default=None, help="force a device index (several ranks on one GPU: gloo only)")
p.add_argument("--hidden", type=int, default=100, help="hidden width (flat_amazon.py:26 uses 100)")
p.add_argument("--narrow", action="store_true",
               help="ShardedGCN(narrow_exchange=True): hub rows cross the links at the class width where the activation-free "
                    "network allows (pytextgcn_amd/narrow.py); needs hidden and class widths that are multiples of 4")
p.add_argument("--rows", action="store_true",
               help="gcn(rows=mask): name the rows of the logits the loop reads (documents only), so the last propagate step "
                    "drops the word rows -- and one collective each way (ShardedGraph.rows_view)")
p.add_argument("--loop-object", action="store_true",
               help="run the epochs through pytextgcn_amd.sharded.FlatLoop (W1's update inside the backward SpMM, activation "
                    "reuse, rows=...: every switch that leaves the numbers alone, behind one object)")
p.add_argument("--fuse-w1", action="store_true",
               help="update this rank's W1 rows inside the backward SpMM (optim.Adam.fuse_into_backward; takes hidden > 128)")
args = p.parse_args()

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
dev_index = args.device if args.device is not None else int(os.environ.get("LOCAL_RANK", 0))
th.cuda.set_device(dev_index)
dev = th.device("cuda", dev_index)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
# RCCL on a high-priority stream: the exchange must be scheduled while the local SpMM grids fill the CUs
init_process_group(args.backend, dev, rank=rank, world_size=world)

seed, lr, dropout, n_classes = 44, 0.05, 0.5, (8 if args.narrow else 6)
th.manual_seed(seed + rank)                        # dropout streams differ per rank
if rank == 0:
    docs, y = synth.synthetic_corpus(args.docs, 4000, n_classes=n_classes, seed=seed)
    perm = np.random.default_rng(seed).permutation(len(docs))
    g = Text2GraphTransformer(min_df=5, window_size=20, rm_stopwords=False).fit_transform(
        docs, y, test_idx=perm[:len(docs) // 10], val_idx=perm[len(docs) // 10:len(docs) // 5])
    meta = th.tensor([g.x.shape[0], g.edge_index.shape[1], g.n_vocab], device=dev)
else:
    meta = th.zeros(3, dtype=th.long, device=dev)
dist.broadcast(meta, 0)
N, E, V = (int(v) for v in meta.tolist())
coo = g.edge_index.t().contiguous().to(dev) if rank == 0 else th.empty(E, 2, dtype=th.long, device=dev)
attr = g.edge_attr.to(dev) if rank == 0 else th.empty(E, dtype=th.float32, device=dev)
labels = g.y.to(dev) if rank == 0 else th.empty(N, dtype=th.long, device=dev)
masks = th.stack([g.train_mask, g.val_mask, g.test_mask]).to(dev).to(th.uint8) if rank == 0 \
    else th.empty(3, N, dtype=th.uint8, device=dev)
for t in (coo, attr, labels, masks):
    dist.broadcast(t, 0)

sg = ShardedGraph(coo.t(), attr, N, hubs=th.arange(N, device=dev) < V)
gcn = ShardedGCN(sg, N, n_classes, n_hidden_gcn=args.hidden, dropout=dropout, narrow_exchange=args.narrow).to(dev)
with th.no_grad():                                  # glorot over the FULL (N, h) matrix
    a = (6.0 / (N + args.hidden)) ** 0.5
    gcn.weights[0].uniform_(-a, a).mul_(sg.real.unsqueeze(1))
y_l = sg.scatter_rows(labels)
train_l, val_l, test_l = (sg.scatter_rows(masks[i].bool()) for i in range(3))
optimizer = optim.Adam(gcn.parameters(), lr=lr, amsgrad=True)
if args.fuse_w1:
    optimizer.fuse_into_backward(gcn.weights[0])    # same steps, no [n_local, h] gradient, no optimizer pass over W1's regular rows


def accuracy(logits, mask):
    hit = th.stack([(logits[mask].argmax(1) == y_l[mask]).sum(), mask.sum()]).double()
    dist.all_reduce(hit)
    return (hit[0] / hit[1]).item()


if args.loop_object:
    from pytextgcn_amd.sharded import FlatLoop
    th.cuda.synchronize()
    t0 = time.time()
    with FlatLoop(gcn, y_l, train_l, val_l, optimizer=optimizer) as loop:
        for epoch in range(args.epochs):
            loss_, val_loss_, pred_val, _ = loop.epoch()              # global losses; class ids of this rank's rows
            hit = th.tensor([float((pred_val == y_l[val_l].cpu().numpy()).sum()), float(len(pred_val))], device=dev).double()
            dist.all_reduce(hit)
            if rank == 0 and (epoch % 10 == 0 or epoch == args.epochs - 1):
                print(f"[{epoch + 1:3d}] loss: {loss_: .3f}, val_loss: {val_loss_: .3f}, val accuracy: "
                      f"{(hit[0] / hit[1]).item(): .3f}", flush=True)
else:
    # --rows: kept tensors (the restricted operators are cached under them); every rank passes a mask in the same call
    rows_train = train_l if args.rows else None
    rows_eval = (train_l | val_l) if args.rows else None
    if args.rows:
        sg.prepare_rows(rows_train), sg.prepare_rows(rows_eval)       # collective, once; the forward pass only looks them up
    th.cuda.synchronize()
    t0 = time.time()
    for epoch in range(args.epochs):
        gcn.train()
        loss = sharded_cross_entropy(sg, gcn(rows=rows_train), y_l, train_l)
        optimizer.zero_grad(set_to_none=True)
        loss.backward()
        gcn.sync_grads()
        optimizer.step()
        gcn.eval()
        with th.no_grad():
            logits = gcn(rows=rows_eval)
            val_loss = sharded_cross_entropy(sg, logits, y_l, val_l)  # flat_amazon.py:110 (this rank's share of the mean)
            acc_val = accuracy(logits, val_l)
        total = th.stack([loss.detach(), val_loss.detach()])
        dist.all_reduce(total)
        if rank == 0 and (epoch % 10 == 0 or epoch == args.epochs - 1):
            print(f"[{epoch + 1:3d}] loss: {total[0].item(): .3f}, val_loss: {total[1].item(): .3f}, val accuracy: {acc_val: .3f}",
                  flush=True)
th.cuda.synchronize()
with th.no_grad():
    acc_test = accuracy(gcn(), test_l)
if rank == 0:
    print(f"{world} rank(s): {args.epochs} epochs in {time.time() - t0:.2f} s; test accuracy {acc_test:.3f}")
dist.barrier()
dist.destroy_process_group()
