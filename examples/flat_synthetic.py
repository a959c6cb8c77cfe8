#!/usr/bin/env python3
"""flat_amazon.py / flat_dbpedia.py re-played on pytextgcn_amd: same plumbing (corpus ->
Text2GraphTransformer -> Data -> GCN(graph) -> CE on train_mask -> Adam(amsgrad) -> eval, metrics on
the host), same hyper-parameter names, on a synthetic corpus because the Amazon / DBpedia CSVs are
not in the reference tree (.MISSING_LARGE_BLOBS:1-3).  Line references: flat_amazon.py.

    python examples/flat_synthetic.py [--docs 5000] [--epochs 50] [--fused]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch as th
from sklearn.metrics import accuracy_score, f1_score

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytextgcn_amd import Text2GraphTransformer, synth  # noqa: E402
from pytextgcn_amd.models import GCN  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--docs", type=int, default=5000)
p.add_argument("--epochs", type=int, default=50)
p.add_argument("--fused", action="store_true", help="pytextgcn_amd.train.FlatLoop: the same epoch with every switch of the package instead of torch's CE / Adam / dropout")
p.add_argument("--reorder", action="store_true",
               help="pytextgcn_amd.reorder_documents: lay the document nodes out by clusters found from the graph (a corpus "
                    "with topical locality whose file is not sorted by class gathers fewer distinct word rows per stretch)")
args = p.parse_args()

seed, lr, dropout, window_size, min_df = 44, 0.05, 0.7, 20, 5          # :22-35,66
np.random.seed(seed)
th.manual_seed(seed)
docs, y = synth.synthetic_corpus(args.docs, 3000, n_classes=6, seed=seed)
perm = np.random.permutation(len(docs))
test_idx, val_idx = perm[:len(docs) // 10], perm[len(docs) // 10:len(docs) // 5]

t0 = time.time()
t2g = Text2GraphTransformer(n_jobs=8, min_df=min_df, window_size=window_size, rm_stopwords=False, verbose=1)
g = t2g.fit_transform(docs, y, test_idx=test_idx, val_idx=val_idx)     # :66-70
print(f"graph: {g}  ({time.time() - t0:.2f} s)")
if args.reorder:
    from pytextgcn_amd import reorder_documents
    g, perm = reorder_documents(g)     # same graph, other numbering: the loop below addresses nodes through the masks only

gcn = GCN(g.x.shape[1], len(np.unique(y)), n_hidden_gcn=100, dropout=dropout)   # :80
criterion = th.nn.CrossEntropyLoss(reduction="mean")                   # :82
device = th.device("cuda")                                             # :84
gcn = gcn.to(device).float()                                           # :85
g = g.to(device)                                                       # :86
y_val, y_train = g.y.cpu()[g.val_mask.cpu()], g.y.cpu()[g.train_mask.cpu()]
th.cuda.synchronize()
t0 = time.time()
if args.fused:
    # the same epoch behind one object: fused loss / optimizer / dropout kernels, W1's update inside the backward SpMM, layer
    # 1's activation kept from the evaluation pass, only the logits rows that are read, predictions (not logits) to the host
    from pytextgcn_amd.train import FlatLoop
    with FlatLoop(gcn, g, lr=lr) as loop:                              # :89 + :99-117
        for epoch in range(args.epochs):
            loss, val_loss, pred_val, pred_train = loop.epoch()        # (val_loss: :110, computed by the same fused pass)
            f1_val = f1_score(y_val, pred_val, average="macro")
            acc_train = accuracy_score(y_train, pred_train)
            if epoch % 10 == 0 or epoch == args.epochs - 1:
                print(f"[{epoch + 1:3d}] loss: {loss: .3f}, val_loss: {val_loss: .3f}, training accuracy: {acc_train: .3f}, "
                      f"val_f1: {f1_val: .3f}")
else:
    optimizer = th.optim.Adam(gcn.parameters(), lr=lr, amsgrad=True)   # :89
    for epoch in range(args.epochs):                                   # :99-117
        gcn.train()
        outputs = gcn(g)[g.train_mask]
        loss = criterion(outputs, g.y[g.train_mask])
        optimizer.zero_grad(set_to_none=True)
        loss.backward()
        optimizer.step()
        gcn.eval()
        with th.no_grad():
            logits = gcn(g)
            val_loss = criterion(logits[g.val_mask], g.y[g.val_mask])  # :110
            pred_val = np.argmax(logits[g.val_mask].cpu().numpy(), axis=1)
            pred_train = np.argmax(logits[g.train_mask].cpu().numpy(), axis=1)
            f1_val = f1_score(y_val, pred_val, average="macro")
            acc_train = accuracy_score(y_train, pred_train)
        if epoch % 10 == 0 or epoch == args.epochs - 1:
            print(f"[{epoch + 1:3d}] loss: {loss.item(): .3f}, val_loss: {val_loss.item(): .3f}, training accuracy: "
                  f"{acc_train: .3f}, val_f1: {f1_val: .3f}")
th.cuda.synchronize()
print(f"{args.epochs} epochs in {time.time() - t0:.2f} s")
with th.no_grad():                                                     # :130-134
    pred_test = np.argmax(gcn(g)[g.test_mask].cpu().numpy(), axis=1)
    y_test = g.y.cpu()[g.test_mask.cpu()]
print(f"Test Accuracy: {accuracy_score(y_test, pred_test): .3f}  F1-Macro: {f1_score(y_test, pred_test, average='macro'): .3f}")
