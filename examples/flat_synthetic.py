#!/usr/bin/env python3
"""flat_amazon.py / flat_dbpedia.py re-played on pytextgcn_amd: same plumbing (corpus ->
Text2GraphTransformer -> Data -> GCN(graph) -> CE on train_mask -> Adam(amsgrad) -> eval, metrics on
the host), same hyper-parameter names, on a synthetic corpus because the Amazon / DBpedia CSVs are
not in the reference tree (.MISSING_LARGE_BLOBS:1-3).  Line references: flat_amazon.py.

    python examples/flat_synthetic.py [--docs 5000] [--epochs 50] [--fused] [--preset amazon|dbpedia]

`--preset dbpedia` takes flat_dbpedia.py's settings instead (flat_dbpedia.py:20-34,70-71,80: dropout 0.5, max_df 0.4, window 5,
documents cut to 15 tokens, hidden width 32, a separate validation split appended to the training documents) with a class count
that is not a multiple of 4 (DBpedia's l3 has 219: odd widths take the padded buffers and column-group products).
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
p.add_argument("--preset", choices=["amazon", "dbpedia"], default="amazon", help="hyper-parameters of flat_amazon.py or of flat_dbpedia.py")
p.add_argument("--reorder", action="store_true",
               help="pytextgcn_amd.reorder_documents: lay the document nodes out by clusters found from the graph (a corpus "
                    "with topical locality whose file is not sorted by class gathers fewer distinct word rows per stretch)")
args = p.parse_args()

seed, lr, min_df = 44, 0.05, 5                                         # :22-35,66 (min_df 100 at DBpedia's 240 k documents)
if args.preset == "amazon":
    dropout, window_size, max_df, max_length, n_hidden, n_classes = 0.7, 20, 0.7, None, 100, 6
else:                                                                  # flat_dbpedia.py:20-34,70-71,80
    dropout, window_size, max_df, max_length, n_hidden, n_classes = 0.5, 5, 0.4, 15, 32, 19
np.random.seed(seed)
th.manual_seed(seed)
docs, y = synth.synthetic_corpus(args.docs, 3000, n_classes=n_classes, seed=seed)
if args.preset == "amazon":
    perm = np.random.permutation(len(docs))
    test_idx, val_idx = perm[:len(docs) // 10], perm[len(docs) // 10:len(docs) // 5]
else:                                       # flat_dbpedia.py:54-66: train, then the validation file, then the test file
    n = len(docs)
    val_idx, test_idx = np.arange(n - n // 5, n - n // 10), np.arange(n - n // 10, n)

t0 = time.time()
t2g = Text2GraphTransformer(n_jobs=8, min_df=min_df, window_size=window_size, rm_stopwords=False, verbose=1, max_df=max_df,
                            **({} if max_length is None else {"max_length": max_length}))
g = t2g.fit_transform(docs, y, test_idx=test_idx, val_idx=val_idx)     # :66-70
print(f"graph: {g}  ({time.time() - t0:.2f} s)")
if args.reorder:
    from pytextgcn_amd import reorder_documents
    g, perm = reorder_documents(g)     # same graph, other numbering: the loop below addresses nodes through the masks only

gcn = GCN(g.x.shape[1], len(np.unique(y)), n_hidden_gcn=n_hidden, dropout=dropout)   # :80
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
