"""CPU oracle, part 4: restatement of textgcn/lib/text2graph.py (Text2GraphTransformer) --
TEST INFRASTRUCTURE (only tests/, smoke() and bench.py's cpu_baseline leg may import it).

Follows the reference line by line, dense intermediates included:
  _encode_input        text2graph.py:20-46   (nltk.RegexpTokenizer(r"\\w+") == re.findall with
                       UNICODE|MULTILINE|DOTALL, which is how nltk implements it; nltk is not
                       installed here)
  fit_transform        text2graph.py:88-204  (CountVectorizer -> dense occurrence matrix :130-131,
                       TfidfTransformer().todense() :145, th.nonzero :148, word-word edges :156-160,
                       weights :162-166, coo :167-171, features :179, masks/labels :180-191)
  node_feats           text2graph.py:226-246
The word-word edges come from oracle/graphbuilder_oracle.c (pinned by the reference's golden vector
and by oracle/_ref).  The rest is PARITY UNPINNED by the reference: its only test of this class
(textgcn/test/test_text2graph.py:10-35) has no assertions and cannot run (3 labels for 4 documents),
and `import textgcn` fails here (torch_geometric, nltk missing: ordinary ModuleNotFoundError).
Two reference quirks are NOT reproduced: the float32 round trip of the node ids at :169-170 (exact
only below 2^24 nodes) and the global `th.set_grad_enabled` toggling at :114,203.
"""
from __future__ import annotations

import re
from typing import List, Optional

import numpy as np
import torch as th
from scipy import sparse as sp
from sklearn.feature_extraction.text import CountVectorizer, TfidfTransformer

from . import graphbuilder_py

_WORD = re.compile(r"\w+", re.UNICODE | re.MULTILINE | re.DOTALL)


class Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def encode_input(docs: List[str], vocabulary: dict, max_len: Optional[int]):
    """text2graph.py:20-46: tokenise, lower-case, keep in-vocabulary words, truncate, pad with -1."""
    sl = slice(None) if max_len is None else slice(max_len)
    toks = [[x.lower() for x in _WORD.findall(doc) if x.lower() in vocabulary][sl] for doc in docs]
    max_sent_len = max(map(len, toks))
    X = np.array([[vocabulary[w] for w in doc] + [-1] * (max_sent_len - len(doc)) for doc in toks],
                 dtype=np.int32).reshape(len(docs), max_sent_len)
    return X, max_sent_len


def node_feats(n_nodes: int, n_vocabs: int, hierarchy_feats) -> th.Tensor:
    feat = sp.identity(n_nodes)
    if hierarchy_feats is not None:
        hf = np.zeros([n_nodes, hierarchy_feats.shape[1]])
        hf[n_vocabs:, :] = hierarchy_feats
        feat = sp.hstack([feat, sp.coo_matrix(hf)])
    ind0, ind1, vals = sp.find(feat)
    inds = th.stack((th.from_numpy(ind0.astype(np.int64)), th.from_numpy(ind1.astype(np.int64))))
    return th.sparse_coo_tensor(inds, th.from_numpy(vals), size=feat.shape, dtype=th.float)


def fit_transform(docs: List[str], y=None, test_idx=None, val_idx=None, hierarchy_feats=None,
                  min_df=5, window_size=20, max_df=1.0, stop_words=None, sparse_features=True,
                  max_length=None) -> Bag:
    test_idx = th.LongTensor([] if test_idx is None else list(test_idx))
    cv = CountVectorizer(stop_words=stop_words, min_df=min_df, max_df=max_df)
    occurrence_mat = cv.fit_transform(docs).toarray()                         # :130-131 (dense)
    n_docs, n_vocabs = occurrence_mat.shape
    n_nodes = n_docs + n_vocabs
    X, max_sent_len = encode_input(docs, cv.vocabulary_, max_length)          # :139
    tfidf_mat = th.from_numpy(np.asarray(TfidfTransformer().fit_transform(occurrence_mat).todense()))  # :145
    docu_coo = th.nonzero(th.from_numpy(occurrence_mat))                      # :148
    docu_coo_sym = th.flip(docu_coo, dims=[1])                                # :150
    edges_coo, edge_ww_weights = map(th.from_numpy,
                                     graphbuilder_py.compute_word_word_edges(X, n_vocabs, window_size))  # :156-160
    edge_weights = th.cat([edge_ww_weights.double(), tfidf_mat[tuple(docu_coo.T)],
                           tfidf_mat[tuple(docu_coo.T)]])                      # :162-166 (promotes to f64)
    coo = th.vstack([edges_coo.long(), docu_coo + th.tensor([n_vocabs, 0]),
                     docu_coo_sym + th.tensor([0, n_vocabs])]).long()         # :167-171 (exact ints)
    feats = node_feats(n_nodes, n_vocabs, hierarchy_feats) if sparse_features else th.eye(n_nodes)  # :179
    test_mask = th.zeros(n_nodes, dtype=th.bool)
    val_mask = th.zeros(n_nodes, dtype=th.bool)
    test_mask[test_idx + n_vocabs] = 1                                         # :183
    if val_idx is not None:
        val_mask[th.LongTensor(list(val_idx)) + n_vocabs] = 1                  # :185
    train_mask = th.logical_not(th.logical_or(test_mask, val_mask))            # :187
    train_mask[:n_vocabs] = 0                                                  # :188
    y_nodes = th.zeros(n_nodes, dtype=th.long)                                 # :190
    if y is not None:
        y_nodes[n_vocabs:] = th.as_tensor(y, dtype=th.long)                    # :191
    return Bag(x=feats.float(), edge_index=coo.T, edge_attr=edge_weights.float(), y=y_nodes,
               test_mask=test_mask, train_mask=train_mask, val_mask=val_mask, n_vocab=n_vocabs,
               vocabulary=dict(cv.vocabulary_), tokens=X, max_sent_len=max_sent_len)
