"""Text2GraphTransformer: drop-in for `textgcn.Text2GraphTransformer`
(textgcn/lib/text2graph.py:49-247; exported at textgcn/__init__.py:1).

Same constructor signature and defaults (text2graph.py:50-52), same `fit_transform(X, y, test_idx,
val_idx, hierarchy_feats)` contract (:88-92), same `load_graph`, `vocabulary`, `node_feats`; the
result is a `pytextgcn_amd.Data` with the fields the reference builds at :192-193 -- node numbering
words [0, V) then documents, edge order [word-word interleaved | doc->word | word->doc],
`edge_index = coo.T` (a non-contiguous view), sparse identity features, boolean masks, pseudo
label 0 on word nodes.  What changes underneath:
  * the word-word PMI edges come from the GPU builder (pytextgcn_amd.graphbuilder -> libtgcn.so)
    instead of the Cython module (text2graph.py:156-160);
  * the occurrence and TF-IDF matrices stay sparse (the reference densifies both, :131,145 --
    n_docs x n_vocab float64); values are identical;
  * node ids stay integers (the reference round-trips them through float32, :169-170, exact only
    below 2^24 nodes); gradients are disabled with a context manager instead of the global switch
    at :114,203; `rm_stopwords=True` uses nltk's English list when nltk is installed and otherwise
    an embedded copy of that list (the reference downloads it at construction, :85).
"""
from __future__ import annotations

import glob
import os
import pickle
import re
import time
from typing import Dict, List, Optional, Union

import numpy as np
import torch as th
from scipy import sparse as sp
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.feature_extraction.text import CountVectorizer, TfidfTransformer

from . import graphbuilder
from .data import Data

# nltk.RegexpTokenizer(r"\w+") compiles its pattern with exactly these flags and calls findall
_WORD = re.compile(r"\w+", re.UNICODE | re.MULTILINE | re.DOTALL)

# nltk.corpus.stopwords.words('english') (NLTK data, 179 entries), for hosts without nltk_data
_NLTK_ENGLISH = (
    "i me my myself we our ours ourselves you you're you've you'll you'd your yours yourself "
    "yourselves he him his himself she she's her hers herself it it's its itself they them their "
    "theirs themselves what which who whom this that that'll these those am is are was were be "
    "been being have has had having do does did doing a an the and but if or because as until "
    "while of at by for with about against between into through during before after above below "
    "to from up down in out on off over under again further then once here there when where why "
    "how all any both each few more most other some such no nor not only own same so than too "
    "very s t can will just don don't should should've now d ll m o re ve y ain aren aren't "
    "couldn couldn't didn didn't doesn doesn't hadn hadn't hasn hasn't haven haven't isn isn't ma "
    "mightn mightn't mustn mustn't needn needn't shan shan't shouldn shouldn't wasn wasn't weren "
    "weren't won won't wouldn wouldn't").split()


def _english_stopwords() -> set:
    try:
        import nltk
        return set(nltk.corpus.stopwords.words("english"))
    except Exception:
        return set(_NLTK_ENGLISH)


def _encode_input(X: List[str], n_jobs, vocabulary: Dict[str, int], verbose, n_docs, max_len):
    """text2graph.py:20-46: `\\w+` tokens, lower-cased, in-vocabulary only, cut to max_len, padded
    with -1 to the longest document; int32 [n_docs, max_sent_len]."""
    if verbose > 0:
        print("text2graph: tokenising (\\w+, lower case, in-vocabulary words only)")
    sl = slice(None) if max_len is None else slice(max_len)

    def encode(chunk):
        return [[vocabulary[t] for t in (x.lower() for x in _WORD.findall(doc)) if t in vocabulary][sl]
                for doc in chunk]

    if n_jobs is not None and n_jobs != 1 and len(X) >= 50000:   # below that the pool start-up costs more
        import joblib as jl                       # the reference tokenises with joblib too (:28,40)
        step = max(500, len(X) // (8 * max(1, n_jobs if n_jobs > 0 else 8)))
        parts = jl.Parallel(n_jobs=n_jobs)(jl.delayed(encode)(X[i:i + step]) for i in range(0, len(X), step))
        docs = [d for part in parts for d in part]
    else:
        docs = encode(X)
    max_sent_len = max(map(len, docs)) if docs else 0
    if verbose > 1:
        print(f"text2graph: longest document has {max_sent_len} tokens")
    out = np.full((len(docs), max(max_sent_len, 1)), -1, dtype=np.int32)
    for i, d in enumerate(docs):
        out[i, :len(d)] = d
    assert out.shape[0] == n_docs
    return out, max(max_sent_len, 1)


class Text2GraphTransformer(BaseEstimator, TransformerMixin):
    def __init__(self, min_df: Union[int, float] = 5, window_size: int = 20, save_path: str = None,
                 n_jobs: int = 1, max_df=1.0, verbose=0, rm_stopwords=True, sparse_features=True,
                 max_length: Optional[int] = None):
        assert min_df > 0                      # same failure mode as the reference (text2graph.py:77)
        # sklearn estimator convention: one attribute per constructor argument, same name
        self.min_df, self.window_size, self.save_path = min_df, window_size, save_path
        self.n_jobs, self.max_df, self.verbose = n_jobs, max_df, verbose
        self.rm_stopwords, self.sparse_features, self.max_length = rm_stopwords, sparse_features, max_length
        self.input, self.cv = None, None
        self.stop_words = _english_stopwords() if rm_stopwords else None

    def _say(self, level: int, msg: str) -> None:
        if self.verbose > level:
            print("text2graph: " + msg)

    def fit_transform(self, X: Union[List[str], str], y=None, test_idx=None, val_idx=None,
                      hierarchy_feats: Union[th.Tensor, None] = None) -> Data:
        with th.no_grad():
            return self._fit_transform(X, y, test_idx, val_idx, hierarchy_feats)

    def _fit_transform(self, X, y, test_idx, val_idx, hierarchy_feats) -> Data:
        test_idx = th.as_tensor([] if test_idx is None else test_idx, dtype=th.long)
        if y is not None:
            y = th.as_tensor(y, dtype=th.long)
        if not isinstance(X, list):                 # a directory of *.txt files (text2graph.py:110-117)
            self._say(0, f"reading *.txt under {X}")
            paths = glob.glob(os.path.join(X, "*.txt"))
            X = [open(path, "r").read() for path in paths]
        self.input = X
        stop = None if self.stop_words is None else sorted(self.stop_words)
        self.cv = CountVectorizer(stop_words=stop, min_df=self.min_df, max_df=self.max_df)
        occ = self.cv.fit_transform(self.input).tocsr()                       # stays sparse
        occ.sort_indices()
        n_docs, n_vocabs = occ.shape
        self._say(1, f"{n_docs} documents, vocabulary of {n_vocabs} words")
        self.n_docs_, self.n_vocabs_, self.n_nodes_ = n_docs, n_vocabs, n_docs + n_vocabs
        tokens, self.max_sent_len_ = _encode_input(self.input, self.n_jobs, self.cv.vocabulary_,
                                                   self.verbose, n_docs, self.max_length)
        self._say(0, "document-word edges (TF-IDF)")
        tfidf = TfidfTransformer().fit_transform(occ).tocsr()
        tfidf.sort_indices()
        # th.nonzero of the dense matrix (text2graph.py:148) = CSR order with sorted column indices
        doc_ids = np.repeat(np.arange(n_docs, dtype=np.int64), np.diff(occ.indptr))
        word_ids = occ.indices.astype(np.int64)
        if tfidf.nnz == occ.nnz and np.array_equal(tfidf.indices, occ.indices):
            dw_weight = tfidf.data
        else:                                                                  # explicit zeros etc.
            dw_weight = np.asarray(tfidf[doc_ids, word_ids]).ravel()
        self._say(0, "word-word edges (PMI over sliding windows)")
        ww_coo, ww_w = graphbuilder.compute_word_word_edges(tokens, n_vocabs, n_docs, self.max_sent_len_,
                                                            self.window_size, self.n_jobs, self.verbose)
        n_ww, n_dw = ww_coo.shape[0], doc_ids.shape[0]
        coo = np.empty((n_ww + 2 * n_dw, 2), dtype=np.int64)
        coo[:n_ww] = ww_coo
        coo[n_ww:n_ww + n_dw, 0], coo[n_ww:n_ww + n_dw, 1] = doc_ids + n_vocabs, word_ids   # doc -> word
        coo[n_ww + n_dw:, 0], coo[n_ww + n_dw:, 1] = word_ids, doc_ids + n_vocabs           # word -> doc
        dw32 = dw_weight.astype(np.float32)                                    # f64 -> f32 as `.float()` at :192
        edge_weights = np.concatenate([ww_w, dw32, dw32])
        self._say(0, f"{coo.shape[0]} directed edges in total")
        feats = self.node_feats(hierarchy_feats) if self.sparse_features else th.eye(self.n_nodes_)
        # masks over all nodes: documents listed in test_idx / val_idx, every other DOCUMENT trains,
        # word nodes belong to no split (text2graph.py:180-188)
        def doc_mask(idx):
            m = th.zeros(self.n_nodes_, dtype=th.bool)
            if idx is not None:
                m[th.as_tensor(idx, dtype=th.long) + n_vocabs] = True
            return m
        test_mask, val_mask = doc_mask(test_idx), doc_mask(val_idx)
        train_mask = ~(test_mask | val_mask)
        train_mask[:n_vocabs] = False
        y_nodes = th.zeros(self.n_nodes_, dtype=th.long)        # pseudo label 0 on word nodes
        if y is not None:
            y_nodes[n_vocabs:] = y
        g = Data(x=feats.float(), edge_index=th.from_numpy(coo).T, edge_attr=th.from_numpy(edge_weights),
                 y=y_nodes, test_mask=test_mask, train_mask=train_mask, val_mask=val_mask,
                 n_vocab=n_vocabs)
        if self.save_path is not None:              # pickle next to the reference's naming (text2graph.py:195-204)
            os.makedirs(self.save_path, exist_ok=True)
            target = os.path.join(self.save_path, f"TGData_{time.time()}.p")
            with open(target, "wb") as out:
                pickle.dump(g, out)
            print(f"text2graph: graph pickled to {target}")
        return g

    @staticmethod
    def load_graph(save_path):
        if not os.path.exists(save_path):
            raise FileNotFoundError(f"no pickled graph at {save_path}")
        with open(save_path, "rb") as src:
            return pickle.load(src)

    @property
    def vocabulary(self) -> Dict[str, int]:
        return self.cv.vocabulary_

    def node_feats(self, hierarchy_feats):
        """Sparse [I_N | H] (text2graph.py:226-246): H holds `hierarchy_feats` on document rows."""
        n = self.n_nodes_
        ar = th.arange(n)
        idx, val, width = [th.stack([ar, ar])], [th.ones(n)], n
        if hierarchy_feats is not None:
            hf = th.as_tensor(np.asarray(hierarchy_feats), dtype=th.float)
            nz = th.nonzero(hf)
            idx.append(th.stack([nz[:, 0] + self.n_vocabs_, nz[:, 1] + n]))
            val.append(hf[nz[:, 0], nz[:, 1]])
            width = n + hf.shape[1]
        return th.sparse_coo_tensor(th.cat(idx, 1), th.cat(val), size=(n, width),
                                    dtype=th.float).coalesce()
