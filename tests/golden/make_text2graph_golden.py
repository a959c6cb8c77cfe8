"""Golden fixture for Text2GraphTransformer from the CPU oracle (oracle/text2graph_oracle.py, which
restates textgcn/lib/text2graph.py with sklearn's CountVectorizer / TfidfTransformer -- the same
library calls the reference makes -- and the pinned graph-builder oracle).  Data only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import text2graph_oracle as TO  # noqa: E402

DOCS = [
    "The cat sat on the mat and the cat ate the fish.",
    "Dogs and cats are pets; the dog barks at the cat!",
    "A mat is on the floor, the floor is flat and the mat is red.",
    "Pets like cats and dogs eat food, fish is food for a cat",
    "Food for the cat; food for the dog. Dogs eat fish?",
    "the floor of the room is flat, a red mat and a red fish",
    "room with a dog: the dog sat on the floor of the room",
    "fish, fish, FISH -- cats eat fish and dogs eat food",
]
Y = [0, 1, 2, 1, 1, 2, 2, 0]


def main():
    b = TO.fit_transform(DOCS, y=Y, test_idx=[6, 7], val_idx=[5], min_df=2, window_size=5,
                         max_df=0.9, stop_words=["the", "a", "is", "and", "of"])
    hf = np.arange(len(DOCS) * 3, dtype=np.float32).reshape(len(DOCS), 3) % 4
    bh = TO.fit_transform(DOCS, y=Y, test_idx=[0], hierarchy_feats=hf, min_df=1, window_size=20,
                          max_length=6)
    xh = bh.x.coalesce()
    np.savez(os.path.join(HERE, "text2graph.npz"), docs=np.array(DOCS), y=np.array(Y),
             edge_index=b.edge_index.numpy(), edge_attr=b.edge_attr.numpy(), y_nodes=b.y.numpy(),
             test_mask=b.test_mask.numpy(), val_mask=b.val_mask.numpy(), train_mask=b.train_mask.numpy(),
             n_vocab=b.n_vocab, tokens=b.tokens, vocab_words=np.array(sorted(b.vocabulary)),
             h_edge_index=bh.edge_index.numpy(), h_edge_attr=bh.edge_attr.numpy(), h_tokens=bh.tokens,
             h_x_indices=xh.indices().numpy(), h_x_values=xh.values().numpy(), h_x_shape=np.array(xh.shape),
             hierarchy_feats=hf)
    print("text2graph.npz:", b.n_vocab, "words,", b.edge_index.shape[1], "edges;", bh.n_vocab, "words,",
          bh.edge_index.shape[1], "edges (hierarchy case)")


if __name__ == "__main__":
    main()
