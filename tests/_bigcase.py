"""BASELINE-size synthetic graphs (configs c4 / c5) and what the ORACLE makes of them, computed once per test session:
the oracle's normalisation of a 50 M / 200 M-edge graph is tens of seconds of host work that several GPU tests share.
Test infrastructure (imports oracle/); nothing under pytextgcn_amd/ imports this."""
import torch

from oracle import gcn_oracle as O
from pytextgcn_amd import synth

SHAPES = {"c4": (2_000_000, 50_000_000), "c5": (8_000_000, 200_000_000)}


class BigCase:
    def __init__(self, name, dev):
        self.name, self.dev = name, dev
        self.N, self.E = SHAPES[name]
        self._g = self._coo = self._csr = self._w64 = None

    @property
    def g(self):
        """The graph on the device (seed 44: bench.py's instance)."""
        if self._g is None:
            if self.name == "c5":
                self._g = synth.power_law_graph(self.N, self.E, seed=44, device=self.dev)
            else:
                self._g = synth.word_doc_graph(self.N, self.E, seed=44, device=self.dev, features="none")
        return self._g

    def oracle_coo(self):
        """(target, source, w_hat) of the oracle's add_remaining_self_loops + gcn_norm (PyG-1.6.3, what the reference
        runs at textgcn/lib/models.py:11-20), float32, in PyG's order: the edges, then one loop per node."""
        if self._coo is None:
            g = self.g
            self._coo = O.normalized_coo(g.edge_index.cpu(), g.edge_attr.cpu(), self.N)
        return self._coo

    def truth_w64(self):
        """FLOAT64 ground truth of the same weights, same order (degrees summed in float64)."""
        if self._w64 is None:
            g = self.g
            ei, w = g.edge_index.cpu(), g.edge_attr.cpu().double()
            assert bool((ei[0] != ei[1]).all())
            deg = torch.ones(self.N, dtype=torch.float64).index_add_(0, ei[1], w)
            dis = deg.pow(-0.5)
            tgt, src, _ = self.oracle_coo()
            self._w64 = torch.cat([w, torch.ones(self.N, dtype=torch.float64)]) * dis[src] * dis[tgt]
        return self._w64

    def oracle_csr(self):
        """The oracle's operator as CSR with the entries of a row sorted by column, ties in edge order (the order the
        plan documents): (rowptr int64, col int32, val float32, order) -- `order` maps CSR positions to oracle_coo()."""
        if self._csr is None:
            tgt, src, nw = self.oracle_coo()
            order = torch.argsort(tgt * self.N + src, stable=True)
            rp = torch.zeros(self.N + 1, dtype=torch.int64)
            rp[1:] = torch.bincount(tgt, minlength=self.N).cumsum(0)
            self._csr = (rp, src[order].to(torch.int32), nw[order], order)
        return self._csr

    def release_device(self):
        self._g = None
