"""1-D row partition of the GCN path across the GPUs of one node (one process per GPU, RCCL).

The reference is single-device (flat_amazon.py:84-86 `gcn.to(device)`, `g.to(device)`; no
torch.distributed anywhere), so this module has no counterpart to mirror: it partitions the
SAME arithmetic (GCNConv propagate, textgcn/lib/models.py:20, and its autograd) so that every rank
owns a block of rows of the operator, of W1 / H1 / logits and of the optimizer state.

Scheme ("Overlap / structure").  Nodes are split into HUBS (replicated operand:
for a TextGCN graph the V word nodes, whose feature block V x F is small) and REGULAR nodes (the
documents; their features never leave the owner).  Both classes are dealt to ranks by degree
(snake order), so every rank gets the same number of rows and of non-zeros.  One SpMM Y = M X is

    all-gather(X_hub shards)            ||  P = A_r @ X_reg(local)      hub rows, partial sums
    reduce-scatter(P) -> own hub rows   ||  Yb = B_r @ [X_hub ; X_reg(local)] (+ bias)
    Y_own = [Yb_hub + RS ; Yb_reg]

with two local operators per rank, both built from the globally normalised M:
    A_r  rows = all hubs, columns = own regular nodes   (entries M[hub, regular in rank r])
    B_r  rows = own hubs + own regular nodes, columns = all hubs + own regular nodes
so only 2 x |hubs| x F floats cross xGMI per SpMM instead of N x F, and both collectives overlap
with a local SpMM.  Edges between regular nodes of different ranks are not representable; with
`hubs=None` every node is a hub, A_r is empty and the scheme degenerates to the plain
all-gather of the row-sharded operand.  M^T uses the same operators when M is symmetric (TextGCN
graphs are, text2graph.py:148-171), otherwise a second pair is built from M^T.

Graphs WITHOUT hub structure (`hubs=None`, BASELINE config c5) have no A_r whose SpMM could cover the gather; there the
default exchange is the PIPELINE (`_Pipeline`): B_r is cut by the origin of its columns into an own-column block, which
starts at once on the rank's own rows, and K stage blocks; the referenced operand rows travel in K all-to-all stages
(every link busy in every stage), and block k is ADDED to the result (`tgcn_spmm_acc`) as soon as stage k has landed --
stage k + 1 is in flight under the compute of stage k ("run local part while halo is in flight").  Opt-in
(`set_pipeline(prefix="auto")`): the rows of the first slots of every rank -- dealt in degree order, read by nearly every
peer -- leave UNPACKED, as contiguous ranges of the operand; one packed stage carries the rest.

Local layout on every rank: rows [0, hp) = own hub shard, rows [hp, hp + rp) = own regular shard
(`owned` maps them to global node ids, -1 = padding row).
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor, nn

from .conv import glorot_


def prepare_hsa_env() -> dict:
    """The ONE place that decides the process environment of a multi-process GPU run; every rank passes it before its
    first HIP call (bench.py main(), `init_process_group`, the examples, the test workers).

    HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool supports only dmabuf IPC; with the legacy mode RCCL's
    P2P / shared-memory transports (and any device-tensor sharing between processes) fail with `hipIpcGetMemHandle:
    invalid argument`.  The image exports it already; a launcher that builds its own environment may have dropped it, so
    it is set here when absent -- which only helps while the HSA runtime has not been initialised yet in this process
    (it reads the variable once).  Returns what a bench record reports under `rccl`."""
    import os
    import warnings
    name = "HSA_ENABLE_IPC_MODE_LEGACY"
    before = os.environ.get(name)
    late = False
    if before is None:
        late = torch.cuda.is_initialized()
        os.environ[name] = "0"
        if late:
            warnings.warn(f"pytextgcn_amd.sharded: {name} was unset and the GPU runtime is already initialised in this "
                          "process; set it to 0 in the launching environment (RCCL between processes needs dmabuf IPC here)")
    return {name: os.environ[name], "was_set_by_the_launcher": before is not None, "set_after_hip_init": late}


def init_process_group(backend: str = "nccl", device: Optional[torch.device] = None, timeout_s: Optional[float] = None,
                       **kw) -> None:
    """torch.distributed.init_process_group for the 1-D partition.  With RCCL (backend "nccl") the communicator's
    kernels go on a HIGH-PRIORITY stream: the local SpMM grids fill all 256 CUs, and a collective enqueued at normal
    priority gets CUs only as SpMM waves drain -- the overlap `all-gather || A_r`, `reduce-scatter || B_r` that
    `ShardedGraph.spmm` is built around would run one phase after the other.  `timeout_s` bounds every collective (a
    dead peer then ends the process instead of hanging it)."""
    import datetime
    import warnings
    prepare_hsa_env()
    if timeout_s is not None:
        kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
    if backend != "nccl":
        dist.init_process_group(backend, **kw)
        return
    if device is not None:
        kw["device_id"] = device
    opts = None
    try:
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    except (AttributeError, TypeError):
        pass
    # a torch that spells an argument differently: drop what it refuses, one argument at a time, and SAY what was lost
    # (without the high-priority stream the exchange queues behind the SpMM grids instead of overlapping them)
    attempts = [dict(kw, pg_options=opts)] if opts is not None else []
    attempts.append(dict(kw))
    if "device_id" in kw:
        attempts.append({k: v for k, v in kw.items() if k != "device_id"})
    last = None
    for i, args in enumerate(attempts):
        try:
            dist.init_process_group("nccl", **args)
        except TypeError as e:
            last = e
            continue
        if opts is None or "pg_options" not in args:
            warnings.warn("pytextgcn_amd.sharded: this torch did not take the high-priority-stream option of the RCCL "
                          "group; the collectives of ShardedGraph.spmm will not overlap the local SpMMs")
        if "device_id" in kw and "device_id" not in args:
            warnings.warn("pytextgcn_amd.sharded: this torch did not take `device_id`; the communicator binds lazily")
        return
    raise last


def let_rccl_ranks_share_a_device(rank: int) -> None:
    """REHEARSAL aid for a box with fewer GPUs than ranks; call it in every rank BEFORE the process group is made.
    RCCL refuses two ranks whose (host hash, PCI bus id) agree ("Duplicate GPU detected").  Giving each process its
    own NCCL_HOSTID makes the ranks look like separate hosts, so the communicator connects them through the socket
    transport (loopback) instead of P2P / shared memory: a few GB/s, i.e. no statement about speed, but it is RCCL
    itself with world size > 1 -- the enqueue order of all_gather_into_tensor / reduce_scatter_tensor /
    all_to_all_single / batched send + recv on the communicator's stream, which gloo (synchronous, host-staged)
    cannot exercise.  Never used by a real one-rank-per-GPU run."""
    import os
    os.environ["NCCL_HOSTID"] = f"tgcn-shared-device-rank-{rank}"
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")


# --------------------------------------------------------------------------------------------------
# local-operator engine: the HIP library.  (tests inject a CPU engine built on the oracle to run the
# partition + exchange logic under gloo without a GPU; the package itself ships no CPU engine.)
# --------------------------------------------------------------------------------------------------
class HipEngine:
    def gcn_norm(self, edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int,
                 add_self_loops: int, degree_sum: str = "reference") -> Tuple[Tensor, Tensor]:
        """(dis, loop_w): deg^-1/2 per node (in-degree at the target incl. the self loop, inf -> 0) and the
        weight of every node's self loop, from the WHOLE edge list -- libtgcn.so `tgcn_gcn_norm`, which walks
        the edges in bounded chunks: no whole-graph plan, no nnz-sized temporaries.  It is the routine
        tgcn_plan_create itself runs, so the factors are bit for bit the single-device plan's."""
        from . import _lib
        from .plan import _require_cuda, _stream_ptr
        lib = _lib.load()
        _require_cuda(edge_index, "edge_index")
        if edge_index.dtype != torch.int64:
            edge_index = edge_index.long()
        dev = edge_index.device
        n_edges = edge_index.size(1)
        w = None
        if edge_weight is not None:
            w = edge_weight.detach().reshape(-1).float().contiguous()
        dis = torch.empty(num_nodes, dtype=torch.float32, device=dev)
        loop_w = torch.empty(num_nodes, dtype=torch.float32, device=dev)
        src, dst = edge_index[0], edge_index[1]
        with torch.cuda.device(dev):
            _lib.check(lib.tgcn_gcn_norm(
                num_nodes, n_edges, src.data_ptr() if n_edges else None, src.stride(0) if n_edges else 1,
                dst.data_ptr() if n_edges else None, dst.stride(0) if n_edges else 1,
                w.data_ptr() if w is not None else None, int(add_self_loops), _lib.DEGREE_SUMS[degree_sum],
                dis.data_ptr(), loop_w.data_ptr(),
                dev.index if dev.index is not None else torch.cuda.current_device(), _stream_ptr(dev)))
        return dis, loop_w

    def make_op(self, row, col, val, n_rows, n_cols):
        from .plan import GraphPlan
        return GraphPlan.from_coo(row, col, val, n_rows, n_cols, with_transpose=False)

    def colsum(self, g: Tensor) -> Tensor:
        from .plan import colsum
        return colsum(g)

    def xw(self, x: Tensor, w: Tensor) -> Tensor:
        from . import dense
        return dense.xw(x, w)                 # fp32 MFMA kernels, with autograd

    def xw_dropout(self, x: Tensor, w: Tensor, p: float) -> Tensor:
        from . import dense
        return dense.xw_dropout(x, w, p)      # dropout fused into the three GEMMs (dropped activation never stored)

    # the three layer-2 products one by one (pytextgcn_amd/narrow.py composes them itself); `keys`: which mask row a
    # matrix row is (tgcn_set_dropout_row_keys)
    def gemm_nn(self, a, b, p=0.0, seed=None, record_mask=False, keys=None, out=None):
        from . import dense
        return dense.gemm_nn(a, b, p, seed, record_mask=record_mask, keys=keys, out=out)

    def gemm_nt(self, a, b, p=0.0, seed=None, mask=None, keys=None):
        from . import dense
        return dense.gemm_nt(a, b, p, seed, note_colsums=True, mask=mask, keys=keys)

    def gemm_tn(self, a, g, p=0.0, seed=None, mask=None, keys=None):
        from . import dense
        return dense.gemm_tn(a, g, p, seed, mask, keys=keys)

    def xw_dropout_keyed(self, x: Tensor, w: Tensor, p: float, seed: Tensor, keys) -> Tensor:
        from . import dense
        return dense.xw_dropout(x, w, p, seed=seed, keys=keys)

    def masked_ce(self, logits: Tensor, y: Tensor, mask: Tensor, count: int, return_pred: bool = False):
        from .functional import masked_cross_entropy
        return masked_cross_entropy(logits, y, mask, count=count, return_pred=return_pred)

    # row movement of the exchange (libtgcn.so `tgcn_rows_*`, csrc/rows.hip).  The C ABI takes raw pointers: what
    # they must point at is asserted here (the library additionally skips indices outside the row counts it is given)
    @staticmethod
    def _rows_ok(x: Tensor, name: str) -> None:
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and (x.size(1) == 0 or x.stride(1) == 1)):
            raise TypeError(f"{name}: a float32 [rows, F] device matrix with unit column stride is required, got "
                            f"{x.dtype} {tuple(x.shape)} strides {tuple(x.stride())} on {x.device}")

    @staticmethod
    def _index_ok(idx: Tensor, like: Tensor, name: str, dtype=torch.int64) -> None:
        if not (idx.is_cuda and idx.device == like.device and idx.dtype == dtype and idx.is_contiguous()):
            raise TypeError(f"{name}: a contiguous {dtype} index tensor on {like.device} is required, got "
                            f"{idx.dtype} (contiguous: {idx.is_contiguous()}) on {idx.device}")

    def rows_gather(self, x: Tensor, idx: Tensor) -> Tensor:
        """x[idx] (rows), packed for sending."""
        from . import _lib
        from .plan import _stream_ptr
        self._rows_ok(x, "rows_gather: x")
        self._index_ok(idx, x, "rows_gather: idx")
        out = torch.empty(idx.numel(), x.size(1), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().tgcn_rows_gather(x.data_ptr(), x.stride(0), x.size(0), idx.data_ptr(), idx.numel(),
                                                x.size(1), out.data_ptr(), out.stride(0), _stream_ptr(x.device)))
        return out

    def rows_scatter_(self, y: Tensor, idx: Tensor, x: Tensor) -> None:
        """y[idx] = x (rows; idx distinct)."""
        from . import _lib
        from .plan import _stream_ptr
        self._rows_ok(x, "rows_scatter_: x")
        self._rows_ok(y, "rows_scatter_: y")
        self._index_ok(idx, y, "rows_scatter_: idx")
        if x.size(0) != idx.numel() or x.size(1) != y.size(1) or x.device != y.device:
            raise ValueError(f"rows_scatter_: {tuple(x.shape)} rows for {idx.numel()} indices into {tuple(y.shape)}")
        _lib.check(_lib.load().tgcn_rows_scatter(x.data_ptr(), x.stride(0), idx.data_ptr(), idx.numel(), x.size(1),
                                                 y.data_ptr(), y.stride(0), y.size(0), _stream_ptr(y.device)))

    def reduce_ranked_(self, y: Tensor, recv: Tensor, inv: Tensor, n_ranks: int, n: int, row0: int, step: int) -> None:
        """y[row0 + j * step] += sum over ranks q, in order, of recv[inv[q, j]] (inv < 0: nothing from q), j < n."""
        from . import _lib
        from .plan import _stream_ptr
        self._rows_ok(y, "reduce_ranked_: y")
        self._index_ok(inv, y, "reduce_ranked_: inv", torch.int32)
        if inv.numel() != n_ranks * n:
            raise ValueError(f"reduce_ranked_: the table has {inv.numel()} entries for {n_ranks} ranks x {n} rows")
        if recv.numel():
            self._rows_ok(recv, "reduce_ranked_: recv")
            if recv.size(1) != y.size(1) or recv.device != y.device:
                raise ValueError("reduce_ranked_: recv and y differ in width or device")
        _lib.check(_lib.load().tgcn_rows_reduce_ranked(
            recv.data_ptr() if recv.numel() else None, recv.stride(0) if recv.numel() else y.size(1),
            recv.size(0) if recv.numel() else 0, inv.data_ptr(), n_ranks, n, y.size(1), y.data_ptr(), y.stride(0),
            y.size(0), row0, step, _stream_ptr(y.device)))


class _Done:
    """A finished transfer (the staged exchanges complete before they return)."""
    @staticmethod
    def wait():
        return True


_DONE = _Done()


def _wrap64(v: int) -> int:
    """A 64-bit constant as the signed value torch.int64 holds."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


_HASH_K = [tuple(_wrap64(k) for k in ks) for ks in (
    (0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0xFF51AFD7ED558CCD),
    (0xD6E8FEB86659FD93, 0xA0761D6478BD642F, 0xE7037ED1A0B428DB, 0x8EBC6AF09C88C6E3))]


def _entry_hash(r: Tensor, c: Tensor, vbits: Tensor, ks) -> int:
    """Order-independent 64-bit fingerprint of a multiset of entries (r, c, bits(v)): the wrapping sum of a
    mixed per-entry hash (int64 arithmetic wraps; integer sums are exact in any order)."""
    x = r * ks[0] + c * ks[1] + vbits * ks[2]
    x = torch.bitwise_xor(x, x >> 29) * ks[3]
    x = torch.bitwise_xor(x, x >> 32)
    return int(x.sum().item())


class Partition:
    """Node -> (owner rank, slot) assignment; a pure function of (graph, world, hubs), so every
    rank computes the same one without communication."""
    _CHUNK = 1 << 24      # edges per pass

    def __init__(self, edge_index: Tensor, num_nodes: int, world: int, hubs: Optional[Tensor]):
        dev = edge_index.device
        N = num_nodes
        self.world, self.n_nodes = world, N
        if hubs is None:
            hub_mask = torch.ones(N, dtype=torch.bool, device=dev)
        elif hubs.dtype == torch.bool:
            hub_mask = hubs.to(dev)
        else:
            hub_mask = torch.zeros(N, dtype=torch.bool, device=dev)
            hub_mask[hubs.to(dev)] = True
        self.hub_mask = hub_mask
        # work a node brings to its owner: a regular node all of its edges (as a column of A_r and as
        # a row of B_r); a hub only its hub-hub edges (its other entries are computed, as partial
        # sums, by the ranks that own the regular endpoints).  The edge list is walked in chunks: nothing
        # of size E is ever materialised next to it.
        E = edge_index.size(1)
        deg_all = torch.zeros(N, dtype=torch.int64, device=dev)
        deg_hh = torch.zeros(N, dtype=torch.int64, device=dev)
        for lo in range(0, E, self._CHUNK):
            s, t = edge_index[0, lo:lo + self._CHUNK], edge_index[1, lo:lo + self._CHUNK]
            deg_all += torch.bincount(s, minlength=N) + torch.bincount(t, minlength=N)
            hh = hub_mask[s] & hub_mask[t]
            deg_hh += torch.bincount(s[hh], minlength=N) + torch.bincount(t[hh], minlength=N)
        self.owner = torch.empty(N, dtype=torch.int64, device=dev)
        self.slot = torch.empty(N, dtype=torch.int64, device=dev)
        sizes = []
        for mask, deg in ((hub_mask, deg_hh), (~hub_mask, deg_all)):
            ids = torch.nonzero(mask).flatten()
            order = ids[torch.argsort(deg[ids], descending=True, stable=True)]
            p = torch.arange(order.numel(), device=dev)
            rnd, j = p // world, p % world
            self.owner[order] = torch.where(rnd % 2 == 0, j, world - 1 - j)    # snake deal
            self.slot[order] = rnd
            sizes.append((order.numel() + world - 1) // world)
        self.hp, self.rp = sizes
        self.n_local = self.hp + self.rp
        # column / row numbering of the local operators
        self.hub_col = self.owner * self.hp + self.slot                # valid where hub_mask
        self.reg_col = world * self.hp + self.slot                     # valid where ~hub_mask
        # regular-regular edges must be rank-local
        bad = 0
        if not bool(hub_mask.all()):
            for lo in range(0, E, self._CHUNK):
                s, t = edge_index[0, lo:lo + self._CHUNK], edge_index[1, lo:lo + self._CHUNK]
                bad += int(((~hub_mask[s]) & (~hub_mask[t]) & (self.owner[s] != self.owner[t])).sum())
        if bad:
            raise ValueError(
                f"{bad} edges join regular (non-hub) nodes owned by different ranks; "
                "enlarge `hubs` (hubs=None replicates every node: plain all-gather partition)")

    def owned(self, rank: int) -> Tensor:
        """Global node id of every local row of `rank` (-1 for padding rows)."""
        out = torch.full((self.n_local,), -1, dtype=torch.int64, device=self.owner.device)
        mine = self.owner == rank
        hub = torch.nonzero(mine & self.hub_mask).flatten()
        reg = torch.nonzero(mine & ~self.hub_mask).flatten()
        out[self.slot[hub]] = hub
        out[self.hp + self.slot[reg]] = reg
        return out


class _Chunk:
    """One row chunk of A_r (hub slots s with s % K == k) and the lists of its pruned reduce-scatter."""
    __slots__ = ("op", "k", "ck", "n_own", "touch_rows", "touch_counts", "recv_counts", "recv_pos", "inv_halo", "inv_dense")

    def __init__(self, op, k, ck, n_own):
        self.op, self.k, self.ck, self.n_own = op, k, ck, n_own
        self.touch_rows = self.touch_counts = self.recv_counts = self.recv_pos = None
        self.inv_halo = self.inv_dense = None      # [W, n_own] int32: row of the receive buffer rank q's partial of row j sits in


class _Direction:
    """The local operators of M (or of M^T) on one rank and what their exchange needs."""

    def __init__(self):
        self.A_entries = None          # (row = owner * hp + slot, col, w) of A_r, kept to cut row chunks
        self.B = None
        self.B_hub = self.B_reg = None # B_r cut at row hp (spmm_adam_w1: the regular rows' sums feed the optimizer)
        self.chunks = {}               # K -> [_Chunk] (K = 1: the whole A_r)
        self.need_cols = None          # halo: gathered-block rows (owner * hp + slot) B_r references, sorted
        self.need_counts = None        #       ... how many of them every rank owns
        self.need_counts_l = self.send_counts_l = None
        self.send_slots = None         #       own hub slots every peer needs, concatenated in rank order
        self.send_counts = None
        self.pipes = {}                # (K, scheme) -> _Pipeline (hub-less graphs: B_r cut by the origin of its columns)

    @property
    def A(self):
        ch = self.chunks.get(1)
        return ch[0].op if ch else None


class _Stage:
    """One stage of the pipelined exchange and the block of B_r whose columns it brings.
    packed stage (`span` is None): the own slots that leave (`send_slots`, peers in rank order, `send_counts` rows each) are
        packed and travel by all_to_all_single; the rows that arrive (`recv_counts`, rank order) form the stage's operand
        as they land -- no scatter;
    unpacked stage (`span` = (lo, hi)): every peer is sent THE SAME contiguous slots x_local[lo:hi] (batched send / recv,
        nothing packed); the operand is the [W * (hi - lo)] block of all ranks' slices (`recv_counts` = hi - lo per rank).
    `op`: [hp x rows of the operand] (None when it holds no entries); `rows`: gathered-block row (owner * hp + slot) of
    every operand row, in order."""
    __slots__ = ("send_slots", "send_counts", "recv_counts", "op", "nnz", "rows", "span", "rows_read")

    def __init__(self, send_slots, send_counts, recv_counts, op, nnz, rows, span=None, rows_read=None):
        self.send_slots, self.send_counts, self.recv_counts, self.op, self.nnz = send_slots, send_counts, recv_counts, op, nnz
        self.rows, self.span = rows, span
        self.rows_read = int(rows.numel()) if rows_read is None else rows_read      # operand rows B_r references


class _Pipeline:
    """B_r of a graph without hub structure (rp == 0: rows = the rank's hp nodes, columns = the gathered block [W * hp]) cut
    by the ORIGIN of its columns:

        own      columns owned by this rank                    operand = the rank's own rows, in place: starts at once
        stage k  columns whose rows arrive in exchange stage k  operand = the stage's receive buffer as it lands

    Which stage a referenced row (owner q, position j of the n rows this rank reads from q) travels in:
        scheme "slices"  k = j * K // n : every stage is an all-to-all over ALL peers carrying 1 / K of each peer's rows --
                         on the xGMI mesh every pair of GPUs has its own link, so every link is busy in every stage;
        scheme "peer"    k = (rank - q) mod W - 1 : stage k is the whole contribution of ONE peer (W - 1 stages; rank r
                         sends to r + k + 1 while it receives from r - k - 1: a ring shift per stage, one link each way).
    `prefix` = a > 0 (scheme "slices"; the same a on every rank): the rows of the first `a` slots of every rank -- slots are
    dealt in DEGREE order, so these are the rows nearly every peer reads -- travel UNPACKED: slots [0, a) are cut into K
    contiguous ranges, stage k sends x_local[a_k : a_k+1] as it stands to every peer (nothing is packed: the pack of the
    rows to send is a third of the compute side at c5) and its operand is the [W * len_k] block of all ranks' ranges; ONE
    packed stage then carries the referenced rows from slot a on.  A prefix row a peer does not read travels for nothing
    (a is chosen so that >= 90 % of the (row, peer) pairs are read, `ShardedGraph.set_pipeline`).
    Both ends evaluate the same formulas, so the lists need no further agreement beyond the halo lists.  The stage blocks
    ACCUMULATE into the result of the own block (`GraphPlan.spmm(accumulate=True)` = `tgcn_spmm_acc`): rows without entries
    in a block are not touched, the order of the additions is the launch order -- deterministic.  Built from B_r's own CSR:
    same entries, same order within (row, block)."""
    SCHEMES = ("slices", "peer")

    def __init__(self, sg: "ShardedGraph", d: _Direction, K: int, scheme: str, prefix: int = 0):
        W, hp, r = sg.world, sg.hp, sg.rank
        if sg.rp != 0:
            raise ValueError("the pipelined exchange serves graphs without hub structure (hubs=None)")
        if scheme not in self.SCHEMES:
            raise ValueError(f"pipeline scheme must be one of {self.SCHEMES}")
        self.scheme, self.world = scheme, W
        self.K = K = (W - 1 if scheme == "peer" else max(1, min(int(K), 16)))
        self.prefix = a = max(0, min(int(prefix), hp)) if scheme == "slices" else 0
        n_pre = K if a > 0 else 0                          # unpacked stages: the prefix in K slot ranges
        Kt = 1 if a > 0 else K                             # packed stages (with a prefix: one, for the rows from slot a on)
        bounds = [a * k // K for k in range(K + 1)] if a > 0 else [0]
        dev = d.need_cols.device

        def stage_of(owner, receiver, j, n):
            """Packed stage of position j (of n) in the list of rows `receiver` reads from `owner` (tensors)."""
            if scheme == "peer":
                return (receiver - owner) % W - 1
            return (j * Kt) // n.clamp_min(1)

        # receive side: the rows this rank reads (need_cols, sorted: owner-major) -- without the prefix rows, which arrive
        # unpacked --, their position in the owner's list
        need_all = d.need_cols
        in_prefix = ((need_all // hp) != r) & ((need_all % hp) < a)
        need = need_all[~in_prefix]
        n_owner = need // hp
        counts = torch.bincount(n_owner, minlength=W)
        starts = torch.cumsum(counts, 0) - counts
        pos = torch.arange(need.numel(), device=dev) - starts[n_owner]
        n_stage = stage_of(n_owner, r, pos, counts[n_owner])
        n_stage = torch.where(n_owner == r, torch.full_like(n_stage, -1), n_stage)
        # column of the stage's receive buffer a needed row lands in: rank order, then list order (all_to_all_single)
        key = (n_stage + 1) * (W * hp) + need                         # (stage, owner, slot): `need` is already owner-major
        order = torch.argsort(key, stable=True)
        per_stage = torch.bincount(n_stage[order] + 1, minlength=Kt + 1)
        stage_start = torch.cumsum(per_stage, 0) - per_stage
        landing = torch.empty_like(need)
        landing[order] = torch.arange(need.numel(), device=dev) - stage_start[n_stage[order] + 1]
        # gathered-block row -> (stage, operand row).  Stage ids: -1 own (operand = x_local in place: the slot itself),
        # 0 .. n_pre - 1 the unpacked prefix ranges (owner * len_k + slot - a_k), n_pre .. the packed stages
        col_stage = torch.full((W * hp,), -2, dtype=torch.int64, device=dev)
        col_land = torch.zeros(W * hp, dtype=torch.int64, device=dev)
        col_stage[need] = torch.where(n_stage >= 0, n_stage + n_pre, n_stage)
        col_land[need] = torch.where(n_owner == r, need - r * hp, landing)
        pre_read = []
        if a > 0:
            pre = need_all[in_prefix]
            p_owner, p_slot = pre // hp, pre % hp
            bt = torch.tensor(bounds, dtype=torch.int64, device=dev)
            p_k = torch.searchsorted(bt, p_slot, right=True) - 1      # range k holds slots [bounds[k], bounds[k + 1])
            col_stage[pre] = p_k
            col_land[pre] = p_owner * (bt[p_k + 1] - bt[p_k]) + (p_slot - bt[p_k])
            pre_read = torch.bincount(p_k, minlength=K).tolist()
        recv_counts = torch.zeros(Kt, W, dtype=torch.int64, device=dev)
        rem = n_owner != r
        recv_counts.index_put_((n_stage[rem], n_owner[rem]), torch.ones_like(need[rem]), accumulate=True)
        # send side: my slots every peer reads (send_slots, peers in rank order; the prefix slots leave unpacked), cut by
        # the SAME formula
        s_peer_all = torch.repeat_interleave(torch.arange(W, device=dev), d.send_counts.to(dev))
        s_keep = (s_peer_all == r) | (d.send_slots >= a)
        s_peer, s_slots = s_peer_all[s_keep], d.send_slots[s_keep]
        s_counts = torch.bincount(s_peer, minlength=W)
        s_starts = torch.cumsum(s_counts, 0) - s_counts
        s_pos = torch.arange(s_peer.numel(), device=dev) - s_starts[s_peer]
        s_stage = stage_of(r, s_peer, s_pos, s_counts[s_peer])
        s_rem = s_peer != r
        send_counts = torch.zeros(Kt, W, dtype=torch.int64, device=dev)
        send_counts.index_put_((s_stage[s_rem], s_peer[s_rem]), torch.ones_like(s_peer[s_rem]), accumulate=True)
        # the blocks, from B_r's own CSR
        rowptr, col, val = d.B.export_csr()
        row = torch.repeat_interleave(torch.arange(hp, device=col.device), (rowptr[1:] - rowptr[:-1]).long())
        col = col.long()
        e_stage, e_col = col_stage[col], col_land[col]
        if bool((e_stage == -2).any()):
            raise RuntimeError("sharded: B_r references a column outside its own need list")
        sel = e_stage == -1
        self.own = sg.engine.make_op(row[sel], e_col[sel], val[sel], hp, hp)
        self.own_nnz = int(sel.sum())
        rc, sc = recv_counts.tolist(), send_counts.tolist()
        self.stages = []
        ar_w = torch.arange(W, device=dev)
        for k in range(n_pre):                             # unpacked: the prefix range k of every rank
            lo, hi = bounds[k], bounds[k + 1]
            n_k = hi - lo
            sel = e_stage == k
            nnz = int(sel.sum())
            op = sg.engine.make_op(row[sel], e_col[sel], val[sel], hp, W * n_k) if (nnz and n_k) else None
            rows = (ar_w.unsqueeze(1) * hp + torch.arange(lo, hi, device=dev).unsqueeze(0)).reshape(-1)
            self.stages.append(_Stage(None, None, [n_k] * W, op, nnz, rows, span=(lo, hi), rows_read=int(pre_read[k])))
        for k in range(Kt):                                # packed
            sel = e_stage == n_pre + k
            nnz = int(sel.sum())
            n_cols = int(sum(rc[k]))
            op = sg.engine.make_op(row[sel], e_col[sel], val[sel], hp, n_cols) if (nnz and n_cols) else None
            slots = s_slots[s_rem & (s_stage == k)].contiguous()           # peers in rank order, list order within a peer
            sg._check_list(slots, hp, "pipeline send slots")
            lo = int(stage_start[k + 1])
            self.stages.append(_Stage(slots, [int(v) for v in sc[k]], [int(v) for v in rc[k]], op, nnz,
                                      need[order[lo:lo + n_cols]]))

    def rows_received(self) -> int:
        """Rows that arrive per SpMM: the other ranks' prefix ranges (read or not) + the packed stages' rows."""
        return (self.world - 1) * self.prefix + int(sum(sum(st.recv_counts) for st in self.stages if st.span is None))


class _RowsView:
    """The local operators restricted to the regular rows a mask keeps (ShardedGraph.rows_view): `B_rows` = B_r with the
    other rows emptied (same operand as B_r), `At` / `At_chunks` = A'_r with the other COLUMNS dropped (same operand and
    row chunks as A'_r, so the reduce lists of the chunks serve unchanged: a row that lost its entries travels as
    zeros), `Bt_reg` [rp x rp] = the regular rows of B'_r over the kept regular columns.  Cut from the operators' own CSR:
    same entries, same order within a row."""

    def __init__(self, sg: "ShardedGraph", keep: Tensor):
        hp, rp, W = sg.hp, sg.rp, sg.world
        eng = sg.engine
        d, dT = sg.dirs[0], sg.dirs[0 if sg.symmetric else 1]

        def cut(op, n_rows, row_keep=None, col_keep=None, col0=0, row0=0, n_rows_out=None, n_cols_out=None):
            rowptr, col, val = op.export_csr()
            counts = (rowptr[1:] - rowptr[:-1]).long()
            row = torch.repeat_interleave(torch.arange(n_rows, device=col.device), counts)
            col = col.long()
            sel = torch.ones_like(col, dtype=torch.bool)
            if row_keep is not None:
                sel &= (row >= row0) & row_keep[(row - row0).clamp_min(0)]
            if col_keep is not None:
                sel &= (col >= col0) & col_keep[(col - col0).clamp_min(0)]
            return eng.make_op(row[sel] - (row0 if n_rows_out is not None else 0), col[sel] - (col0 if n_cols_out is not None else 0),
                               val[sel], n_rows_out if n_rows_out is not None else n_rows,
                               n_cols_out if n_cols_out is not None else op.n_cols), int(sel.sum())

        self.keep = keep
        self.B_rows, kept_b = cut(d.B, hp + rp, row_keep=keep, row0=hp)
        at, kept_a = cut(dT.A, W * hp, col_keep=keep)
        self.At = at
        self.At_chunks = []
        for ch in dT.chunks[sg.rs_chunks]:
            c = _Chunk(at if sg.rs_chunks == 1 else cut(ch.op, W * ch.ck, col_keep=keep)[0], ch.k, ch.ck, ch.n_own)
            for name in ("touch_rows", "touch_counts", "recv_counts", "recv_pos", "inv_halo", "inv_dense"):
                setattr(c, name, getattr(ch, name))
            self.At_chunks.append(c)
        self.Bt_reg, kept_bt = cut(dT.B, hp + rp, row_keep=torch.ones_like(keep), row0=hp, col_keep=keep, col0=W * hp,
                                   n_rows_out=rp, n_cols_out=rp)
        self.kept_entries = {"B": kept_b, "At": kept_a, "Bt_reg": kept_bt}


class ShardedGraph:
    _CHUNK = 1 << 23      # edges per pass of the local-operator construction (transients stay O(chunk))
    EXCHANGES = ("collective", "p2p", "halo")          # forms every partition serves; "pipeline": hub-less graphs only
    PIPE_STAGES_MAX = 8

    def __init__(self, edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int,
                 group=None, hubs: Optional[Tensor] = None, add_self_loops=True,
                 normalize: bool = True, engine=None, symmetric: Optional[bool] = None,
                 degree_sum: Optional[str] = None):
        """`symmetric`: None = find out (an edge-multiset fingerprint of M against M^T); True / False skips
        the test (TextGCN graphs are symmetric by construction, text2graph.py:148-171).
        `degree_sum`: "reference" (PyG's sequential fp32 sums in edge order and its association: the reference's bits --
        the package default; the operator is then not bitwise symmetric and M^T gets its own operators) or "accurate"
        (float64 degree sums, symmetric association: one pair of operators serves M and M^T of a symmetric graph).
        None = the package default (pytextgcn_amd.set_degree_sum).
        Either way the weights are bit for bit the single-device plan's of the same mode.
        Construction is COLLECTIVE: every rank of `group` must build the graph at the same time (the index lists of
        the halo exchange are swapped between the ranks)."""
        self.group = group if group is not None else dist.group.WORLD
        self._setup(edge_index, edge_weight, num_nodes, dist.get_world_size(self.group), dist.get_rank(self.group),
                    hubs, add_self_loops, normalize, engine, symmetric, degree_sum)
        # Forms of the exchange (all give the same sums; "p2p" and "halo" add the ranks' partial rows in rank order,
        # bit for bit alike; RCCL's reduce-scatter adds them in an order of its own):
        #   "collective"  RCCL all-gather + reduce-scatter of the whole hub block;
        #   "p2p"         the same rows as direct pairwise transfers (batched send / recv, all-to-all + local sum):
        #                 on a full xGMI mesh every pair of GPUs has its own link;
        #   "halo"        index lists built here, once: a rank receives only the hub rows its B_r references and sends
        #                 only the partial rows its A_r touches (all_to_all_single with split sizes).  For graphs
        #                 without hub structure (hubs=None) this is the true halo exchange.
        # TGCN_EXCHANGE pins the form, TGCN_RS_CHUNKS the number of row chunks A_r runs in (each chunk's
        # reduce-scatter starts when its rows are finished); bench.py times a few steps of each and keeps the fastest.
        #   "pipeline"    graphs without hub structure only (hubs=None; their DEFAULT): the halo rows travel in K stages and
        #                 the column block of stage k is added to the result while stage k + 1 is in flight (`_Pipeline`);
        #                 TGCN_PIPE_STAGES / TGCN_PIPE_SCHEME pin K and the scheme, TGCN_PIPE_PREFIX = rows per rank (or
        #                 "auto") that travel unpacked ahead of one packed stage (`set_pipeline`; default 0).
        import os
        self._build_halo_lists()
        self.set_rs_chunks(int(os.environ.get("TGCN_RS_CHUNKS", "1")))
        self.pipe_prefix = 0
        self.set_pipeline(os.environ.get("TGCN_PIPE_STAGES"), os.environ.get("TGCN_PIPE_SCHEME", "slices"),
                          prefix=os.environ.get("TGCN_PIPE_PREFIX", "0"))
        # a graph without hub structure has nothing to cover a whole-operand all-gather with: only the referenced rows
        # travel, pipelined under the column blocks; a hub partition overlaps its two collectives with A_r / B_r
        self.exchange = os.environ.get("TGCN_EXCHANGE", "pipeline" if self.rp == 0 else "collective")

    @property
    def exchange(self) -> str:
        return self._exchange

    @exchange.setter
    def exchange(self, form: str) -> None:
        if form == "pipeline":
            if self.rp != 0:
                raise ValueError('exchange="pipeline" serves graphs without hub structure (hubs=None); a hub partition '
                                 f"takes one of {self.EXCHANGES}")
        elif form not in self.EXCHANGES:
            raise ValueError(f"the exchange form must be one of {self.EXCHANGES + ('pipeline',)}, got {form!r}")
        self._exchange = form

    PIPE_PREFIX_READ_SHARE = 0.9       # the unpacked prefix reaches as far as this share of its (row, peer) pairs is read

    def set_pipeline(self, K=None, scheme: str = "slices", prefix=None) -> None:
        """Number of stages and scheme of the pipelined exchange (`_Pipeline`; hub-less graphs).  K = None: from the bytes
        -- a stage should still carry ~32 MB per peer at the hidden width (256 floats) so that the links run at their
        rate, at most PIPE_STAGES_MAX stages.  `prefix`: rows per rank that are sent to every peer UNPACKED, ahead of the
        stages (0 = none; None = keep the current choice; "auto" = the longest prefix of the degree-ordered slots of
        which the peers read >= PIPE_PREFIX_READ_SHARE of the (row, peer) pairs, the smallest such over the ranks).  The
        blocks are cut on first use and kept per (K, scheme, prefix); local once the numbers are fixed (the halo lists are
        agreed on already), but every rank must choose the same ones -- K = None and prefix = "auto" agree through one
        all-reduce each (collective)."""
        if scheme not in _Pipeline.SCHEMES:
            raise ValueError(f"pipeline scheme must be one of {_Pipeline.SCHEMES}")
        if self.rp != 0:
            self.pipe_stages, self.pipe_scheme, self.pipe_prefix = 0, scheme, 0
            return
        if prefix == "auto":
            prefix = self._auto_prefix()
        if prefix is not None:
            self.pipe_prefix = max(0, min(int(prefix), self.hp))
        if K is None or K == "":
            d = self.dirs[0]
            own = d.need_counts_l[self.rank] if d.need_counts_l is not None else 0
            per_peer = (sum(d.need_counts_l) - own) / max(1, self.world - 1) if d.need_counts_l is not None else 0
            if self.group is not None and self.world > 1:            # the same K everywhere: the largest rank decides
                t = torch.tensor([float(per_peer)], device=self._comm_device())
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                per_peer = float(t.item())
            K = int(per_peer * 1024 // (32 << 20))
        self.pipe_stages = max(1, min(int(K), self.PIPE_STAGES_MAX))
        self.pipe_scheme = scheme

    def _auto_prefix(self) -> int:
        """The longest prefix [0, a) of this rank's (degree-ordered) slots of which the peers read at least
        PIPE_PREFIX_READ_SHARE of the (slot, peer) pairs, over both directions; the minimum over the ranks (collective when
        there is a group); a multiple of 256 rows; 0 when it would cover less than 1 / 64 of the shard."""
        W, hp, r = self.world, self.hp, self.rank
        if W < 2:
            return 0
        a = hp
        for d in self.dirs:
            if d.send_slots is None:
                return 0
            peer = torch.repeat_interleave(torch.arange(W, device=d.send_slots.device), d.send_counts.to(d.send_slots.device))
            pop = torch.bincount(d.send_slots[peer != r], minlength=hp).double()      # peers that read slot s
            share = torch.cumsum(pop, 0) / ((W - 1) * torch.arange(1, hp + 1, device=pop.device, dtype=torch.float64))
            ok = torch.nonzero(share >= self.PIPE_PREFIX_READ_SHARE).flatten()
            a = min(a, int(ok[-1]) + 1 if ok.numel() else 0)
        if self.group is not None:
            t = torch.tensor([float(a)], device=self._comm_device())
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            a = int(t.item())
        a -= a % 256
        return a if a * 64 >= hp else 0

    def _pipeline(self, d: _Direction) -> _Pipeline:
        key = (self.pipe_stages, self.pipe_scheme, self.pipe_prefix)
        pipe = d.pipes.get(key)
        if pipe is None:
            if d.send_slots is None:
                raise RuntimeError("the pipelined exchange needs the halo lists (for_rank(..., halo_lists=True))")
            pipe = d.pipes[key] = _Pipeline(self, d, *key)
        return pipe

    def drop_unused_pipelines(self) -> None:
        """Release the column blocks of every (K, scheme) but the current one (bench.py times a few)."""
        keep = (self.pipe_stages, self.pipe_scheme, self.pipe_prefix)
        for d in self.dirs:
            for key in [k for k in d.pipes if k != keep]:
                pipe = d.pipes.pop(key)
                for op in [pipe.own] + [st.op for st in pipe.stages]:
                    close = getattr(op, "close", None)
                    if close is not None:
                        close()

    @classmethod
    def for_rank(cls, edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int, world: int, rank: int,
                 hubs: Optional[Tensor] = None, add_self_loops=True, normalize: bool = True, engine=None,
                 symmetric: Optional[bool] = None, degree_sum: Optional[str] = None,
                 halo_lists: bool = False) -> "ShardedGraph":
        """What rank `rank` of a `world`-rank group would build, WITHOUT a process group: the partition, the
        normalisation and the local operators A_r / B_r (`ops`, `dirs`), cut exactly as the collective constructor
        cuts them, plus `need_cols` (the gathered-block rows its B_r reads) and, with `halo_lists`, the gather-side
        lists of the halo exchange (`_build_halo_lists_offline`).  What needs the peers (`spmm`, the reduce-side lists)
        is absent.  For inspecting and timing one rank's share of a partition on a single
        device -- tests at BASELINE size, tools/sim_shard_compute.py."""
        self = cls.__new__(cls)
        self.group = None
        self._setup(edge_index, edge_weight, num_nodes, int(world), int(rank), hubs, add_self_loops, normalize, engine,
                    symmetric, degree_sum)
        self._exchange = "collective"
        self.pipe_stages, self.pipe_scheme, self.pipe_prefix = 0, "slices", 0
        if halo_lists:
            self._build_halo_lists_offline(edge_index)
            self.set_pipeline(None)
        return self

    def _build_halo_lists_offline(self, edge_index: Tensor) -> None:
        """The gather-side lists of the halo exchange as `_build_halo_lists` leaves them, computed WITHOUT the peers:
        which gathered-block rows every rank's B_q reads follows from the partition and the edge list alone (one pass,
        a [W, W * hp] table of flags), so the slots this rank would be asked for are known without asking."""
        p, W, hp, r = self.part, self.world, self.hp, self.rank
        dev = edge_index.device
        for k, d in enumerate(self.dirs):
            asked = torch.zeros(W, hp, dtype=torch.bool, device=dev)          # asked[q, s]: rank q reads own hub slot s
            E = edge_index.size(1)
            for lo in range(0, E, self._CHUNK):
                s, t = edge_index[0, lo:lo + self._CHUNK], edge_index[1, lo:lo + self._CHUNK]
                if k == 1:
                    s, t = t, s
                if self._loops:
                    keep = s != t
                    s, t = s[keep], t[keep]
                sel = p.hub_mask[s] & (p.owner[s] == r)                       # B_q reads hub columns only from the block
                asked[p.owner[t[sel]], p.slot[s[sel]]] = True
            if self._loops:
                own_hub = self.owned[:hp][self.real[:hp]]
                asked[r, p.slot[own_hub]] = True                              # the own hubs' loops
            q_idx, slots = torch.nonzero(asked, as_tuple=True)                 # rank-major, slots ascending
            d.send_slots, d.send_counts = slots.contiguous(), torch.bincount(q_idx, minlength=W)
            owner = d.need_cols // hp
            d.need_counts = torch.bincount(owner, minlength=W)
            d.need_counts_l = [int(v) for v in d.need_counts.tolist()]
            d.send_counts_l = [int(v) for v in d.send_counts.tolist()]

    def _setup(self, edge_index, edge_weight, num_nodes, world, rank, hubs, add_self_loops, normalize, engine, symmetric,
               degree_sum) -> None:
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside a world of {world}")
        self.world, self.rank = world, rank
        self.engine = engine if engine is not None else HipEngine()
        self.num_nodes = num_nodes
        self.device = edge_index.device
        if degree_sum is None:
            from .plan import default_degree_sum
            degree_sum = default_degree_sum()
        if degree_sum not in ("accurate", "reference"):
            raise ValueError('degree_sum must be "accurate" or "reference"')
        self.degree_sum = degree_sum
        part = Partition(edge_index, num_nodes, self.world, hubs)
        self.part = part
        self.hp, self.rp, self.n_local = part.hp, part.rp, part.n_local
        if self.hp == 0:
            raise ValueError("a sharded graph needs at least one hub node per rank (hubs=None makes every node one)")
        self.owned = part.owned(self.rank)
        self.real = self.owned >= 0
        self._all_real = None                                 # resolved at the first bias gradient (colsum_real)
        # Normalisation over the WHOLE edge list (two vectors of N floats), then every rank cuts its own
        # two operators out of the edge list chunk by chunk: no rank ever holds the whole-graph plan, its
        # CSR or anything else of size nnz beyond the edge list it was handed.
        # PyG adds the loops inside gcn_norm, so GCNConv(normalize=False) never sees them
        loops = (max(0, min(int(add_self_loops), 2)) if normalize else 0)
        self._dis = self._loop_w = None
        if normalize:
            self._dis, self._loop_w = self.engine.gcn_norm(edge_index, edge_weight, num_nodes, loops, degree_sum)
        self._loops = loops
        self.symmetric = bool(symmetric) if symmetric is not None else \
            self._is_symmetric(edge_index, edge_weight)
        self.dirs = [self._local_ops(edge_index, edge_weight, transpose=False)]
        if not self.symmetric:
            self.dirs.append(self._local_ops(edge_index, edge_weight, transpose=True))
        self._dis = self._loop_w = None
        self.plan = self.dirs[0].B            # the larger local operator (for reporting)
        self.rs_chunks = 1
        self._xbuf = {}
        self._stage = {}
        self._recv = {}

    @property
    def ops(self):
        """[(A_r, B_r)] of M (and of M^T when M is not symmetric)."""
        return [(d.A, d.B) for d in self.dirs]

    @classmethod
    def from_data(cls, g, group=None, **kw) -> "ShardedGraph":
        """From the graph object Text2GraphTransformer returns (text2graph.py:192-193): the word nodes
        [0, g.n_vocab) become the replicated hubs, the documents stay with their owners."""
        n = g.x.shape[0] if getattr(g, "x", None) is not None else int(g.edge_index.max()) + 1
        n_vocab = int(getattr(g, "n_vocab", 0) or 0)
        hubs = torch.arange(n, device=g.edge_index.device) < n_vocab if n_vocab > 0 else None
        return cls(g.edge_index, g.edge_attr, n, group=group, hubs=hubs, **kw)

    # ---- construction ---------------------------------------------------------------------------
    def _chunks(self, edge_index: Tensor, edge_weight: Optional[Tensor]):
        """(source, target, w_hat) of the non-loop entries of M, a chunk of edges at a time: w_hat = w * (dis[s]
        * dis[t]) -- the association of tgcn_plan_create, which keeps a symmetric graph bitwise symmetric ((dis[s] *
        w) * dis[t] in the reference-order mode).  `dis` comes from the routine tgcn_plan_create runs itself and the
        products are the same IEEE multiplications, so the weights are the single-device plan's bit for bit."""
        E = edge_index.size(1)
        for lo in range(0, E, self._CHUNK):
            hi = min(E, lo + self._CHUNK)
            s, t = edge_index[0, lo:hi], edge_index[1, lo:hi]
            w = edge_weight[lo:hi].float() if edge_weight is not None else \
                torch.ones(hi - lo, dtype=torch.float32, device=edge_index.device)
            if self._loops:                                    # input loops are replaced (add_remaining_self_loops)
                keep = s != t
                s, t, w = s[keep], t[keep], w[keep]
            if self._dis is not None:
                if self.degree_sum == "reference":             # PyG's association (gcn_conv.py: dis[row] * w * dis[col])
                    w = (self._dis[s] * w) * self._dis[t]
                else:
                    w = w * (self._dis[s] * self._dis[t])
            yield s, t, w

    def _is_symmetric(self, edge_index: Tensor, edge_weight: Optional[Tensor]) -> bool:
        """M == M^T as edge multisets {(target, source, bits(w_hat))}: two independent 64-bit fingerprints of
        the entries against those of the swapped entries (the loops are symmetric by construction).  Equal
        multisets give equal operators up to the order duplicates are summed in."""
        sums = [[0, 0], [0, 0]]
        for s, t, w in self._chunks(edge_index, edge_weight):
            bits = w.view(torch.int32).long()
            for i, ks in enumerate(_HASH_K):
                sums[i][0] = (sums[i][0] + _entry_hash(t, s, bits, ks)) & ((1 << 64) - 1)
                sums[i][1] = (sums[i][1] + _entry_hash(s, t, bits, ks)) & ((1 << 64) - 1)
        return all(a == b for a, b in sums)

    def _local_ops(self, edge_index: Tensor, edge_weight: Optional[Tensor], transpose: bool) -> _Direction:
        """A_r and B_r of M (or of M^T), cut out of the edge list chunk by chunk."""
        p, r, W, hp, rp = self.part, self.rank, self.world, self.hp, self.rp
        dev = edge_index.device
        parts_a, parts_b = [], []
        need = torch.zeros(W * hp, dtype=torch.bool, device=dev)      # gathered-block rows B_r reads
        for s, t, w in self._chunks(edge_index, edge_weight):
            if transpose:
                s, t = t, s
            t_hub, s_hub = p.hub_mask[t], p.hub_mask[s]
            t_mine, s_mine = p.owner[t] == r, p.owner[s] == r
            # A: hub rows (gathered numbering) <- own regular columns
            if rp > 0:
                a = t_hub & ~s_hub & s_mine
                parts_a.append((p.hub_col[t[a]], p.slot[s[a]], w[a]))
            # B: own rows <- all hubs + own regular columns (hub <- regular entries all live in A)
            b = t_mine & (s_hub | (s_mine & ~t_hub))
            tb, sb = t[b], s[b]
            sb_hub = p.hub_mask[sb]
            need[p.hub_col[sb[sb_hub]]] = True
            parts_b.append((torch.where(p.hub_mask[tb], p.slot[tb], hp + p.slot[tb]),
                            torch.where(sb_hub, p.hub_col[sb], p.reg_col[sb]), w[b]))
        if self._loops:
            # one loop per own node, after the edges (the tail position add_remaining_self_loops gives them)
            own = self.owned[self.real]
            lrow = torch.nonzero(self.real).flatten()
            own_hub = p.hub_mask[own]
            lcol = torch.where(own_hub, p.hub_col[own], p.reg_col[own])
            need[lcol[own_hub]] = True
            lw = self._loop_w[own]
            if self._dis is not None:
                if self.degree_sum == "reference":
                    lw = (self._dis[own] * lw) * self._dis[own]
                else:
                    lw = lw * (self._dis[own] * self._dis[own])
            parts_b.append((lrow, lcol, lw))

        def cat(parts, k):
            if not parts:
                return torch.empty(0, dtype=torch.float32 if k == 2 else torch.int64, device=dev)
            return torch.cat([q[k] for q in parts])
        d = _Direction()
        if rp > 0:
            d.A_entries = (cat(parts_a, 0), cat(parts_a, 1), cat(parts_a, 2))
            d.chunks[1] = [_Chunk(self.engine.make_op(*d.A_entries, W * hp, rp), 0, hp, hp)]
        del parts_a
        d.B = self.engine.make_op(cat(parts_b, 0), cat(parts_b, 1), cat(parts_b, 2), hp + rp, W * hp + rp)
        d.need_cols = torch.nonzero(need).flatten()
        return d

    # ---- index lists of the halo exchange (built once; collective) ---------------------------------
    def _comm_device(self):
        """Where small index tensors travel: RCCL moves device memory only, a host-serviced backend host memory."""
        return self.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")

    def _swap_lists(self, idx: Tensor, counts: Tensor) -> Tuple[Tensor, Tensor]:
        """Every rank hands every peer q the slice of `idx` that concerns q (`counts[q]` entries, rank order) and
        gets the peers' slices for itself: (received indices concatenated in rank order, their counts)."""
        cd = self._comm_device()
        counts = counts.to(cd)
        got_counts = torch.empty_like(counts)
        dist.all_to_all_single(got_counts, counts, group=self.group)
        out_sizes, in_sizes = [int(v) for v in got_counts.tolist()], [int(v) for v in counts.tolist()]
        got = torch.empty(sum(out_sizes), dtype=torch.int64, device=cd)
        dist.all_to_all_single(got, idx.to(cd).contiguous(), out_sizes, in_sizes, group=self.group)
        return got.to(self.device), got_counts.to(self.device)

    def _build_halo_lists(self) -> None:
        hp, W = self.hp, self.world
        for d in self.dirs:
            owner = d.need_cols // hp
            d.need_counts = torch.bincount(owner, minlength=W)
            # tell every owner which of its slots this rank reads; learn which of the own slots the peers read
            d.send_slots, d.send_counts = self._swap_lists(d.need_cols - owner * hp, d.need_counts)
            d.need_counts_l = [int(v) for v in d.need_counts.tolist()]
            d.send_counts_l = [int(v) for v in d.send_counts.tolist()]
            # the row-movement kernels take these lists as raw pointers: check them once, here
            self._check_list(d.send_slots, hp, "send_slots")                  # rows of the own shard
            self._check_list(d.need_cols, W * hp, "need_cols", distinct=True)  # rows of the gathered block

    @staticmethod
    def _check_list(idx: Tensor, n_rows: int, name: str, distinct: bool = False) -> None:
        if idx.numel() == 0:
            return
        lo, hi = int(idx.min()), int(idx.max())
        if lo < 0 or hi >= n_rows:
            raise IndexError(f"sharded: index list `{name}` spans [{lo}, {hi}] for a buffer of {n_rows} rows")
        if distinct and int(torch.unique(idx).numel()) != idx.numel():
            raise ValueError(f"sharded: index list `{name}` holds a row twice")

    def set_rs_chunks(self, K: int) -> None:
        """Run A_r as K row chunks (hub slot s belongs to chunk s % K, so the chunks are alike in weight); the
        reduce-scatter of a chunk is issued as soon as its rows are finished and overlaps the remaining chunks.
        Collective (the pruned reduce-scatter's index lists are swapped)."""
        K = max(1, min(int(K), self.hp))
        W, hp, rp = self.world, self.hp, self.rp
        for d in self.dirs:
            if d.A_entries is None:
                continue
            row, col, w = d.A_entries
            owner, slot = row // hp, row % hp
            if K not in d.chunks:
                ck = (hp + K - 1) // K
                chunks = []
                for k in range(K):
                    sel = (slot % K) == k
                    op = self.engine.make_op(owner[sel] * ck + slot[sel] // K, col[sel], w[sel], W * ck, rp)
                    chunks.append(_Chunk(op, k, ck, len(range(k, hp, K))))
                d.chunks[K] = chunks
            for ch in d.chunks[K]:
                if ch.touch_rows is not None:
                    continue
                sel = (slot % K) == ch.k
                touched = torch.unique(owner[sel] * ch.ck + slot[sel] // K)        # sorted: rank-major
                ch.touch_rows = touched
                t_owner = touched // ch.ck
                counts = torch.bincount(t_owner, minlength=W)
                got, got_counts = self._swap_lists(touched - t_owner * ch.ck, counts)
                ch.touch_counts = [int(v) for v in counts.tolist()]
                ch.recv_counts = [int(v) for v in got_counts.tolist()]
                ch.recv_pos = got                          # positions inside the chunk's own rows, rank order
                # the same lists as tables for the one-pass reduction kernel (tgcn_rows_reduce_ranked)
                n_own = ch.n_own
                inv = torch.full((W, max(n_own, 1)), -1, dtype=torch.int32, device=self.device)
                src = torch.repeat_interleave(torch.arange(W, device=self.device), got_counts.to(self.device))
                inv[src, got] = torch.arange(got.numel(), dtype=torch.int32, device=self.device)
                ch.inv_halo = inv[:, :n_own].contiguous()
                self._check_list(touched, W * ch.ck, "touch_rows", distinct=True)   # rows of the chunk's partial sums
                self._check_list(got, max(n_own, 1), "recv_pos")
                ar = torch.arange(n_own, dtype=torch.int32, device=self.device)
                ch.inv_dense = (torch.arange(W, dtype=torch.int32, device=self.device).unsqueeze(1) * ch.ck + ar).contiguous()
        self.rs_chunks = K

    def drop_unused_chunks(self) -> None:
        """Release the A_r row-chunk operators of every chunk count but the current one (bench.py builds K = 1, 2, 4
        to time them; the losers are dead weight afterwards).  `A_entries` stay, so another count can be cut again."""
        for d in self.dirs:
            for K in [k for k in d.chunks if k != self.rs_chunks and k != 1]:
                for ch in d.chunks.pop(K):
                    close = getattr(ch.op, "close", None)
                    if close is not None:
                        close()

    # ---- data movement ---------------------------------------------------------------------------
    def scatter_rows(self, full: Tensor) -> Tensor:
        """Rows of a replicated [N, ...] tensor that this rank owns, in local order (padding rows
        are zero / False)."""
        out = torch.zeros((self.n_local,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
        out[self.real] = full[self.owned[self.real]]
        return out

    def gather_rows(self, local: Tensor) -> Tensor:
        """Inverse of scatter_rows: the full [N, ...] tensor on every rank (metrics, checkpoints)."""
        parts = [torch.empty_like(local) for _ in range(self.world)]
        dist.all_gather(parts, local.contiguous(), group=self.group)
        full = torch.zeros((self.num_nodes,) + tuple(local.shape[1:]), dtype=local.dtype,
                           device=local.device)
        for q in range(self.world):
            own = self.part.owned(q)
            real = own >= 0
            full[own[real]] = parts[q][real]
        return full

    # ---- the distributed SpMM --------------------------------------------------------------------
    def _stream_ordered(self, t: Tensor) -> bool:
        """Does the group's backend order point-to-point transfers of `t` against the work queued on the
        current stream?  RCCL does (its operations are enqueued on a stream that first waits for the
        caller's).  A host-serviced backend (gloo) does not for DEVICE memory: ProcessGroupGloo's
        send / recv hand the tensor's raw data pointer to a host thread, which reads and writes the
        buffer whenever the socket is ready -- before the kernel that produces the shard has run, and
        behind the back of the device's L2 for the receiving side.  (Its collectives and all-to-all do
        stage CUDA tensors through pinned memory with stream events; send / recv do not.)  Host
        tensors are always fine."""
        return (not t.is_cuda) or dist.get_backend(self.group) == "nccl"

    def _host_stage(self, name: str, shape, dtype) -> Tensor:
        """Pinned host buffer for the staged exchange, one per (role, shape)."""
        key = (name, tuple(shape), dtype)
        buf = self._stage.get(key)
        if buf is None:
            buf = torch.empty(shape, dtype=dtype, pin_memory=True)
            self._stage[key] = buf
        return buf

    _NARROW = 128     # widths up to here run the sub-group SpMM kernels, which take ONE operand buffer

    def _xbuf_key(self, x_local: Tensor, halo: bool, d: Optional["_Direction"]):
        return (x_local.size(1), x_local.dtype, x_local.device, halo, self.dirs.index(d) if (halo and d is not None) else 0)

    def _gather_buffer(self, x_local: Tensor, halo: bool, d: Optional[_Direction] = None) -> Tensor:
        """The gathered hub block [W * hp, F], one per width (layer-1 / layer-2 widths alternate).  The halo form
        has its own, zero-filled once: it only ever writes the rows B_r references, and the rest must stay finite
        (the dense hot block of a local operator multiplies EVERY operand row by a possibly zero weight).
        Narrow widths (the class width of layer 2) get rp more rows behind the block: B_r's whole operand [hubs ;
        own regular rows] in one buffer, which is what the sub-group kernels gather from (the copy of the own rows
        is rp x F floats; at wide widths B_r reads them in place as a split operand)."""
        F = x_local.size(1)
        # the halo form writes only the rows the direction's B_r references: M and M^T of an asymmetric graph read
        # different rows, so each direction keeps its own block (a row left over from the other direction would be
        # multiplied by a zero weight of the dense hot block -- harmless only while it is finite)
        return self._operand_buffer(self._xbuf_key(x_local, halo, d))[:self.world * self.hp]

    def _operand_buffer(self, key) -> Tensor:
        """The whole buffer behind `_gather_buffer` (key = `_xbuf_key`): [W * hp (+ rp at narrow widths), F]."""
        xbuf = self._xbuf.get(key)
        if xbuf is None:
            F, dtype, device, halo = key[:4]
            make = torch.zeros if halo else torch.empty
            rows = self.world * self.hp + (self.rp if F <= self._NARROW else 0)
            xbuf = make(rows, F, dtype=dtype, device=device)
            self._xbuf[key] = xbuf
        return xbuf

    def operand_buffer(self, F: int, device, dtype=torch.float32) -> Tensor:
        """The operand buffer of width F of the collective / pairwise forms (the one `spmm` gathers into): rows [0, W * hp)
        the hub block, at narrow widths followed by rp rows for the rank's own regular rows."""
        return self._operand_buffer((F, dtype, device, False, 0))

    def _whole_operand(self, d: _Direction, x_local: Tensor) -> Optional[Tensor]:
        """[gathered hubs ; own regular rows] as one tensor when the width calls for it (see _gather_buffer)."""
        if self.rp == 0 or x_local.size(1) > self._NARROW:
            return None
        whole = self._xbuf[self._xbuf_key(x_local, self.exchange == "halo", d)]
        whole[self.world * self.hp:].copy_(x_local[self.hp:])
        return whole

    def _start_gather(self, d: _Direction, x_local: Tensor):
        """Start bringing the hub rows this rank's B_r reads into the gather buffer.  Returns (buffer, finish):
        `finish()` makes the current stream wait for the rows."""
        hp = self.hp
        shard = x_local[:hp]
        direct = self._stream_ordered(x_local)
        if self._exchange == "pipeline":
            raise RuntimeError("the pipelined exchange has no gather step of its own (ShardedGraph._spmm_pipeline)")
        if self.exchange == "collective":
            xbuf = self._gather_buffer(x_local, False)
            work = dist.all_gather_into_tensor(xbuf, shard, group=self.group, async_op=True)
            return xbuf, work.wait
        if self.exchange == "p2p":
            xbuf = self._gather_buffer(x_local, False)
            works = self._all_gather_p2p(xbuf, shard) if direct else self._all_gather_p2p_staged(xbuf, shard)
            return xbuf, lambda: [w.wait() for w in works]
        # halo: only the rows somebody reads travel
        xbuf = self._gather_buffer(x_local, True, d)
        pack = self._rows_gather(shard, d.send_slots)
        recv, work = self._all_to_all_v(pack, d.need_counts_l, d.send_counts_l, direct, role="gather")

        def finish():
            work.wait()
            self._rows_scatter(xbuf, d.need_cols, recv)
        return xbuf, finish

    def _rows_gather(self, x: Tensor, idx: Tensor) -> Tensor:
        f = getattr(self.engine, "rows_gather", None)
        return f(x, idx) if (f is not None and x.is_cuda) else x.index_select(0, idx)

    def _rows_scatter(self, y: Tensor, idx: Tensor, x: Tensor) -> None:
        f = getattr(self.engine, "rows_scatter_", None)
        if f is not None and y.is_cuda:
            f(y, idx, x)
        else:
            y.index_copy_(0, idx, x)

    def _recv_buffer(self, role, shape, dtype, device, zero: bool = False) -> Tensor:
        """Receive buffer of one collective of the distributed SpMM, kept per (role, shape): `role` names the collective
        inside one SpMM (the gather, the reduce of chunk k), so two transfers in flight never share one; consecutive
        SpMMs reuse it in stream order (the consumer of call n is enqueued before the collective of call n + 1, and the
        communicator's stream waits for the caller's at enqueue)."""
        key = (role, tuple(shape), dtype, device)
        buf = self._recv.get(key)
        if buf is None:
            if len(self._recv) >= 64:
                self._recv.clear()
            # `zero`: a buffer part of which no transfer ever writes (the own range of an unpacked pipeline stage) must
            # still hold finite values: the dense hot block of an operator multiplies EVERY operand row by a weight
            buf = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device)
            self._recv[key] = buf
        return buf

    def _all_to_all_v(self, src: Tensor, out_sizes, in_sizes, direct: bool, role="a2a"):
        """all_to_all_single with split sizes (rows); device tensors over a host-serviced backend go through pinned
        host memory with blocking copies (see `_stream_ordered`).  Returns (received rows, work)."""
        shape = (sum(out_sizes),) + tuple(src.shape[1:])
        if direct:
            dst = self._recv_buffer(role, shape, src.dtype, src.device)
            return dst, dist.all_to_all_single(dst, src, out_sizes, in_sizes, group=self.group, async_op=True)
        h_src = self._host_stage("a2av_src", src.shape, src.dtype)
        h_dst = self._host_stage("a2av_dst", shape, src.dtype)
        h_src.copy_(src)                                            # synchronous: `src` is complete
        dist.all_to_all_single(h_dst, h_src, out_sizes, in_sizes, group=self.group)
        dst = torch.empty(shape, dtype=src.dtype, device=src.device)
        dst.copy_(h_dst)
        return dst, _DONE

    def _start_reduce(self, ch: _Chunk, partial: Tensor):
        """Start the reduce-scatter of one chunk of hub partial sums [W * ck, F].  Returns finish(y_hub): waits and adds
        the sums of this rank's rows of the chunk (slots k, k + K, ...) to `y_hub`."""
        W, ck, F = self.world, ch.ck, partial.size(1)
        direct = self._stream_ordered(partial)
        K = self.rs_chunks

        def add_to(y_hub: Tensor, acc: Tensor):
            if K == 1:
                y_hub += acc
            else:
                y_hub[ch.k::K] += acc[:ch.n_own]
        if self.exchange == "collective":
            out = self._recv_buffer(("rs", K, ch.k), (ck, F), partial.dtype, partial.device)
            work = dist.reduce_scatter_tensor(out, partial, group=self.group, async_op=True)

            def finish(y_hub):
                work.wait()
                add_to(y_hub, out)
            return finish
        ranked = getattr(self.engine, "reduce_ranked_", None) if partial.is_cuda else None
        if self.exchange == "p2p":
            sizes = [ck] * W
            recv, work = self._all_to_all_v(partial, sizes, sizes, direct, role=("rs", K, ch.k))

            def finish(y_hub):
                work.wait()
                if ranked is not None:                      # one pass: zero + the ranks' rows in rank order, then += into y
                    ranked(y_hub, recv, ch.inv_dense, W, ch.n_own, ch.k, K)
                    return
                acc = torch.zeros(ck, F, dtype=partial.dtype, device=partial.device)
                for q in range(W):                          # the ranks' partial rows, added in rank order
                    acc += recv[q * ck:(q + 1) * ck]
                add_to(y_hub, acc)
            return finish
        # halo: only rows with entries travel; absent rows are exact zeros in the other forms
        pack = self._rows_gather(partial, ch.touch_rows)
        recv, work = self._all_to_all_v(pack, ch.recv_counts, ch.touch_counts, direct, role=("rs", K, ch.k))

        def finish(y_hub):
            work.wait()
            if ranked is not None:
                ranked(y_hub, recv, ch.inv_halo, W, ch.n_own, ch.k, K)
                return
            acc = torch.zeros(ck, F, dtype=partial.dtype, device=partial.device)
            off = 0
            for q in range(W):                              # rank order; a rank's rows are distinct
                n = ch.recv_counts[q]
                if n:
                    acc.index_add_(0, ch.recv_pos[off:off + n], recv[off:off + n])
                off += n
            add_to(y_hub, acc)
        return finish

    def spmm(self, x_local: Tensor, bias: Optional[Tensor] = None, transpose: bool = False) -> Tensor:
        d = self.dirs[1 if (transpose and not self.symmetric) else 0]
        hp, rp = self.hp, self.rp
        if x_local.shape[0] != self.n_local:
            raise ValueError(f"operand has {x_local.shape[0]} rows, this rank owns {self.n_local}")
        x_local = x_local.contiguous()
        if self._exchange == "pipeline":
            return self._spmm_pipeline(d, x_local, bias)
        xbuf, gathered = self._start_gather(d, x_local)
        pending = []
        if d.A is not None:
            # hub rows x own regular columns, chunk by chunk: a chunk's reduce-scatter is issued behind its SpMM and
            # runs while the next chunks (and then B_r) are computed
            xr = x_local[hp:]
            for ch in d.chunks[self.rs_chunks]:
                pending.append(self._start_reduce(ch, ch.op.spmm(xr)))
        whole = self._whole_operand(d, x_local)                # narrow widths: own rows copied behind the hub block
        gathered()
        y = self._apply_B(d, x_local, bias, xbuf, whole)                    # overlaps the reduce-scatter
        for finish in pending:
            finish(y[:hp])
        return y

    def _spmm_pipeline(self, d: _Direction, x_local: Tensor, bias: Optional[Tensor]) -> Tensor:
        """The distributed SpMM of a graph without hub structure: every stage's transfer is posted up front, in order (the
        communicator's stream runs them one after the other beside the compute stream) -- an unpacked stage sends its
        range of x_local as it stands, a packed one packs its rows first --, the own-column block starts at once, and block
        k is added as soon as stage k has landed."""
        pipe = self._pipeline(d)
        posted = [self._post_stage(pipe, k, x_local, ("pipe", self.dirs.index(d))) for k in range(len(pipe.stages))]
        y = pipe.own.spmm(x_local, bias)                       # overlaps the first stage
        for st, (recv, works) in zip(pipe.stages, posted):
            for w_ in works:
                w_.wait()
            if st.op is not None:
                st.op.spmm(recv, out=y, accumulate=True)       # overlaps the next stage
        return y

    def _post_stage(self, pipe: _Pipeline, k: int, x_local: Tensor, tag) -> Tuple[Tensor, list]:
        """Start the transfer of stage k of `pipe`: (the stage's operand buffer, the works to wait for)."""
        st = pipe.stages[k]
        direct = self._stream_ordered(x_local)
        role = (tag, pipe.K, pipe.prefix, k)
        if st.span is None:
            pack = self._rows_gather(x_local, st.send_slots)
            recv, work = self._all_to_all_v(pack, st.recv_counts, st.send_counts, direct, role=role)
            return recv, [work]
        lo, hi = st.span
        buf = self._recv_buffer(role, (self.world * (hi - lo), x_local.size(1)), x_local.dtype, x_local.device, zero=True)
        works = []
        if self.world > 1 and hi > lo:
            piece = x_local[lo:hi]                          # (the own range is not copied: no block reads it)
            works = self._all_gather_p2p(buf, piece, hi - lo, own=False) if direct else \
                self._all_gather_p2p_staged(buf, piece, hi - lo)
        return buf, works

    def _apply_B(self, d: _Direction, x_local: Tensor, bias: Optional[Tensor], xbuf: Tensor, whole: Optional[Tensor]) -> Tensor:
        if whole is not None:
            return d.B.spmm(whole, bias)
        # split operand: hub columns from the gathered block, own regular columns straight from x_local
        return d.B.spmm(xbuf, bias, x2=x_local[self.hp:] if self.rp > 0 else None)

    def local_step(self, d: _Direction, x_local: Tensor, bias: Optional[Tensor], gathered: Tensor, rs_out: Tensor) -> Tensor:
        """Every launch one distributed SpMM makes on this rank EXCEPT the collectives, whose results are handed in
        (`gathered`: the hub block [W * hp, F] -- for narrow widths the leading part of a [W * hp + rp, F] buffer --,
        `rs_out`: the reduce-scattered hub sums [hp, F]): the compute side of the step, for measurements on one
        device (tools/sim_shard_compute.py)."""
        hp, rp = self.hp, self.rp
        if d.A is not None:
            xr = x_local[hp:]
            for ch in d.chunks[self.rs_chunks]:
                ch.op.spmm(xr)
        whole = None
        if rp > 0 and x_local.size(1) <= self._NARROW:
            whole = gathered._base if gathered._base is not None else gathered
            whole[self.world * hp:].copy_(x_local[hp:])
        y = self._apply_B(d, x_local, bias, gathered, whole)
        if d.A is not None:
            y[:hp] += rs_out
        return y

    # ---- the backward SpMM of layer 1 with the optimizer inside (one-hot features: d W1 = M^T d H1) ------------
    def _split_B(self, d: _Direction) -> None:
        """B_r as two operators, cut at local row hp: the hub slice (its sums still wait for the other ranks'
        partial rows) and the regular rows (final as they leave the kernel, so `tgcn_spmm_adam_split` can spend them
        on the optimizer).  Built once, from B_r's own CSR: same entries, same order within a row."""
        if d.B_reg is not None or d.B_hub is not None:
            return
        if not hasattr(d.B, "export_csr"):
            raise RuntimeError("the fused W1 update needs the HIP engine's operators")
        hp, rp, W = self.hp, self.rp, self.world
        rowptr, col, val = d.B.export_csr()
        counts = (rowptr[1:] - rowptr[:-1]).long()
        row = torch.repeat_interleave(torch.arange(hp + rp, device=col.device), counts)
        cut = int(rowptr[hp].item())
        col = col.long()
        if hp > 0:
            d.B_hub = self.engine.make_op(row[:cut], col[:cut], val[:cut], hp, W * hp + rp)
        if rp > 0:
            d.B_reg = self.engine.make_op(row[cut:] - hp, col[cut:], val[cut:], rp, W * hp + rp)

    def spmm_adam_w1(self, g_local: Tensor, adam, hub_block: Optional[Tensor] = None, pending=None) -> None:
        """W1_local <- Adam(W1_local, M^T @ g) with the regular rows' update INSIDE the SpMM that computes their gradient
        (`GraphPlan.spmm_adam`, split operand: gathered hub block | own rows) -- the [rp, F] gradient and the optimizer's
        pass over those rows disappear, as `optim.Adam.fuse_into_backward` does on one device.  The hub slice takes the
        plain road: its gradient rows are complete only after the reduce-scatter, then one Adam pass over hp rows.
        `adam(rows, grad_or_None, op, g1, g2)` is the optimizer's closure (pytextgcn_amd.optim.Adam._fused_update_sharded).
        The exchange is the one of `spmm(..., transpose=True)`: same collectives, same order on every rank.
        `hub_block` [W * hp, F]: the operand's hub rows of EVERY rank, already present here (the narrow exchange forms
        them locally, pytextgcn_amd/narrow.py) -- nothing is gathered, `g_local` is then this rank's REGULAR rows [rp, F]
        only, and `pending` the reduce closures of A_r's partial sums if the caller started them already."""
        d = self.dirs[1 if not self.symmetric else 0]
        hp, rp = self.hp, self.rp
        if g_local.size(1) <= self._NARROW:
            raise ValueError("spmm_adam_w1 serves the wide (hidden) width only")
        g_local = g_local.contiguous()
        if self._exchange == "pipeline":
            # no regular rows: every gradient row is complete only after the last stage block -- the plain road
            if hub_block is not None or g_local.shape[0] != self.n_local:
                raise ValueError("the pipelined exchange takes the rank's whole operand")
            adam(slice(0, hp), self._spmm_pipeline(d, g_local, None), None, None, None)
            return
        self._split_B(d)
        if hub_block is None:
            if g_local.shape[0] != self.n_local:
                raise ValueError(f"operand has {g_local.shape[0]} rows, this rank owns {self.n_local}")
            xbuf, gathered = self._start_gather(d, g_local)
            own = g_local[hp:] if rp > 0 else None
        else:
            if g_local.shape[0] != rp or hub_block.shape != (self.world * hp, g_local.size(1)):
                raise ValueError("with `hub_block` [W * hp, F] the operand is the rank's regular rows [rp, F]")
            xbuf, gathered, own = hub_block, (lambda: None), (g_local if rp > 0 else None)
        if pending is None:
            pending = []
            if d.A is not None:
                for ch in d.chunks[self.rs_chunks]:
                    pending.append(self._start_reduce(ch, ch.op.spmm(own)))
        gathered()
        y_hub = d.B_hub.spmm(xbuf, None, x2=own) if d.B_hub is not None else None
        if d.B_reg is not None:
            adam(slice(hp, hp + rp), None, d.B_reg, xbuf, own)        # overlaps the reduce-scatter
        if y_hub is not None:
            for finish in pending:
                finish(y_hub)
            adam(slice(0, hp), y_hub, None, None, None)

    def _all_gather_p2p(self, xbuf: Tensor, shard: Tensor, n: Optional[int] = None, own: bool = True):
        """All-gather as W - 1 direct sends and receives per rank, batched into one group call (`n` rows per rank, default
        the whole shard: every peer is sent THE SAME source rows, nothing is packed).  Only for transfers the backend
        orders on the stream (`_stream_ordered`)."""
        hp = self.hp if n is None else n
        ranks = dist.get_process_group_ranks(self.group)
        ops = []
        for q in range(self.world):
            if q != self.rank:
                ops.append(dist.P2POp(dist.isend, shard, ranks[q], group=self.group))
                ops.append(dist.P2POp(dist.irecv, xbuf[q * hp:(q + 1) * hp], ranks[q], group=self.group))
        if own:
            xbuf[self.rank * hp:(self.rank + 1) * hp].copy_(shard)
        return dist.batch_isend_irecv(ops) if ops else []

    def _all_gather_p2p_staged(self, xbuf: Tensor, shard: Tensor, n: Optional[int] = None):
        """The same exchange for device tensors over a host-serviced backend: the shard goes to pinned
        host memory with a BLOCKING copy (ordered after the kernel that produced it), the transfers run
        host to host, and the gathered block returns with a blocking host-to-device copy (ordered before
        the SpMM that reads it, and visible to it: the copy engine, not a host store through the PCIe
        BAR, writes the rows).  No overlap with the local SpMM -- this form only serves rehearsals of
        the N > 1 code path on boxes without RCCL peers."""
        hp, W = (self.hp if n is None else n), self.world
        ranks = dist.get_process_group_ranks(self.group)
        send = self._host_stage("ag_send", shard.shape, shard.dtype)
        recv = self._host_stage("ag_recv", xbuf.shape, xbuf.dtype)
        send.copy_(shard)                                           # device -> host, synchronous
        recv[self.rank * hp:(self.rank + 1) * hp].copy_(send)
        ops = []
        for q in range(W):
            if q != self.rank:
                ops.append(dist.P2POp(dist.isend, send, ranks[q], group=self.group))
                ops.append(dist.P2POp(dist.irecv, recv[q * hp:(q + 1) * hp], ranks[q], group=self.group))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        xbuf.copy_(recv)                                            # host -> device, synchronous
        return []

    # ---- the rows somebody reads (ShardedGCN.forward(rows=...)) ------------------------------------------------------
    def prepare_rows(self, rows: Tensor) -> Optional["_RowsView"]:
        """COLLECTIVE: the local operators of a LAST propagate step whose result is read on `rows` only (a bool mask over this
        rank's n_local rows that the caller keeps), built on every rank at the same time; None unless NO rank reads a hub
        row -- in a TextGCN graph the hubs are the words and the rows read are documents (flat_amazon.py:101,109-114), so
        the hub rows of the result need not exist at all:

            forward   y[rows] = B_r[rows, :] @ [gathered hubs ; own rows]      A_r and its reduce-scatter disappear;
            backward  g is zero outside `rows`, so its hub rows are:           nothing to gather;
                      d x[hubs] = RS(A'_r[:, rows] @ g_reg),  d x[regular] = B'_r[regular, rows] @ g_reg.

        One collective of each propagate step and the entries of the unread rows go.  The ranks agree on whether any hub
        row is read (one all-reduce of a flag) HERE, once per mask and chunk count; `ShardedGCN.forward(rows=mask)` then
        only LOOKS the mask up (`rows_view`) -- which collectives a step contains never depends on a per-rank cache state.
        `sharded.FlatLoop` prepares its two masks itself."""
        if rows.dtype != torch.bool or rows.dim() != 1 or rows.numel() != self.n_local:
            raise ValueError(f"rows must be a bool mask over this rank's {self.n_local} rows")
        if rows.device != self.real.device:
            raise ValueError("rows must live where the graph does")
        cache = self.__dict__.setdefault("_rows_views", {})
        hp = self.hp
        no = torch.zeros(1, dtype=torch.float32, device=self._comm_device())
        if self.rp == 0 or bool((rows[:hp] & self.real[:hp]).any()):
            no += 1.0
        dist.all_reduce(no, op=dist.ReduceOp.MAX, group=self.group)
        view = None if float(no.item()) > 0.0 else _RowsView(self, rows[hp:] & self.real[hp:])
        while len(cache) >= 16:                                # (oldest first: a mask in use is prepared again, collectively)
            cache.pop(next(iter(cache)))
        cache[id(rows)] = (rows, rows._version, self.rs_chunks, view)
        return view

    def rows_view(self, rows: Tensor) -> Optional["_RowsView"]:
        """LOOKUP of what `prepare_rows(rows)` built (local, no collective).  A mask that was not prepared -- or was edited
        in place, or prepared under another chunk count -- raises: preparing it here would enter a collective that peers
        whose cache still holds it never reach."""
        hit = self.__dict__.get("_rows_views", {}).get(id(rows))
        if hit is not None and hit[0] is rows and hit[1] == rows._version and hit[2] == self.rs_chunks:
            return hit[3]
        if rows.dtype != torch.bool or rows.dim() != 1 or rows.numel() != self.n_local:
            raise ValueError(f"rows must be a bool mask over this rank's {self.n_local} rows")
        raise RuntimeError("ShardedGraph: this `rows` mask has not been prepared (or was edited in place / prepared under "
                           "another TGCN_RS_CHUNKS since): call `sg.prepare_rows(mask)` on EVERY rank once (collective), keep "
                           "the tensor and pass the same one every epoch")

    def exchange_rows(self) -> dict:
        """Rows per SpMM this rank receives in each form (what the halo lists prune), for reports."""
        d = self.dirs[0]
        out = {"gather_all": (self.world - 1) * self.hp,
               "gather_halo": int(sum(d.need_counts_l)) - d.need_counts_l[self.rank]}
        if self.rp == 0 and self.pipe_stages:
            out["pipeline_stages"] = self.pipe_stages
        if d.A is not None:
            chs = d.chunks[self.rs_chunks]
            out["reduce_all"] = (self.world - 1) * self.hp
            out["reduce_halo"] = int(sum(sum(c.recv_counts) - c.recv_counts[self.rank] for c in chs))
        return out

    def colsum_real(self, g_local: Tensor) -> Tensor:
        """Column sums over this rank's real rows (padding rows carry zero gradient by
        construction, but are masked anyway)."""
        if self._all_real is None:
            self._all_real = bool(self.real.all().item())          # static: one synchronisation per graph
        if self._all_real:
            # no padding rows on this rank: the matrix as it is -- which lets the HIP engine hand back the sums its
            # producer kernel left (plan.note_colsum: cross-entropy backward, nt GEMM epilogue) without reading it
            return self.engine.colsum(g_local)
        return self.engine.colsum(g_local * self.real.unsqueeze(1).to(g_local.dtype))

    def allreduce_(self, tensors: List[Tensor]) -> None:
        """Sum small replicated tensors (W2, b1, b2 gradients, loss terms) in ONE flat all-reduce."""
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.all_reduce(flat, group=self.group)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


def _fused_optimizer_for(param):
    from .optim import fused_optimizer_for
    return fused_optimizer_for(param)


class _ShardedPropagate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sg: ShardedGraph, xw_local: Tensor, bias: Optional[Tensor]):
        ctx.sg = sg
        ctx.has_bias = bias is not None
        # the operand IS the rank's W1 shard and its optimizer asked for the update inside this backward
        # (pytextgcn_amd.optim.Adam.fuse_into_backward): see ShardedGraph.spmm_adam_w1
        ctx.fused_param = xw_local if (isinstance(xw_local, nn.Parameter)
                                       and _fused_optimizer_for(xw_local) is not None) else None
        return sg.spmm(xw_local.detach(), None if bias is None else bias.detach())

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        sg = ctx.sg
        g = grad_out.contiguous()
        d_xw = None
        if ctx.needs_input_grad[1]:
            p = getattr(ctx, "fused_param", None)
            opt = _fused_optimizer_for(p) if p is not None else None
            if opt is not None:
                opt.assert_no_pending_update(p)                # (before the first collective of this pass)
            # every rank takes the same branch: the registration and the shapes are the same on all of them
            if opt is None or not opt._fused_update_sharded(p, sg, g):
                d_xw = sg.spmm(g, None, transpose=True)
        d_bias = sg.colsum_real(g) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return None, d_xw, d_bias


class _ShardedPropagateCached(torch.autograd.Function):
    """_ShardedPropagate whose forward value was computed earlier from the same (xw, bias) versions:
    no local SpMM and no exchange in the forward, the regular backward."""

    @staticmethod
    def forward(ctx, sg: ShardedGraph, xw_local: Tensor, bias: Optional[Tensor], value: Tensor):
        ctx.sg = sg
        ctx.has_bias = bias is not None
        ctx.fused_param = xw_local if (isinstance(xw_local, nn.Parameter)
                                       and _fused_optimizer_for(xw_local) is not None) else None
        return value.detach()

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        return _ShardedPropagate.backward(ctx, grad_out) + (None,)


def sharded_propagate(sg: ShardedGraph, xw_local: Tensor, bias: Optional[Tensor]) -> Tensor:
    return _ShardedPropagate.apply(sg, xw_local, bias)


class _ShardedPropagateRows(torch.autograd.Function):
    """The last propagate step when only `view`'s rows of the result are read and none of them is a hub row
    (ShardedGraph.rows_view): forward = all-gather + B_r's kept rows (every other row of the result holds the bias),
    backward = A'_r on the kept rows + its reduce-scatter -- one collective each way instead of two."""

    @staticmethod
    def forward(ctx, sg: ShardedGraph, xw_local: Tensor, bias: Optional[Tensor], view: _RowsView):
        ctx.sg, ctx.view = sg, view
        ctx.has_bias = bias is not None
        d = sg.dirs[0]
        x = xw_local.detach().contiguous()
        if x.shape[0] != sg.n_local:
            raise ValueError(f"operand has {x.shape[0]} rows, this rank owns {sg.n_local}")
        b = None if bias is None else bias.detach()
        xbuf, gathered = sg._start_gather(d, x)
        whole = sg._whole_operand(d, x)
        gathered()
        if whole is not None:
            return view.B_rows.spmm(whole, b)
        return view.B_rows.spmm(xbuf, b, x2=x[sg.hp:])

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        sg, view = ctx.sg, ctx.view
        g = grad_out.contiguous()
        d_xw = None
        if ctx.needs_input_grad[1]:
            gr = g[sg.hp:]                                     # rows nobody read carry no gradient: the hub rows are zero
            pending = [sg._start_reduce(ch, ch.op.spmm(gr)) for ch in view.At_chunks]
            d_xw = torch.cat([torch.zeros(sg.hp, g.size(1), dtype=g.dtype, device=g.device), view.Bt_reg.spmm(gr)])
            for finish in pending:
                finish(d_xw[:sg.hp])
        d_bias = sg.colsum_real(g) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return None, d_xw, d_bias, None


class _LocalFeatureBlock(torch.autograd.Function):
    """H_local @ w and d w = H_local^T @ d out for this rank's rows of the hierarchy block H of [I | H]
    (text2graph.py:237-241): two local operators, no exchange -- the sum of the ranks' d w is taken with the other small
    gradients (`ShardedGCN.sync_grads`)."""

    @staticmethod
    def forward(ctx, op, op_t, w: Tensor):
        ctx.op_t = op_t
        return op.spmm(w.detach().contiguous())

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        return None, None, ctx.op_t.spmm(grad_out.contiguous())


class ShardedGCN(nn.Module):
    """The GCN of textgcn/lib/models.py:6-25 for one-hot features (text2graph.py:179: X = I, so
    X @ W1 = W1), with W1 row-sharded like the graph: `layers_w[0]` is [n_local, hidden].  The
    small dense parameters (b1, W2, b2, ...) are replicated; `sync_grads()` sums their gradients
    over ranks after backward (W1's gradient rows are owned and need no reduction)."""

    def __init__(self, sg: ShardedGraph, in_channels, out_channels, n_gcn=2, n_hidden_gcn=64,
                 activation=nn.ReLU, dropout=0.5, narrow_exchange: bool = False, keyed_dropout: Optional[bool] = None,
                 hierarchy_feats: Optional[Tensor] = None):
        """`hierarchy_feats`: this rank's rows [n_local, F_h] (local order: `sg.scatter_rows(H)`; dense or sparse COO) of the
        hierarchy block H of the features [I_N | H] (text2graph.py:237-241; perlevel_amazon.py:122,156); then
        `in_channels == num_nodes + F_h`, X @ W1 = W1[:N] + H @ W1[N:], the rows W1[N:] are a small replicated parameter
        (`weight_h`) whose gradient H^T dXW is summed over the ranks with the other small gradients.
        `narrow_exchange` (opt-in; two layers, a hub partition, widths that are multiples of 4): hub rows cross the
        links at the class width wherever the activation-free network allows it -- pytextgcn_amd/narrow.py; fp32-equal to
        the plain exchange (1e-5), not bit-equal.  `keyed_dropout`: the fused dropout's mask is a function of a node's
        position in the PARTITION and of a seed common to the group (so any rank can evaluate a hub row's mask) instead of
        the local row and a per-rank seed; implied by `narrow_exchange`, available on its own so that the two exchanges
        can be run on identical masks."""
        super().__init__()
        F_h = 0
        if hierarchy_feats is not None:
            if hierarchy_feats.dim() != 2 or hierarchy_feats.size(0) != sg.n_local:
                raise ValueError(f"hierarchy_feats must hold this rank's {sg.n_local} rows (sg.scatter_rows of the block)")
            F_h = int(hierarchy_feats.size(1))
        if in_channels != sg.num_nodes + F_h:
            raise ValueError("ShardedGCN implements one-hot features, alone or followed by a hierarchy block [I | H]: "
                             "in_channels must equal num_nodes (+ the block's width)")
        if F_h and narrow_exchange:
            raise ValueError("narrow_exchange serves the one-hot features alone")
        self.narrow_exchange = bool(narrow_exchange)
        self.keyed_dropout = self.narrow_exchange if keyed_dropout is None else bool(keyed_dropout)
        if self.narrow_exchange:
            if n_gcn != 2 or sg.rp == 0 or out_channels % 4 or n_hidden_gcn % 4:
                raise ValueError("narrow_exchange serves the two-layer network over a hub partition (hubs given, regular "
                                 "rows on every rank) with hidden and class widths that are multiples of 4")
            if not self.keyed_dropout:
                raise ValueError("narrow_exchange needs keyed_dropout (every rank evaluates the hub rows' mask)")
        self._seed_base, self._seed_calls = None, 0
        self.sg = sg
        self.activation = activation()
        self.dropout = dropout
        dims = [n_hidden_gcn] * (n_gcn - 1) + [out_channels]
        self.weights = nn.ParameterList([nn.Parameter(torch.zeros(sg.n_local, dims[0]))])
        self.biases = nn.ParameterList([nn.Parameter(torch.zeros(dims[0]))])
        for i in range(1, len(dims)):
            self.weights.append(nn.Parameter(glorot_(torch.empty(dims[i - 1], dims[i]))))
            self.biases.append(nn.Parameter(torch.zeros(dims[i])))
        self.in_channels = in_channels
        self.weight_h = None
        self._h_ops = None
        if F_h:
            # the rows W1[N:] of the reference's (N + F_h) x h weight: replicated, glorot bound of the whole matrix
            bound = math.sqrt(6.0 / (in_channels + dims[0]))
            self.weight_h = nn.Parameter(torch.empty(F_h, dims[0]).uniform_(-bound, bound))
            hf = hierarchy_feats
            if hf.is_sparse:
                hf = hf.coalesce()
                (r, c), v = hf.indices(), hf.values().float()
            else:
                r, c = torch.nonzero(hf, as_tuple=True)
                v = hf[r, c].float()
            keep = sg.real.to(r.device)[r]                       # padding rows own nothing
            r, c, v = r[keep], c[keep], v[keep]
            self._h_ops = (sg.engine.make_op(r, c, v, sg.n_local, F_h), sg.engine.make_op(c, r, v, F_h, sg.n_local))
        self.sync_replicated_parameters()

    def sync_replicated_parameters(self) -> None:
        """The replicated parameters (W2 ..., the hierarchy rows W1[N:]; the biases start at zero) are drawn from each rank's
        own generator: make them the first rank's everywhere (COLLECTIVE; called by the constructor, and to be called again
        after moving the module to its device if the ranks initialise them differently afterwards).  Loading a state dict
        (`load_full_state_dict`) overwrites them anyway."""
        sg = self.sg
        if sg.group is None or sg.world == 1:
            return
        small = list(self.weights)[1:] + list(self.biases) + ([self.weight_h] if self.weight_h is not None else [])
        flat = torch.cat([p.detach().reshape(-1) for p in small])
        buf = flat.to(sg._comm_device()) if flat.device != sg._comm_device() else flat.clone()
        dist.broadcast(buf, src=dist.get_global_rank(sg.group, 0), group=sg.group)
        buf = buf.to(flat.device)
        off = 0
        with torch.no_grad():
            for p in small:
                n = p.numel()
                p.copy_(buf[off:off + n].view_as(p))
                off += n

    @classmethod
    def for_data(cls, sg: ShardedGraph, g, out_channels, **kw) -> "ShardedGCN":
        """The model for the graph object Text2GraphTransformer returns (text2graph.py:192-193), built like the reference
        builds its own (`model(g.x.shape[1], n_classes, ...)`, flat_amazon.py:80): `g.x` is the sparse identity, or
        [I_N | H] with the hierarchy block (text2graph.py:226-246), whose rows are handed to this rank."""
        from .conv import is_sparse_identity, split_identity_block
        x, N = g.x, sg.num_nodes
        if x.size(0) != N or not x.is_sparse:
            raise ValueError("ShardedGCN serves the sparse one-hot features of text2graph.py:226-246 ([I_N] or [I_N | H])")
        if x.size(1) == N:
            if not is_sparse_identity(x):
                raise ValueError("g.x is square but not the identity")
            return cls(sg, N, out_channels, **kw)
        H = split_identity_block(x)
        if H is None:
            raise ValueError("g.x is wider than it is tall but its first N columns are not the identity")
        # this rank's rows of the block, selected on the sparse indices (the dense [N, F_h] block would be 4 GB at
        # N = 2 M, F_h = 500): global row id -> local slot through the partition
        Hc = H.coalesce()
        (r, c), v = Hc.indices(), Hc.values()
        p = sg.part
        mine = p.owner.to(r.device)[r] == sg.rank
        r, c, v = r[mine], c[mine], v[mine]
        slot = p.slot.to(r.device)[r]
        local = torch.where(p.hub_mask.to(r.device)[r], slot, sg.hp + slot)
        feats = torch.sparse_coo_tensor(torch.stack([local, c]), v, (sg.n_local, H.size(1))).coalesce()
        return cls(sg, x.size(1), out_channels, hierarchy_feats=feats, **kw)

    def load_full_state_dict(self, sd: dict) -> None:
        """From a single-device GCN state_dict (`layers.{i}.weight` / `.bias`, PyG-1.6.3 layout)."""
        N = self.sg.num_nodes
        with torch.no_grad():
            for i, (w, b) in enumerate(zip(self.weights, self.biases)):
                fw = sd[f"layers.{i}.weight"].to(w.device)
                if i == 0:
                    w.copy_(self.sg.scatter_rows(fw[:N]))
                    if self.weight_h is not None:
                        self.weight_h.copy_(fw[N:])
                else:
                    w.copy_(fw)
                b.copy_(sd[f"layers.{i}.bias"].to(b.device))

    def full_state_dict(self) -> dict:
        sd = {}
        for i, (w, b) in enumerate(zip(self.weights, self.biases)):
            if i == 0:
                full = self.sg.gather_rows(w.detach())
                sd["layers.0.weight"] = full if self.weight_h is None else torch.cat([full, self.weight_h.detach()])
            else:
                sd[f"layers.{i}.weight"] = w.detach().clone()
            sd[f"layers.{i}.bias"] = b.detach().clone()
        return sd

    def _common_seed(self) -> Tensor:
        """The dropout seed of this forward pass, the SAME on every rank of the group (keyed dropout): a base drawn by
        the group's first rank and broadcast once (collective: every rank reaches its first training forward together),
        advanced by a counter that moves in lock step.  A device int64[1], written without a synchronisation."""
        dev = self.weights[0].device
        if self._seed_base is None:
            t = torch.empty(1, dtype=torch.int64, device=dev if dist.get_backend(self.sg.group) == "nccl" else "cpu").random_()
            dist.broadcast(t, src=dist.get_global_rank(self.sg.group, 0), group=self.sg.group)
            self._seed_base = int(t.item())
        self._seed_calls += 1
        return torch.full((1,), _wrap64(self._seed_base + self._seed_calls * 0x9E3779B97F4A7C15), dtype=torch.int64, device=dev)

    def forward(self, g=None, rows: Optional[Tensor] = None) -> Tensor:
        """Logits of this rank's rows, [n_local, out_channels] (padding rows hold the bias).
        `rows` (as in `GCN.forward`; a bool mask over the rank's n_local rows that the caller keeps): the rows of the
        result that will be READ.  When no rank reads a hub row (`ShardedGraph.rows_view`; the words of a TextGCN graph)
        the last propagate step runs without A_r and its reduce-scatter, its backward without the all-gather, and every
        row outside `rows` holds the last bias.  Every rank passes a mask in the same call, or none does, and the mask was
        handed to `sg.prepare_rows(mask)` on every rank before (collective, once; `sharded.FlatLoop` does it itself)."""
        eng = self.sg.engine
        fused_drop = self.training and 0.0 < self.dropout < 1.0
        view = self.sg.rows_view(rows) if (rows is not None and len(self.weights) > 1) else None
        if self.narrow_exchange:
            from . import conv
            from .narrow import NarrowGCN2
            seed = self._common_seed() if fused_drop else None
            if self.training and self.dropout >= 1.0:
                raise ValueError("narrow_exchange: dropout must be < 1")
            cache = self.__dict__.setdefault("_narrow_cache", {}) if conv._REUSE else None
            return NarrowGCN2.apply(self.sg, self.weights[0], self.biases[0], self.weights[1], self.biases[1],
                                    float(self.dropout) if fused_drop else 0.0, seed, cache, view)
        x = self._layer1()
        last = len(self.weights) - 1
        for i in range(1, len(self.weights)):
            if fused_drop and self.keyed_dropout and hasattr(eng, "xw_dropout_keyed"):
                from .narrow import own_row_keys
                xw = eng.xw_dropout_keyed(x, self.weights[i], float(self.dropout), self._common_seed(), own_row_keys(self.sg))
            elif fused_drop and hasattr(eng, "xw_dropout"):
                xw = eng.xw_dropout(x, self.weights[i], float(self.dropout))
            else:
                x = nn.functional.dropout(x, p=self.dropout, training=self.training)
                xw = eng.xw(x, self.weights[i])
            if view is not None and i == last:
                x = _ShardedPropagateRows.apply(self.sg, xw, self.biases[i], view)
            else:
                x = sharded_propagate(self.sg, xw, self.biases[i])
        return x

    def _layer1(self) -> Tensor:
        """M @ W1 + b1 on the one-hot features.  With pytextgcn_amd.enable_activation_reuse() the value of the
        preceding call is handed out again while W1 and b1 are unchanged (the eval forward of epoch k
        and the training forward of epoch k + 1, flat_amazon.py:100-109): one distributed SpMM and its
        exchange less per epoch, bitwise the same result.  Every rank takes the same branch (the
        version counters move in lock step), so the collectives stay matched."""
        from . import conv
        w, b = self.weights[0], self.biases[0]
        if self.weight_h is not None:
            # [I | H] features: X @ W1 = W1[:N] + H @ W1[N:] on this rank's rows; the operand is then a value, not the
            # parameter (no optimizer inside the backward SpMM, no reuse across calls)
            return sharded_propagate(self.sg, w + _LocalFeatureBlock.apply(self._h_ops[0], self._h_ops[1], self.weight_h), b)
        if not conv._REUSE:
            return sharded_propagate(self.sg, w, b)
        key = (w.data_ptr(), w._version, b.data_ptr(), b._version)
        hit = getattr(self, "_reuse_cache", None)
        if hit is not None and hit[0] == key:
            if torch.is_grad_enabled() and (w.requires_grad or b.requires_grad):
                return _ShardedPropagateCached.apply(self.sg, w, b, hit[1])
            return hit[1].detach()
        out = sharded_propagate(self.sg, w, b)
        self._reuse_cache = (key, out.detach())
        return out

    def sync_grads(self) -> None:
        small = list(self.weights)[1:] + list(self.biases) + ([self.weight_h] if self.weight_h is not None else [])
        grads = [p.grad for p in small if p.grad is not None]
        if grads:
            self.sg.allreduce_(grads)


def sharded_cross_entropy(sg: ShardedGraph, logits_local: Tensor, y_local: Tensor,
                          mask_local: Tensor, return_pred: bool = False):
    """CrossEntropyLoss(reduction='mean') over the GLOBAL masked rows (flat_amazon.py:82,101-102):
    local sum of row losses divided by the global row count, so that summing the replicated
    gradients over ranks reproduces the single-device gradient.  Returns this rank's share of the
    loss; all-reduce it (sum) for the value the reference prints.  `return_pred=True` adds the arg-max
    of every local row (same pass on the HIP engine)."""
    # masks are static: count (and sync) once per mask object / version; the entry holds the tensor, so
    # its id() cannot be recycled while the entry lives
    cache = sg.__dict__.setdefault("_count_cache", {})
    hit = cache.get(id(mask_local))
    if hit is not None and hit[0] is mask_local and hit[1] == mask_local._version:
        cached = hit[2]
    else:
        c = mask_local.sum().to(torch.float64).reshape(1)
        dist.all_reduce(c, group=sg.group)
        cached = int(c.item())
        if len(cache) > 16:
            cache.clear()
        cache[id(mask_local)] = (mask_local, mask_local._version, cached)
    if hasattr(sg.engine, "masked_ce"):                  # fused HIP kernel, global divisor
        return sg.engine.masked_ce(logits_local, y_local, mask_local, cached, return_pred)
    cnt = torch.tensor(float(cached), device=logits_local.device)
    if bool(mask_local.any()):
        part = nn.functional.cross_entropy(logits_local[mask_local], y_local[mask_local], reduction="sum")
    else:
        part = logits_local.sum() * 0.0
    return (part / cnt, logits_local.argmax(1)) if return_pred else part / cnt


class FlatLoop:
    """The epoch of flat_amazon.py:99-117 on the 1-D partition behind one object (the multi-GPU counterpart of
    `pytextgcn_amd.train.FlatLoop`): training step (forward, cross-entropy over the GLOBAL training rows, backward, the small
    gradients summed over the ranks, Adam(amsgrad)) + evaluation pass, with W1's update inside the backward SpMM, layer 1 kept
    from the evaluation pass for the next training pass, and -- `needed_rows_only` -- the last propagate step on the rows
    that are read (`ShardedGCN.forward(rows=...)`: one collective less each way when no rank reads a hub row).

        loop = sharded.FlatLoop(model, y_local, train_local, val_local, lr=lr)
        loss, val_loss, pred_val, pred_train = loop.epoch()       # global losses; class ids of THIS rank's rows (numpy)

    Every rank calls `epoch()` together.  `optimizer`: None = `pytextgcn_amd.optim.Adam` (GPU); any torch optimizer over
    `model.parameters()` otherwise.  `enable_activation_reuse` is ON while the loop lives (`close()` / `with`)."""

    def __init__(self, model: ShardedGCN, y_local: Tensor, train_local: Tensor, val_local: Tensor, lr: float = 0.05,
                 amsgrad: bool = True, weight_decay: float = 0.0, needed_rows_only: bool = True, optimizer=None):
        from . import conv
        self.model, self.sg = model, model.sg
        self.y, self.train_mask, self.val_mask = y_local, train_local, val_local
        if optimizer is None:
            from .optim import Adam
            optimizer = Adam(model.parameters(), lr=lr, amsgrad=amsgrad, weight_decay=weight_decay)
        self.optimizer = optimizer
        if hasattr(optimizer, "fuse_into_backward"):
            optimizer.fuse_into_backward(model.weights[0])     # (declines by itself where it does not apply)
        self._saved = conv._REUSE
        conv.enable_activation_reuse(True)
        self.rows_train = train_local if needed_rows_only else None
        self.rows_eval = (train_local | val_local) if needed_rows_only else None
        if needed_rows_only and len(model.weights) > 1:
            for m in (self.rows_train, self.rows_eval):            # collective, once: forward(rows=...) only looks them up
                self.sg.prepare_rows(m)
        self.epochs = 0

    def epoch(self):
        model, sg, opt = self.model, self.sg, self.optimizer
        model.train()
        loss = sharded_cross_entropy(sg, model(rows=self.rows_train), self.y, self.train_mask)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        model.sync_grads()
        opt.step()
        model.eval()
        with torch.no_grad():
            val_loss, pred = sharded_cross_entropy(sg, model(rows=self.rows_eval), self.y, self.val_mask, return_pred=True)
            both = torch.stack([loss.detach(), val_loss.detach()]).float()      # this rank's shares of the two means
            dist.all_reduce(both, group=sg.group)
            pred_val, pred_train = pred[self.val_mask].cpu().numpy(), pred[self.train_mask].cpu().numpy()
        self.epochs += 1
        lo = both.tolist()
        return lo[0], lo[1], pred_val, pred_train

    def close(self) -> None:
        if self._saved is not None:
            from . import conv
            conv.enable_activation_reuse(self._saved)
            self._saved = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
