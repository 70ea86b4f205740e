"""Sparse adjacency containers exchanged between the DGG and the graph-conv layers.

The reference passes torch sparse COO [N,N] tensors and densifies them at every hand-over
(model.py:1249-1251, 1264, 1274; dgm.py:1298).  Here the learned graph stays in a fixed-width ELL layout
(idx int32 [N,K], values fp32 [N,K]) end to end; `.to_dense()` / `.to_sparse()` exist so that reference-style
callers (`unnorm_adj.to_dense()`, model.py:1274) keep working on small graphs.
"""
import torch

from . import ops


class AllPairs:
    """Candidate set = every ordered pair (the reference equivalent is a complete in_adj whose row sums are the
    prior degrees): carries the prior-degree vector read by the k-net (dgm.py:1568)."""

    def __init__(self, prior_degree):
        self.prior_degree = prior_degree
        n = prior_degree.shape[0]
        self.shape = (n, n)
        self.device = prior_degree.device


class EllAdjacency:
    """Row-major fixed-width sparse matrix: entry (i, idx[i,r]) = values[i,r]; idx == -1 marks padding.
    layout (ops.ChunkLayout, optional): CHUNKED rows -- the learned degrees of an all-pairs graph outgrew the 64-rank list, so row i is
    the chunks [cptr[i], cptr[i+1]) of idx / values [chunks,64] (rank r of the row = entry r % 64 of its chunk r / 64); the reference
    has no width limit (dense [N,N] rows, dgm.py:1402-1421)."""

    def __init__(self, idx, values, n_cols, rs=None, k=None, score=None, normalized=False, part=None, owner=None, partp=None, layout=None):
        self.idx, self._values, self.n_cols = idx, values, n_cols
        self.layout = layout if (layout is not None and layout.wide) else None
        # (payload partition, row sums of the unnormalised weights) when `values` are the normalised values that partition was
        # built with (DGG_LearnableK_debug.forward_conv): the backward of A @ X then runs by destination, without atomics
        self.partp = partp
        self.owner = owner          # the DGG module that produced it (its ELL-width bound is checked when the matrix is densified)
        self.rs, self.k, self.score, self.normalized = rs, k, score, normalized
        self.part = part            # destination-ordered partition of the active entries (ops.part_build), if one was built
        self.shape = (idx.shape[0] if self.layout is None else self.layout.rows, n_cols)
        self.device = idx.device

    # --- reference-style accessors ---------------------------------------------------------------------------
    def values(self):
        return self._values

    def ell(self):
        return self.idx, self._values

    def coalesce(self):
        return self

    def to_dense(self):
        """Differentiable densification (small graphs / tests only)."""
        if self.owner is not None:
            self.owner.check_ell_bound()
        valid = self.idx >= 0
        cols = self.idx.clamp(min=0).long()
        out = torch.zeros(self.shape, device=self.device, dtype=self._values.dtype)
        vals = torch.where(valid, self._values, torch.zeros_like(self._values))
        if self.layout is not None:              # chunked rows: a chunk's entries land in the row of its node
            rows = self._chunk_rows().unsqueeze(1).expand_as(self.idx)
            return out.index_put((rows.reshape(-1), cols.reshape(-1)), vals.reshape(-1), accumulate=True)
        return out.scatter_add(1, cols, vals)

    def _chunk_rows(self):
        return self.layout.cnode.long()[:self.idx.shape[0]]

    def to_sparse(self):
        if self.owner is not None:
            self.owner.check_ell_bound()
        valid = self.idx >= 0
        base = torch.arange(self.shape[0], device=self.device) if self.layout is None else self._chunk_rows()
        rows = base.unsqueeze(1).expand_as(self.idx)[valid]
        return torch.sparse_coo_tensor(torch.stack([rows, self.idx[valid].long()]), self._values[valid], self.shape)

    def to_csr(self):
        """CsrAdjacency with the stored entries (idx >= 0) as its pattern and the same (differentiable) values: rows of any width for
        the layers that take the learned graph as a separate module"""
        valid = self.idx >= 0
        base = torch.arange(self.idx.shape[0], device=self.device) if self.layout is None else self._chunk_rows()
        erow = base.unsqueeze(1).expand_as(self.idx)[valid]
        rowptr = torch.zeros(self.shape[0] + 1, device=self.device, dtype=torch.int64)
        rowptr[1:] = torch.bincount(erow, minlength=self.shape[0]).cumsum(0)
        return CsrAdjacency(rowptr, self.idx[valid].contiguous(), erow.to(torch.int32).contiguous(), self._values[valid], self.shape[0], k=self.k)

    # --- fast path -------------------------------------------------------------------------------------------
    def row_sums(self):
        if self.rs is None:
            per = self._values.detach().sum(1)
            self.rs = per if self.layout is None else torch.zeros(self.shape[0], device=self.device, dtype=per.dtype).index_add_(0, self._chunk_rows(), per)
        return self.rs

    def normalize(self):
        """D^-1/2 A D^-1/2 with ROW sums on both sides (normalize_adj, model.py:1205-1219)."""
        if self.layout is not None:              # chunked rows as a separate module: the CSR kernels (rows of any width)
            return self.to_csr().normalize()
        rs = self.row_sums()
        ahat = ops.EllNormalizeFn.apply(self._values, self.idx, rs, self.part)
        return EllAdjacency(self.idx, ahat, self.n_cols, rs=None, k=self.k, score=self.score, normalized=True, part=self.part,
                            owner=self.owner)

    def matmul(self, X, act=ops.ACT_NONE):
        """act(A @ X) (torch.mm(adj, x), model.py:594; act = ReLU fuses GCNConv's activation into the aggregation)."""
        # weights produced by the DGG ramp: an exact zero is a saturated ramp whose gradient vanishes too
        if self.layout is not None and self.partp is None:
            if not ops.backward_will_follow(self._values, X):      # inference (no_grad / a hipGraph capture): the chunked aggregation kernel
                return ops.spmm_fwd(self.idx, self._values.detach(), X.detach().contiguous(), act, layout=self.layout)
            out = self.to_csr().matmul(X)
            return torch.relu(out) if act == ops.ACT_RELU else out
        return ops.EllSpmmFn.apply(self._values, self.idx, X, self.k is not None, self.part, act, self.partp, self.layout)

    __matmul__ = matmul


class CsrAdjacency:
    """Sparse matrix with the pattern of a coalesced COO in_adj (rowptr int64 [N+1], col int32 [E], erow int32 [E]) and
    its own values [E]: what the `DGG` class returns (every candidate edge kept, dgm.py:1804-1815), rows of any width."""

    def __init__(self, rowptr, col, erow, values, n, k=None):
        self.rowptr, self.col, self.erow, self._values, self.k = rowptr, col, erow, values, k
        self.shape = (n, n)
        self.device = values.device

    def values(self):
        return self._values

    def coalesce(self):
        return self

    def indices(self):
        return torch.stack([self.erow.long(), self.col.long()])

    def to_sparse(self):
        return torch.sparse_coo_tensor(self.indices(), self._values, self.shape)

    def to_dense(self):
        out = torch.zeros(self.shape, device=self.device, dtype=self._values.dtype)
        return out.index_put((self.erow.long(), self.col.long()), self._values, accumulate=True)

    def normalize(self):
        return CsrAdjacency(self.rowptr, self.col, self.erow, ops.CsrNormalizeFn.apply(self._values, self.rowptr, self.col),
                            self.shape[0], k=self.k)

    def matmul(self, X):
        return ops.CsrSpmmFn.apply(self._values, self.rowptr, self.col, X)

    __matmul__ = matmul


# The candidate graph is DATA: the same sparse tensor object comes in on every forward (train_small_graphs.py builds it once), and
# turning it into the CSR arrays costs ~15 tiny launches (coalesce, coo -> csr, casts, a float64 cumsum, two index ops) -- a sixth of
# the Pubmed-shape step.  Both conversions are cached per tensor OBJECT (weak reference) and version of its values: a new tensor,
# or one modified in place, is converted again.  Not during a hipGraph capture of an unseen tensor (nothing is cached from there).
_CSR_CACHE = {}


def _cached(kind, in_adj, make):
    import weakref
    key = (kind, id(in_adj))
    ver = in_adj._values()._version if in_adj.is_sparse else in_adj._version
    ent = _CSR_CACHE.get(key)
    if ent is not None and ent[0]() is in_adj and ent[1] == ver:
        return ent[2]
    out = make()
    if not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing():
        for k_ in [k_ for k_, v in _CSR_CACHE.items() if v[0]() is None]:
            del _CSR_CACHE[k_]
        _CSR_CACHE[key] = (weakref.ref(in_adj), ver, out)
    return out


def csr_pattern(in_adj):
    """coalesced sparse COO -> (rowptr int64, col int32, erow int32)"""
    def make():
        a = in_adj.coalesce()
        ind = a.indices()
        rowptr = torch._convert_indices_from_coo_to_csr(ind[0], a.shape[0], out_int32=False)
        return rowptr, ind[1].to(torch.int32).contiguous(), ind[0].to(torch.int32).contiguous()
    return _cached("pattern", in_adj, make)


def ell_from_dense(A, K=ops.DEFAULT_K):
    """Dense [N,N] -> ELL keeping the K largest entries per row (exact when every row has <= K non-zeros)."""
    assert (A != 0).sum(1).max() <= K, "row has more non-zeros than the ELL width"
    idx, val = ops.select_scores(A.abs().contiguous(), K)
    vals = torch.gather(A, 1, idx.clamp(min=0).long())
    valid = (idx >= 0) & (vals != 0)
    idx = torch.where(valid, idx, torch.full_like(idx, -1))
    return EllAdjacency(idx, torch.where(valid, vals, torch.zeros_like(vals)), A.shape[1])


def csr_candidates(in_adj):
    """torch sparse COO [N,N] (coalesced, self loops included by the caller as in model.py:1249-1264) ->
    (rowptr int64 [N+1], col int32 [E], deg fp32 [N] = row sums of the stored values)."""
    def make():
        a = in_adj.coalesce()
        N = a.shape[0]
        ind = a.indices()
        rowptr = torch._convert_indices_from_coo_to_csr(ind[0], N, out_int32=False)
        col = ind[1].to(torch.int32).contiguous()
        # row sums, deterministic (index_add_ uses float atomics: the last bit would change from call to call, and the raw
        # degrees are an input of the u-v-deg scorers): differences of a float64 running sum at the row boundaries
        cs = torch.cat([a.values().new_zeros(1, dtype=torch.float64), a.values().double().cumsum(0)])
        deg = (cs[rowptr[1:]] - cs[rowptr[:-1]]).float()
        return rowptr, col, deg
    return _cached("candidates", in_adj, make)
