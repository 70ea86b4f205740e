"""Node-range sharding of the DGG hot path across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no distributed code; the all-pairs similarity shards by rows: every
output row i needs x_i, all candidate x_j, k_i and the row sums rs_j of its selected neighbours, and nothing else
(sort / ramp act on dim=-1, reference dgm.py:1404-1420).  Rank r owns rows [r*ceil(N/G), ...) of X, of the ELL
adjacency and of the conv output.  Collectives per step:

  forward   all-gather xp [N,h]  (projected features of the OWN rows; the scoring needs every candidate's xp_j)
            all-gather X  [N,d]  (needed only by the SpMM gather: issued asynchronously behind xp, it crosses the
                                  fabric while the top-k, soft-k and partition kernels run)
            all-gather rs [N]    (row sums for the symmetric-ish normalisation, model.py:1215-1218)
  backward  all-reduce da [N]   (d loss / d rs^-1/2: neighbour-side terms land on non-owner ranks)
            all-reduce of the replicated weight gradients (one flat bucket, ~35k floats)
            [reduce-scatter dX [N,d] only when the input features need a gradient]

Every rank projects only its own rows (xp = leaky(X We^T + be)) and gathers the rest.  The weight
gradient of the projection is formed from each rank's PARTIAL dxp against the full X / xp and summed by the weight
all-reduce, so the [N,h] gradient itself never crosses the fabric.

REPLICATED FEATURES (`x_full=`): when the DGG input is DATA (GCN_DGG / SAGE_DGG / GCNII_DGG all feed it the raw node
features, model.py:1266, 720) it never changes between steps, so it is placed on every GPU once at load (N*d*4 bytes:
0.4 GB of the 288 GB for 800k nodes) and NO feature tensor crosses the fabric per step: every rank projects all N rows
itself (N*d*h*2 flop: cheaper than receiving (G-1)/G of xp over xGMI) and the forward's only collective is the all-gather
of the row sums.  The gathers above remain the path for inputs that are activations (`x_grad`, DGG on hidden layers).

`kern` is the kernel namespace (dgg_amd.ops on the GPU; tests substitute a CPU stand-in built on the oracle so
that the partition / collective logic is exercised with gloo, world_size 2, without a GPU).
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(N, world, rank):
    per = (N + world - 1) // world
    r0 = min(rank * per, N)
    return r0, min(r0 + per, N), per


class _Gather:
    """A (possibly still running) all-gather of row shards: .get() waits and returns the [N, ...] tensor.  On RCCL the
    wait is a stream dependency (no host block); shards are padded to `per` rows for the fixed-size collective."""

    def __init__(self, t_local, N, per, group, async_op):
        pad = per - t_local.shape[0]
        self.src = t_local.contiguous() if pad == 0 else torch.cat([t_local, t_local.new_zeros((pad,) + tuple(t_local.shape[1:]))])
        self.out = self.src.new_empty((dist.get_world_size(group) * per,) + tuple(t_local.shape[1:]))
        self.N = N
        self.work = dist.all_gather_into_tensor(self.out, self.src, group=group, async_op=async_op)

    def get(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.out[:self.N]


def _all_gather_rows(t_local, N, per, group):
    """[n_loc, ...] -> [N, ...]"""
    if dist.get_world_size(group) == 1 and os.environ.get("DGG_FORCE_COLLECTIVES") != "1":
        return t_local
    return _Gather(t_local, N, per, group, False).get()


class ShardedDGGConv:
    """One DGG (all-pairs, u-v-dist / x / k_times_edge_prob) + normalise + GCNConv layer, forward and backward,
    on a row shard.  Parameters are a dict with the reference's names (dgm.py:1097-1143, model.py:583)."""

    PARAM_KEYS = ("We", "be", "Wk", "bk", "W1", "b1", "Wmu", "bmu", "Wp", "bp", "Wc")

    def __init__(self, kern, N, group=None, K=64, t=-0.05, noise_mode=2, seed=(1234, 0), mode=0, algo=0, x_grad=False, x_full=None):
        self.kern, self.N, self.group = kern, N, group
        assert x_full is None or not x_grad, "replicated features are data: they cannot take a gradient"
        self.x_full = x_full                                 # [N,d] static node features present on every rank, or None
        self.K, self.t, self.noise_mode, self.seed, self.mode, self.algo, self.x_grad = K, t, noise_mode, seed, mode, algo, x_grad
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # DGG_FORCE_COLLECTIVES=1 issues every collective even in a 1-rank group (exercises the RCCL calls on a 1-GPU box)
        self.coll = self.world > 1 or (dist.is_initialized() and os.environ.get("DGG_FORCE_COLLECTIVES") == "1")
        self.emulate = None
        self.r0, self.r1, self.per = shard_bounds(N, self.world, self.rank)

    def emulate_rank(self, world, rank):
        """TIMING DIAGNOSTIC (bench.py --emulate-world): do the work of `rank` of `world` in a single process -- own row range
        against all N columns, replicated features -- with the collectives left out and the other ranks' row sums faked by
        tiling the own ones.  Results are not meaningful; the kernel sequence and sizes are those of the real rank."""
        assert self.x_full is not None and not dist.is_initialized()
        self.world, self.rank, self.coll, self.emulate = world, rank, False, (world, rank)
        self.r0, self.r1, self.per = shard_bounds(self.N, world, rank)

    def forward(self, x_local, deg_full, P):
        kern = self.kern
        s = {}
        repl = self.x_full is not None
        xp = kern.linear_fwd(self.x_full if repl else x_local, P["We"], P["be"], 1, 0)
        if self.coll and not repl:                  # xp first (the top-k waits for it), X streams in behind it
            g_xp = _Gather(xp, self.N, self.per, self.group, True)
            g_X = _Gather(x_local, self.N, self.per, self.group, True)
        s["xk"] = xk = kern.linear_fwd(x_local, P["Wk"], P["bk"], 1, 0)
        s["mu_sd"] = mu_sd = kern.degree_stats(deg_full)
        deg_local = deg_full[self.r0:self.r1].contiguous()
        s["k"], s["z"], s["u"], s["feat"] = kern.knet_x_fwd(xk, deg_local, mu_sd, P["W1"], P["b1"], P["Wmu"], P["bmu"],
                                                          P["Wp"].reshape(-1), P["bp"])
        s["xp"] = xp = g_xp.get() if (self.coll and not repl) else xp
        s["idx"], s["val"] = kern.allpairs_topk(xp, self.K, self.t, self.noise_mode, None, self.seed,
                                                rows=(self.r0, self.r1), algo=self.algo, k_limit=s["k"])
        s["w"], rs_local = kern.softk_fwd(s["idx"], s["val"], s["k"], self.mode)
        # destination-bucket partition of the active entries: the backward's column-side terms run on it (no atomics)
        s["part"] = kern.part_build(s["idx"], s["w"], self.N) if hasattr(kern, "part_build") else None
        s["rs"] = rs = _all_gather_rows(rs_local, self.N, self.per, self.group) if self.coll else rs_local
        if self.emulate is not None:
            s["rs"] = rs = rs_local.repeat(self.world)[:self.N].contiguous()
        s["ahat"] = kern.normalize_fwd(s["idx"], s["w"], rs, self.r0)
        s["X"] = X = self.x_full if repl else (g_X.get() if self.coll else x_local)
        s["Y"] = kern.spmm_fwd(s["idx"], s["ahat"], X)
        s["Z"] = kern.linear_fwd(s["Y"], P["Wc"], None, 2, 1)
        self.saved = s
        return s["Z"]

    def backward(self, dZ, x_local, P):
        """-> dict of parameter gradients (summed over ranks) and, if x_grad, 'x' = d loss / d x_local."""
        kern, s = self.kern, self.saved
        if hasattr(kern, "zero_pool"):                   # every zero-initialised accumulator of the backward from ONE filled buffer
            ncols, h, d = s["xp"].shape[0], s["xp"].shape[1], s["X"].shape[1]
            need = ncols * (h + 2) + 2 * sum(int(v.numel()) for v in P.values()) + (ncols * d if self.x_grad else 0) + 65536
            with kern.zero_pool(s["xp"].device, need):
                return self._backward(dZ, x_local, P)
        return self._backward(dZ, x_local, P)

    def _backward(self, dZ, x_local, P):
        kern, s = self.kern, self.saved
        g = {}
        dY, g["Wc"], _ = kern.linear_bwd(s["Y"], P["Wc"], s["Z"], dZ, 2, 1, True, False)
        part = s.get("part")
        fused = None
        if not self.x_grad and part is not None and hasattr(kern, "sddmm_norm"):
            fused = kern.sddmm_norm(s["idx"], s["ahat"], s["w"], s["rs"], s["X"], dY, self.r0, part, True)
        if fused is not None:                            # SDDMM + row side of the normalisation backward in one pass
            (dA, da), dX = fused, None
        else:
            if part is not None and self.x_grad:         # dX through the destination-ordered partition (no entry-wise atomics)
                dA, dX = kern.spmm_bwd(s["idx"], s["ahat"], s["X"], dY, True, True, part=part, part_cols=self.N)
            else:
                dA, dX = kern.spmm_bwd(s["idx"], s["ahat"], s["X"], dY, self.x_grad, True)
            da = kern.norm_bwd_da(s["idx"], s["w"], s["rs"], dA, self.r0, part) if part is not None else \
                kern.norm_bwd_da(s["idx"], s["w"], s["rs"], dA, self.r0)
        if self.coll:
            dist.all_reduce(da, group=self.group)
        fused = kern.softk_edge_bwd(s["xp"], s["idx"], s["val"], s["k"], dA, s["rs"], da, self.r0, self.t, self.noise_mode != 0,
                                    self.mode, True, part) if (part is not None and hasattr(kern, "softk_edge_bwd")) else None
        if fused is not None:                            # ramp + normalisation backward inside the row kernel of the score backward
            dxp, dk, _ = fused
        else:
            dval, dk = kern.softk_bwd(s["idx"], s["val"], s["k"], dA, s["rs"], da, self.r0, self.mode, True)
            dxp = kern.edge_bwd(s["xp"], s["idx"], s["val"], dval, self.r0, self.t, self.noise_mode != 0, part) \
                if part is not None else kern.edge_bwd(s["xp"], s["idx"], s["val"], dval, self.r0, self.t, self.noise_mode != 0)
        dX1, g["We"], g["be"] = kern.linear_bwd(s["X"], P["We"], s["xp"], dxp, 1, 0, self.x_grad, True)
        dxk, g["W1"], g["b1"], g["Wmu"], g["bmu"], dWp, g["bp"] = kern.knet_x_bwd(
            s["xk"].shape[1], s["mu_sd"], P["W1"], P["Wmu"], P["bmu"], P["Wp"].reshape(-1), s["z"], s["u"], s["feat"], dk)
        g["Wp"] = dWp.reshape(P["Wp"].shape)
        dx2, g["Wk"], g["bk"] = kern.linear_bwd(x_local, P["Wk"], s["xk"], dxk, 1, 0, self.x_grad, True)
        if self.coll:
            flat = torch.cat([g[k].reshape(-1) for k in self.PARAM_KEYS])
            dist.all_reduce(flat, group=self.group)
            o = 0
            for k in self.PARAM_KEYS:
                n = g[k].numel()
                g[k] = flat[o:o + n].view_as(g[k])
                o += n
        if self.x_grad:
            dXf = dX + dX1                               # [N,d] partial: neighbour-side terms of every rank
            if self.coll:
                pad = self.world * self.per - self.N
                if pad:
                    dXf = torch.cat([dXf, dXf.new_zeros((pad, dXf.shape[1]))])
                out = dXf.new_empty((self.per, dXf.shape[1]))
                dist.reduce_scatter_tensor(out, dXf.contiguous(), group=self.group)
                dXf = out[: self.r1 - self.r0]
            g["x"] = dXf + dx2
        return g
