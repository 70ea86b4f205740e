"""Node-range sharding of the DGG hot path across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no distributed code; the all-pairs similarity shards by rows: every
output row i needs x_i, all candidate x_j, k_i and the row sums rs_j of its selected neighbours, and nothing else
(sort / ramp act on dim=-1, reference dgm.py:1404-1420).  Rank r owns rows [r*ceil(N/G), ...) of X, of the ELL
adjacency and of the conv output.

The graph conv is evaluated as relu(A (x W)) instead of the reference's relu((A x) W) (model.py:594-598; equal up to
fp32 reassociation): the aggregation gathers the out_features-wide projected rows H = X Wc, not the d-wide inputs, and
the backward's three column-walking terms (SDDMM, transposed SpMM, neighbour side of the normalisation backward) share
ONE gathered cotangent row per edge (ops.conv_bwd_cols).  Every per-node array a rank gathers from is therefore a
PROJECTION of the features: xp (scoring) and H (aggregation).

Two exchange schemes:

REPLICATED FEATURES (`x_full=`): when the DGG input is DATA (GCN_DGG / SAGE_DGG / GCNII_DGG all feed it the raw node
features, model.py:1266, 720) it never changes between steps, so it is placed on every GPU once at load (N*d*4 bytes:
0.4 GB of the 288 GB for 800k nodes) and NO feature tensor crosses the fabric per step: every rank projects all N rows
itself ([xp | H] = one GEMM that reads X once) and forms the weight gradients from its PARTIAL [dxp | dH] against the
full X (summed by the weight all-reduce).  Per-step collectives: all-gather of the row sums (N*4 B), all-reduce of da
(N*4 B), one flat all-reduce of the replicated weight gradients (~35k floats).

HYBRID (`x_full=` with `hybrid=True`; bench.py's default for several GPUs): as REPLICATED FEATURES for the scoring side -- every
rank projects xp of all N nodes itself (the search needs it at once, and N*d*h*2 flop is cheaper than receiving 7/8 of it over
xGMI) and forms dWe from its partial dxp against the full X -- but the AGGREGATION side is sharded: every rank projects H = X Wc
for its own rows only and all-gathers it ASYNCHRONOUSLY (H is first read by the aggregation, after the k-net, the search and the
partition: ~0.3 ms of kernels at 62 500 rows per rank to hide N*F*4 bytes behind), and the partial dH [N,F] -- complete after the
first backward kernel, needed only by the last -- is reduce-scattered behind the score backward, so that dWc is a product over
the rank's own rows.  Half of the replicated full-N GEMM work (projection and weight gradient) leaves the critical path.

GATHERED PROJECTIONS (inputs that are activations, `x_grad`; or data that is not replicated): every rank projects its
own rows and all-gathers [xp | H] (xp first -- the top-k waits for it -- H asynchronously behind it: it is needed only
by the aggregation, so it crosses xGMI while the top-k, soft-k, partition and normalisation kernels run); the backward
reduce-scatters the partial [dxp | dH] (the adjoint of that all-gather) and forms weight and input gradients from the
rank's own rows.

`kern` is the kernel namespace (dgg_amd.ops on the GPU; tests substitute a CPU stand-in built on the oracle so
that the partition / collective logic is exercised with gloo, world_size 2 and 3, without a GPU).
"""
import os
import weakref

import torch
import torch.distributed as dist


def shard_bounds(N, world, rank):
    per = (N + world - 1) // world
    r0 = min(rank * per, N)
    return r0, min(r0 + per, N), per


class _Gather:
    """A (possibly still running) all-gather of row shards: .get() waits and returns the [N, ...] tensor.  On RCCL the
    wait is a stream dependency (no host block); shards are padded to `per` rows for the fixed-size collective.  `bufs`
    (a dict owned by the layer) keeps the padded source and the gathered output between steps: no allocation, no
    torch.cat per step."""

    def __init__(self, t_local, N, per, group, async_op, bufs=None, key=None):
        world = dist.get_world_size(group)
        tail = tuple(t_local.shape[1:])
        n_loc = t_local.shape[0]
        if n_loc == per:
            src = t_local.contiguous()
        else:
            src = self._buf(bufs, (key, "src"), (per,) + tail, t_local)
            src[:n_loc].copy_(t_local)
            src[n_loc:].zero_()
        self.out = self._buf(bufs, (key, "out"), (world * per,) + tail, t_local)
        self.N = N
        self.work = dist.all_gather_into_tensor(self.out, src, group=group, async_op=async_op)

    @staticmethod
    def _buf(bufs, key, shape, like):
        if bufs is None or key[0] is None:
            return like.new_empty(shape)
        b = bufs.get(key)
        if b is None or tuple(b.shape) != tuple(shape) or b.device != like.device:
            b = bufs[key] = like.new_empty(shape)
        return b

    def get(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.out[:self.N]


def _all_gather_rows(t_local, N, per, group, bufs=None, key=None):
    """[n_loc, ...] -> [N, ...]"""
    if dist.get_world_size(group) == 1 and os.environ.get("DGG_FORCE_COLLECTIVES") != "1":
        return t_local
    return _Gather(t_local, N, per, group, False, bufs, key).get()


class ShardedDGGConv:
    """One DGG (all-pairs, u-v-dist / x / k_times_edge_prob) + normalise + GCNConv layer, forward and backward,
    on a row shard.  Parameters are a dict with the reference's names (dgm.py:1097-1143, model.py:583)."""

    PARAM_KEYS = ("We", "be", "Wk", "bk", "W1", "b1", "Wmu", "bmu", "Wp", "bp", "Wc")

    def __init__(self, kern, N, group=None, K=64, t=-0.05, noise_mode=2, seed=(1234, 0), mode=0, algo=0, x_grad=False, x_full=None,
                 hybrid=False, cand=None):
        self.kern, self.N, self.group = kern, N, group
        # cand = (rowptr int64 [N+1], col int32 [E]): the candidates of row i are the stored entries of in_adj (edge-list mode, the
        # live class's semantics dgm.py:1613-1614) instead of all N columns; every candidate is scored (per-pair hash noise), the rest
        # of the step is the same.  One rank only: a citation graph's step is launch-bound, not something to shard.
        self.cand = cand
        # scorer (edge-list candidates only): None = exp(t ||xp_i - xp_j||) (u-v-dist), or a dict with the edge-MLP scorer's terms in
        # the per-node / per-edge form of dgg_edge_mlp_fwd (reference dgm.py:1628-1719): Wcat [2hw,h], wdu / wdv / wex [hw] or None,
        # b1 [hw], w2 [hw], b2 [1], erow int32 [E], ex_in [E] or None, ex_mode, t_ex, act.  backward() then also returns
        # g["scorer"] = {Wcat, wdu, wdv, wex, b1, w2, b2} (the rest of the step is the same)
        self.scorer = None
        assert cand is None or noise_mode in (0, 2, 3), "edge-list candidates: noise_mode none / hash / symmetric hash"
        assert x_full is None or not x_grad, "replicated features are data: they cannot take a gradient"
        assert not hybrid or x_full is not None, "the hybrid scheme replicates the features for the scoring side"
        self.x_full = x_full                                 # [N,d] static node features present on every rank, or None
        self.hybrid = bool(hybrid)                           # replicated xp, all-gathered H, reduce-scattered dH (module docstring)
        # Two independent kernels of the step run beside their neighbours on a SECOND stream (captured into the same hipGraph): the
        # partition's sort (read by the backward only) beside the forward aggregation, and the k-net backward (needs only dk, MFMA-
        # bound) beside the per-destination kernel of the score backward (gather-bound).  DGG_OVERLAP=0 keeps everything on one stream.
        # DGG_OVERLAP: 2 (default) = only the k-net backward beside the score backward's column kernel; 1 = also the partition's sort
        # beside the aggregation; 3 = only the sort; 0 = one stream.  Round 6, same box, ms per step: 1.233 (2), 1.234-1.237 (1),
        # 1.244-1.246 (0), 1.252-1.255 (3): the sort beside the gather-bound aggregation LOSES 8 us (it stretched from 36 to 168 us and
        # took the aggregation with it), the MFMA-bound k-net backward beside the gather-bound column kernel wins 11
        ov = os.environ.get("DGG_OVERLAP", "2")
        self.overlap = ov != "0" and hasattr(kern, "partp_sort")
        self.overlap_sort, self.overlap_knet = ov in ("1", "3"), ov in ("1", "2")
        self._side = None
        self.K, self.t, self.noise_mode, self.seed, self.mode, self.algo, self.x_grad = K, t, noise_mode, seed, mode, algo, x_grad
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # DGG_FORCE_COLLECTIVES=1 issues every collective even in a 1-rank group (exercises the RCCL calls on a 1-GPU box)
        self.coll = self.world > 1 or (dist.is_initialized() and os.environ.get("DGG_FORCE_COLLECTIVES") == "1")
        self.emulate = None
        self.r0, self.r1, self.per = shard_bounds(N, self.world, self.rank)
        assert cand is None or self.world == 1, "edge-list candidates run on one rank"
        self.bufs = {}                                       # collective staging buffers, kept between steps
        # Rows wider than the 64-rank list (all-pairs candidates, ranked noise): the learned degree is unbounded (dgm.py:1580-1584) and
        # the reference ramps over the whole dense row (dgm.py:1402-1421).  wide_rows: "off" = the [rows,64] list whatever k (exact
        # while k_i + 9.5 <= 64; callers enforce the bound), "auto" = chunked rows (ops.chunk_layout: ceil(k_i + 8.5) + 1 ranks of every
        # row in chunks of 64) from the forward in which some row needs them -- ONE readback of the chunk count per forward --, "on" =
        # chunked rows always.  wide_cap = (chunks, lists per wavefront): a FIXED capacity instead of the readback (hipGraph capture;
        # the flags of the last forward are in self.wide_meta, read by check_wide()).
        self.wide_rows = "off"
        self.wide_cap = None
        self.wide_meta = None
        self.wide_sticky = None                              # int32[1] on the device: overflow flags of every forward under wide_cap
        # symmetric noise (noise_mode 5, the ranked symmetric generator): sym_fallback = redo a forward it cannot settle under the
        # symmetric per-pair hash (noise_mode 3) and stay there (sym_hash); off by default for the bare engine (bench.py times the
        # generator it names), on for the nn.Module mirror
        self.sym_fallback = False
        self.sym_hash = False
        # force_chunked: rows that fit the 64-rank list go through the chunked rows' evaluator as well (one chunk per row).  Set by the module
        # once the ranked symmetric generator has failed on its data -- latents spread over several noise scales -- because the 64-rank
        # entry of the per-pair hash noise guesses ONE threshold for the graph from the noise law alone and loses every row to its
        # exhaustive fallback on such data (145-170 ms at N = 100 000, features x4), while the chunked rows' front end filters on
        # G + lpub_i with a threshold per row (6-9 ms on the same data; 2-3x the 64-rank entry where that one works)
        self.force_chunked = False
        self.tight_bound = "off"                             # ranked search: row-minimum distance bound in its stop tests (_row_bound)
        self._tight_on, self._tight_n, self.tight_probe = False, 0, None

    def check_generator(self):
        """raises if the ranked symmetric noise generator (noise_mode 5) could not settle every row inside its workspace in any
        forward since the last call (one synchronisation; the rows concerned came back empty)"""
        err, self.rsym_err = getattr(self, "rsym_err", None), None
        if err is not None and bool(err.any()):
            raise RuntimeError("ShardedDGGConv: the ranked symmetric noise generator ran out of workspace for its dense tier; "
                               "use noise_mode 3 (per-pair hash) for this data")

    def check_wide(self):
        """raises if a forward under a fixed chunk capacity (wide_cap) could not hold every row's ranks since the last call (one
        synchronisation).  The flags come from a device word that the layout kernel only ever ORs into (`wide_sticky`): every replay of a
        captured step is covered, not just the last one.  An overflowing forward is memory-safe -- the layout is cut to the capacity on
        the device -- but its cut rows are wrong: re-capture with a larger wide_cap."""
        st = self.wide_sticky
        meta, self.wide_meta = self.wide_meta, None
        if st is None:
            return
        flags = int(st.item())
        if flags:
            st.zero_()
            total, widest = (int(v) for v in meta[:2].cpu()) if meta is not None else (-1, -1)
            raise RuntimeError(f"ShardedDGGConv: the chunked rows outgrew their fixed capacity {self.wide_cap} (flags {flags}: 2 = more chunks than "
                               f"the capacity, 1 = a row wider than its lists, 4 = a learned degree is NaN; last forward: {total} chunks needed, "
                               f"widest row {widest}): re-capture with a larger wide_cap")

    WIDE_NOISE = {0: 0, 2: 2, 3: 3, 4: 4, 5: 3}      # noise_mode -> the generator wide rows are evaluated under (ranked symmetric: the
                                                     # symmetric per-pair hash -- same law, another realisation; no wide-row form of its own)

    def _chunk_layout(self, k):
        """-> ops.ChunkLayout for this forward's learned degrees, or None: every row fits the 64-rank list (or wide_rows is off)"""
        kern = self.kern
        if self.wide_rows == "off" or not hasattr(kern, "chunk_layout") or self.noise_mode not in self.WIDE_NOISE or self.cand is not None or self.K != 64:
            return None
        if self.wide_cap is not None:
            if self.wide_sticky is None or self.wide_sticky.device != k.device:
                # (allocated by the first forward that runs with a capacity -- normally an eager warm-up; inside a capture the fill below
                #  would be replayed, i.e. the word would only cover one replay)
                self.wide_sticky = torch.zeros((1,), device=k.device, dtype=torch.int32)
            lay = kern.chunk_layout(k, maxm=int(self.wide_cap[1]), ccap=int(self.wide_cap[0]), sticky=self.wide_sticky)
            self.wide_meta = lay.meta
            return lay
        assert not (k.is_cuda and torch.cuda.is_current_stream_capturing()), \
            "ShardedDGGConv: chunked rows inside a hipGraph capture need a fixed capacity (wide_cap = (chunks, lists))"
        lay = kern.chunk_layout(k, ncols=self.N)
        self.last_layout = (lay.chunks, lay.maxm)
        if not lay.wide and self.wide_rows == "auto" and not self.force_chunked:
            return None
        return lay

    def _row_bound(self, xp, k):
        """-> lpub [own rows] or None.  The ranked search bounds the score of the ranks it has not reached by their noise alone, i.e. as
        if they sat at distance 0; on latents whose distances spread over several noise scales it then walks ~ exp(spread / 0.3) times
        deeper than it has to (bench.py data_regimes: 2 blocks of 64 ranks per row on unit-scale features, 2 000 at x16).  With a
        lower bound of the distance to the row's nearest OTHER node (kern.rowmin_logp_bound: one fp16-MFMA sweep over all pairs, ~1.4 ms
        at N = 100 000) the same stop test uses G + log p_max(i).  tight_bound: "off" | "on" | "auto" = on when a pilot walk (~1000
        sampled rows, 64-block budget, every 16th forward and never inside a capture) puts the search's cost above 2.5 x the sweep's."""
        kern = self.kern
        if self.tight_bound == "off" or not hasattr(kern, "rowmin_logp_bound") or xp.shape[1] not in (16, 32, 64, 128) or self.t >= 0:
            return None
        if self.tight_bound == "auto":
            capturing = xp.is_cuda and torch.cuda.is_current_stream_capturing()
            if not capturing and self._tight_n % 16 == 0:
                seed = self.seed
                if torch.is_tensor(seed):
                    sd = seed.cpu()
                    seed = (int(sd[0]) & 0xFFFFFFFF, int(sd[1]) & 0xFFFFFFFF)
                pr = kern.ranked_probe(xp, k, self.t, seed, rows=(self.r0, self.r1), stride=max(1, (self.r1 - self.r0) // 1024), max_blocks=64)
                est = kern.ranked_cost_estimate(pr, self.N, rows=self.r1 - self.r0)
                sweep_us = 1400.0 * (self.N / 1e5) * ((self.r1 - self.r0) / 1e5) * (xp.shape[1] + 16) / 80.0
                self._tight_on = est > 2.5 * sweep_us and self.N >= 8192
                self.tight_probe = dict(pr, ranked_us_estimate=est, sweep_us_estimate=sweep_us)
            if not capturing:
                self._tight_n += 1
            if not self._tight_on:
                return None
        return kern.rowmin_logp_bound(xp, self.t, rows=(self.r0, self.r1))

    def emulate_rank(self, world, rank):
        """TIMING DIAGNOSTIC (bench.py --emulate-world): do the work of `rank` of `world` in a single process -- own row range
        against all N columns, replicated features -- with the collectives left out and the other ranks' row sums faked by
        tiling the own ones.  Results are not meaningful; the kernel sequence and sizes are those of the real rank."""
        assert self.x_full is not None and not dist.is_initialized()
        self.world, self.rank, self.coll, self.emulate = world, rank, False, (world, rank)
        self.r0, self.r1, self.per = shard_bounds(self.N, world, rank)

    # ------------------------------------------------------------------------------------------------------------------
    def _project(self, x_local, P):
        """-> xp (rows of `Xall`), H (rows of `Xall`; hybrid: the rank's OWN rows), xk (own rows).  One GEMM that reads X once when
        the kernel namespace offers it (ops.linear_fwd_multi), three plain calls otherwise."""
        kern = self.kern
        repl = self.x_full is not None
        Xall = self.x_full if repl else x_local
        single = not repl or (self.world == 1 and self.emulate is None)      # the own rows are all the rows that are projected
        if self.hybrid and not single:
            if hasattr(kern, "linear_fwd_multi"):            # (the multi entry reaches the persistent register kernel for large N)
                (xp,) = kern.linear_fwd_multi(Xall, [(P["We"], P["be"], 1, 0)])
                xk, H = kern.linear_fwd_multi(x_local, [(P["Wk"], P["bk"], 1, 0), (P["Wc"], None, 0, 1)])
                return xp, H, xk
            xp = kern.linear_fwd(Xall, P["We"], P["be"], 1, 0)
            return xp, kern.linear_fwd(x_local, P["Wc"], None, 0, 1), kern.linear_fwd(x_local, P["Wk"], P["bk"], 1, 0)
        if hasattr(kern, "linear_fwd_multi"):
            if single:
                xp, xk, H = kern.linear_fwd_multi(Xall, [(P["We"], P["be"], 1, 0), (P["Wk"], P["bk"], 1, 0), (P["Wc"], None, 0, 1)])
                return xp, H, xk
            xp, H = kern.linear_fwd_multi(Xall, [(P["We"], P["be"], 1, 0), (P["Wc"], None, 0, 1)])
            return xp, H, kern.linear_fwd(x_local, P["Wk"], P["bk"], 1, 0)
        xp = kern.linear_fwd(Xall, P["We"], P["be"], 1, 0)
        H = kern.linear_fwd(Xall, P["Wc"], None, 0, 1)
        return xp, H, kern.linear_fwd(x_local, P["Wk"], P["bk"], 1, 0)

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream()
        return self._side

    def _hyb(self):
        """the hybrid exchange is live: several ranks (or their single-process emulation) on replicated features"""
        return self.hybrid and self.x_full is not None and (self.coll or self.emulate is not None)

    def forward(self, x_local, deg_full, P):
        """One forward per backward: `saved` holds views of the persistent collective buffers (the gathered xp / H / row sums),
        which the next forward overwrites."""
        kern = self.kern
        s = {}
        self._fwd_gen = getattr(self, "_fwd_gen", 0) + 1
        repl = self.x_full is not None
        xp, H, xk = self._project(x_local, P)
        s["xp_loc"], s["H_loc"] = xp, H
        hyb = self._hyb()
        if self.coll and not repl:                  # xp first (the top-k waits for it), H streams in behind it
            g_xp = _Gather(xp, self.N, self.per, self.group, True, self.bufs, "xp")
            g_H = _Gather(H, self.N, self.per, self.group, True, self.bufs, "H")
        elif hyb and self.coll:                     # H of the own rows only: gathered behind the k-net, the search and the partition
            g_H = _Gather(H, self.N, self.per, self.group, True, self.bufs, "H")
        s["xk"] = xk
        # mean / std of the prior degrees (dgm.py:1569-1570): an INPUT statistic -- recomputed only when the degree tensor changes
        # (keyed on the tensor OBJECT and its version counter, not on its address: a new tensor is always re-read)
        ref = getattr(self, "_stats_ref", None)
        if ref is None or ref() is not deg_full or self._stats_ver != deg_full._version:
            self._stats_ref, self._stats_ver, self._stats = weakref.ref(deg_full), deg_full._version, kern.degree_stats(deg_full)
        s["mu_sd"] = mu_sd = self._stats
        deg_local = deg_full[self.r0:self.r1].contiguous()
        s["deg_local"] = deg_local
        if hasattr(kern, "knet_x_bwd_fused") and xk.shape[1] in getattr(kern, "KNET_MFMA_WIDTHS", ()):
            # k-net on the matrix cores: the forward saves only u; the backward re-runs layer 1 from xk and forms every gradient in one pass
            s["k"], s["u"] = kern.knet_x_fwd_slim(xk, deg_local, mu_sd, P["W1"], P["b1"], P["Wmu"], P["bmu"], P["Wp"].reshape(-1), P["bp"])
            s["z"] = s["feat"] = None
        else:
            s["k"], s["z"], s["u"], s["feat"] = kern.knet_x_fwd(xk, deg_local, mu_sd, P["W1"], P["b1"], P["Wmu"], P["bmu"],
                                                              P["Wp"].reshape(-1), P["bp"])
        s["xp"] = xp = g_xp.get() if (self.coll and not repl) else xp
        if self.cand is not None and self.scorer is not None:
            # edge-MLP scorer: first layer split into per-node products AB = xp [Wa | Wb]^T (MFMA GEMM) + per-edge terms
            rowptr, col = self.cand
            sc = self.scorer
            s["AB"] = AB = kern.linear_fwd(xp, sc["Wcat"], None, 0, 0)
            s["sdeg"] = sdeg = deg_full if sc["wdu"] is not None else None
            p_edge, s["ex"] = kern.edge_mlp_fwd(AB, xp, sc["erow"], col, sdeg, sc["ex_in"], sc["ex_mode"], sc["t_ex"], sc["wdu"], sc["wdv"],
                                                sc["wex"], sc["b1"], sc["w2"], sc["b2"], sc["act"])
            s["idx"], s["val"], s["eid"] = kern.edgelist_topk_p(p_edge, self.N, rowptr, col, self.K, self.noise_mode, None, self.seed)
            s["w"], rs_local = kern.softk_fwd(s["idx"], s["val"], s["k"], self.mode)
        elif self.cand is not None:
            rowptr, col = self.cand
            # (self.overflow: optional int32[1] device flag the caller owns -- set when the K-wide list drops a weighted rank)
            got = kern.edgelist_topk_softk(xp, rowptr, col, s["k"], self.mode, self.K, self.t, self.noise_mode, None, self.seed,
                                           overflow=getattr(self, "overflow", None)) if hasattr(kern, "edgelist_topk_softk") else None
            if got is not None:                     # search + ramp in one launch
                s["idx"], s["val"], s["w"], rs_local = got
            else:
                s["idx"], s["val"] = kern.edgelist_topk(xp, rowptr, col, self.K, self.t, self.noise_mode, None, self.seed)
                s["w"], rs_local = kern.softk_fwd(s["idx"], s["val"], s["k"], self.mode)
        elif self.noise_mode == 4 and self.K == 64 and hasattr(kern, "allpairs_topk_softk") and xp.shape[1] in (8, 16, 32, 64, 128):
            # ranked noise: the ramp is applied inside the search kernel, while the settled list is still in registers
            s["layout"] = lay = self._chunk_layout(s["k"]) if xp.shape[1] in (16, 32, 64, 128) else None
            lkw_ = {}
            lp = self._row_bound(xp, s["k"])        # upper bounds of log p over a row's other nodes, when the walk is deep enough to pay for them
            if lp is not None:
                lkw_["lpub"] = lp
            if lay is not None:                     # rows wider than 64 ranks: ceil(k_i + 8.5) + 1 ranks of every row, in chunks of 64
                s["idx"], s["val"], s["w"], rs_local = kern.allpairs_topk_wide(xp, s["k"], lay, self.mode, self.t, self.seed, rows=(self.r0, self.r1), **lkw_)
            else:
                s["idx"], s["val"], s["w"], rs_local = kern.allpairs_topk_softk(xp, s["k"], self.mode, self.t, self.seed, rows=(self.r0, self.r1), **lkw_)
        else:
            # the generators without a row-wise early-stopping search (unperturbed scores, per-pair hash noise, the ranked symmetric
            # generator).  Rows wider than 64 ranks: chunked rows through the threshold-buffer evaluator (any width; the reference's
            # defaults symmetric_noise=True / perturb_edge_prob=False train into this regime like every other configuration)
            lay = self._chunk_layout(s["k"]) if (xp.shape[1] in (16, 32, 64, 128) and self.mode in (0, 1)) else None
            s["layout"] = lay
            if lay is not None:
                s["idx"], s["val"], s["w"], rs_local = kern.allpairs_topk_wide(xp, s["k"], lay, self.mode, self.t, self.seed, rows=(self.r0, self.r1),
                                                                             noise_mode=self.WIDE_NOISE[self.noise_mode])
            else:
                nm = 3 if (self.noise_mode == 5 and self.sym_hash) else self.noise_mode
                if nm == 5 and not hasattr(kern, "rsym_status"):      # (a stand-in kernel namespace without the status plumbing)
                    st = None
                else:
                    # sym_fallback: a forward the ranked symmetric generator cannot settle (its dense tier overflows: data-dependent)
                    # is redone under the symmetric per-pair hash, and the layer stays with it (one flag read back per forward; not
                    # inside a capture, not across ranks -- every rank would have to take the same decision)
                    st = {"sym_fallback": self.sym_fallback and self.world == 1} if nm == 5 else None
                kw = {} if st is None else {"status": st}
                s["idx"], s["val"] = kern.allpairs_topk(xp, self.K, self.t, nm, None, self.seed,
                                                        rows=(self.r0, self.r1), algo=self.algo, k_limit=s["k"], **kw)
                if st and st.get("rsym_fell_back"):
                    self.sym_hash = True
                if st and st.get("rsym_err") is not None:      # ranked symmetric generator out of workspace: device flag, read by check_generator()
                    self.rsym_err = st["rsym_err"] if getattr(self, "rsym_err", None) is None else (self.rsym_err | st["rsym_err"])
                    self.rsym_last = st                 # (all three status words of this forward: the module mirror keeps its own tally)
                s["w"], rs_local = kern.softk_fwd(s["idx"], s["val"], s["k"], self.mode)
        s["rs"] = rs = _all_gather_rows(rs_local, self.N, self.per, self.group, self.bufs, "rs") if self.coll else rs_local
        if self.emulate is not None:
            s["rs"] = rs = rs_local.repeat(self.world)[:self.N].contiguous()
        # destination-bucket partition of the active entries: the backward's column-side terms run on it (no atomics)
        # (payload form -- records carry w rs_i^-1/2 and the score, no slot map, normalize_adj fused into its fill pass -- when the
        # namespace offers it and covers the shape)
        use_p = hasattr(kern, "partp_build") and xp.shape[1] in (16, 32, 64, 128) and H.shape[1] in (16, 32, 64, 128) and self.mode in (0, 1)
        ov = use_p and self.overlap and self.overlap_sort and s["idx"].is_cuda
        # want_backward = False (set by the autograd node under torch.no_grad() / frozen parameters): the per-bucket sort of the
        # partition -- read by the backward's column kernels only -- is not launched at all (an eval forward used to
        # leave it running on the side stream with nothing ever joining it)
        nobwd = use_p and not getattr(self, "want_backward", True)
        lay = s.get("layout")
        lkw = {} if lay is None else {"layout": lay}
        assert lay is None or use_p, "chunked rows run on the payload partition (latent / conv widths 16, 32, 64, 128; soft modes)"
        got = (kern.partp_build(s["idx"], s["w"], s["val"], rs_local, self.N, rs, phase=1, **lkw) if (ov or nobwd) else
               kern.partp_build(s["idx"], s["w"], s["val"], rs_local, self.N, rs, **lkw)) if use_p else None
        s["partp"], s["ahat"] = got if got is not None else (None, None)
        s["side_join"] = False
        s["partp_sorted"] = not nobwd
        if nobwd and got is not None:
            s["partp"].args = None                  # (phase 1 only: ahat is complete, the records stay in bucket order, unread)
        elif ov and got is not None:                # the sort runs beside the aggregation; the backward joins before its first column kernel
            main, side = torch.cuda.current_stream(), self._side_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                kern.partp_sort(s["partp"])
            if not torch.cuda.is_current_stream_capturing():
                s["partp"].ws.record_stream(side)
            s["side_join"] = True
        s["part"] = kern.part_build(s["idx"], s["w"], self.N) if (s["partp"] is None and hasattr(kern, "part_build")) else None
        if s["ahat"] is None:
            s["ahat"] = kern.normalize_fwd(s["idx"], s["w"], rs, self.r0)
        if self.coll and (not repl or hyb):
            H = g_H.get()
        elif hyb:                                   # emulation: the other ranks' rows of H are stale filler in a persistent buffer
            Hf = _Gather._buf(self.bufs, ("H", "emu"), (self.N, H.shape[1]), H)
            if not self.bufs.get(("H", "emu_init")):
                Hf.copy_(H.repeat((self.N + H.shape[0] - 1) // H.shape[0], 1)[:self.N])
                self.bufs[("H", "emu_init")] = True
            Hf[self.r0:self.r1].copy_(H)
            H = Hf
        s["H"] = H
        s["Z"] = kern.spmm_fwd(s["idx"], s["ahat"], H, 2, **lkw)    # relu(A (x Wc))
        s["gen"] = self._fwd_gen
        self.saved = s
        return s["Z"]

    def backward(self, dZ, x_local, P, dA_ext=None):
        """-> dict of parameter gradients (summed over ranks) and, if x_grad, 'x' = d loss / d x_local.
        dA_ext [rows,K] (optional): cotangent of the NORMALISED adjacency (saved["ahat"]) from consumers other than this layer's own
        aggregation (GCN_DGG's second layer reads the same adjacency); added to the aggregation's own before the score backward."""
        kern, s = self.kern, self.saved
        assert s.get("gen") == getattr(self, "_fwd_gen", None), "ShardedDGGConv: backward() must follow the forward() it differentiates " \
            "(a second forward has overwritten the gathered buffers)"
        assert s.get("partp_sorted", True), "ShardedDGGConv: this forward ran with want_backward = False (its partition was not sorted)"
        if hasattr(kern, "zero_pool"):                   # every zero-initialised accumulator of the backward from ONE filled buffer
            ncols, h, F = s["xp"].shape[0], s["xp"].shape[1], s["H"].shape[1]
            rows = s["idx"].shape[0]
            # (payload path: dH / da / dxp / dA are written by their owner wavefronts, only the weight gradients are accumulated into)
            need = 3 * sum(int(v.numel()) for v in P.values()) + 65536
            if s.get("partp") is None:
                need += rows * self.K + ncols * (h + F + 2)
            if self.scorer is not None:                  # row-major dA (zero outside the partition), dAB + parameter sums, dWcat, dxp
                need += rows * self.K + ncols * (2 * self.scorer["Wcat"].shape[0] + h) + 4 * int(self.scorer["Wcat"].numel()) + 4096
            with kern.zero_pool(s["xp"].device, need):
                return self._backward(dZ, x_local, P, dA_ext)
        return self._backward(dZ, x_local, P, dA_ext)

    def _reduce_scatter_rows(self, t, key, async_op=False):
        """[N, c] partial sums on every rank -> the rank's own rows [r1-r0, c], summed over ranks.  async_op: returns a callable
        that waits (a stream dependency on RCCL) and yields the rows -- the transfer runs behind the kernels issued meanwhile."""
        pad = self.world * self.per - self.N
        src = t
        if pad:
            src = _Gather._buf(self.bufs, (key, "rs_src"), (self.world * self.per, t.shape[1]), t)
            src[:self.N].copy_(t)
            src[self.N:].zero_()
        out = _Gather._buf(self.bufs, (key, "rs_out"), (self.per, t.shape[1]), t)
        src = src.contiguous()
        work = dist.reduce_scatter_tensor(out, src, group=self.group, async_op=async_op)
        if not async_op:
            return out[: self.r1 - self.r0]

        def get(_keep=src):
            work.wait()
            return out[: self.r1 - self.r0]
        return get

    def _backward(self, dZ, x_local, P, dA_ext=None):
        kern, s = self.kern, self.saved
        g = {}
        repl = self.x_full is not None
        part = s.get("part")
        G = kern.act_bwd(s["Z"], dZ, 2)                  # cotangent of A H
        # one gather of G per edge for SDDMM + transposed SpMM + neighbour side of da; its companion is the fused score backward
        # (which adds the row side of da in registers), so both must cover the shape
        partp = s.get("partp")
        if partp is not None:
            if s.get("side_join"):                  # the partition's sort ran on the side stream
                torch.cuda.current_stream().wait_stream(self._side_stream())
                s["side_join"] = False
            # (dA of the entries outside the partition is masked by the row kernel -- ahat_rows is 0 there -- so it is not zero-filled)
            # Small graphs (the record-ordered dA fits an XCD's L2: Pubmed, 5 MB): no row-major dA -- the row kernel gathers the
            # record-ordered copy through the slot -> record map the partition's sort left (Pubmed step 0.359 -> 0.349 ms).  At
            # N = 100 000 (25.6 MB) the 4.1 M four-byte gathers cost the row kernel more (+70 us) than the scattered stores cost the
            # node kernel (-47 us): 1.264 -> 1.285 ms, so large graphs keep the scattered row-major copy.  DGG_DA_MAP=0/1 forces one.
            use_map = getattr(kern, "DA_MAP", False) and kern.partp_has_map(s["idx"].shape[0]) and self.scorer is None and s.get("layout") is None
            kw = {"want_dA": False} if use_map else {}
            if dA_ext is not None:
                kw["dA_ext"] = dA_ext
            pc = kern.conv_bwd_cols_p(s["idx"], s["H"], G, partp, s["rs"], zero_dA=self.scorer is not None, **kw)
            assert pc is not None
            dA, dA_rec, dH, da = pc
            if self.scorer is not None:          # (`da` holds the neighbour-side sums; the row side: sqrt(rs_i) sum_r dA_ir ahat_ir)
                return self._scorer_backward(g, dA, dH, da, x_local, P, cols_only=True)
            # (da first: both collectives share one RCCL stream and da is an operand of the very next kernel; behind the 25-128 MB
            #  reduce-scatter it would wait for that transfer -- ADVICE round 4)
            if self.coll:
                dist.all_reduce(da, group=self.group)
            if self._hyb() and self.coll:           # dH [N,F] partial is complete here and needed only by the last kernel of the step
                dH = self._reduce_scatter_rows(dH, "dH", async_op=True)
            # the activation derivative of the two LeakyReLU projections is applied by the kernels that PRODUCE dxp / dxk (they hold
            # xp_j / xk in registers): the fused weight-gradient product then reads no forward output for the mask (51 MB less)
            pre = self._premask(x_local)
            if self.overlap and self.overlap_knet and self.mode == 0 and s["z"] is None and s["idx"].is_cuda:
                # row kernel -> dk; then the k-net backward on the side stream beside the per-destination kernel on this one
                dxp, dk, st = kern.softk_edge_bwd_p(s["xp"], s["idx"], s["val"], s["k"], dA, dA_rec, s["rs"], da, self.r0, self.t,
                                                    self.noise_mode != 0, self.mode, True, partp, ahat_rows=s["ahat"],
                                                    out_act=1 if pre else 0, phase=1)
                main, side = torch.cuda.current_stream(), self._side_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    kn = kern.knet_x_bwd_fused(s["xk"], s["deg_local"], s["mu_sd"], P["W1"], P["b1"], P["Wmu"], P["bmu"],
                                               P["Wp"].reshape(-1), s["u"], dk, out_act=1 if pre else 0)
                kern.softk_edge_bwd_p(s["xp"], s["idx"], s["val"], s["k"], dA, dA_rec, s["rs"], da, self.r0, self.t,
                                      self.noise_mode != 0, self.mode, True, partp, ahat_rows=s["ahat"], out_act=1 if pre else 0,
                                      phase=2, state=st)
                main.wait_stream(side)
                if not torch.cuda.is_current_stream_capturing():
                    dk.record_stream(side)
                    for t_ in kn:
                        t_.record_stream(main)
                return self._weight_grads(g, dxp, dH, dk, x_local, P, premasked=pre, knet=kn)
            dxp, dk = kern.softk_edge_bwd_p(s["xp"], s["idx"], s["val"], s["k"], dA, dA_rec, s["rs"], da, self.r0, self.t,
                                            self.noise_mode != 0, self.mode, True, partp, ahat_rows=s["ahat"], out_act=1 if pre else 0)
            return self._weight_grads(g, dxp, dH, dk, x_local, P, premasked=pre)
        cols = None
        if self.scorer is None and dA_ext is None and part is not None and hasattr(kern, "conv_bwd_cols") and hasattr(kern, "softk_edge_bwd") and \
                s["xp"].shape[1] in (16, 32, 64, 128) and self.mode in (0, 1):
            cols = kern.conv_bwd_cols(s["idx"], s["ahat"], s["H"], G, part, s["rs"], True)
        if cols is not None:
            dA, dH, da = cols
            ahat_rows = s["ahat"]
        else:
            dA, dH = kern.spmm_bwd(s["idx"], s["ahat"], s["H"], G, True, True)
            if dA_ext is not None:                       # (the slot-wise total; the masked entries carry no weight either way)
                dA = dA + torch.where(s["idx"] >= 0, dA_ext, torch.zeros_like(dA_ext))
            da = kern.norm_bwd_da(s["idx"], s["w"], s["rs"], dA, self.r0, part) if part is not None else \
                kern.norm_bwd_da(s["idx"], s["w"], s["rs"], dA, self.r0)
            ahat_rows = None
        if self.scorer is not None:
            return self._scorer_backward(g, dA, dH, da, x_local, P)
        if self.coll:
            dist.all_reduce(da, group=self.group)
        if self._hyb() and self.coll:
            dH = self._reduce_scatter_rows(dH, "dH", async_op=True)
        fused = None
        if part is not None and hasattr(kern, "softk_edge_bwd"):
            fused = kern.softk_edge_bwd(s["xp"], s["idx"], s["val"], s["k"], dA, s["rs"], da, self.r0, self.t, self.noise_mode != 0,
                                        self.mode, True, part, ahat_rows=ahat_rows)
        if fused is not None:                            # ramp + normalisation backward inside the row kernel of the score backward
            dxp, dk, _ = fused
        else:
            assert ahat_rows is None
            dval, dk = kern.softk_bwd(s["idx"], s["val"], s["k"], dA, s["rs"], da, self.r0, self.mode, True)
            dxp = kern.edge_bwd(s["xp"], s["idx"], s["val"], dval, self.r0, self.t, self.noise_mode != 0, part) \
                if part is not None else kern.edge_bwd(s["xp"], s["idx"], s["val"], dval, self.r0, self.t, self.noise_mode != 0)
        return self._weight_grads(g, dxp, dH, dk, x_local, P)

    def _scorer_backward(self, g, dA, dH, da, x_local, P, cols_only=False):
        """score backward of the edge-MLP scorer (one rank): ramp + normalisation backward by rows, the MLP's backward on the selected
        edges (dgg_edge_mlp_bwd: per-node dAB, parameter sums), AB's GEMM backward into dxp; the k-net and the fused weight gradients
        as for the distance scorer.  da [N]: d loss / d (rs^-1/2), both sides (dgg_norm_bwd_da), or with cols_only the neighbour-side
        sums (the row side is then formed inside the row kernel, dgg_softk_bwd_rows)."""
        kern, s, sc = self.kern, self.saved, self.scorer
        if cols_only:
            dval, dk = kern.softk_bwd(s["idx"], s["val"], s["k"], dA, s["rs"], da, self.r0, self.mode, True, ahat_rows=s["ahat"])
        else:
            dval, dk = kern.softk_bwd(s["idx"], s["val"], s["k"], dA, s["rs"], da, self.r0, self.mode, True)
        kw = {}
        if s.get("partp") is not None and self.cand is not None and getattr(kern, "EMLP_BWD_PARTP", False):
            kw = dict(partp=s["partp"], w=s["w"], nrec_max=int(self.cand[1].numel()))       # (a selected entry is a candidate edge)
        dAB, dpar, dex = kern.edge_mlp_bwd(s["AB"], s["idx"], s["eid"], s["val"], dval, s["sdeg"], s["ex"], sc["wdu"], sc["wdv"], sc["wex"],
                                           sc["b1"], sc["w2"], sc["b2"], sc["act"], self.noise_mode != 0, need_dex=sc["ex_mode"] == 2, **kw)
        dxp, dWcat, _ = kern.linear_bwd(s["xp"], sc["Wcat"], s["AB"], dAB, 0, 0, True, False)
        if sc["ex_mode"] == 2:                              # exp(t ||xp_u - xp_v||) as an edge feature also depends on the projection
            dxp = dxp + kern.edge_bwd(s["xp"], s["idx"], s["val"], dex, self.r0, sc["t_ex"], False)
        hw = sc["Wcat"].shape[0] // 2
        pick = lambda t_, a: None if t_ is None else dpar[a * hw:(a + 1) * hw]  # noqa: E731
        g["scorer"] = dict(Wcat=dWcat, wdu=pick(sc["wdu"], 0), wdv=pick(sc["wdv"], 1), wex=pick(sc["wex"], 2), b1=dpar[3 * hw:4 * hw],
                           w2=dpar[4 * hw:5 * hw], b2=dpar[5 * hw:5 * hw + 1])
        return self._weight_grads(g, dxp, dH, dk, x_local, P, premasked=False)

    def _premask(self, x_local):
        """True when the step's weight gradients go through linear_bwd_multi (no input gradient) and the producers of dxp / dxk can
        apply LeakyReLU' themselves (ranked / k_times_edge_prob path on the HIP kernels)"""
        kern = self.kern
        return bool(getattr(kern, "PREMASK", False)) and hasattr(kern, "linear_bwd_multi") and not self.x_grad and self.mode == 0 and \
            self.saved["z"] is None and not (self.coll and self.x_full is None)

    def _weight_grads(self, g, dxp, dH, dk, x_local, P, premasked=False, knet=None):
        kern, s = self.kern, self.saved
        repl = self.x_full is not None
        # weight gradients of the two projections.  Replicated features: partial [dxp | dH] of all N nodes against the full X (the
        # weight all-reduce sums the ranks); gathered projections: reduce-scatter the partials, then the rank's own rows only.
        hyb = self._hyb()
        if self.coll and not repl:
            both = self._reduce_scatter_rows(torch.cat([dxp, dH], 1), "dproj")
            h = dxp.shape[1]
            dxp_g, dH_g, Xg, xp_g = both[:, :h].contiguous(), both[:, h:].contiguous(), x_local, s["xp_loc"]
        else:
            dxp_g, dH_g, Xg, xp_g = dxp, dH, (self.x_full if repl else x_local), s["xp"]
        if knet is not None:                         # already run (beside the score backward's column kernel)
            dxk, g["W1"], g["b1"], g["Wmu"], g["bmu"], dWp, g["bp"] = knet
        elif s["z"] is None and premasked:
            dxk, g["W1"], g["b1"], g["Wmu"], g["bmu"], dWp, g["bp"] = kern.knet_x_bwd_fused(
                s["xk"], s["deg_local"], s["mu_sd"], P["W1"], P["b1"], P["Wmu"], P["bmu"], P["Wp"].reshape(-1), s["u"], dk, out_act=1)
        elif s["z"] is None:
            dxk, g["W1"], g["b1"], g["Wmu"], g["bmu"], dWp, g["bp"] = kern.knet_x_bwd_fused(
                s["xk"], s["deg_local"], s["mu_sd"], P["W1"], P["b1"], P["Wmu"], P["bmu"], P["Wp"].reshape(-1), s["u"], dk)
        else:
            dxk, g["W1"], g["b1"], g["Wmu"], g["bmu"], dWp, g["bp"] = kern.knet_x_bwd(
                s["xk"].shape[1], s["mu_sd"], P["W1"], P["Wmu"], P["bmu"], P["Wp"].reshape(-1), s["z"], s["u"], s["feat"], dk)
        g["Wp"] = dWp.reshape(P["Wp"].shape)
        same_rows = Xg.shape[0] == x_local.shape[0] and (not repl or self.world == 1) and self.emulate is None
        if hyb:
            # hybrid: dWe from the rank's PARTIAL dxp of all N nodes against the full X (summed by the weight all-reduce); dWk and dWc
            # from the rank's own rows -- dH_own = the reduce-scattered sum over ranks (emulation: the rank's own rows of its partial)
            dH_own = dH_g() if callable(dH_g) else dH_g[self.r0:self.r1]
            ym, am = (None, 0) if premasked else (True, 1)
            if hasattr(kern, "linear_bwd_multi"):
                ((g["We"], g["be"]),) = kern.linear_bwd_multi(Xg, [(P["We"], xp_g if ym else None, dxp_g, am, 0, True)])
                (g["Wk"], g["bk"]), (g["Wc"], _) = kern.linear_bwd_multi(
                    x_local, [(P["Wk"], s["xk"] if ym else None, dxk, am, 0, True), (P["Wc"], None, dH_own, 0, 1, False)])
            else:
                _, g["We"], g["be"] = kern.linear_bwd(Xg, P["We"], xp_g if ym else None, dxp_g, am, 0, False, True)
                _, g["Wk"], g["bk"] = kern.linear_bwd(x_local, P["Wk"], s["xk"] if ym else None, dxk, am, 0, False, True)
                _, g["Wc"], _ = kern.linear_bwd(x_local, P["Wc"], None, dH_own, 0, 1, False, False)
            # (dWc comes from the ALREADY SUMMED dH of the own rows: in the flat weight all-reduce below every rank's part is its share
            # of the total, so the sum over ranks is the whole gradient -- as for every other weight)
            dX1 = dX2 = dX3 = None
        elif hasattr(kern, "linear_bwd_multi") and not self.x_grad and same_rows:
            # one pass over X for the three weight gradients (leaky masks applied on the operand load)
            ym, am = (None, 0) if premasked else (True, 1)          # premasked: dxp / dxk already carry LeakyReLU'
            (g["We"], g["be"]), (g["Wk"], g["bk"]), (g["Wc"], _) = kern.linear_bwd_multi(
                Xg, [(P["We"], xp_g if ym else None, dxp_g, am, 0, True), (P["Wk"], s["xk"] if ym else None, dxk, am, 0, True),
                     (P["Wc"], None, dH_g, 0, 1, False)])
            dX1 = dX2 = dX3 = None
        elif hasattr(kern, "linear_bwd_multi") and not self.x_grad:
            ym, am = (None, 0) if premasked else (True, 1)
            (g["We"], g["be"]), (g["Wc"], _) = kern.linear_bwd_multi(
                Xg, [(P["We"], xp_g if ym else None, dxp_g, am, 0, True), (P["Wc"], None, dH_g, 0, 1, False)])
            _, g["Wk"], g["bk"] = kern.linear_bwd(x_local, P["Wk"], s["xk"] if ym else None, dxk, am, 0, False, True)
            dX1 = dX2 = dX3 = None
        else:
            dX1, g["We"], g["be"] = kern.linear_bwd(Xg, P["We"], xp_g, dxp_g, 1, 0, self.x_grad, True)
            dX3, g["Wc"], _ = kern.linear_bwd(Xg, P["Wc"], None, dH_g, 0, 1, self.x_grad, False)
            dX2, g["Wk"], g["bk"] = kern.linear_bwd(x_local, P["Wk"], s["xk"], dxk, 1, 0, self.x_grad, True)
        if self.coll:
            flat = _Gather._buf(self.bufs, ("wgrad", "flat"), (sum(int(g[k].numel()) for k in self.PARAM_KEYS),), g["We"])
            o = 0
            for k in self.PARAM_KEYS:
                n = g[k].numel()
                flat[o:o + n].copy_(g[k].reshape(-1))
                o += n
            dist.all_reduce(flat, group=self.group)
            o = 0
            for k in self.PARAM_KEYS:              # (copies: `flat` is a persistent bucket, overwritten by the next step's all-reduce)
                n = g[k].numel()
                g[k] = flat[o:o + n].view_as(g[k]).clone()
                o += n
        if self.x_grad:
            g["x"] = dX1 + dX3 + dX2                     # all three on the rank's own rows
        return g
