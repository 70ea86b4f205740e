"""GPU parity of the CHUNKED rows: all-pairs rows wider than the 64-rank list (learned degrees k_i + 9.5 > 64).

The reference ramps over its whole dense row (dgm.py:1402-1421) with an unbounded learned degree (dgm.py:1580-1584); this build
settles L_i = ceil(k_i + 8.5) + 1 ranks of row i in M_i = ceil(L_i / 64) chunks of 64 (include/dgg_hip.h, dgg_chunk_layout).  Bars as
everywhere: indices, scores, ramp weights and row sums BIT-EXACT against the oracle (which scores all N columns of a row and keeps
K = 64 * max M_i); activations 1e-5; gradients 2e-4 of the gradient's maximum.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import dgg_amd  # noqa: F401
    return torch.device("cuda:0")


def Nn(t):
    return t.detach().cpu().numpy()


def chunked_to_rows(lay, a, fill):
    """[chunks,64] array of a chunked layout -> dense-by-rank [rows, 64 * maxM] (numpy), `fill` where a row has no chunk"""
    cptr, cnode = Nn(lay.cptr).astype(np.int64), Nn(lay.cnode).astype(np.int64)
    a = Nn(a)[:len(cnode)] if torch.is_tensor(a) else a
    M = int((cptr[1:] - cptr[:-1]).max())
    out = np.full((lay.rows, 64 * M), fill, a.dtype)
    c = np.arange(len(cnode))
    m = c - cptr[cnode]
    for mm in range(M):
        sel = m == mm
        out[cnode[sel], 64 * mm:64 * mm + 64] = a[sel]
    return out


def rank_limit(k, cap):
    L = np.ceil(k.astype(np.float32) + np.float32(8.5)) + 1
    return np.minimum(L, cap).astype(np.int64)


@pytest.mark.parametrize("N,h,kmax,mode", [(1500, 32, 300.0, 0), (2600, 64, 150.0, 0), (700, 16, 600.0, 1), (1100, 128, 200.0, 0)])
def test_chunked_ranked_search_bit_exact(dev, N, h, kmax, mode):
    """layout + search + ramp on rows of 1 .. ~10 chunks against the oracle: every settled rank, its score, its weight and the row
    sums, bit for bit; ranks beyond L_i come back empty; chunks are laid out in node order"""
    from dgg_amd import ops
    g = torch.Generator().manual_seed(N + h)
    xp = (torch.randn(N, h, generator=g) * 0.7).to(dev)
    k = (1.0 + (kmax - 1.0) * torch.rand(N, generator=g) ** 2).to(dev)          # many narrow rows, a tail of wide ones
    k[:5] = torch.tensor([1.0, 54.5, 54.6, 118.49, 118.51], device=dev)        # chunk boundaries of L = ceil(k + 8.5) + 1
    lay = ops.chunk_layout(k)
    kc = Nn(k)
    L = rank_limit(kc, 64 * ops.CHUNK_MAXM)
    Mi = (L + 63) // 64
    cptr = Nn(lay.cptr).astype(np.int64)
    assert np.array_equal(cptr[1:] - cptr[:-1], Mi) and lay.chunks == int(Mi.sum()) and lay.maxm == int(Mi.max())
    assert np.array_equal(Nn(lay.cnode), np.repeat(np.arange(N), Mi))
    assert lay.wide
    idx, val, w, rs = ops.allpairs_topk_wide(xp, k, lay, mode=mode, seed=(77, 3))
    K = 64 * int(Mi.max())
    ri, rv = O.allpairs_topk(Nn(xp), K=K, noise_mode=O.NOISE_RANKED, seed=(77, 3))
    r = np.arange(K)[None, :]
    keep = r < L[:, None]
    gi, gv, gw = chunked_to_rows(lay, idx, -1), chunked_to_rows(lay, val, 0.0), chunked_to_rows(lay, w, 0.0)
    assert np.array_equal(gi, np.where(keep, ri, -1)), "settled ranks differ from the oracle"
    assert np.array_equal(gv, np.where(keep, rv, np.float32(0))), "scores differ from the oracle"
    wo, rso = O.softk(np.where(keep, ri, -1).astype(np.int32), rv, kc, mode=mode)
    assert np.array_equal(gw, wo), "ramp weights differ from the oracle"
    assert np.array_equal(Nn(rs), rso), "row sums differ from the oracle's butterfly over the per-lane chunk sums"
    # a fixed capacity (what a captured hipGraph uses): same result, the spare chunks empty, no overflow flag
    cap = lay.chunks + 37
    lay2 = ops.chunk_layout(k, maxm=lay.maxm, ccap=cap)
    idx2, val2, w2, rs2 = ops.allpairs_topk_wide(xp, k, lay2, mode=mode, seed=(77, 3))
    assert int(lay2.meta[2]) == 0 and int(lay2.meta[0]) == lay.chunks
    assert torch.equal(idx2[:lay.chunks], idx) and torch.equal(w2[:lay.chunks], w) and torch.equal(rs2, rs)
    assert bool((idx2[lay.chunks:] == -1).all()) and bool((w2[lay.chunks:] == 0).all()) and bool((lay2.cnode[lay.chunks:] == 0).all())
    lay3 = ops.chunk_layout(k, maxm=lay.maxm, ccap=lay.chunks - 1)
    assert int(lay3.meta[2]) & 2, "a capacity below the chunk count raises the flag"
    with pytest.raises(RuntimeError, match="ranks"):
        ops.chunk_layout(torch.full((10,), 64.0 * ops.CHUNK_MAXM, device=dev), maxm=ops.CHUNK_MAXM)      # (a caller-imposed limit)
    with pytest.raises(RuntimeError, match="NaN"):
        ops.chunk_layout(torch.tensor([3.0, float("nan")], device=dev))


def test_chunked_search_with_single_chunk_rows_equals_the_list(dev):
    """every k_i + 9.5 <= 64: the chunked layout IS the [N,64] list and the wide search returns what the 64-rank kernel returns"""
    from dgg_amd import ops
    N, h = 3000, 64
    g = torch.Generator().manual_seed(3)
    xp = torch.randn(N, h, generator=g).to(dev)
    k = (20 + 30 * torch.rand(N, generator=g)).to(dev)
    lay = ops.chunk_layout(k)
    assert not lay.wide and lay.maxm == 1
    a = ops.allpairs_topk_wide(xp, k, lay, seed=(5, 6))
    b = ops.allpairs_topk_softk(xp, k, seed=(5, 6))
    for x_, y_ in zip(a, b):
        assert torch.equal(x_, y_)


def _check_rows_against_oracle(lay, xp, k, idx, val, w, rs, noise, seed, mode, rows=None):
    """rows of a chunked result (all of them, or the given ones) against the oracle's top-(64 M_i) of the row + ramp + row sum: bit for
    bit; ranks beyond L_i (and beyond the number of columns) empty"""
    xp_c, kc = Nn(xp), Nn(k)
    N = xp_c.shape[0]
    cptr = Nn(lay.cptr).astype(np.int64)
    idx_c, val_c, w_c, rs_c = Nn(idx), Nn(val), Nn(w), Nn(rs)
    todo = range(lay.rows) if rows is None else rows
    # rows of up to 4 chunks in one oracle call, wider ones row by row (the oracle's insertion list is O(K) per entering column)
    small = [r for r in todo if cptr[r + 1] - cptr[r] <= 4]
    big = [r for r in todo if cptr[r + 1] - cptr[r] > 4]
    ref = {}
    if small and len(small) == lay.rows - len(big) and rows is None:
        ri, rv = O.allpairs_topk(xp_c, K=256, noise_mode=noise, seed=seed)
        for r in small:
            ref[r] = (ri[r], rv[r])
    else:
        for r in small:
            ri, rv = O.allpairs_topk(xp_c, K=256, noise_mode=noise, seed=seed, rows=(int(r), int(r) + 1))
            ref[r] = (ri[0], rv[0])
    for r in big:
        ri, rv = O.allpairs_topk(xp_c, K=int(64 * (cptr[r + 1] - cptr[r])), noise_mode=noise, seed=seed, rows=(int(r), int(r) + 1))
        ref[r] = (ri[0], rv[0])
    for r in todo:
        M = int(cptr[r + 1] - cptr[r])
        K = 64 * M
        L = int(rank_limit(kc[r:r + 1], K)[0])
        ri, rv = ref[r][0][:K], ref[r][1][:K]
        keep = (np.arange(K) < L) & (ri >= 0)
        gi, gv, gw = idx_c[cptr[r]:cptr[r + 1]].reshape(-1), val_c[cptr[r]:cptr[r + 1]].reshape(-1), w_c[cptr[r]:cptr[r + 1]].reshape(-1)
        assert np.array_equal(gi, np.where(keep, ri, -1)), f"row {r} (k = {kc[r]:.2f}, {M} chunks): settled ranks differ from the oracle"
        assert np.array_equal(gv, np.where(keep, rv, np.float32(0))), f"row {r}: scores differ from the oracle"
        wo, rso = O.softk(np.where(keep, ri, -1).astype(np.int32)[None, :], rv[None, :], kc[r:r + 1], mode=mode)
        assert np.array_equal(gw, wo[0]), f"row {r}: ramp weights differ from the oracle"
        assert rs_c[r] == rso[0], f"row {r}: row sum differs from the oracle's butterfly over the per-lane chunk sums"


@pytest.mark.parametrize("noise", ["none", "hash", "hash_sym"])
@pytest.mark.parametrize("N,h,mode", [(1500, 32, 0), (900, 64, 1), (1300, 16, 0), (700, 128, 0)])
def test_anywidth_rows_every_generator_bit_exact(dev, noise, N, h, mode):
    """rows wider than 64 ranks under the generators WITHOUT an early-stopping search -- unperturbed scores
    (the reference script's default perturb_edge_prob=False), per-pair hash noise, symmetric per-pair hash noise (symmetric_noise=True)
    -- through the threshold-buffer evaluator (dgg_allpairs_topk_anywide): narrow rows, rows of 2-10 chunks, rows beyond 32 chunks (more
    than 2048 ranks) and a row whose learned degree exceeds the number of columns (it keeps them all, as the reference's dense row
    does), every rank / score / weight / row sum against the oracle bit for bit; a fixed capacity gives the same arrays"""
    from dgg_amd import ops
    nm = {"none": ops.NOISE_NONE, "hash": ops.NOISE_HASH, "hash_sym": ops.NOISE_HASH_SYM}[noise]
    om = {"none": O.NOISE_NONE, "hash": O.NOISE_HASH, "hash_sym": O.NOISE_HASH_SYM}[noise]
    g = torch.Generator().manual_seed(N + h)
    xp = (torch.randn(N, h, generator=g) * 0.7).to(dev)
    k = (1.0 + 200.0 * torch.rand(N, generator=g) ** 2).to(dev)
    k[:5] = torch.tensor([1.0, 54.5, 54.6, 118.49, 118.51], device=dev)
    k[5:9] = torch.tensor([500.0, 0.45 * N, 0.8 * N, 3.0 * N], device=dev)     # 8+ chunks ... more ranks than columns exist
    lay = ops.chunk_layout(k)
    cptr = Nn(lay.cptr).astype(np.int64)
    assert lay.wide and lay.maxm == ops.chunk_maxm_for(N) and int((cptr[1:] - cptr[:-1])[8]) == lay.maxm
    idx, val, w, rs = ops.allpairs_topk_wide(xp, k, lay, mode=mode, seed=(77, 3), noise_mode=nm)
    _check_rows_against_oracle(lay, xp, k, idx, val, w, rs, om, (77, 3), mode)
    lay2 = ops.chunk_layout(k, maxm=lay.maxm, ccap=lay.chunks + 11)
    idx2, val2, w2, rs2 = ops.allpairs_topk_wide(xp, k, lay2, mode=mode, seed=(77, 3), noise_mode=nm)
    assert int(lay2.meta[2]) == 1, "only the 'more ranks than 64 maxm' flag of the row beyond every column"
    assert torch.equal(idx2[:lay.chunks], idx) and torch.equal(val2[:lay.chunks], val) and torch.equal(w2[:lay.chunks], w) and torch.equal(rs2, rs)
    assert bool((idx2[lay.chunks:] == -1).all()) and bool((w2[lay.chunks:] == 0).all())


def test_ranked_rows_beyond_the_register_lists(dev):
    """ranked generator, rows of MORE than 32 chunks (2048 ranks; VERDICT round 5: 'and stop at 2038 ranks'): the register-list search
    settles the rows of up to 32 chunks and leaves the wider ones to the threshold-buffer walk -- all of them bit-exact"""
    from dgg_amd import ops
    N, h = 6000, 32
    g = torch.Generator().manual_seed(9)
    xp = (torch.randn(N, h, generator=g) * 0.7).to(dev)
    k = (1.0 + 120.0 * torch.rand(N, generator=g) ** 2).to(dev)
    heavy = [3, 1000, 2500, 5999]
    k[heavy] = torch.tensor([2040.0, 2600.0, 4000.0, 2.0 * N], device=dev)
    k[7] = 2030.0                                                 # 32 chunks: the widest row of the register lists
    lay = ops.chunk_layout(k)
    cptr = Nn(lay.cptr).astype(np.int64)
    M = cptr[1:] - cptr[:-1]
    assert lay.maxm > ops.CHUNK_MAXM and M[7] == 32 and all(M[r] > 32 for r in heavy)
    idx, val, w, rs = ops.allpairs_topk_wide(xp, k, lay, seed=(5, 6))
    rng = np.random.default_rng(1)
    rows = sorted(set(heavy + [7, 0, N - 1] + [int(v) for v in rng.integers(0, N, 40)]))
    _check_rows_against_oracle(lay, xp, k, idx, val, w, rs, O.NOISE_RANKED, (5, 6), 0, rows=rows)


def test_fixed_capacity_overflow_is_memory_safe_and_sticky(dev):
    """a captured step's chunk capacity that the learned degrees outgrow must not write outside the arrays, and the
    report must survive later forwards that fit.  The layout is CUT to the capacity on the device (cptr clamped); the arrays' guard
    chunks stay untouched; the flags are ORed into a word the host clears"""
    from dgg_amd import ops, _lib
    N, h = 1200, 32
    g = torch.Generator().manual_seed(4)
    xp = torch.randn(N, h, generator=g).to(dev)
    k = (20.0 + 150.0 * torch.rand(N, generator=g)).to(dev)
    need = ops.chunk_layout(k).chunks
    cap, guard = need - 40, 16
    sticky = torch.zeros(1, dtype=torch.int32, device=dev)
    for nm in (ops.NOISE_RANKED, ops.NOISE_HASH_SYM, ops.NOISE_NONE):
        lay = ops.chunk_layout(k, maxm=4, ccap=cap, sticky=sticky)
        cptr = Nn(lay.cptr).astype(np.int64)
        assert int(lay.meta[2]) & 2 and int(lay.meta[0]) == need and cptr.max() == cap and (np.diff(cptr) >= 0).all()
        # the arrays as a caller with a fixed capacity holds them, followed by guard chunks that nothing may touch
        idx = torch.full((cap + guard, 64), 12345, dtype=torch.int32, device=dev)
        val = torch.full((cap + guard, 64), 7.0, device=dev)
        w = torch.full((cap + guard, 64), 7.0, device=dev)
        rs = torch.empty(N, device=dev)
        L = _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        import ctypes as C
        P_ = lambda t_: C.c_void_p(t_.data_ptr())  # noqa: E731
        if nm == ops.NOISE_RANKED:
            _lib.check(L.dgg_allpairs_topk_ranked_wide(P_(xp), N, h, 0, N, ops.T_DIST, 1, 2, None, P_(k), 0, 4, P_(lay.cptr), cap, P_(idx), P_(val), P_(w),
                                                       P_(rs), None, C.c_void_p(st)), "ranked_wide")
        else:
            nb = int(L.dgg_allpairs_anywide_ws_bytes(cap, N, N, h))
            ws = torch.empty(nb + 4096, dtype=torch.uint8, device=dev)
            ws[nb:] = 0x5A
            _lib.check(L.dgg_allpairs_topk_anywide(P_(xp), N, h, 0, N, ops.T_DIST, nm, 1, 2, None, P_(k), 0, 4, 0, P_(lay.cptr), cap, P_(idx), P_(val),
                                                   P_(w), P_(rs), None, P_(ws), nb, C.c_void_p(st)), "anywide")
            assert bool((ws[nb:] == 0x5A).all()), "the evaluator wrote beyond its workspace"
        torch.cuda.synchronize()
        assert bool((idx[cap:] == 12345).all()) and bool((val[cap:] == 7.0).all()) and bool((w[cap:] == 7.0).all()), "guard chunks overwritten"
        assert bool((idx[:cap] != 12345).all()), "every chunk inside the capacity is written"
        # rows in front of the cut are complete and exact
        full = np.nonzero(np.diff(cptr) == np.diff(Nn(ops.chunk_layout(k).cptr).astype(np.int64)))[0]
        lay_v = ops.ChunkLayout(lay.cptr, lay.cnode, lay.meta, cap, 4, N)
        _check_rows_against_oracle(lay_v, xp, k, idx[:cap], val[:cap], w[:cap], rs, {ops.NOISE_RANKED: O.NOISE_RANKED, ops.NOISE_HASH_SYM: O.NOISE_HASH_SYM,
                                   ops.NOISE_NONE: O.NOISE_NONE}[nm], (1, 2), 0, rows=[int(v) for v in full[:: max(1, len(full) // 25)]])
    assert int(sticky.item()) & 2
    lay_ok = ops.chunk_layout(k * 0.2, maxm=4, ccap=cap, sticky=sticky)     # a later forward that fits does not clear the report
    assert int(lay_ok.meta[2]) == 0 and int(sticky.item()) & 2


def _wide_step(dev, N, d, h, scale=150.0, seed=(1234, 0), cap=None, noise_mode=None):
    import bench
    from dgg_amd import ops
    from dgg_amd.parallel import ShardedDGGConv
    P = bench.make_params(d, h, dev)
    P["Wp"] = (P["Wp"] * scale).contiguous()                      # a spread of learned degrees: half the rows narrow, the rest 2-4 chunks
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, d, generator=g).to(dev)
    deg = (2 + 100 * torch.rand(N, generator=g) ** 3).to(dev)
    layer = ShardedDGGConv(ops, N, K=64, noise_mode=ops.NOISE_RANKED if noise_mode is None else noise_mode, seed=seed)
    layer.wide_rows = "auto"
    layer.wide_cap = cap
    Z = layer.forward(x, deg, P)
    grads = layer.backward(torch.ones_like(Z), x, P)
    return layer, x, deg, P, Z, grads


@pytest.mark.parametrize("noise", ["none", "hash", "hash_sym", "ranked_sym"])
def test_chunked_step_other_generators_match_the_oracle(dev, noise):
    """the whole step (generator -> normalize_adj -> relu(A (x Wc)), forward and backward) on chunked rows under the reference's DEFAULT
    noise settings (perturb_edge_prob=False; symmetric_noise=True -- the ranked symmetric generator evaluates wide rows under the
    symmetric per-pair hash) and the per-pair hash: lists / weights / row sums / normalised weights bit-exact, Z 1e-5, every
    parameter gradient 2e-4 of max against the oracle's backward"""
    from dgg_amd import ops
    from test_hip_parity import _full_size_gradient_parity
    nm = {"none": ops.NOISE_NONE, "hash": ops.NOISE_HASH, "hash_sym": ops.NOISE_HASH_SYM, "ranked_sym": ops.NOISE_RANKED_SYM}[noise]
    om = {"none": O.NOISE_NONE, "hash": O.NOISE_HASH, "hash_sym": O.NOISE_HASH_SYM, "ranked_sym": O.NOISE_HASH_SYM}[noise]
    N, d, h = 2500, 48, 32
    layer, x, deg, P, Z, grads = _wide_step(dev, N, d, h, noise_mode=nm)
    s = layer.saved
    lay = s["layout"]
    assert lay is not None and lay.wide and lay.maxm >= 3
    kc = Nn(s["k"])
    K = 64 * lay.maxm
    L = rank_limit(kc, K)
    ri, rv = O.allpairs_topk(Nn(s["xp"]), K=K, noise_mode=om, seed=(1234, 0))
    keep = np.arange(K)[None, :] < L[:, None]
    ri = np.where(keep, ri, -1).astype(np.int32)
    rv = np.where(keep, rv, np.float32(0)).astype(np.float32)
    gi, gv = chunked_to_rows(lay, s["idx"], -1), chunked_to_rows(lay, s["val"], 0.0)
    gw, ga = chunked_to_rows(lay, s["w"], 0.0), chunked_to_rows(lay, s["ahat"], 0.0)
    assert np.array_equal(gi, ri) and np.array_equal(gv, rv)
    wo, rso = O.softk(ri, rv, kc)
    assert np.array_equal(gw, wo) and np.array_equal(Nn(s["rs"]), rso)
    assert np.array_equal(ga, O.normalize(ri, wo, rso))
    Zo = np.maximum(O.spmm(ri, ga, Nn(s["H"])), 0)
    np.testing.assert_allclose(Nn(Z), Zo, rtol=1e-5, atol=1e-5)
    dense = dict(s, idx=torch.from_numpy(gi), val=torch.from_numpy(gv), w=torch.from_numpy(gw), ahat=torch.from_numpy(ga))
    _full_size_gradient_parity(dense, grads, x, deg, P, perturb=noise != "none")


def test_module_sends_spread_latents_under_a_chosen_hash_generator_through_the_chunked_rows(dev):
    """args.dgg_sym_generator = "hash" (the caller's choice of the per-pair hash generator for symmetric noise): on unit-scale features the
    64-rank entry evaluates the forward; on features x8 -- latents spread over many noise scales, where that entry's one guessed threshold
    loses every row to the exhaustive fallback -- the module's pilot (_hash_spread_now) routes the forward through the chunked rows'
    per-row front end although every row fits the list; the model trains on either (finite output, finite gradients on the generator's and the layers' parameters)."""
    from argparse import Namespace
    import dgg_amd
    from dgg_amd.adjacency import AllPairs
    N, d = 9000, 48
    args = Namespace(extra_edge_dim=0, extra_k_dim=1, dgg_hard=False, deg_mean=20.0, deg_std=4.0, dgg_mode_edge_net="u-v-dist",
                     dgg_mode_k_net="x", dgg_mode_k_select="k_times_edge_prob", debug_step=3, perturb_edge_prob=True,
                     symmetric_noise=True, stochastic_k=False, dgg_adj_input="input_adj", n_dgg_layers=1, dgg_sym_generator="hash")
    g = torch.Generator().manual_seed(8)
    prior = (10 + 20 * torch.rand(N, generator=g)).to(dev)
    for scale, want in ((1.0, False), (8.0, True)):
        torch.manual_seed(0)
        m = dgg_amd.GCN_DGG(nfeat=d, nhid=32, nclass=7, dropout=0.0, args=args).to(dev)
        m.train()
        x = (torch.randn(N, d, generator=g) * scale).to(dev)
        out = m(x, AllPairs(prior))
        out = out[0] if isinstance(out, tuple) else out
        out.square().mean().backward()
        dg = [mod for mod in m.modules() if isinstance(mod, dgg_amd.DGG_LearnableK_debug)][0]
        fl = dg.__dict__["_fused_layer"]
        print(f"scale {scale}: pilot {dg.__dict__['_hash_state']}, force_chunked {fl.force_chunked}")
        assert fl.force_chunked == want and (fl.saved["layout"] is not None) == want
        grads = {n_: p.grad for n_, p in m.named_parameters() if p.grad is not None}       # (parameters of unused scorer modes have none)
        assert bool(torch.isfinite(out).all()) and all(bool(torch.isfinite(v).all()) for v in grads.values())
        assert any("node_encode_for_edges" in n_ for n_ in grads) and any("k_net" in n_ for n_ in grads) and len(grads) >= 8, sorted(grads)


def test_forced_chunked_evaluation_of_narrow_rows_equals_the_list(dev):
    """ShardedDGGConv.force_chunked (set by the module once the ranked symmetric generator has failed on its data: spread latents, where the
    64-rank entry of the per-pair hash noise loses every row to its exhaustive fallback): rows that all fit the 64-rank list evaluated by
    the chunked rows' per-row front end (one chunk per row) -- the same lists, scores, weights, row sums, output and gradients as the
    64-rank entry, under symmetric hash noise, on benchmark-like and on spread (x4) features"""
    import bench
    from dgg_amd import ops
    from dgg_amd.parallel import ShardedDGGConv
    N, d, h = 3000, 48, 32
    P = bench.make_params(d, h, dev)
    g = torch.Generator().manual_seed(2)
    deg = (2 + 30 * torch.rand(N, generator=g)).to(dev)
    for scale in (1.0, 4.0):
        x = (torch.randn(N, d, generator=g) * scale).to(dev)
        res = []
        for force in (False, True):
            layer = ShardedDGGConv(ops, N, K=64, noise_mode=ops.NOISE_HASH_SYM, seed=(99, 7))
            layer.wide_rows = "auto"
            layer.force_chunked = force
            Z = layer.forward(x, deg, P)
            grads = layer.backward(torch.ones_like(Z), x, P)
            s = layer.saved
            assert (s["layout"] is not None) == force and float(s["k"].max()) + 9.5 <= 64
            if force:
                assert s["layout"].maxm == 1 and s["layout"].chunks == N
            res.append((Z, grads, s["idx"][:N], s["val"][:N], s["w"][:N], s["rs"]))
        (Za, ga, *la), (Zb, gb, *lb) = res
        for u, v in zip(la, lb):
            assert torch.equal(u, v)
        assert torch.equal(Za, Zb)
        assert set(ga) == set(gb)
        for kk in ga:
            if ga[kk] is not None:
                assert float((ga[kk] - gb[kk]).abs().max()) <= 2e-6 * float(ga[kk].abs().max()) + 1e-30, kk


def test_chunked_step_matches_the_oracle(dev):
    """generator -> normalize_adj -> relu(A (x Wc)), forward and backward, with rows of 1-4 chunks through ShardedDGGConv (the engine of
    the fused layer and of bench.py): lists / scores / weights / row sums / normalised weights bit-exact against the oracle on the
    dense-by-rank arrays, Z to 1e-5, EVERY parameter gradient against the oracle's backward (2e-4 of max)"""
    from test_hip_parity import _full_size_gradient_parity
    N, d, h = 2500, 48, 32
    layer, x, deg, P, Z, grads = _wide_step(dev, N, d, h)
    s = layer.saved
    lay = s["layout"]
    assert lay is not None and lay.wide and lay.maxm >= 3
    kc = Nn(s["k"])
    assert (kc < 54.5).mean() > 0.2 and kc.max() > 150, (kc.min(), kc.max())
    L = rank_limit(kc, 64 * lay.maxm)
    K = 64 * lay.maxm
    xp_c = Nn(s["xp"])
    ri, rv = O.allpairs_topk(xp_c, K=K, noise_mode=O.NOISE_RANKED, seed=(1234, 0))
    keep = np.arange(K)[None, :] < L[:, None]
    ri = np.where(keep, ri, -1).astype(np.int32)
    rv = np.where(keep, rv, np.float32(0)).astype(np.float32)
    gi, gv = chunked_to_rows(lay, s["idx"], -1), chunked_to_rows(lay, s["val"], 0.0)
    gw, ga = chunked_to_rows(lay, s["w"], 0.0), chunked_to_rows(lay, s["ahat"], 0.0)
    assert np.array_equal(gi, ri) and np.array_equal(gv, rv)
    wo, rso = O.softk(ri, rv, kc)
    assert np.array_equal(gw, wo) and np.array_equal(Nn(s["rs"]), rso)
    # (entries whose ramp is exactly 0 are outside the partition: their normalised weight is written 0, as the oracle's product is)
    assert np.array_equal(ga, O.normalize(ri, wo, rso))
    Zo = np.maximum(O.spmm(ri, ga, Nn(s["H"])), 0)
    np.testing.assert_allclose(Nn(Z), Zo, rtol=1e-5, atol=1e-5)
    dense = dict(s, idx=torch.from_numpy(gi), val=torch.from_numpy(gv), w=torch.from_numpy(gw), ahat=torch.from_numpy(ga))
    _full_size_gradient_parity(dense, grads, x, deg, P)
    # a fixed capacity (hipGraph capture): the same step, the spare chunks empty; too small a capacity is reported, not truncated
    layer2, _, _, _, Z2, grads2 = _wide_step(dev, N, d, h, cap=(lay.chunks + 100, 4))
    layer2.check_wide()
    assert torch.equal(Z2, Z)
    for k_ in grads:
        np.testing.assert_allclose(Nn(grads2[k_]), Nn(grads[k_]), rtol=0, atol=2e-4 * float(grads[k_].abs().max()))
    layer3, *_ = _wide_step(dev, N, d, h, cap=(lay.chunks - 5, 4))
    with pytest.raises(RuntimeError, match="capacity"):
        layer3.check_wide()


def _args(**kw):
    from argparse import Namespace
    base = dict(extra_edge_dim=0, extra_k_dim=1, dgg_hard=False, deg_mean=3.899, deg_std=5.288, dgg_mode_edge_net="u-v-dist",
                dgg_mode_k_net="x", dgg_mode_k_select="k_times_edge_prob", debug_step=3, perturb_edge_prob=True, symmetric_noise=False,
                stochastic_k=False, dgg_adj_input="input_adj", n_dgg_layers=1)
    base.update(kw)
    return Namespace(**base)


def _ranked_noise(N, seed):
    import ctypes as C
    G = np.empty((N, N), np.float32)
    L = O.lib()
    for i in range(N):
        L.ora_ranked_row(C.c_uint32(seed[0]), C.c_uint32(seed[1]), C.c_uint32(i), C.c_int64(N), O._p(G[i]))
    return G


@pytest.mark.parametrize("mode", ["k_times_edge_prob", "k_only"])
def test_module_chunked_rows_match_the_dense_formulation(dev, mode):
    """DGG_LearnableK_debug on all-pairs candidates with a degree prior of ~90 (ceil(k + 8.5) ~ 100 ranks per row: two chunks) as a
    SEPARATE module: the returned adjacency (EllAdjacency with a chunk layout) densified, and its autograd, against the reference-
    shaped dense formulation in float64 (oracle/dense_ref.py: sort of the whole row + ramp, dgm.py:1402-1435) fed with the ranked
    generator's own noise matrix: every entry 1e-5 (near-tie rank swaps excepted, as for the CSR form), every gradient 2e-4.  A prior
    of ~19 on the same nodes stays on the 64-rank list."""
    import dgg_amd
    from oracle import dense_ref as D
    N, d, h = 400, 24, 16
    torch.manual_seed(1)
    m = dgg_amd.DGG_LearnableK_debug(in_dim=d, latent_dim=h, args=_args(dgg_mode_k_select=mode)).to(dev)
    with torch.no_grad():
        m.k_net.k_project.weight.mul_(0.3)
    m.set_seed(4321, 17)
    x = torch.randn(N, d, generator=torch.Generator().manual_seed(2)).to(dev).requires_grad_(True)
    deg = 90.0 * (1 + 0.1 * torch.randn(N, generator=torch.Generator().manual_seed(3)))
    small = m(x.detach(), dgg_amd.AllPairs((deg * 0.2).to(dev)))
    assert isinstance(small, dgg_amd.EllAdjacency) and small.layout is None
    m.check_ell_bound()
    adj = m(x, dgg_amd.AllPairs(deg.to(dev)))
    assert isinstance(adj, dgg_amd.EllAdjacency) and adj.layout is not None and adj.layout.maxm == 2
    kk = Nn(adj.k)
    assert kk.max() + 8.5 > 64
    dense = adj.to_dense()
    assert tuple(dense.shape) == (N, N)
    cot = torch.randn(N, N, generator=torch.Generator().manual_seed(4)).to(dev)
    (dense * cot).sum().backward()
    P = {"We": m.node_encode_for_edges[0].weight, "be": m.node_encode_for_edges[0].bias, "Wk": m.node_encode_for_k[0].weight,
         "bk": m.node_encode_for_k[0].bias, "W1": m.k_embed[0].weight, "b1": m.k_embed[0].bias, "Wmu": m.k_net.k_mu.weight,
         "bmu": m.k_net.k_mu.bias, "Wp": m.k_net.k_project.weight, "bp": m.k_net.k_project.bias}
    Pd = {k_: v.detach().cpu().double().requires_grad_(True) for k_, v in P.items()}
    xd = x.detach().cpu().double().requires_grad_(True)
    G = torch.from_numpy(_ranked_noise(N, (4321, 17))).double()
    rows, cols = torch.arange(N).repeat_interleave(N), torch.arange(N).repeat(N)
    Ad, kd = D.dgg_dense(xd, rows, cols, deg.double(), Pd, G)
    if mode == "k_only":                                     # (dgm.py:1423-1435: the ramp alone at the sorted positions)
        p = torch.exp(torch.log(torch.exp(-0.05 * torch.cdist(torch.nn.functional.leaky_relu(torch.nn.functional.linear(xd, Pd["We"], Pd["be"])),
                                                             torch.nn.functional.leaky_relu(torch.nn.functional.linear(xd, Pd["We"], Pd["be"])),
                                                             compute_mode="donot_use_mm_for_euclid_dist")) + 1e-8) + G)
        order = torch.sort(p, dim=-1, descending=True).indices
        ramp = 1 - 0.5 * (1 + torch.tanh(torch.arange(N, dtype=torch.float64)[None, :] - kd[:, None]))
        Ad = torch.zeros_like(p).scatter(1, order, ramp)
    np.testing.assert_allclose(kk, kd.detach().numpy(), rtol=1e-5, atol=1e-4)
    diff = np.abs(Nn(dense) - Ad.detach().numpy())
    nswap = int((diff > 1e-5).sum())
    assert nswap <= 4 and diff.max() < 2e-2, (nswap, diff.max())
    gtol = 2e-4 if nswap == 0 else 2e-3
    (Ad * cot.cpu().double()).sum().backward()
    for k_, v in P.items():
        if Pd[k_].grad is None:
            continue
        ref = Pd[k_].grad.numpy()
        err = np.abs(Nn(v.grad).reshape(ref.shape) - ref).max() / max(np.abs(ref).max(), 1e-12)
        assert err <= gtol, f"grad {k_}: {err:.3e}"
    err = np.abs(Nn(x.grad) - xd.grad.numpy()).max() / np.abs(xd.grad.numpy()).max()
    assert err <= gtol, f"grad x: {err:.3e}"
    # the same adjacency through the layers that take it as a separate module: normalize_adj + aggregation on the CSR kernels
    na = adj.normalize()
    y = na.matmul(x.detach())
    Ahat = D.normalize_dense(Ad.detach())
    np.testing.assert_allclose(Nn(y), (Ahat @ xd.detach()).numpy(), rtol=1e-4, atol=2e-5 + 1e-3 * (nswap > 0))


@pytest.mark.parametrize("hard", [False, True])
def test_gcn_dgg_chunked_fused_layer_matches_the_separate_modules(dev, hard):
    """GCN_DGG on all-pairs candidates whose learned degrees need two to three chunks: the fused first layer (generator + normalize_adj
    + GCNConv as one node on the step engine, the second layer reading the same chunked adjacency) against the separate modules
    (chunked generator -> CSR normalise / aggregate): same lists, log-probabilities 1e-5, every parameter gradient 3e-4"""
    import dgg_amd
    N, d, h, C = 1800, 48, 32, 7
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, d, generator=g).to(dev)
    deg = (60 + 90 * torch.rand(N, generator=g)).to(dev)
    y = torch.randint(0, C, (N,), generator=g).to(dev)
    outs = []
    for fused in ([True, False] if not hard else [False]):
        torch.manual_seed(11)
        m = dgg_amd.GCN_DGG(nfeat=d, nhidden=h, nclass=C, args=_args(dgg_fused_layer=fused, dgg_hard=hard)).to(dev).eval()
        m.dggs[0].set_seed(99, 1)
        logp, adj, _ = m(x, dgg_amd.AllPairs(deg))
        assert isinstance(adj, dgg_amd.EllAdjacency) and adj.layout is not None and adj.layout.maxm >= 2
        loss = torch.nn.functional.nll_loss(logp, y)
        loss.backward()
        m.dggs[0].check_ell_bound()
        assert torch.isfinite(logp).all()
        outs.append((logp.detach(), adj, {n_: p_.grad.detach().clone() for n_, p_ in m.named_parameters() if p_.grad is not None}))
    if hard:
        return                                               # (straight-through values on chunked rows: runs, finite)
    (lf, af, gf), (ls, as_, gs) = outs
    assert torch.equal(af.idx, as_.idx), "fused layer and separate modules disagree on the neighbour lists"
    np.testing.assert_allclose(Nn(lf), Nn(ls), rtol=1e-5, atol=1e-5)
    assert set(gf) == set(gs)
    for n_ in gf:
        err = float((gf[n_] - gs[n_]).abs().max() / gs[n_].abs().max().clamp(min=1e-30))
        assert err <= 3e-4, f"grad {n_}: {err:.3e}"


def test_learned_degrees_keep_training_on_chunked_rows(dev):
    """GCN_DGG on all-pairs candidates, the script's optimiser groups (train_small_graphs.py:399-418): Adam moves the learned degrees
    past the 64-rank list within a few steps (k = relu(kp sd + mu) + 1 is unbounded, dgm.py:1580-1584).  From that forward on the
    rows are chunked -- same generator, same search -- and training goes on: no bound, nothing raised, the loss falls."""
    import dgg_amd
    N, d, h, C = 3000, 64, 64, 7
    g = torch.Generator().manual_seed(5)
    deg = 24 + 16 * torch.rand(N, generator=g)
    x = torch.randn(N, d, generator=g)
    y = (x[:, :C] + 0.3 * torch.randn(N, C, generator=g)).argmax(1).to(dev)
    x = x.to(dev)
    A = dgg_amd.AllPairs(deg.to(dev))
    torch.manual_seed(11)
    m = dgg_amd.GCN_DGG(nfeat=d, nhidden=h, nclass=C, args=_args()).to(dev).train()
    opt = torch.optim.Adam([{"params": m.params1, "weight_decay": 0.01}, {"params": m.params2, "weight_decay": 5e-4}], lr=0.01)
    hist = []
    for step in range(40):
        opt.zero_grad()
        logp, adj, _ = m(x, A)
        loss = torch.nn.functional.nll_loss(logp, y)
        loss.backward()
        hist.append((float(adj.k.max()), adj.layout is not None, float(loss.detach())))
        m.dggs[0].check_ell_bound()
        opt.step()
    first = next((s_ for s_, (km, _, _) in enumerate(hist) if km + 9.5 > 64), None)
    print("learned degrees exceed the list from step", first, "; k_max after 40 steps", hist[-1][0], "; loss", hist[0][2], "->", hist[-1][2])
    assert first is not None and 1 <= first < 20
    assert all(not w_ for _, w_, _ in hist[:first]) and all(w_ for _, w_, _ in hist[first:])
    assert hist[-1][0] > 100 and np.isfinite(hist[-1][2]) and hist[-1][2] < hist[0][2]


def test_full_size_chunked_step(dev):
    """BASELINE size (N = 100 000, d = 128, h = 64) with learned degrees far beyond the list (k ~ 134 on average, k_max > 256: rows of
    1-5 chunks): 256 sampled rows of the chunked search against the oracle (which scores all N columns of a row) bit for bit --
    indices, scores, weights, row sums --, list invariants over the whole graph, Z on the sampled rows, and the gradient of EVERY
    parameter against the oracle's backward over the WHOLE graph (2e-4 of max)."""
    import bench
    from dgg_amd import ops
    from dgg_amd.parallel import ShardedDGGConv
    from test_hip_parity import _full_size_gradient_parity
    N, d, h = 100_000, 128, 64
    P = bench.make_params(d, h, dev)
    P["Wp"] = (P["Wp"] * 40.0).contiguous()
    g = torch.Generator(device="cpu").manual_seed(1000)
    x = torch.randn(N, d, generator=g).to(dev)
    deg = (20 + 800 * torch.rand(N, generator=torch.Generator().manual_seed(7)) ** 4).to(dev)
    layer = ShardedDGGConv(ops, N, K=64, noise_mode=ops.NOISE_RANKED, seed=(1234, 0))
    layer.wide_rows = "auto"
    Z = layer.forward(x, deg, P)
    grads = layer.backward(torch.ones_like(Z), x, P)
    s = layer.saved
    lay = s["layout"]
    kc = Nn(s["k"])
    print(f"k mean {kc.mean():.1f} max {kc.max():.1f}; {lay.chunks} chunks, widest row {lay.maxm}")
    assert lay is not None and kc.max() >= 256 and (kc < 54.5).any()
    K = 64 * lay.maxm
    L = rank_limit(kc, K)
    gi, gv = chunked_to_rows(lay, s["idx"], -1), chunked_to_rows(lay, s["val"], 0.0)
    gw, ga = chunked_to_rows(lay, s["w"], 0.0), chunked_to_rows(lay, s["ahat"], 0.0)
    keep = np.arange(K)[None, :] < L[:, None]
    assert np.array_equal(gi >= 0, keep), "every row keeps exactly the ranks that can carry weight"
    vv = np.where(keep, gv, -1.0)
    assert (vv[:, :-1] >= vv[:, 1:]).all(), "scores are sorted"
    srt = np.sort(np.where(keep, gi, -np.arange(1, K + 1)[None, :]), axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all(), "duplicate column in a row"
    xp_c, H_c, rs_c = Nn(s["xp"]), Nn(s["H"]), Nn(s["rs"])
    rng = np.random.default_rng(0)
    rows = np.unique(np.concatenate([[0, N - 1, int(kc.argmax()), int(kc.argmin())], rng.integers(0, N, 256)]))
    assert len(rows) >= 200
    for r in rows:
        ri, rv = O.allpairs_topk(xp_c, K=K, noise_mode=O.NOISE_RANKED, seed=(1234, 0), rows=(int(r), int(r) + 1))
        m_ = keep[r]
        assert np.array_equal(gi[r][m_], ri[0][m_]) and np.array_equal(gv[r][m_], rv[0][m_]), f"row {r} (k = {kc[r]:.2f})"
        wo, rso = O.softk(np.where(m_, ri[0], -1)[None, :].astype(np.int32), rv, kc[r:r + 1])
        assert np.array_equal(gw[r], wo[0]) and rs_c[r] == rso[0], f"row {r}: weights / row sum"
        ai = np.float32(1.0) / np.sqrt(rs_c[r])
        aj = (np.float32(1.0) / np.sqrt(rs_c[np.where(m_, ri[0], 0)])).astype(np.float32)
        assert np.array_equal(ga[r], np.where(m_ & (wo[0] != 0), ((ai * wo[0]).astype(np.float32) * aj).astype(np.float32), np.float32(0)))
        zr = np.zeros(H_c.shape[1], np.float64)
        for q in np.nonzero(m_)[0]:
            zr += np.float64(ga[r, q]) * H_c[gi[r, q]]
        np.testing.assert_allclose(Nn(Z[r]), np.maximum(zr, 0), rtol=1e-5, atol=1e-5)
    dense = dict(s, idx=torch.from_numpy(gi), val=torch.from_numpy(gv), w=torch.from_numpy(gw), ahat=torch.from_numpy(ga))
    _full_size_gradient_parity(dense, grads, x, deg, P)


@pytest.mark.parametrize("noise", ["ranked", "none", "hash", "hash_sym"])
def test_full_size_rows_of_any_width_bit_exact(dev, noise):
    """N = 100 000 with learned degrees up to beyond 2 500 (rows of 1 .. 40+ chunks; under the ranked generator
    the rows of more than 32 chunks take the threshold-buffer walk, the others the register lists) under every generator that has a
    wide-row form: 200+ sampled rows -- the widest ones included -- against the oracle (all N columns of the row scored, its top 64 M_i
    kept) bit for bit: indices, scores, ramp weights, row sums; list invariants over the whole graph"""
    from dgg_amd import ops
    nm = {"ranked": ops.NOISE_RANKED, "none": ops.NOISE_NONE, "hash": ops.NOISE_HASH, "hash_sym": ops.NOISE_HASH_SYM}[noise]
    om = {"ranked": O.NOISE_RANKED, "none": O.NOISE_NONE, "hash": O.NOISE_HASH, "hash_sym": O.NOISE_HASH_SYM}[noise]
    N, h = 100_000, 64
    g = torch.Generator().manual_seed(21)
    xp = (torch.randn(N, h, generator=g) * 0.8).to(dev)
    k = (20.0 + 600.0 * torch.rand(N, generator=g) ** 6).contiguous()           # mean ~ 105, a thin tail of wide rows
    wide = torch.randperm(N, generator=g)[:24]
    k[wide] = torch.linspace(1900.0, 3400.0, 24)                               # around and beyond the 32-chunk register lists
    k = k.to(dev)
    lay = ops.chunk_layout(k, ncols=N)
    kc = Nn(k)
    cptr = Nn(lay.cptr).astype(np.int64)
    M = cptr[1:] - cptr[:-1]
    assert kc.max() >= 2500 and lay.maxm > ops.CHUNK_MAXM and (M == 1).any()
    idx, val, w, rs = ops.allpairs_topk_wide(xp, k, lay, seed=(1234, 5), noise_mode=nm)
    torch.cuda.synchronize()
    # invariants over the whole graph: exactly L_i ranks kept per row, scores sorted inside a row, no duplicate column
    L = rank_limit(kc, 64 * lay.maxm)
    kept = (idx >= 0).sum(1)
    per_row = torch.zeros(N, dtype=torch.int64, device=dev).index_add_(0, lay.cnode.long(), kept)
    assert np.array_equal(Nn(per_row), L), "every row keeps exactly the ranks that can carry weight"
    rng = np.random.default_rng(3)
    rows = sorted(set([int(v) for v in wide] + [0, N - 1, int(kc.argmin())] + [int(v) for v in rng.integers(0, N, 200)]))
    assert len(rows) >= 200
    _check_rows_against_oracle(lay, xp, k, idx, val, w, rs, om, (1234, 5), 0, rows=rows)


@pytest.mark.parametrize("symmetric_noise,perturb", [(False, True), (True, True), (False, False), (True, False)])
def test_full_size_gcn_dgg_trains_past_the_list(dev, symmetric_noise, perturb):
    """GCN_DGG on all-pairs candidates at N = 100 000 with the script's optimiser groups
    (train_small_graphs.py:399-418) takes 200 Adam steps without raising under EVERY noise setting of the reference -- its defaults
    symmetric_noise=True / perturb_edge_prob=False included (train_small_graphs.py:153-162): the learned degrees leave the 64-rank list
    within a few steps and the rows become chunked (asymmetric noise: k_max beyond 10 000 = rows of 150+ chunks by step 100); the loss
    falls, every weighted rank is kept (check_ell_bound every step), memory stays bounded (no step's state outlives it), and a hipGraph
    capture of the trained model's forward replays the chunked layout"""
    import warnings
    import dgg_amd
    N, d, h, C = 100_000, 128, 64, 7
    g = torch.Generator().manual_seed(5)
    deg = 24 + 16 * torch.rand(N, generator=g)
    x = torch.randn(N, d, generator=g)
    y = (x[:, :C] + 0.3 * torch.randn(N, C, generator=g)).argmax(1).to(dev)
    x = x.to(dev)
    A = dgg_amd.AllPairs(deg.to(dev))
    torch.manual_seed(11)
    m = dgg_amd.GCN_DGG(nfeat=d, nhidden=h, nclass=C, args=_args(symmetric_noise=symmetric_noise, perturb_edge_prob=perturb)).to(dev).train()
    with torch.no_grad():
        m.dggs[0].k_net.k_project.weight.mul_(0.1)               # the benchmark's initialisation: k in ~[24, 41] at step 0
    opt = torch.optim.Adam([{"params": m.params1, "weight_decay": 0.01}, {"params": m.params2, "weight_decay": 5e-4}], lr=0.01)
    hist = []
    import gc
    gc.collect()                                                 # (what earlier tests of the session left behind is not this test's peak)
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated() / 2 ** 30
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                          # (symmetric noise: the one-time notice of the generator switch)
        for step in range(200):
            opt.zero_grad()
            logp, adj, _ = m(x, A)
            loss = torch.nn.functional.nll_loss(logp, y)
            loss.backward()
            hist.append((float(adj.k.max()), None if adj.layout is None else adj.layout.maxm, float(loss.detach())))
            m.dggs[0].check_ell_bound()
            opt.step()
    first = next((s_ for s_, (km, _, _) in enumerate(hist) if km + 9.5 > 64), None)
    kmax = max(km for km, _, _ in hist)
    peak = torch.cuda.max_memory_allocated() / 2 ** 30 - base
    print(f"N = 100 000, symmetric_noise={symmetric_noise}, perturb_edge_prob={perturb}: learned degrees exceed the list from step {first}; "
          f"largest k {kmax:.0f} (widest row {max(w_ or 1 for _, w_, _ in hist)} chunks); loss {hist[0][2]:.3f} -> {hist[-1][2]:.3f}; "
          f"peak memory {peak:.1f} GiB")
    assert hist[0][1] is None, "inside the list at initialisation"
    # (the largest learned degree may dip below the list's 54.5 again for a step or two right after it first crossed it)
    assert first is not None and first < 30 and all(w_ is not None for _, w_, _ in hist[first + 10:]), f"rows left the chunked form again: {hist[first:first + 12]}"
    assert all((w_ is not None) == (km + 9.5 > 64) for km, w_, _ in hist), "a row wider than the list without the chunked layout (or the reverse)"
    assert kmax > 700, f"largest learned degree {kmax}"
    assert np.isfinite(hist[-1][2]) and hist[-1][2] < hist[0][2], f"loss {hist[0][2]} -> {hist[-1][2]}"
    assert peak < 60, f"peak memory {peak:.1f} GiB above the start: a step's state must not outlive the step (reference cycles through the autograd node)"
    # inference under a hipGraph: the capture replays the last eager layout as a fixed capacity
    m.eval()
    m.dggs[0].set_seed(7, 7)
    with torch.no_grad():
        ref = m(x, A)[0]
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            out = m(x, A)[0]
        gph.replay()
        torch.cuda.synchronize()
    m.dggs[0].check_ell_bound()
    assert torch.equal(out, ref)


def test_no_grad_forward_skips_the_partition_sort(dev):
    """ctx.needs_input_grad is (True, ...) under torch.no_grad() too, so 'a backward will follow' has to be decided by
    the caller of the autograd node (ops.backward_will_follow): an eval forward under no_grad leaves the partition unsorted (nothing on
    the side stream), frozen parameters likewise, and the training step that follows matches one that never saw the eval forward"""
    import copy
    import dgg_amd
    N, d, h = 3000, 32, 32
    torch.manual_seed(2)
    m = dgg_amd.DGG_LearnableK_debug(in_dim=d, latent_dim=h, args=_args()).to(dev)
    conv = dgg_amd.GCNConv(d, 16).to(dev)
    m2, conv2 = copy.deepcopy(m), copy.deepcopy(conv)
    x = torch.randn(N, d, generator=torch.Generator().manual_seed(3)).to(dev)
    A = dgg_amd.AllPairs((10 + 10 * torch.rand(N, generator=torch.Generator().manual_seed(4))).to(dev))
    m.set_seed(3, 4)
    m2.set_seed(3, 4)
    with torch.no_grad():
        Ze, _ = m.forward_conv(x, A, conv.W)
    layer = m._fused_layer
    assert layer.want_backward is False and layer.saved["partp_sorted"] is False and layer.saved["side_join"] is False
    for p_ in list(m.parameters()) + list(conv.parameters()):
        p_.requires_grad_(False)
    m.forward_conv(x, A, conv.W)
    assert layer.want_backward is False and layer.saved["partp_sorted"] is False, "frozen parameters, data input: no backward can follow"
    for p_ in list(m.parameters()) + list(conv.parameters()):
        p_.requires_grad_(True)
    Z, _ = m.forward_conv(x, A, conv.W)
    assert layer.want_backward is True and layer.saved["partp_sorted"] is True and torch.equal(Z, Ze)
    Z.sum().backward()
    Z2, _ = m2.forward_conv(x, A, conv2.W)
    Z2.sum().backward()
    assert torch.equal(Z, Z2)
    for (n1, p1), (_, p2) in zip(m.named_parameters(), m2.named_parameters()):
        if p2.grad is not None:
            np.testing.assert_allclose(Nn(p1.grad), Nn(p2.grad), rtol=0, atol=2e-4 * float(p2.grad.abs().max()) + 1e-12, err_msg=n1)
    np.testing.assert_allclose(Nn(conv.W.grad), Nn(conv2.W.grad), rtol=0, atol=2e-4 * float(conv2.W.grad.abs().max()))


@pytest.mark.parametrize("noise", ["ranked", "none", "hash_sym"])
def test_anywidth_rows_on_a_row_shard_and_with_a_device_seed(dev, noise):
    """the any-width evaluators on a ROW SHARD (rows [r0, r1) of the graph against all N columns: what a rank of the sharded layer
    calls) and with the noise seed read from DEVICE memory (what a captured step uses) return the rows of the whole-graph call bit for
    bit; so does the ranked search with the rows' nearest-neighbour bound"""
    from dgg_amd import ops
    nm = {"ranked": ops.NOISE_RANKED, "none": ops.NOISE_NONE, "hash_sym": ops.NOISE_HASH_SYM}[noise]
    N, h = 5000, 32
    g = torch.Generator().manual_seed(12)
    xp = (torch.randn(N, h, generator=g) * 0.7).to(dev)
    k = (5.0 + 300.0 * torch.rand(N, generator=g) ** 3).to(dev)
    k[[10, 2600, 4999]] = torch.tensor([2300.0, 2900.0, 2100.0], device=dev)          # rows beyond 32 chunks on both sides of the cut
    lay = ops.chunk_layout(k, ncols=N)
    full = ops.allpairs_topk_wide(xp, k, lay, seed=(21, 4), noise_mode=nm)
    cptr = lay.cptr.long()
    r0, r1 = 2500, 4100
    ks = k[r0:r1].contiguous()
    lay_s = ops.chunk_layout(ks, ncols=N)
    dseed = torch.tensor([21, 4], dtype=torch.int32, device=dev)
    lp = ops.rowmin_logp_bound(xp, rows=(r0, r1)) if noise == "ranked" else None
    part = ops.allpairs_topk_wide(xp, ks, lay_s, seed=dseed, noise_mode=nm, rows=(r0, r1), lpub=lp)
    c0, c1 = int(cptr[r0]), int(cptr[r1])
    assert lay_s.chunks == c1 - c0
    for a_, b_ in zip(full[:3], part[:3]):
        assert torch.equal(a_[c0:c1], b_), "shard differs from the rows of the whole-graph call"
    assert torch.equal(full[3][r0:r1], part[3])


@pytest.mark.parametrize("noise", ["hash", "hash_sym"])
@pytest.mark.parametrize("knob", ["DGG_ANYWIDE_HASH_TARGET=0.5", "DGG_ANYWIDE_HASH_TARGET=0.95", "DGG_ANYWIDE_HASH_FRONT=0", "DGG_ANYWIDE_HASH_BOUND=0"])
def test_hash_wide_rows_guess_and_verify_falls_back_exactly(dev, noise, knob, monkeypatch):
    """The hash generators' wide rows: a per-row threshold guessed from a pilot, one integer sweep, verification -- and the moving-threshold
    scan for the rows whose guess fails.  Forced here: a target of 0.5 L + 32 candidates (every wide row fails: all of them fall back), of
    0.95 L + 32 (about half fail), the front end off, the distance bound off (the front end needs it): the same bits every time"""
    from dgg_amd import ops
    nm = {"hash": ops.NOISE_HASH, "hash_sym": ops.NOISE_HASH_SYM}[noise]
    N, h = 6000, 64
    g = torch.Generator().manual_seed(31)
    xp = (torch.randn(N, h, generator=g) * 0.8).to(dev)
    k = (5.0 + 400.0 * torch.rand(N, generator=g) ** 2).to(dev)
    lay = ops.chunk_layout(k, ncols=N)
    ref = ops.allpairs_topk_wide(xp, k, lay, seed=(8, 9), noise_mode=nm)
    name, value = knob.split("=")
    if name == "DGG_ANYWIDE_HASH_BOUND":
        monkeypatch.setattr(ops, "ANYWIDE_HASH_BOUND", False)
    else:
        monkeypatch.setenv(name, value)
    got = ops.allpairs_topk_wide(xp, k, lay, seed=(8, 9), noise_mode=nm)
    for a_, b_ in zip(ref, got):
        assert torch.equal(a_, b_)
    rows = [0, 17, N - 1] + [int(v) for v in np.random.default_rng(2).integers(0, N, 12)]
    _check_rows_against_oracle(lay, xp, k, *got, {"hash": O.NOISE_HASH, "hash_sym": O.NOISE_HASH_SYM}[noise], (8, 9), 0, rows=rows)


@pytest.mark.parametrize("knob", ["", "DGG_ANYWIDE_PLAIN_FEW=0"])
def test_unperturbed_wide_rows_front_end_and_fallback_agree(dev, knob, monkeypatch):
    """unperturbed scores on chunked rows: the matrix-core front end (radius per row from a sampled sweep, one full fp16-MFMA sweep,
    candidates scored and verified, failing rows redone -- by one wavefront per row when they are few, by the tiled exhaustive scan
    otherwise: DGG_ANYWIDE_PLAIN_FEW=0 forces the latter) against the exhaustive scan alone -- same bits -- on data with a tight cluster
    and far outliers (rows whose guessed radius fails), and against the oracle on sampled rows"""
    from dgg_amd import ops
    N, h = 9000, 64
    g = torch.Generator().manual_seed(41)
    xp = torch.randn(N, h, generator=g) * 0.8
    xp[2000:2600] *= 0.03                                        # a tight blob
    xp[2600:2620] = xp[2600:2620] * 0.01 + 4.0                   # outliers far from everything
    xp[100] = xp[7]                                              # duplicate rows
    xp = xp.to(dev)
    k = (5.0 + 500.0 * torch.rand(N, generator=g) ** 2).to(dev)
    k[[3, 2005, 2610]] = torch.tensor([3000.0, 2500.0, 40.0], device=dev)
    lay = ops.chunk_layout(k, ncols=N)
    if knob:
        monkeypatch.setenv(*knob.split("="))
    got = ops.allpairs_topk_wide(xp, k, lay, seed=(0, 0), noise_mode=ops.NOISE_NONE)
    monkeypatch.setenv("DGG_ANYWIDE_PLAIN_FRONT", "0")
    ref = ops.allpairs_topk_wide(xp, k, lay, seed=(0, 0), noise_mode=ops.NOISE_NONE)
    for a_, b_ in zip(ref, got):
        assert torch.equal(a_, b_)
    rows = [0, 3, 7, 100, 2005, 2300, 2610, N - 1] + [int(v) for v in np.random.default_rng(4).integers(0, N, 10)]
    _check_rows_against_oracle(lay, xp, k, *got, O.NOISE_NONE, (0, 0), 0, rows=rows)
