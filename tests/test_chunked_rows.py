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
        ops.chunk_layout(torch.full((10,), 64.0 * ops.CHUNK_MAXM, device=dev))


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


def _wide_step(dev, N, d, h, scale=150.0, seed=(1234, 0), cap=None):
    import bench
    from dgg_amd import ops
    from dgg_amd.parallel import ShardedDGGConv
    P = bench.make_params(d, h, dev)
    P["Wp"] = (P["Wp"] * scale).contiguous()                      # a spread of learned degrees: half the rows narrow, the rest 2-4 chunks
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, d, generator=g).to(dev)
    deg = (2 + 100 * torch.rand(N, generator=g) ** 3).to(dev)
    layer = ShardedDGGConv(ops, N, K=64, noise_mode=ops.NOISE_RANKED, seed=seed)
    layer.wide_rows = "auto"
    layer.wide_cap = cap
    Z = layer.forward(x, deg, P)
    grads = layer.backward(torch.ones_like(Z), x, P)
    return layer, x, deg, P, Z, grads


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
