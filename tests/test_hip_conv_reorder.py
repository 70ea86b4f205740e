"""GPU parity of the graph-conv path that aggregates PROJECTED features: relu(A (x W)) instead of relu((A x) W)
(reference model.py:594-598; equal up to fp32 reassociation, bar 1e-5 on activations).

Kernels under test, all through the C ABI: dgg_ell_spmm_act_fwd (narrow rows, fused ReLU), dgg_act_bwd,
dgg_ell_conv_bwd_part (SDDMM + transposed SpMM + neighbour side of the normalisation backward from ONE gathered row per
entry) and the row side of the normalisation backward formed inside dgg_softk_edge_bwd_part.  Checker: the CPU oracle.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu
K = 64


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import dgg_amd  # noqa: F401
    return torch.device("cuda:0")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def Nn(t):
    return t.detach().cpu().numpy()


def _graph(rng, N, h, kmax=30):
    xp = rng.standard_normal((N, h)).astype(np.float32)
    k = (3 + kmax * rng.random(N)).astype(np.float32)
    idx, val = O.allpairs_topk(xp, K=K, noise_mode=O.NOISE_HASH, seed=(8, 8))
    w, rs = O.softk(idx, val, k)
    return xp, k, idx, val, w, rs, O.normalize(idx, w, rs)


@pytest.mark.parametrize("F", [16, 32, 64])
@pytest.mark.parametrize("act", [0, 2])
def test_spmm_narrow_rows_with_fused_relu(dev, F, act):
    from dgg_amd import ops
    rng = np.random.default_rng(100 + F)
    N = 777
    _, _, idx, _, _, _, ahat = _graph(rng, N, 32)
    H = rng.standard_normal((N, F)).astype(np.float32)
    ref = O.spmm(idx, ahat, H)
    if act == 2:
        ref = np.maximum(ref, 0)
    got = Nn(ops.spmm_fwd(T(idx, dev), T(ahat, dev), T(H, dev), act))
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    if act == 2:
        assert (got >= 0).all() and ((got == 0) == (ref == 0)).mean() > 0.999
    # a row range (sharded use: local rows, global columns) and an empty / padded adjacency
    got2 = Nn(ops.spmm_fwd(T(idx[100:300], dev), T(ahat[100:300], dev), T(H, dev), act))
    np.testing.assert_allclose(got2, ref[100:300], rtol=1e-5, atol=1e-6)
    none = Nn(ops.spmm_fwd(T(np.full((5, K), -1, np.int32), dev), T(np.zeros((5, K), np.float32), dev), T(H, dev), act))
    assert (none == 0).all()


def test_act_bwd(dev):
    from dgg_amd import ops
    rng = np.random.default_rng(3)
    y = rng.standard_normal((1000, 48)).astype(np.float32)
    dy = rng.standard_normal((1000, 48)).astype(np.float32)
    assert np.array_equal(Nn(ops.act_bwd(T(y, dev), T(dy, dev), 2)), np.where(y > 0, dy, 0).astype(np.float32))
    assert np.array_equal(Nn(ops.act_bwd(T(y, dev), T(dy, dev), 1)), np.where(y > 0, dy, np.float32(0.01) * dy).astype(np.float32))


@pytest.mark.parametrize("F", [16, 32, 64, 128])
def test_conv_backward_through_the_partition(dev, F):
    """dA, dH == the oracle's SpMM backward (dA on the active entries); da == neighbour side of the oracle's normalisation
    backward; two row shards with their own partitions add up to the whole"""
    from dgg_amd import ops
    rng = np.random.default_rng(200 + F)
    N = 900
    _, _, idx, _, w, rs, ahat = _graph(rng, N, 32)
    H = rng.standard_normal((N, F)).astype(np.float32)
    G = rng.standard_normal((N, F)).astype(np.float32)
    rdA, rdH = O.spmm_bwd(idx, ahat, H, G)
    rdA = np.where(ahat == 0, 0.0, rdA).astype(np.float32)
    part = ops.part_build(T(idx, dev), T(w, dev), N)
    got = ops.conv_bwd_cols(T(idx, dev), T(ahat, dev), T(H, dev), T(G, dev), part, T(rs, dev), want_da=True)
    assert got is not None
    dA, dH, da = got
    np.testing.assert_allclose(Nn(dA), rdA, rtol=1e-4, atol=1e-4 * np.abs(rdA).max())
    np.testing.assert_allclose(Nn(dH), rdH, rtol=1e-4, atol=1e-4 * np.abs(rdH).max())
    # neighbour side of da: da_j = sum_{(i,r)->j} dA_ir w_ir a_i  (float64 restatement)
    a = 1.0 / np.sqrt(rs.astype(np.float64))
    da_ref = np.zeros(N)
    m = (idx >= 0) & (w != 0)
    np.add.at(da_ref, idx[m], (rdA.astype(np.float64) * w * a[:, None])[m])
    np.testing.assert_allclose(Nn(da), da_ref, rtol=3e-4, atol=3e-4 * np.abs(da_ref).max())
    # without da
    dA2, dH2, none = ops.conv_bwd_cols(T(idx, dev), T(ahat, dev), T(H, dev), T(G, dev), part)
    assert none is None
    np.testing.assert_allclose(Nn(dA2), rdA, rtol=1e-4, atol=1e-4 * np.abs(rdA).max())
    # row shards
    r0 = 317
    totH, totda = 0, 0
    for lo, hi in [(0, r0), (r0, N)]:
        pt = ops.part_build(T(idx[lo:hi], dev), T(w[lo:hi], dev), N)
        dA_s, dH_s, da_s = ops.conv_bwd_cols(T(idx[lo:hi], dev), T(ahat[lo:hi], dev), T(H, dev), T(G[lo:hi], dev), pt, T(rs, dev), want_da=True)
        np.testing.assert_allclose(Nn(dA_s), rdA[lo:hi], rtol=1e-4, atol=1e-4 * np.abs(rdA).max())
        totH, totda = totH + dH_s, totda + da_s
    np.testing.assert_allclose(Nn(totH), rdH, rtol=1e-4, atol=1e-4 * np.abs(rdH).max())
    np.testing.assert_allclose(Nn(totda), da_ref, rtol=3e-4, atol=3e-4 * np.abs(da_ref).max())


@pytest.mark.parametrize("h", [16, 64, 128])
def test_row_side_of_da_inside_the_score_backward(dev, h):
    """softk_edge_bwd(ahat_rows=...) on the neighbour-side da == softk_edge_bwd on the complete da"""
    from dgg_amd import ops
    rng = np.random.default_rng(300 + h)
    N = 800
    xp, k, idx, val, w, rs, ahat = _graph(rng, N, h)
    dA = (rng.standard_normal((N, K)) * (w != 0)).astype(np.float32)
    da_full = Nn(ops.norm_bwd_da(T(idx, dev), T(w, dev), T(rs, dev), T(dA, dev)))
    a = 1.0 / np.sqrt(rs.astype(np.float64))
    j = np.maximum(idx, 0)
    row_part = (dA.astype(np.float64) * w * a[j] * (idx >= 0)).sum(1)
    da_cols = (da_full - row_part).astype(np.float32)
    for lo, hi in [(0, N), (211, 650)]:
        sl = slice(lo, hi)
        part = ops.part_build(T(idx[sl], dev), T(w[sl], dev), N)
        args = (T(xp, dev), T(idx[sl], dev), T(val[sl], dev), T(k[sl], dev), T(dA[sl], dev), T(rs, dev))
        ref = ops.softk_edge_bwd(*args, T(da_full, dev), lo, ops.T_DIST, True, 0, True, part, want_dval=True)
        got = ops.softk_edge_bwd(*args, T(da_cols, dev), lo, ops.T_DIST, True, 0, True, part, want_dval=True, ahat_rows=T(ahat[sl], dev))
        for g_, r_ in zip(got, ref):
            np.testing.assert_allclose(Nn(g_), Nn(r_), rtol=2e-4, atol=2e-4 * max(float(r_.abs().max()), 1e-9))
