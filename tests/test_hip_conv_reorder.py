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


@pytest.mark.parametrize("N,d,outs", [(1000, 128, (64, 64, 64)), (333, 70, (32, 64)), (257, 128, (64, 64)), (130, 24, (32,)),
                                      (500, 128, (128, 128)), (77, 40, (64, 96, 32)),
                                      # N >= 32768, d = 128: the persistent kernel (weights resident in LDS, rows through v_permlane32_swap)
                                      (40_000, 128, (64, 64, 64)), (33_001, 128, (64, 64)), (32_800, 128, (32,)), (50_003, 128, (64, 32, 32, 32))])
def test_fused_projections_are_bit_identical_to_separate_calls(dev, N, d, outs):
    """dgg_linear_fwd_multi (X read once) == one dgg_linear_fwd per layer, bit for bit; mixed layouts / activations / biases"""
    from dgg_amd import ops
    rng = np.random.default_rng(N + d)
    x = T(rng.standard_normal((N, d)).astype(np.float32), dev)
    layers = []
    for s_, o in enumerate(outs):
        lay = s_ % 2 if len(outs) > 1 else 0
        W = T((rng.standard_normal((o, d) if lay == 0 else (d, o)) * 0.2).astype(np.float32), dev)
        b = T((rng.standard_normal(o) * 0.1).astype(np.float32), dev) if s_ != 1 else None
        layers.append((W, b, [ops.ACT_LEAKY, ops.ACT_NONE, ops.ACT_RELU][s_ % 3], lay))
    got = ops.linear_fwd_multi(x, layers)
    for y, (W, b, act, lay) in zip(got, layers):
        ref = ops.linear_fwd(x, W, b, act, lay)
        assert torch.equal(y, ref)
        sel = np.r_[0:min(N, 1500), max(N - 100, 0):N]           # the oracle on the first and last rows (the separate call covers all)
        xo = O.linear(Nn(x)[sel], Nn(W), None if b is None else Nn(b), act, w_layout=lay)
        assert np.array_equal(Nn(y)[sel], xo)


@pytest.mark.parametrize("N,d,outs", [(2100, 128, (64, 64, 64)), (700, 128, (64, 64)), (300, 40, (32, 64, 32)), (65, 128, (128, 128)),
                                      # d = 128, widths in multiples of 64, N >= 4096: the wide-tile kernel (permuted operands, 16-byte loads)
                                      (40_000, 128, (64, 64, 64)), (5_001, 128, (64, 128)), (100_003, 128, (64, 64)), (4_100, 128, (64,))])
def test_fused_weight_gradients_match_separate_calls(dev, N, d, outs):
    """dgg_gemm_tn_multi == one dgg_linear_bwd (weight / bias gradient) per layer, up to summation order"""
    from dgg_amd import ops
    rng = np.random.default_rng(N + d + len(outs))
    x = T(rng.standard_normal((N, d)).astype(np.float32), dev)
    layers = []
    for s_, o in enumerate(outs):
        lay = s_ % 2
        act = [ops.ACT_LEAKY, ops.ACT_NONE, ops.ACT_RELU][s_ % 3]
        W = T((rng.standard_normal((o, d) if lay == 0 else (d, o)) * 0.2).astype(np.float32), dev)
        y = ops.linear_fwd(x, W, None, act, lay)
        dy = T(rng.standard_normal((N, o)).astype(np.float32), dev)
        layers.append((W, y if act != ops.ACT_NONE else None, dy, act, lay, s_ != 1))
    got = ops.linear_bwd_multi(x, layers)
    for (dW, db), (W, y, dy, act, lay, need_db) in zip(got, layers):
        _, rdW, rdb = ops.linear_bwd(x, W, y, dy, act, lay, need_dx=False, need_db=need_db)
        np.testing.assert_allclose(Nn(dW), Nn(rdW), rtol=2e-4, atol=2e-4 * float(rdW.abs().max()))
        assert (db is None) == (not need_db)
        if need_db:
            np.testing.assert_allclose(Nn(db), Nn(rdb), rtol=2e-4, atol=2e-4 * float(rdb.abs().max()))


@pytest.mark.parametrize("h,F", [(64, 64), (16, 32), (128, 128), (32, 16)])
def test_payload_partition_matches_the_slot_map_path(dev, h, F):
    """16-byte payload records (no slot map): conv backward and fused ramp / normalisation / score backward == the slot-map
    kernels (themselves oracle-checked above), whole block and row shards (local rows, global columns)"""
    from dgg_amd import ops
    rng = np.random.default_rng(400 + h + F)
    N = 1100
    xp, k, idx, val, w, rs, ahat = _graph(rng, N, h)
    H = rng.standard_normal((N, F)).astype(np.float32)
    G = rng.standard_normal((N, F)).astype(np.float32)
    # (500, 700): fewer than 16 records per destination node -> the lane-group node kernels (a rank of 8's regime)
    for lo, hi in [(0, N), (300, 811), (500, 700)]:
        sl = slice(lo, hi)
        a = dict(idx=T(idx[sl], dev), w=T(w[sl], dev), val=T(val[sl], dev), ahat=T(ahat[sl], dev), k=T(k[sl], dev), G=T(G[sl], dev))
        part = ops.part_build(a["idx"], a["w"], N)
        partp = ops.partp_build(a["idx"], a["w"], a["val"], T(rs[sl], dev), N)
        assert partp is not None
        dA0, dH0, da0 = ops.conv_bwd_cols(a["idx"], a["ahat"], T(H, dev), a["G"], part, T(rs, dev), want_da=True)
        dA1, dArec, dH1, da1 = ops.conv_bwd_cols_p(a["idx"], T(H, dev), a["G"], partp, T(rs, dev))
        assert torch.equal(dA0, dA1)                                 # same dot products, same order
        np.testing.assert_allclose(Nn(dH1), Nn(dH0), rtol=1e-5, atol=1e-5 * float(dH0.abs().max()))
        np.testing.assert_allclose(Nn(da1), Nn(da0), rtol=1e-4, atol=1e-4 * float(da0.abs().max()))
        # dA_rec is a permutation of the active entries of dA
        act = Nn(a["w"]) != 0
        assert np.allclose(np.sort(Nn(dArec)[: act.sum()]), np.sort(Nn(dA1)[act]))
        for mode in (0, 1):
            ref = ops.softk_edge_bwd(T(xp, dev), a["idx"], a["val"], a["k"], dA0, T(rs, dev), da0, lo, ops.T_DIST, True, mode, True, part,
                                     ahat_rows=a["ahat"])
            got = ops.softk_edge_bwd_p(T(xp, dev), a["idx"], a["val"], a["k"], dA1, dArec, T(rs, dev), da1, lo, ops.T_DIST, True, mode, True,
                                       partp, ahat_rows=a["ahat"])
            assert got is not None
            np.testing.assert_allclose(Nn(got[1]), Nn(ref[1]), rtol=2e-4, atol=2e-4 * max(float(ref[1].abs().max()), 1e-9))
            np.testing.assert_allclose(Nn(got[0]), Nn(ref[0]), rtol=2e-4, atol=2e-4 * max(float(ref[0].abs().max()), 1e-9))


@pytest.mark.parametrize("N,h,clustered", [(12_000, 64, False), (9_000, 16, True), (10_000, 32, "wild"), (8_200, 128, False)])
def test_unperturbed_allpairs_with_pilot_guess_is_bit_exact(dev, N, h, clustered):
    """dgg_allpairs_topk, no perturbation, MFMA-bounded kernel (algo 2) at sizes where the pilot guesses each row's 64-NN
    radius (N >= 8192): neighbour lists and scores equal the oracle's bit for bit -- uniform data and a mixture of a dense
    cluster, a sparse halo and isolated points (rows whose guess must come out very differently)"""
    from dgg_amd import ops
    rng = np.random.default_rng(N + h)
    xp = (rng.standard_normal((N, h)) * 0.6).astype(np.float32)
    if clustered == "wild":                                          # features outside the fp16 range of the Gram bound, tiny norms
        xp[17, 3] = 1.0e5
        xp[4000:4003] *= 3.0e4
        xp[5000:5100] *= 1e-6
    elif clustered:
        xp[:3000] *= 0.05                                            # tight cluster: its 64-NN radius is tiny
        xp[3000:3040] = xp[3000:3040] * 0.01 + 7.0                   # 40 far-away points: fewer than 64 close neighbours
    idx, val = ops.allpairs_topk(T(xp, dev), K, noise_mode=ops.NOISE_NONE, algo=2)
    ridx, rval = O.allpairs_topk(xp, K=K, noise_mode=O.NOISE_NONE)
    assert np.array_equal(Nn(idx), ridx) and np.array_equal(Nn(val), rval)


@pytest.mark.parametrize("n,F,residual", [(700, 128, True), (1333, 256, False), (2257, 2048, True)])
def test_bf16_variant_layer_without_the_concatenation(dev, n, F, residual):
    """The variant GCNII layer (support = cat[hi, h0], model.py:37-44) through ops.GcniiVariantBf16Fn -- split A operand, h0 packed once,
    [d hi | d h0] from one product with the elementwise terms in its epilogue, weight gradient over the two transposed halves --
    against ops.GcniiBf16Fn on the explicit concatenation: the same bf16 operands in the same contraction order, so the output is
    bit-identical and so are the gradients (the old path adds the same two fp32 terms in the same order)."""
    from dgg_amd import ops
    rng = np.random.default_rng(n + F)
    theta, alpha = 0.405, 0.5

    def run(fn):
        hi = T(rng0.standard_normal((n, F)).astype(np.float32), dev).requires_grad_(True)
        h0 = T(rng0.standard_normal((n, F)).astype(np.float32), dev).requires_grad_(True)
        W = T((rng0.standard_normal((2 * F, F)) / np.sqrt(2 * F)).astype(np.float32), dev).requires_grad_(True)
        inp = T(rng0.standard_normal((n, F)).astype(np.float32), dev).requires_grad_(True) if residual else None
        cot = T(rng0.standard_normal((n, F)).astype(np.float32), dev)
        out = fn(hi, h0, W, inp)
        (out * cot).sum().backward()
        return [out.detach(), hi.grad, h0.grad, W.grad] + ([inp.grad] if residual else [])

    rng0 = np.random.default_rng(5)
    a = run(lambda hi, h0, W, inp: ops.GcniiVariantBf16Fn.apply(hi, h0, W, inp, theta, alpha))
    rng0 = np.random.default_rng(5)
    b = run(lambda hi, h0, W, inp: ops.GcniiBf16Fn.apply(torch.cat([hi, h0], 1), W, hi, h0, inp, theta, alpha))
    for name, x, y in zip(["out", "d hi", "d h0", "d weight", "d input"], a, b):
        assert torch.equal(x, y), f"{name}: max difference {float((x - y).abs().max()):.3e}"
    # the packs of h0 ride on the tensor: a second layer of the stack reuses them, an in-place update invalidates them
    h0 = T(rng.standard_normal((n, F)).astype(np.float32), dev)
    p1 = ops._h0_packs(h0)
    assert ops._h0_packs(h0)[0] is p1[0]
    h0.mul_(2.0)
    p2 = ops._h0_packs(h0)
    assert p2[0] is not p1[0] and torch.equal(p2[0].float(), h0.bfloat16().float())


@pytest.mark.parametrize("n,K,F,variant,residual", [(700, 256, 128, True, True), (333, 128, 128, False, True), (1500, 4096, 2048, True, True)])
def test_bf16_gcnii_layer_product(dev, n, K, F, variant, residual):
    """BASELINE configs[4] (bf16 fwd+bwd): GraphConvolution with gemm_dtype=bfloat16 runs the layer product and its autograd on the
    hand-written bf16 MFMA kernel (dgg_bf16.hip).  Tolerance against the fp32 layer: 1e-2 of the tensor's max (bf16 rounds both
    operands to 8 significant bits: ~4e-3 relative per product, averaged down over the contraction); and against a float64
    product of the bf16-ROUNDED operands: 1e-5 (the kernel itself adds only fp32 accumulation error)."""
    import dgg_amd
    from dgg_amd import ops
    rng = np.random.default_rng(n + K)
    S = T(rng.standard_normal((n, K)).astype(np.float32), dev)
    W = T((rng.standard_normal((K, F)) / np.sqrt(K)).astype(np.float32), dev)
    # the raw kernel against float64 on the rounded operands (padding of an odd contraction length included)
    A16, B16 = ops.pack_bf16(S), ops.pack_bf16(W, transpose=True)
    got = ops.gemm_nt_bf16(A16, B16, 0.5)
    ref = 0.5 * (A16.double() @ B16.double().t())
    assert float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-6
    At, Gt = ops.pack_bf16(S, transpose=True), ops.pack_bf16(got, transpose=True)      # contraction over n (padded to 64)
    dW = ops.gemm_nt_bf16(At, Gt)
    refW = At.double() @ Gt.double().t()
    assert float((dW.double() - refW).abs().max()) <= 1e-5 * float(refW.abs().max()) + 1e-6
    # the layer, bf16 vs fp32, forward and gradients
    in_f = K // 2 if variant else K
    if variant and in_f != F or (not variant and residual and K != F):
        residual = False
    layers = []
    for dt in (None, torch.bfloat16):
        torch.manual_seed(3)
        L = dgg_amd.GraphConvolution(in_f, F, residual=residual, variant=variant).to(dev)
        L.gemm_dtype = dt
        layers.append(L)
    N0 = n
    idx = torch.arange(N0, device=dev, dtype=torch.int32)[:, None].repeat(1, 64)
    idx[:, 1:] = -1
    from dgg_amd.adjacency import EllAdjacency
    vals = torch.zeros((N0, 64), device=dev)
    vals[:, 0] = 1.0
    outs, grads = [], []
    for L in layers:
        x = T(rng.standard_normal((N0, in_f)).astype(np.float32), dev) if not outs else xin.detach().clone()
        xin = x
        x = x.requires_grad_(True)
        h0 = (x * 0.5).detach().requires_grad_(True)
        out = L(x, EllAdjacency(idx, vals, N0), h0, 0.5, 0.1, 1)
        out.square().sum().backward()
        outs.append(out.detach())
        grads.append((x.grad.detach(), h0.grad.detach(), L.weight.grad.detach()))
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())  # noqa: E731
    assert rel(outs[1], outs[0]) <= 1e-2
    for g16, g32 in zip(grads[1], grads[0]):
        assert rel(g16, g32) <= 2e-2


def test_wide_knet_products_on_the_bf16_matrix_cores(dev):
    """latent 256 (the k-net of wide latents runs layer by layer as GEMMs): `bf16=True` moves k_embed and k_mu, forward and backward, to
    the bf16 kernel -- learned degrees within 1e-2 of the fp32 path's, gradients within 1e-1 in the Frobenius norm (measured 4-7e-2: pre-activations that cross the LeakyReLU kink)"""
    from dgg_amd import ops
    rng = np.random.default_rng(4)
    N, h = 900, 256
    xk = T(rng.standard_normal((N, h)).astype(np.float32), dev)
    deg = T((10 + 5 * rng.random(N)).astype(np.float32), dev)
    W1, b1 = T((rng.standard_normal((h // 2, h + 1)) / 16).astype(np.float32), dev), T((0.1 * rng.standard_normal(h // 2)).astype(np.float32), dev)
    Wmu, bmu = T((rng.standard_normal((h // 4, h // 2)) / 11).astype(np.float32), dev), T((0.1 * rng.standard_normal(h // 4)).astype(np.float32), dev)
    Wp, bp = T((rng.standard_normal(h // 4) / 8).astype(np.float32), dev), T(np.array([0.05], np.float32), dev)
    mu_sd = ops.degree_stats(deg)
    dk = T(rng.standard_normal(N).astype(np.float32), dev)
    outs = []
    for bf in (False, True):
        k, z, u, feat = ops.knet_x_fwd(xk, deg, mu_sd, W1, b1, Wmu, bmu, Wp, bp, bf16=bf)
        outs.append((k,) + tuple(ops.knet_x_bwd(h, mu_sd, W1, Wmu, bmu, Wp, z, u, feat, dk, bf16=bf)))
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())  # noqa: E731
    assert 1e-6 < rel(outs[1][0], outs[0][0]) <= 1e-2
    # (pre-activations at the LeakyReLU's kink land on the other side under 8-bit operands: single gradient entries then differ by the
    #  factor 100 of the two slopes, so the gradients are held in the Frobenius norm)
    nrm = lambda a, b: float((a - b).norm() / b.norm())           # noqa: E731
    for a, b in zip(outs[1][1:], outs[0][1:]):
        assert nrm(a.reshape(b.shape), b) <= 1e-1, nrm(a.reshape(b.shape), b)


def _stack_inputs(n, F, L, dev, seed=3):
    """h0, a normalised ELL adjacency with a few empty slots / zero weights, L variant weights, a cotangent"""
    from dgg_amd import ops
    rng = np.random.default_rng(seed)
    K = 64
    h0 = np.maximum(rng.standard_normal((n, F)), 0).astype(np.float32)
    idx = np.stack([rng.choice(n, K, replace=False) for _ in range(n)]).astype(np.int32)
    w = rng.random((n, K)).astype(np.float32)
    w[:, 20:] = 0.0                                               # saturated ramp beyond rank 20
    idx[:, 40:] = -1                                              # empty slots
    w[idx < 0] = 0.0
    rs = w.sum(1)
    ahat = (w / np.sqrt(rs)[:, None] / np.sqrt(rs)[np.maximum(idx, 0)]).astype(np.float32)
    Ws = [(rng.standard_normal((2 * F, F)) / np.sqrt(2 * F)).astype(np.float32) for _ in range(L)]
    cot = rng.standard_normal((n, F)).astype(np.float32)
    idx_t = T(idx, dev)
    part = ops.part_build(idx_t, T(w, dev), n)
    return T(h0, dev), T(ahat, dev), idx_t, part, [T(W, dev) for W in Ws], T(cot, dev)


@pytest.mark.parametrize("n,F,L,residual,gathers", [(700, 256, 4, True, False), (1100, 512, 3, False, False), (1100, 512, 3, True, True)])
def test_gcnii_stack_bf16_matches_the_layers_one_by_one(dev, n, F, L, residual, gathers, monkeypatch):
    """ops.GcniiStackBf16Fn (dropout off) against the same stack through the per-layer pieces it replaces -- EllSpmmFn, GcniiVariantBf16Fn,
    torch.relu under torch autograd.  gathers False: the same bf16 operands in the same contraction order, so the forward is identical up
    to the order of the fp32 epilogue terms (5e-5), and every gradient (h0, the adjacency values, each weight) agrees to 5e-4 of its max
    (d h0 is accumulated over the layers in another order; the ReLU mask is read off the activation).  gathers True (the default for
    widths in multiples of 512): the aggregation, the SDDMM and the transposed aggregation read bf16 copies of what they gather -- the
    aggregated activations are rounded to 8 significant bits as well: forward 1e-2, gradients 3e-2 of max."""
    import math
    from dgg_amd import ops
    monkeypatch.setattr(ops, "STACK_BF16_GATHERS", gathers)
    ftol, gtol = (1e-2, 3e-2) if gathers else (5e-5, 5e-4)       # (the epilogue runs as its own pass: another order of its fp32 terms)
    h0, ahat, idx, part, Ws, cot = _stack_inputs(n, F, L, dev)
    lamda, alpha = 0.5, 0.3

    def leaves():
        return h0.clone().requires_grad_(True), ahat.clone().requires_grad_(True), [W.clone().requires_grad_(True) for W in Ws]

    a_h0, a_ah, a_W = leaves()
    y = ops.GcniiStackBf16Fn.apply(a_h0, a_ah, idx, part, True, residual, 0.0, lamda, alpha, (1, 2), True, *a_W)
    (y * cot).sum().backward()
    b_h0, b_ah, b_W = leaves()
    x = b_h0
    for l, W in enumerate(b_W, 1):
        hi = ops.EllSpmmFn.apply(b_ah, idx, x, True, part, ops.ACT_NONE)
        x = torch.relu(ops.GcniiVariantBf16Fn.apply(hi, b_h0, W, x if residual else None, math.log(lamda / l + 1), alpha))
    (x * cot).sum().backward()
    rel = lambda u, v: float((u.double() - v.double()).abs().max() / v.double().abs().max())  # noqa: E731
    if gathers:
        # (8-bit rounding of the aggregated activations moves pre-activations that sit at the ReLU's kink across it: single elements of
        #  the gradients then differ by a whole term, so the gradients are held in the Frobenius norm)
        rel = lambda u, v: float((u.double() - v.double()).norm() / v.double().norm())  # noqa: E731,F811
    assert rel(y, x) <= ftol, rel(y, x)
    assert gathers == (rel(y, x) > 2e-4), "the bf16 gather copies are in use exactly when asked for"
    assert rel(a_h0.grad, b_h0.grad) <= gtol and rel(a_ah.grad, b_ah.grad) <= gtol, (rel(a_h0.grad, b_h0.grad), rel(a_ah.grad, b_ah.grad))
    for u, v in zip(a_W, b_W):
        assert rel(u.grad, v.grad) <= gtol, rel(u.grad, v.grad)


def _np_drop_keep(s0, s1, n_elem, p):
    """numpy restatement of dgg_common.h drop_keep (mix32 twice, 24-bit threshold)"""
    def mix32(x):
        x = x.astype(np.uint32)
        x ^= x >> np.uint32(16); x = (x * np.uint32(0x7feb352d)).astype(np.uint32)
        x ^= x >> np.uint32(15); x = (x * np.uint32(0x846ca68b)).astype(np.uint32)
        x ^= x >> np.uint32(16)
        return x
    e = np.arange(n_elem, dtype=np.uint32)
    x = mix32(mix32(e ^ np.uint32(s0)) ^ np.uint32(s1))
    return (x >> np.uint32(8)) >= np.uint32(int(p * 16777216.0))


def test_gcnii_stack_bf16_dropout_is_the_counter_based_mask(dev):
    """Training mode: the stack applies dropout(h0) in front of the first layer and dropout(relu(.)) in every layer's epilogue with the
    counter-based mask of dgg_common.h (element e of layer l kept iff hash24(s0, s1 ^ l*0x9E3779B9, e) >= p 2^24).  The masks are
    restated in numpy, the stack in torch (float64, GEMM operands rounded to bf16 as the kernel rounds them) with those masks under
    torch autograd: output 1e-4, gradients of h0 / adjacency values / weights 3e-2 of max (the kernel's backward products round the
    cotangent to bf16 as well), keep rate 1 - p to 1 %."""
    import math
    from dgg_amd import ops
    n, F, L, p, lamda, alpha = 600, 256, 3, 0.3, 0.5, 0.4
    s0, s1 = 12345, 678
    h0, ahat, idx, part, Ws, cot = _stack_inputs(n, F, L, dev, seed=8)
    a_h0, a_ah = h0.clone().requires_grad_(True), ahat.clone().requires_grad_(True)
    a_W = [W.clone().requires_grad_(True) for W in Ws]
    y = ops.GcniiStackBf16Fn.apply(a_h0, a_ah, idx, part, True, True, p, lamda, alpha, (s0, s1), True, *a_W)
    (y * cot).sum().backward()
    y2 = ops.GcniiStackBf16Fn.apply(h0, ahat, idx, part, True, True, p, lamda, alpha, (s0, s1), False, *Ws)
    assert torch.equal(y, y2), "same seeds, same mask"
    masks = [torch.from_numpy(_np_drop_keep(s0, (s1 ^ (0x9E3779B9 * l)) & 0xFFFFFFFF, n * F, p).reshape(n, F)) for l in range(L + 1)]
    assert abs(float(masks[1].double().mean()) - (1 - p)) < 0.01
    r16 = lambda t_: t_.float().bfloat16().double()                # noqa: E731
    d_h0 = h0.detach().cpu().double().requires_grad_(True)
    d_ah = ahat.detach().cpu().double().requires_grad_(True)
    d_W = [W.detach().cpu().double().requires_grad_(True) for W in Ws]
    idc = idx.cpu().long()
    valid = idc >= 0
    A = torch.zeros(n, n, dtype=torch.float64).index_put((torch.arange(n)[:, None].expand_as(idc)[valid], idc[valid]), d_ah[valid], accumulate=True)

    class R16(torch.autograd.Function):                            # rounding with a straight-through gradient
        @staticmethod
        def forward(ctx, t_):
            return r16(t_)

        @staticmethod
        def backward(ctx, g_):
            return g_

    x = d_h0 * masks[0] / (1 - p)
    for l, W in enumerate(d_W, 1):
        theta = math.log(lamda / l + 1)
        hi = A @ x
        out = theta * (R16.apply(torch.cat([hi, d_h0], 1)) @ R16.apply(W)) + (1 - theta) * ((1 - alpha) * hi + alpha * d_h0) + x
        x = torch.relu(out) * masks[l] / (1 - p)
    (x * cot.cpu().double()).sum().backward()
    rel = lambda u, v: float((u.detach().cpu().double() - v).abs().max() / v.abs().max())  # noqa: E731
    assert rel(y, x.detach()) <= 1e-4, rel(y, x.detach())
    live = (ahat != 0).cpu()                                       # (skip_zero: an exact zero weight is a saturated ramp, its gradient is dropped)
    assert float(a_ah.grad.cpu()[~live].abs().max()) == 0.0
    assert rel(a_h0.grad, d_h0.grad) <= 3e-2 and rel(a_ah.grad.cpu()[live], d_ah.grad[live]) <= 3e-2
    for u, v in zip(a_W, d_W):
        assert rel(u.grad, v.grad) <= 3e-2


def _config4_models(dev, kinds=(None, torch.bfloat16, "stack")):
    """GCNIIppi_DGG at BASELINE configs[4]'s shape, one per entry of `kinds`, all from the same seed: None = fp32 layer products,
    torch.bfloat16 = the bf16 MFMA kernel layer by layer, "stack" = the fused stack with the generator's k-net in bf16 as well."""
    from argparse import Namespace
    import dgg_amd
    d, hid, C, L = 50, 2048, 121, 9
    args = Namespace(extra_edge_dim=0, extra_k_dim=1, dgg_hard=False, deg_mean=3.899, deg_std=5.288, dgg_mode_edge_net="u-v-dist",
                     dgg_mode_k_net="x", dgg_mode_k_select="k_times_edge_prob", debug_step=3, perturb_edge_prob=True,
                     symmetric_noise=False, stochastic_k=False, dgg_adj_input="input_adj", n_dgg_layers=1)
    models = []
    for dt in kinds:
        torch.manual_seed(0)
        m = dgg_amd.GCNIIppi_DGG(nfeat=d, nlayers=L, nhidden=hid, nclass=C, dropout=0.0, lamda=0.5, alpha=0.5, variant=True, args=args).to(dev)
        with torch.no_grad():
            for dg in m.dggs:
                dg.k_net.k_project.weight.mul_(0.1)
        for conv in m.convs:
            conv.gemm_dtype = torch.bfloat16 if dt == "stack" else dt
        m.fused_stack = dt == "stack"                               # models[1]: the layers one by one (forward hooks below); [2]: ops.GcniiStackBf16Fn
        if dt == "stack":                                           # ... and the generator's latent-2048 k-net products in bf16 as well
            for dg in m.dggs:
                dg.gemm_dtype = torch.bfloat16
        m.train()                                                   # training mode: the DGG perturbs the scores (noise=True)
        models.append(m)
    return models, (d, hid, C, L)


def test_config4_ppi_gcniippi_dgg_bf16_end_to_end(dev):
    """BASELINE configs[4] end to end: GCNIIppi_DGG (hidden 2048, 9 variant GCNII layers with residual, DGG at latent 2048 on
    edge-list candidates; reference model.py:887-965, train_ppi.py:43-44, 204-219) on two PPI-shaped graphs, forward + backward,
    with the GCNII layer products on the bf16 MFMA kernel (gemm_dtype = bfloat16):
      (a) against the SAME model in fp32 (same parameters, same noise seed, identical neighbour lists): outputs within 1e-2 and
          every gradient within 3e-2 of the tensor's max (bf16 rounds both GEMM operands to 8 significant bits; the gradient of
          the first layer has passed through the bf16 backward products of all nine layers);
      (b) against a float64 restatement of the layer stack (model.py:32-44, 942-957) that rounds the GEMM operands to bf16 exactly
          as the kernel does, on the adjacency the module produced: EVERY LAYER, fed the module's own fp32 input of that layer,
          within 1e-5 of its output's max (the kernel adds only fp32 accumulation error), and the whole 9-layer stack end to end
          within 1e-4 (fp32 aggregation / epilogue / fc layers against float64 through nine residual layers: measured 3.6e-5)."""
    import math
    from bench import pubmed_graph
    models, (d, hid, C, L) = _config4_models(dev)
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())  # noqa: E731
    for n in (1300, 640):
        rows, cols = pubmed_graph(n, n * 14, seed=n)
        keep = rows != cols
        A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows[keep], cols[keep]])), torch.ones(int(keep.sum())), (n, n)).coalesce().to(dev)
        g = torch.Generator().manual_seed(n)
        x = torch.randn(n, d, generator=g).to(dev)
        y = (torch.rand(n, C, generator=g) < 0.3).float().to(dev)
        outs, grads, adjs, taps = [], [], [], []
        hooks = [con.register_forward_hook(lambda mod, inp, out: taps.append((inp[0].detach(), inp[2].detach(), out.detach())))
                 for con in models[1].convs]
        for m in models:
            for p_ in m.parameters():
                p_.grad = None
            torch.manual_seed(77)                                   # the module draws its noise seed from torch's generator
            out, unnorm = m._body(x, A, None, None)
            prob = torch.sigmoid(out)
            torch.nn.functional.binary_cross_entropy(prob, y).backward()
            outs.append(prob.detach())
            grads.append({k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None})
            adjs.append(unnorm)
        assert torch.equal(adjs[0].idx, adjs[1].idx), "the two runs must select the same graph (same seed, fp32 DGG in both)"
        # the fused stack (dropout 0): the same products as the layers one by one, its three gather kernels on bf16 copies of the
        # activations / of d hi (ops.STACK_BF16_GATHERS: one more 8-bit rounding per layer)
        assert torch.equal(adjs[2].idx, adjs[1].idx) and rel(outs[2], outs[1]) <= 1e-2, rel(outs[2], outs[1])
        assert set(grads[2]) == set(grads[1])
        for k in grads[1]:       # (the k-net's own weights: its two wide products run on bf16 operands in model [2], on fp32 in [1])
            assert rel(grads[2][k], grads[1][k]) <= (1e-1 if ".k_" in k else 3e-2), (k, rel(grads[2][k], grads[1][k]))
        assert rel(outs[2], outs[0]) <= 1e-2
        assert rel(outs[1], outs[0]) <= 1e-2
        assert set(grads[0]) == set(grads[1]) and len(grads[0]) >= L + 4
        for k in grads[0]:
            assert rel(grads[1][k], grads[0][k]) <= 3e-2, k          # (measured: up to 2.2e-2, on the first layer's weight)
        # (b) float64 restatement with bf16-rounded GEMM operands, on the module's own normalised adjacency
        for hk in hooks:
            hk.remove()
        m = models[1]
        Ahat = adjs[1].normalize().to_dense().double()
        r16 = lambda t_: t_.float().bfloat16().double()            # noqa: E731  (round to nearest even, as pack_bf16)
        assert len(taps) == L
        with torch.no_grad():
            for i, (con, (inp, h0t, outt)) in enumerate(zip(m.convs, taps)):      # layer by layer, on the module's own inputs
                theta = math.log(m.lamda / (i + 1) + 1)
                hi = Ahat @ inp.double()
                want = theta * (r16(torch.cat([hi, h0t.double()], 1)) @ r16(con.weight)) + (1 - theta) * ((1 - m.alpha) * hi + m.alpha * h0t.double()) + inp.double()
                assert rel(outt, want) <= 1e-5, (i, rel(outt, want))
        with torch.no_grad():
            h = torch.relu(x.double() @ m.fcs[0].weight.double().t() + m.fcs[0].bias.double())
            h0 = h
            for i, con in enumerate(m.convs):
                theta = math.log(m.lamda / (i + 1) + 1)
                hi = Ahat @ h
                support = torch.cat([hi, h0], 1)
                r = (1 - m.alpha) * hi + m.alpha * h0
                h = torch.relu(theta * (r16(support) @ r16(con.weight)) + (1 - theta) * r + h)
            ref = torch.sigmoid(h @ m.fcs[-1].weight.double().t() + m.fcs[-1].bias.double())
        assert rel(outs[1], ref) <= 1e-4, rel(outs[1], ref)


def test_config4_ppi_twenty_graph_batch_bf16_against_fp32(dev):
    """The batch bench.py times for BASELINE configs[4]: 20 PPI-shaped graphs of 591..3480 nodes (the bench's own sizes, seed 0),
    one optimiser step per graph as train_ppi.py:204-219 does.  The fused bf16 stack against the SAME model with fp32 layer
    products, graph by graph from the same parameters and the same noise seed: identical neighbour lists, outputs within 1e-2,
    every layer gradient within 4e-2 of the tensor's max and the k-net's own weights (whose two wide products run on bf16
    operands) within 1.5e-1.  Measured worst case over the 20 graphs on MI355X: outputs 6.0e-3, layer gradients 2.94e-2, k-net
    1.05e-1 -- the two-graph test above holds 3e-2 / 1e-1 on its two graphs; the maximum over ten times as many draws sits at
    those limits, so the batch test states its own."""
    from bench import pubmed_graph
    models, (d, hid, C, L) = _config4_models(dev, kinds=(None, "stack"))
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())  # noqa: E731
    sizes = np.random.default_rng(0).integers(591, 3481, size=20)
    assert sizes.min() >= 591 and sizes.max() <= 3480
    worst = {"out": 0.0, "grad": 0.0, "knet": 0.0}
    for gi, n in enumerate(int(v) for v in sizes):
        rows, cols = pubmed_graph(n, n * 14, seed=n)
        keep = rows != cols
        A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows[keep], cols[keep]])), torch.ones(int(keep.sum())), (n, n)).coalesce().to(dev)
        g = torch.Generator().manual_seed(n)
        x = torch.randn(n, d, generator=g).to(dev)
        y = (torch.rand(n, C, generator=g) < 0.3).float().to(dev)
        outs, grads, adjs = [], [], []
        for m in models:
            for p_ in m.parameters():
                p_.grad = None
            torch.manual_seed(1000 + gi)
            out, unnorm = m._body(x, A, None, None)
            prob = torch.sigmoid(out)
            torch.nn.functional.binary_cross_entropy(prob, y).backward()
            outs.append(prob.detach())
            grads.append({k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None})
            adjs.append(unnorm)
        assert torch.equal(adjs[0].idx, adjs[1].idx), (gi, n)
        assert set(grads[0]) == set(grads[1]) and len(grads[0]) >= L + 4
        worst["out"] = max(worst["out"], rel(outs[1], outs[0]))
        for k in grads[0]:
            key = "knet" if ".k_" in k else "grad"
            worst[key] = max(worst[key], rel(grads[1][k], grads[0][k]))
    assert worst["out"] <= 1e-2 and worst["grad"] <= 4e-2 and worst["knet"] <= 1.5e-1, worst


def test_bf16_weight_packs_are_never_stale(dev):
    """two GraphConvolution layers of different size built back to back (the second parameter may land on the first one's freed
    address, both at the same autograd version) and an update through `p.data` (which does not move the version counter): the bf16
    product must use the CURRENT weights every time"""
    import dgg_amd
    from dgg_amd.adjacency import EllAdjacency
    n = 200
    idx = torch.arange(n, device=dev, dtype=torch.int32)[:, None].repeat(1, 64)
    idx[:, 1:] = -1
    vals = torch.zeros((n, 64), device=dev)
    vals[:, 0] = 1.0
    adj = EllAdjacency(idx, vals, n)

    def run(L, x):
        return L(x, adj, x * 0.5, 0.5, 0.1, 1)

    for width in (128, 64, 256, 128):
        L = dgg_amd.GraphConvolution(width, width, residual=True, variant=True).to(dev)
        L.gemm_dtype = torch.bfloat16
        x = torch.randn(n, width, device=dev)
        a = run(L, x)
        L.gemm_dtype = None
        b = run(L, x)
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()), width
        L.gemm_dtype = torch.bfloat16
        L.weight.data.mul_(-2.0)                                    # does not bump L.weight._version
        L.gemm_dtype = None
        b2 = run(L, x)
        L.gemm_dtype = torch.bfloat16
        a2 = run(L, x)
        assert float((a2 - b2).abs().max()) <= 2e-2 * float(b2.abs().max()), "stale bf16 pack after an update through .data"
        del L


@pytest.mark.parametrize("mode", [0, 1, 3])
def test_ramp_fused_into_the_ranked_search(dev, mode):
    """dgg_allpairs_topk_ranked_softk == dgg_allpairs_topk(k_limit) followed by dgg_softk_fwd, bit for bit (whole block and a row shard)"""
    from dgg_amd import ops
    rng = np.random.default_rng(17 + mode)
    N, h = 3000, 64
    xp = T((rng.standard_normal((N, h)) * 0.6).astype(np.float32), dev)
    k = T((3 + 40 * rng.random(N)).astype(np.float32), dev)
    for r0, r1 in [(0, N), (700, 1901)]:
        kk = k[r0:r1].contiguous()
        idx, val = ops.allpairs_topk(xp, K, noise_mode=ops.NOISE_RANKED, seed=(5, 9), rows=(r0, r1), k_limit=kk)
        w, rs = ops.softk_fwd(idx, val, kk, mode)
        got = ops.allpairs_topk_softk(xp, kk, mode, seed=(5, 9), rows=(r0, r1))
        for a, b in zip(got, (idx, val, w, rs)):
            assert torch.equal(a, b)


@pytest.mark.parametrize("N,lo,hi,hub", [(1300, 0, 1300, False), (1300, 400, 977, False), (5000, 0, 5000, True), (70, 0, 70, False)])
def test_payload_partition_is_a_csc_view_of_the_active_entries(dev, N, lo, hi, hub):
    """dgg_partp_build_norm: the records are exactly the active entries (idx >= 0, w != 0) of the block, grouped by destination node
    (nodeptr), each carrying w rs_i^-1/2 and the score.  `hub`: one node receives an edge from every row, so that its bucket
    exceeds what the sort keeps in registers (the re-reading form of pp_sort)."""
    from dgg_amd import ops
    rng = np.random.default_rng(5 + N)
    rows = hi - lo
    idx = rng.integers(0, N, size=(rows, K)).astype(np.int32)
    idx[rng.random((rows, K)) < 0.3] = -1
    w = rng.random((rows, K)).astype(np.float32)
    w[rng.random((rows, K)) < 0.2] = 0.0
    if hub:
        idx[:, 7] = 1234
        w[:, 7] = 0.5
        idx[:, 8:40] = rng.integers(1152, 1280, size=(rows, 32)).astype(np.int32)      # ~ 160 000 records in one 128-node range
        w[:, 8:40] = 0.25
    val = rng.random((rows, K)).astype(np.float32)
    rs = (0.5 + rng.random(N)).astype(np.float32)
    part, ahat = ops.partp_build(T(idx, dev), T(w, dev), T(val, dev), T(rs[lo:hi], dev), N, T(rs, dev))
    nodeptr, recs = ops.partp_records(part)
    nodeptr, recs = Nn(nodeptr), Nn(recs)
    act = (idx >= 0) & (w != 0)
    assert nodeptr[0] == 0 and nodeptr[-1] == act.sum() and np.all(np.diff(nodeptr) >= 0)
    assert np.array_equal(np.repeat(np.arange(N), np.diff(nodeptr)), recs[:, 1]), "records are not grouped by destination node"
    ai = (np.float32(1.0) / np.sqrt(rs[lo:hi])).astype(np.float32)
    wa = (ai[:, None] * w).astype(np.float32)
    r_, c_ = np.nonzero(act)
    exp = np.stack([(r_ * 64 + c_).astype(np.int32), idx[r_, c_], wa[r_, c_].view(np.int32), val[r_, c_].view(np.int32)], 1)
    order = np.lexsort((exp[:, 0], exp[:, 1]))
    got_order = np.lexsort((recs[:, 0], recs[:, 1]))
    assert np.array_equal(exp[order], recs[got_order])
    aj = (np.float32(1.0) / np.sqrt(rs)).astype(np.float32)
    exp_ahat = np.where(act, (wa * aj[np.where(act, idx, 0)]).astype(np.float32), np.float32(0.0))
    assert np.array_equal(Nn(ahat), exp_ahat)


def test_normalisation_fused_into_the_partition_build(dev):
    """dgg_partp_build_norm writes the bits of dgg_ell_normalize_fwd (row shard: own rows, global row sums)"""
    from dgg_amd import ops
    rng = np.random.default_rng(23)
    N = 1300
    xp, k, idx, val, w, rs, ahat = _graph(rng, N, 32)
    for lo, hi in [(0, N), (400, 977)]:
        sl = slice(lo, hi)
        got = ops.partp_build(T(idx[sl], dev), T(w[sl], dev), T(val[sl], dev), T(rs[sl], dev), N, T(rs, dev))
        assert got is not None
        ref = ops.normalize_fwd(T(idx[sl], dev), T(w[sl], dev), T(rs, dev), lo)
        assert torch.equal(got[1], ref)
        assert np.array_equal(Nn(got[1]), ahat[sl])


@pytest.mark.gpu
@pytest.mark.parametrize("R,Cc", [(130, 200), (64, 64), (257, 36), (33, 50), (4096, 2048), (7, 3)])
def test_bf16_packs_are_the_rounded_copies(dev, R, Cc):
    """dgg_pack_bf16 / dgg_pack_bf16_both: plain and transposed bf16 copies, zero padded to multiples of 64, bit for bit the
    round-to-nearest-even conversion torch makes -- on shapes that take the 64 x 64-tile kernels and on shapes that do not"""
    from dgg_amd import ops
    g = torch.Generator().manual_seed(R * 1000 + Cc)
    x = torch.randn(R, Cc, generator=g).to(dev)
    xb = x.to(torch.bfloat16)
    plain, tr = ops.pack_bf16(x), ops.pack_bf16(x, transpose=True)
    both = ops.pack_bf16_both(x)
    for p_ in (plain, both[0]):
        assert p_.shape == (R, (Cc + 63) // 64 * 64)
        assert torch.equal(p_[:, :Cc], xb) and not bool(p_[:, Cc:].any())
    for t_ in (tr, both[1]):
        assert t_.shape == (Cc, (R + 63) // 64 * 64)
        assert torch.equal(t_[:, :R], xb.t()) and not bool(t_[:, R:].any())


def _hub_graph(n, K, deg, hubs, seed):
    """ELL block [n,K] whose first `deg` entries per row are valid; the columns in `hubs` receive an entry from EVERY row (in-degree n:
    runs of the destination-ordered partition far longer than a chunk), the rest are uniform"""
    g = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, n, (n, K), generator=g, dtype=torch.int32)
    for q, hcol in enumerate(hubs):
        idx[:, q] = hcol
    idx[:, deg:] = -1
    a = torch.rand(n, K, generator=g) + 0.1
    a[:, deg:] = 0
    return idx, a


@pytest.mark.gpu
@pytest.mark.parametrize("b16", [False, True])
@pytest.mark.parametrize("n,deg,hubs", [(300, 9, (5, 250)), (1000, 29, ()), (700, 3, (0,)), (65, 1, ())])
def test_transposed_aggregation_with_hub_columns(dev, n, deg, hubs, b16):
    """dX += A^T dY on run-aligned chunks of the partition (dgg_ell_spmm_t_part / _b16): columns whose runs are longer than a chunk
    (split, added atomically), runs that end exactly at a chunk's end, short graphs -- against the dense product, into a non-zero dX"""
    import ctypes as C
    from dgg_amd import _lib, ops
    F, K = 512, 32
    idx, a = _hub_graph(n, K, deg, hubs, seed=n + deg)
    g = torch.Generator().manual_seed(3)
    dY = torch.randn(n, F, generator=g)
    dX0 = torch.randn(n, F, generator=g)
    if b16:
        dY = dY.to(torch.bfloat16).float()                      # (so that the bf16 copy is exact)
    A = torch.zeros(n, n, dtype=torch.float64)
    rows = torch.arange(n)[:, None].expand(n, K)
    valid = idx >= 0
    A.index_put_((rows[valid], idx[valid].long()), a[valid].double(), accumulate=True)
    ref = dX0.double() + A.t() @ dY.double()
    idx_d, a_d, dX = idx.to(dev), a.to(dev), dX0.to(dev).clone()
    part = ops.part_build(idx_d, a_d, n)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if b16:
        dyb = dY.to(dev).to(torch.bfloat16)
        _lib.check(_lib.lib().dgg_ell_spmm_t_part_b16(p(a_d), p(dyb), n, K, F, p(part), n, p(dX), st), "spmm_t_b16")
    else:
        dyd = dY.to(dev)
        _lib.check(_lib.lib().dgg_ell_spmm_t_part(p(a_d), p(dyd), n, K, F, p(part), n, p(dX), st), "spmm_t")
    err = float((dX.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err <= 1e-5, err


@pytest.mark.gpu
@pytest.mark.parametrize("n,deg,F", [(300, 9, 512), (1000, 29, 2048), (65, 1, 1024)])
def test_sliced_sddmm_matches_the_whole_row_kernel(dev, n, deg, F):
    """dgg_ell_sddmm_b16_sliced (512-feature slices + the slice sum; overwrite and accumulate) == dgg_ell_sddmm_b16 up to the fp32
    summation order, padding entries 0"""
    import ctypes as C
    from dgg_amd import _lib
    K = 32
    idx, a = _hub_graph(n, K, deg, (), seed=n)
    a[:, 0] = 0                                                  # a zero-weight entry: skipped by skip_zero, its dA is 0
    g = torch.Generator().manual_seed(4)
    xb = torch.randn(n, F, generator=g).to(dev).to(torch.bfloat16)
    dyb = torch.randn(n, F, generator=g).to(dev).to(torch.bfloat16)
    idx_d, a_d = idx.to(dev), a.to(dev)
    L = _lib.lib()
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = torch.empty(n, K, device=dev)
    _lib.check(L.dgg_ell_sddmm_b16(p(idx_d), p(a_d), p(xb), p(dyb), n, K, F, 1, p(ref), st), "sddmm")
    ws = torch.empty(int(L.dgg_ell_sddmm_b16_ws_floats(n, K, F)), device=dev)
    got = torch.full((n, K), 7.0, device=dev)
    _lib.check(L.dgg_ell_sddmm_b16_sliced(p(idx_d), p(a_d), p(xb), p(dyb), n, K, F, 1, p(ws), p(got), 0, st), "sddmm_sliced")
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 1e-5 * scale
    assert not bool(got[:, deg:].any()) and not bool(got[:, 0].any())
    _lib.check(L.dgg_ell_sddmm_b16_sliced(p(idx_d), p(a_d), p(xb), p(dyb), n, K, F, 1, p(ws), p(got), 1, st), "sddmm_sliced")
    assert float((got - 2 * ref).abs().max()) <= 2e-5 * scale
    # a series of calls that leave their slice sums in the workspace (dA NULL), one reduction at the end
    for acc in (0, 1, 1):
        _lib.check(L.dgg_ell_sddmm_b16_sliced(p(idx_d), p(a_d), p(xb), p(dyb), n, K, F, 1, p(ws), None, acc, st), "sddmm_sliced")
    _lib.check(L.dgg_ell_sddmm_slices_sum(p(ws), n, K, F, p(got), 0, st), "slices_sum")
    assert float((got - 3 * ref).abs().max()) <= 3e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("n,F", [(200, 256), (333, 512)])
def test_dsupport_halves_accumulate_into_dh0(dev, n, F):
    """dgg_gcnii_dsupport_bf16_b: [d hi | d h0] = theta g W^T + [c1 | c2] g as one product per half; bf16 copy of d hi with and without
    the fp32 one; accumulate_dh0 adds to what dh0 holds"""
    import ctypes as C
    from dgg_amd import _lib, ops
    g_ = torch.Generator().manual_seed(n)
    g = torch.randn(n, F, generator=g_).to(dev)
    W = (torch.randn(2 * F, F, generator=g_) / 16).to(dev)
    theta, alpha = 0.37, 0.2
    Gp, Wp = ops.pack_bf16(g), ops.pack_bf16(W)
    full = theta * (Gp.float() @ Wp.float().t())               # [n, 2F] on the rounded operands
    c1, c2 = (1 - theta) * (1 - alpha), (1 - theta) * alpha
    ref_hi, ref_h0 = full[:, :F] + c1 * g, full[:, F:] + c2 * g
    L = _lib.lib()
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dhi, dh0 = torch.empty(n, F, device=dev), torch.empty(n, F, device=dev)
    dhib = torch.empty(n, F, device=dev, dtype=torch.bfloat16)
    _lib.check(L.dgg_gcnii_dsupport_bf16_b(p(Gp), p(Wp), n, F, p(g), theta, alpha, p(dhi), p(dh0), p(dhib), 0, st), "dsupport")
    tol = 2e-3 * float(ref_hi.abs().max())
    assert float((dhi - ref_hi).abs().max()) <= tol and float((dh0 - ref_h0).abs().max()) <= tol
    assert torch.equal(dhib, dhi.to(torch.bfloat16))
    base = torch.randn(n, F, generator=g_).to(dev)
    dh0b, dhib2 = base.clone(), torch.empty_like(dhib)
    _lib.check(L.dgg_gcnii_dsupport_bf16_b(p(Gp), p(Wp), n, F, p(g), theta, alpha, None, p(dh0b), p(dhib2), 1, st), "dsupport")
    assert torch.equal(dhib2, dhib)
    assert float((dh0b - (base + dh0)).abs().max()) <= 1e-5 * float(dh0.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["ring8", "ring4", "staged"])
def test_bf16_product_shapes_and_pipeline_edges(dev, mode, monkeypatch):
    """dgg_gemm_nt_bf16 over contraction lengths of 1-5 K-steps (the ring's prologue holds three stages: every way of starting and
    draining it), row / column counts inside, at and across tile edges, both tile heights -- against the fp32 product of the bf16
    operands; the eight- and four-wavefront ring loops and the register-staged loop"""
    from dgg_amd import ops
    if mode == "ring4":
        monkeypatch.setenv("DGG_BF16_WAVES", "4")
    if mode == "staged":
        monkeypatch.setenv("DGG_BF16_RING", "0")
    g = torch.Generator().manual_seed(11)
    for tile in ("64", "128"):
        monkeypatch.setenv("DGG_BF16_TILE", tile)
        for M in (1, 63, 64, 130, 257):
            for N in (32, 128, 200):
                for Kc in (64, 128, 192, 256, 320):
                    A = torch.randn(M, Kc, generator=g).to(dev).to(torch.bfloat16)
                    B = torch.randn(N, Kc, generator=g).to(dev).to(torch.bfloat16)
                    got = ops.gemm_nt_bf16(A, B, 0.5)
                    ref = 0.5 * (A.float() @ B.float().t())
                    assert got.shape == (M, N)
                    assert float((got - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1.0), (mode, tile, M, N, Kc)
