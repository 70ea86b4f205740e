"""Row-sharded DGG step (dgg_amd.parallel.ShardedDGGConv) exercised WITHOUT a GPU: 2 processes, gloo backend.

The HIP kernels cannot run here, so the kernel namespace is substituted by a CPU stand-in (numpy + the oracle for
the all-pairs stage).  What is tested is the partition / collective logic: the 2-rank result (forward rows of each
rank, summed parameter gradients, reduce-scattered dX) must equal the single-process result on the same inputs.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class CpuKern:
    """numpy restatement of the dgg_amd.ops signatures used by ShardedDGGConv (test stand-in only)."""

    @staticmethod
    def _t(a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def linear_fwd(self, x, W, b, act, w_layout):
        y = x.numpy() @ (W.numpy().T if w_layout == 0 else W.numpy())
        if b is not None:
            y = y + b.numpy()
        if act == 1:
            y = np.where(y > 0, y, 0.01 * y)
        elif act == 2:
            y = np.maximum(y, 0)
        return self._t(y.astype(np.float32))

    def linear_bwd(self, x, W, y, dy, act, w_layout, need_dx, need_db):
        g = dy.numpy().astype(np.float64)
        if act == 1:
            g = np.where(y.numpy() > 0, g, 0.01 * g)
        elif act == 2:
            g = np.where(y.numpy() > 0, g, 0.0)
        Wn = W.numpy().astype(np.float64)
        dW = g.T @ x.numpy() if w_layout == 0 else x.numpy().T.astype(np.float64) @ g
        dx = (g @ Wn if w_layout == 0 else g @ Wn.T) if need_dx else None
        return (self._t(dx.astype(np.float32)) if need_dx else None, self._t(dW.astype(np.float32)),
                self._t(g.sum(0).astype(np.float32)) if need_db else None)

    def degree_stats(self, deg):
        d = deg.numpy().astype(np.float64)
        return self._t(np.array([d.mean(), d.std(ddof=1)], np.float32))

    def knet_x_fwd(self, xk, deg, mu_sd, W1, b1, Wmu, bmu, Wp, bp):
        mu, sd = mu_sd.numpy()
        nd = (deg.numpy() - mu) / (sd + 1e-5)
        feat = np.concatenate([xk.numpy(), nd[:, None]], 1).astype(np.float32)
        pre = feat @ W1.numpy().T + b1.numpy()
        z = np.where(pre > 0, pre, 0.01 * pre)
        m = z @ Wmu.numpy().T + bmu.numpy()
        u = (m @ Wp.numpy() + bp.numpy()[0]) * sd + mu
        k = np.maximum(u, 0) + 1
        return self._t(k.astype(np.float32)), self._t(z.astype(np.float32)), self._t(u.astype(np.float32)), self._t(feat)

    def knet_x_bwd(self, h, mu_sd, W1, Wmu, bmu, Wp, z, u, feat, dk):
        sd = float(mu_sd.numpy()[1])
        dkp = np.where(u.numpy() > 0, dk.numpy() * sd, 0.0).astype(np.float64)
        m = z.numpy() @ Wmu.numpy().T + bmu.numpy()
        dm = dkp[:, None] * Wp.numpy()[None, :]
        dz = dm @ Wmu.numpy()
        dp1 = np.where(z.numpy() > 0, dz, 0.01 * dz)
        dxk = dp1 @ W1.numpy()[:, :h]
        f32 = lambda a: self._t(np.asarray(a, np.float32))  # noqa: E731
        return (f32(dxk), f32(dp1.T @ feat.numpy()), f32(dp1.sum(0)), f32(dm.T @ z.numpy()), f32(dm.sum(0)),
                f32((dkp[:, None] * m).sum(0)[None, :]), f32([dkp.sum()]))

    def allpairs_topk(self, xp, K, t, noise_mode, G, seed, rows=None, algo=0, k_limit=None):
        sys.path.insert(0, ROOT)
        from oracle import oracle as O
        idx, val = O.allpairs_topk(xp.numpy(), K=K, t=t, noise_mode=noise_mode, seed=seed, rows=rows)
        if k_limit is not None:                          # ranks the ramp zeroes exactly (include/dgg_hip.h, k_limit)
            L = np.minimum(np.ceil(k_limit.numpy() + np.float32(8.5)) + 1, K)
            cut = np.arange(K)[None, :] >= L[:, None]
            idx[cut], val[cut] = -1, 0.0
        return self._t(idx), self._t(val)

    def edgelist_topk(self, xp, rowptr, col, K, t, noise_mode, G, seed):
        sys.path.insert(0, ROOT)
        from oracle import oracle as O
        idx, val = O.edgelist_topk(xp.numpy(), rowptr.numpy(), col.numpy(), K=K, t=t, noise_mode=noise_mode, seed=seed)
        return self._t(idx), self._t(val)

    # edge-MLP scorer (oracle/oracle.c, ora_edge_mlp_*)
    @staticmethod
    def _n(a):
        return None if a is None else a.detach().numpy()

    def edge_mlp_fwd(self, AB, xp, erow, col, deg, ex_in, ex_mode, t_ex, wdu, wdv, wex, b1, w2, b2, act=1):
        from oracle import oracle as O
        n = self._n
        p, ex = O.edge_mlp_fwd(n(AB), n(xp), n(erow), n(col), n(deg), n(ex_in), ex_mode, t_ex, n(wdu), n(wdv), n(wex), n(b1), n(w2),
                               float(b2.reshape(-1)[0]), act)
        return self._t(p), (self._t(ex) if ex_mode else None)

    def edgelist_topk_p(self, p_edge, N, rowptr, col, K, noise_mode, G, seed):
        from oracle import oracle as O
        return tuple(self._t(a) for a in O.edgelist_topk_p(p_edge.numpy(), N, rowptr.numpy(), col.numpy(), K, noise_mode, None, seed))

    def edge_mlp_bwd(self, AB, idx, eid, val, dval, deg, ex, wdu, wdv, wex, b1, w2, b2, act=1, perturb=False, need_dex=False):
        from oracle import oracle as O
        n = self._n
        dAB, dpar, dex = O.edge_mlp_bwd(n(AB), n(idx), n(eid), n(val), n(dval), n(deg), n(ex), n(wdu), n(wdv), n(wex), n(b1), n(w2),
                                        float(b2.reshape(-1)[0]), act, perturb)
        return self._t(dAB), self._t(dpar), (self._t(dex) if need_dex else None)

    @staticmethod
    def _ramp(K, k):
        r = np.arange(K, dtype=np.float64)[None, :]
        th = np.tanh(r - k[:, None].astype(np.float64))
        return 1 - 0.5 * (1 + th), 0.5 * (1 - th * th)

    def softk_fwd(self, idx, val, k, mode):
        f, _ = self._ramp(idx.shape[1], k.numpy())
        w = np.where(idx.numpy() >= 0, val.numpy() * f if mode == 0 else f, 0.0).astype(np.float32)
        return self._t(w), self._t(w.sum(1).astype(np.float32))

    def normalize_fwd(self, idx, w, rs, row0=0):
        a = 1.0 / np.sqrt(rs.numpy().astype(np.float64))
        n = idx.shape[0]
        j = np.maximum(idx.numpy(), 0)
        out = np.where(idx.numpy() >= 0, a[row0:row0 + n, None] * w.numpy() * a[j], 0.0)
        return self._t(out.astype(np.float32))

    def spmm_fwd(self, idx, ahat, X, act=0):
        j = np.maximum(idx.numpy(), 0)
        y = np.einsum("nk,nkf->nf", ahat.numpy().astype(np.float64), X.numpy()[j])
        return self._t((np.maximum(y, 0) if act == 2 else y).astype(np.float32))

    def act_bwd(self, y, dy, act):
        assert act == 2
        return self._t(np.where(y.numpy() > 0, dy.numpy(), 0.0).astype(np.float32))

    def spmm_bwd(self, idx, ahat, X, dY, need_dx=True, skip_zero=False):
        j = np.maximum(idx.numpy(), 0)
        dA = np.einsum("nf,nkf->nk", dY.numpy().astype(np.float64), X.numpy()[j]) * (idx.numpy() >= 0)
        dX = None
        if need_dx:
            dX = np.zeros(X.shape, np.float64)
            np.add.at(dX, j.reshape(-1), (ahat.numpy()[:, :, None] * dY.numpy()[:, None, :]).reshape(-1, X.shape[1]))
            dX = self._t(dX.astype(np.float32))
        return self._t(dA.astype(np.float32)), dX

    def norm_bwd_da(self, idx, w, rs, dA, row0=0):
        a = 1.0 / np.sqrt(rs.numpy().astype(np.float64))
        n = idx.shape[0]
        j = np.maximum(idx.numpy(), 0)
        g = dA.numpy().astype(np.float64) * w.numpy() * (idx.numpy() >= 0)
        da = np.zeros(rs.shape[0], np.float64)
        da[row0:row0 + n] += (g * a[j]).sum(1)
        np.add.at(da, j.reshape(-1), (g * a[row0:row0 + n, None]).reshape(-1))
        return self._t(da.astype(np.float32))

    def softk_bwd(self, idx, val, k, dA, rs=None, da=None, row0=0, mode=0, normalized=False):
        n, K = idx.shape
        dw = dA.numpy().astype(np.float64)
        j = np.maximum(idx.numpy(), 0)
        if normalized:
            a = 1.0 / np.sqrt(rs.numpy().astype(np.float64))
            ai = a[row0:row0 + n]
            drs = -0.5 * da.numpy()[row0:row0 + n] * ai / rs.numpy()[row0:row0 + n]
            dw = dw * ai[:, None] * a[j] + drs[:, None]
        f, dfdk = self._ramp(K, k.numpy())
        valid = idx.numpy() >= 0
        dval = np.where(valid, dw * f, 0.0) if mode == 0 else np.zeros_like(dw)
        dk = (np.where(valid, dw * (val.numpy() if mode == 0 else 1.0) * dfdk, 0.0)).sum(1)
        return self._t(dval.astype(np.float32)), self._t(dk.astype(np.float32))

    def edge_bwd(self, xp, idx, val, dval, row0=0, t=-0.05, perturb=False):
        X = xp.numpy().astype(np.float64)
        n, K = idx.shape
        out = np.zeros_like(X)
        j = np.maximum(idx.numpy(), 0)
        diff = X[row0:row0 + n, None, :] - X[j]
        dist = np.sqrt((diff ** 2).sum(-1))
        p = np.exp(t * dist)
        g = dval.numpy().astype(np.float64) * (idx.numpy() >= 0)
        dp = g * val.numpy() / (p + 1e-8) if perturb else g
        with np.errstate(divide="ignore", invalid="ignore"):
            dd = np.where(dist > 0, dp * t * p / dist, 0.0)
        e = dd[:, :, None] * diff
        out[row0:row0 + n] += e.sum(1)
        np.add.at(out, j.reshape(-1), -e.reshape(-1, X.shape[1]))
        return self._t(out.astype(np.float32))


def make_inputs(N=150, d=12, h=16):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, d, generator=g)
    deg = 6 + 6 * torch.rand(N, generator=g)
    h2, h4 = h // 2, h // 4
    P = dict(We=torch.randn(h, d, generator=g) * 0.3, be=torch.randn(h, generator=g) * 0.1,
             Wk=torch.randn(h, d, generator=g) * 0.3, bk=torch.randn(h, generator=g) * 0.1,
             W1=torch.randn(h2, h + 1, generator=g) * 0.3, b1=torch.randn(h2, generator=g) * 0.1,
             Wmu=torch.randn(h4, h2, generator=g) * 0.3, bmu=torch.randn(h4, generator=g) * 0.1,
             Wp=torch.randn(1, h4, generator=g) * 0.3, bp=torch.tensor([0.05]), Wc=torch.rand(d, 5, generator=g))
    cot = torch.randn(N, 5, generator=g)
    return x, deg, P, cot


def run_step(x_local, deg, P, cot_local, N, group, x_full=None, noise_mode=2, hybrid=False):
    sys.path.insert(0, ROOT)
    from dgg_amd.parallel import ShardedDGGConv
    layer = ShardedDGGConv(CpuKern(), N, group=group, K=64, noise_mode=noise_mode, seed=(5, 6), x_grad=x_full is None, x_full=x_full,
                           hybrid=hybrid)
    Z = layer.forward(x_local, deg, P)
    g = layer.backward(cot_local, x_local, P)
    return Z, g


def _worker(rank, world, port, ret, replicated=False, N=150, noise_mode=2):
    hybrid = replicated == "hybrid"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from dgg_amd.parallel import shard_bounds
    x, deg, P, cot = make_inputs(N)
    r0, r1, _ = shard_bounds(N, world, rank)
    Z, g = run_step(x[r0:r1].contiguous(), deg, P, cot[r0:r1].contiguous(), N, None, x if replicated else None, noise_mode, hybrid)
    ret[rank] = (r0, r1, Z.numpy(), {k: v.numpy() for k, v in g.items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("replicated", [False, True, "hybrid"])
@pytest.mark.parametrize("world,N,noise_mode", [(2, 150, 2), (3, 151, 2), (2, 150, 5)])
def test_sharded_step_matches_single_process(replicated, world, N, noise_mode):
    """replicated=True: the node features are data present on every rank (no per-step exchange of projections, every rank
    projects all rows); replicated=False: every rank projects its own rows, all-gathers [xp | H] and reduce-scatters the
    partial [dxp | dH]; "hybrid" (bench.py's default for several GPUs): replicated features, every rank projects xp of all rows
    but H of its own rows only -- H is all-gathered asynchronously, the partial dH reduce-scattered asynchronously, dWc formed
    from the own rows.  world 3 with N = 151: uneven shards (51 + 51 + 49), padded collectives.  noise_mode 5: the ranked
    symmetric generator -- a rank's rows take noise from owners on the other rank; keyed on global ids, nothing is exchanged."""
    x, deg, P, cot = make_inputs(N)
    Z1, g1 = run_step(x, deg, P, cot, N, None, None, noise_mode)          # world 1 (no process group)
    if replicated:
        g1.pop("x")
    port = 29500 + os.getpid() % 2000 + {False: 0, True: 1, "hybrid": 23}[replicated] + 2 * world + 7 * noise_mode
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret, replicated, N, noise_mode)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret[0][0] == 0 and ret[world - 1][1] == N and all(ret[r][1] == ret[r + 1][0] for r in range(world - 1))
    Z2 = np.concatenate([ret[r][2] for r in range(world)])
    np.testing.assert_allclose(Z2, Z1.numpy(), rtol=1e-5, atol=1e-6)
    for k in g1:
        if k == "x":
            got = np.concatenate([ret[r][3]["x"] for r in range(world)])
        else:
            got = ret[0][3][k]
            for r in range(1, world):
                np.testing.assert_allclose(ret[r][3][k], got, rtol=0, atol=0)   # all-reduced: identical on every rank
        ref = g1[k].numpy()
        np.testing.assert_allclose(got.reshape(ref.shape), ref, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(ref).max()))


def test_shard_bounds_cover_every_row_once():
    sys.path.insert(0, ROOT)
    from dgg_amd.parallel import shard_bounds
    for N in (1, 7, 64, 100_000, 100_001):
        for world in (1, 2, 3, 8):
            seen = 0
            for r in range(world):
                r0, r1, per = shard_bounds(N, world, r)
                assert r0 == min(seen, N) and r1 >= r0 and r1 - r0 <= per
                seen = r1
            assert seen == N


def random_candidates(N, seed=3, lo=0, hi=12, wide=()):
    """CSR candidate lists with self loops: `lo..hi` neighbours a row, rows in `wide` get 100 (more than the ELL width)"""
    rng = np.random.default_rng(seed)
    deg = rng.integers(lo, hi + 1, N)
    for r in wide:
        deg[r] = min(100, N - 1)
    rows, cols = [], []
    for i in range(N):
        c = np.unique(np.append(rng.choice(N, deg[i], replace=False), i)).astype(np.int32)
        rows.append(np.full(c.shape, i))
        cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    rowptr = np.zeros(N + 1, np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return torch.from_numpy(np.cumsum(rowptr)), torch.from_numpy(cols.astype(np.int32))


@pytest.mark.parametrize("noise_mode,ext", [(0, False), (2, False), (3, False), (2, True)])
def test_edge_list_step_matches_dense_autograd(noise_mode, ext):
    """ShardedDGGConv with edge-list candidates (cand = CSR of in_adj; dgm.py:1613-1614) on the oracle stand-in against torch autograd
    of the dense formulation restricted to the lists the step selected: relu(D^-1/2 A D^-1/2 (x Wc)) with A = score * ramp on the
    selected entries, gradients of every parameter and of x.  ext: the normalised adjacency has a second consumer whose cotangent
    enters the backward as dA_ext (GCN_DGG's second layer)."""
    sys.path.insert(0, ROOT)
    from dgg_amd.parallel import ShardedDGGConv
    N, d, h = 90, 12, 16
    x, deg, P, cot = make_inputs(N, d, h)
    rowptr, col = random_candidates(N, wide=(7,))
    lay = ShardedDGGConv(CpuKern(), N, K=64, noise_mode=noise_mode, seed=(5, 6), x_grad=True, cand=(rowptr, col))
    Z = lay.forward(x, deg, P)
    cotA = torch.randn(N, 64, generator=torch.Generator().manual_seed(9)) if ext else None
    g = lay.backward(cot, x, P, dA_ext=cotA)
    s = lay.saved
    idx = s["idx"].numpy()
    lens = (rowptr[1:] - rowptr[:-1]).numpy()
    assert np.array_equal((idx >= 0).sum(1), np.minimum(lens, 64)), "every candidate of a row is selected (up to the ELL width)"
    for i in (0, 7, 50):
        assert set(idx[i][idx[i] >= 0]) <= set(col[rowptr[i]:rowptr[i + 1]].numpy())
    # dense autograd restatement on the selected pattern, float64
    xd = x.double().requires_grad_(True)
    Pd = {k_: v.double().requires_grad_(True) for k_, v in P.items()}
    lk = lambda t_: torch.where(t_ > 0, t_, 0.01 * t_)  # noqa: E731
    xp, xk = lk(xd @ Pd["We"].T + Pd["be"]), lk(xd @ Pd["Wk"].T + Pd["bk"])
    mu, sd = deg.double().mean(), deg.double().std()
    feat = torch.cat([xk, ((deg.double() - mu) / (sd + 1e-5))[:, None]], 1)
    m = lk(feat @ Pd["W1"].T + Pd["b1"]) @ Pd["Wmu"].T + Pd["bmu"]
    k = torch.relu((m @ Pd["Wp"].reshape(-1) + Pd["bp"][0]) * sd + mu) + 1
    np.testing.assert_allclose(k.detach().numpy(), s["k"].numpy(), rtol=1e-5)
    valid = torch.from_numpy(idx >= 0)
    j = torch.from_numpy(np.maximum(idx, 0)).long()
    dist = ((xp[:, None, :] - xp[j]) ** 2).sum(-1).clamp_min(1e-30).sqrt()
    p = torch.exp(-0.05 * dist)
    sel_p = p if noise_mode == 0 else torch.exp(torch.log(p + 1e-8) + (torch.log(s["val"].double().clamp_min(1e-300)) - torch.log(p + 1e-8)).detach())
    r = torch.arange(64, dtype=torch.float64)[None, :]
    ramp = 1 - 0.5 * (1 + torch.tanh(r - k[:, None]))
    w = torch.where(valid, sel_p * ramp, torch.zeros_like(p))
    np.testing.assert_allclose(w.detach().numpy(), s["w"].numpy(), rtol=2e-5, atol=1e-7)
    a = w.sum(1).rsqrt()
    ahat = a[:, None] * w * a[j]
    H = xd @ Pd["Wc"]
    Zd = torch.relu((ahat[:, :, None] * H[j]).sum(1))
    np.testing.assert_allclose(Zd.detach().numpy(), Z.numpy(), rtol=1e-4, atol=1e-5)
    loss = (Zd * cot.double()).sum()
    if ext:
        loss = loss + (ahat * cotA.double()).sum()
    loss.backward()
    for k_, v in g.items():
        ref = (xd.grad if k_ == "x" else Pd[k_].grad).numpy()
        np.testing.assert_allclose(v.numpy().reshape(ref.shape), ref, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=k_)


@pytest.mark.parametrize("mode_name,noise_mode", [("u-v-deg", 0), ("u-v-deg", 2), ("u-v-deg-dist", 2), ("u-v-A_uv", 0), ("edge_conv", 2)])
def test_edge_list_step_with_edge_mlp_scorer_matches_dense_autograd(mode_name, noise_mode):
    """ShardedDGGConv with an edge-MLP scorer on edge-list candidates (reference dgm.py:1628-1719; the scorer the training script
    defaults to) on the oracle stand-in against torch autograd of the same formulas on the selected entries: every layer parameter, the
    scorer's own terms (Wcat, wdu, wdv, wex, b1, w2, b2) and x."""
    sys.path.insert(0, ROOT)
    from dgg_amd.parallel import ShardedDGGConv
    N, d, h = 90, 12, 16
    x, deg, P, cot = make_inputs(N, d, h)
    rowptr, col = random_candidates(N)
    E = col.shape[0]
    erow = torch.repeat_interleave(torch.arange(N, dtype=torch.int32), (rowptr[1:] - rowptr[:-1]))
    gen = torch.Generator().manual_seed(21)
    hw = h // 2 if mode_name == "edge_conv" else h
    rnd = lambda *sh: torch.randn(*sh, generator=gen) * 0.3  # noqa: E731
    sc = dict(Wcat=rnd(2 * hw, h), wdu=None, wdv=None, wex=None, b1=rnd(hw), w2=rnd(hw), b2=rnd(1), erow=erow, ex_in=None, ex_mode=0, t_ex=0.0,
              act=0 if mode_name == "edge_conv" else 1)
    if mode_name in ("u-v-deg", "u-v-deg-dist"):
        sc.update(wdu=rnd(hw) * 0.1, wdv=rnd(hw) * 0.1)
    if mode_name == "u-v-deg-dist":
        sc.update(wex=rnd(hw), ex_mode=2, t_ex=-1.0)
    if mode_name == "u-v-A_uv":
        sc.update(wex=rnd(hw), ex_mode=1, ex_in=torch.rand(E, generator=gen))
    lay = ShardedDGGConv(CpuKern(), N, K=64, noise_mode=noise_mode, seed=(5, 6), x_grad=True, cand=(rowptr, col))
    lay.scorer = sc
    Z = lay.forward(x, deg, P)
    g = lay.backward(cot, x, P)
    s = lay.saved
    idx, eid = s["idx"].numpy(), s["eid"].numpy()
    # dense autograd restatement on the selected entries, float64
    xd = x.double().requires_grad_(True)
    Pd = {k_: v.double().requires_grad_(True) for k_, v in P.items()}
    Sd = {k_: sc[k_].double().requires_grad_(True) for k_ in ("Wcat", "wdu", "wdv", "wex", "b1", "w2", "b2") if sc[k_] is not None}
    lk = lambda t_: torch.where(t_ > 0, t_, 0.01 * t_)  # noqa: E731
    xp, xk = lk(xd @ Pd["We"].T + Pd["be"]), lk(xd @ Pd["Wk"].T + Pd["bk"])
    mu, sd = deg.double().mean(), deg.double().std()
    feat = torch.cat([xk, ((deg.double() - mu) / (sd + 1e-5))[:, None]], 1)
    m = lk(feat @ Pd["W1"].T + Pd["b1"]) @ Pd["Wmu"].T + Pd["bmu"]
    k = torch.relu((m @ Pd["Wp"].reshape(-1) + Pd["bp"][0]) * sd + mu) + 1
    valid = torch.from_numpy(idx >= 0)
    j = torch.from_numpy(np.maximum(idx, 0)).long()
    AB = xp @ Sd["Wcat"].T
    z = AB[:, None, :hw] + AB[j][:, :, hw:] + Sd["b1"]
    if "wdu" in Sd:
        z = z + deg.double()[:, None, None] * Sd["wdu"] + deg.double()[j][:, :, None] * Sd["wdv"]
    if sc["ex_mode"] == 1:
        z = z + sc["ex_in"].double()[torch.from_numpy(np.maximum(eid, 0)).long()][:, :, None] * Sd["wex"]
    if sc["ex_mode"] == 2:
        dist = ((xp[:, None, :] - xp[j]) ** 2).sum(-1).clamp_min(1e-30).sqrt()
        z = z + torch.exp(sc["t_ex"] * dist)[:, :, None] * Sd["wex"]
    hid = lk(z) if sc["act"] == 1 else z
    p = torch.sigmoid((hid * Sd["w2"]).sum(-1) + Sd["b2"])
    sel_p = p if noise_mode == 0 else torch.exp(torch.log(p + 1e-8) + (torch.log(s["val"].double().clamp_min(1e-300)) - torch.log(p + 1e-8)).detach())
    r = torch.arange(64, dtype=torch.float64)[None, :]
    w = torch.where(valid, sel_p * (1 - 0.5 * (1 + torch.tanh(r - k[:, None]))), torch.zeros_like(p))
    np.testing.assert_allclose(w.detach().numpy(), s["w"].numpy(), rtol=2e-5, atol=1e-7)
    a = w.sum(1).rsqrt()
    ahat = a[:, None] * w * a[j]
    Zd = torch.relu((ahat[:, :, None] * (xd @ Pd["Wc"])[j]).sum(1))
    np.testing.assert_allclose(Zd.detach().numpy(), Z.numpy(), rtol=1e-4, atol=1e-5)
    (Zd * cot.double()).sum().backward()
    gs = g.pop("scorer")
    for k_, v in Sd.items():
        ref = v.grad.numpy()
        np.testing.assert_allclose(gs[k_].numpy().reshape(ref.shape), ref, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg="scorer " + k_)
    for k_, v in g.items():
        ref = (xd.grad if k_ == "x" else Pd[k_].grad).numpy()
        np.testing.assert_allclose(v.numpy().reshape(ref.shape), ref, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=k_)
