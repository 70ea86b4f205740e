"""ctypes front-end of the CPU oracle (oracle/dgg_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product package
(dgg_amd/) never imports it.  All arrays are numpy, C-contiguous; fp32 / int32 / int64.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdgg_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "dgg_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdgg_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.ora_exp.restype = C.c_float
        _lib.ora_exp.argtypes = [C.c_float]
        _lib.ora_log.restype = C.c_float
        _lib.ora_log.argtypes = [C.c_float]
        _lib.ora_tanh.restype = C.c_float
        _lib.ora_tanh.argtypes = [C.c_float]
        _lib.ora_pair_u24.restype = C.c_uint32
        _lib.ora_pair_u24.argtypes = [C.c_uint32] * 4 + [C.c_int]
        _lib.ora_gumbel_u24.restype = C.c_float
        _lib.ora_gumbel_u24.argtypes = [C.c_uint32]
        _lib.ora_noise.restype = C.c_float
        _lib.ora_noise.argtypes = [C.c_uint32] * 4 + [C.c_int]
        _lib.ora_pair_score.restype = C.c_float
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return a.ctypes.data_as(C.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


NOISE_NONE, NOISE_EXPLICIT, NOISE_HASH, NOISE_HASH_SYM, NOISE_RANKED, NOISE_RANKED_SYM = 0, 1, 2, 3, 4, 5
ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2


def exp(x):
    return lib().ora_exp(float(x))


def log(x):
    return lib().ora_log(float(x))


def tanh(x):
    return lib().ora_tanh(float(x))


def noise_matrix(N, s0, s1, symmetric=False, rows=None):
    L = lib()
    rows = range(N) if rows is None else rows
    out = np.empty((len(rows), N), np.float32)
    for a, i in enumerate(rows):
        for j in range(N):
            out[a, j] = L.ora_noise(s0, s1, i, j, int(symmetric))
    return out


def linear(x, W, b=None, act=ACT_NONE, w_layout=0):
    x, W = f32(x), f32(W)
    N, d = x.shape
    out = W.shape[0] if w_layout == 0 else W.shape[1]
    y = np.empty((N, out), np.float32)
    bb = f32(b) if b is not None else None
    lib().ora_linear(_p(x), C.c_int64(N), C.c_int(d), _p(W), _p(bb), C.c_int(out), C.c_int(w_layout), C.c_int(act), _p(y))
    return y


def linear_bwd(x, W, y, dy, act=ACT_NONE, w_layout=0, need_dx=True):
    x, W, y, dy = f32(x), f32(W), f32(y), f32(dy)
    N, d = x.shape
    out = W.shape[0] if w_layout == 0 else W.shape[1]
    dx = np.empty_like(x) if need_dx else None
    dW = np.empty_like(W)
    db = np.empty((out,), np.float32)
    lib().ora_linear_bwd(_p(x), C.c_int64(N), C.c_int(d), _p(W), C.c_int(out), C.c_int(w_layout), C.c_int(act),
                         _p(y), _p(dy), _p(dx), _p(dW), _p(db))
    return dx, dW, db


def degree_stats(deg):
    deg = f32(deg)
    mu, sd = C.c_float(), C.c_float()
    lib().ora_degree_stats(_p(deg), C.c_int64(deg.shape[0]), C.byref(mu), C.byref(sd))
    return np.float32(mu.value), np.float32(sd.value)


def knet_x(xk, deg, mu, sd, W1, b1, Wmu, bmu, Wp, bp, save=False):
    xk, deg = f32(xk), f32(deg)
    N, h = xk.shape
    W1, b1, Wmu, bmu, Wp, bp = map(f32, (W1, b1, Wmu, bmu, Wp, bp))
    h2, h4 = W1.shape[0], Wmu.shape[0]
    k = np.empty((N,), np.float32)
    z = np.empty((N, h2), np.float32) if save else None
    m = np.empty((N, h4), np.float32) if save else None
    u = np.empty((N,), np.float32) if save else None
    lib().ora_knet_x(_p(xk), C.c_int64(N), C.c_int(h), _p(deg), C.c_float(mu), C.c_float(sd), _p(W1), _p(b1), C.c_int(h2),
                     _p(Wmu), _p(bmu), C.c_int(h4), _p(Wp), _p(bp), _p(k), _p(z), _p(m), _p(u))
    return (k, z, m, u) if save else k


def knet_x_bwd(xk, deg, mu, sd, W1, Wmu, Wp, z, m, u, dk):
    xk, deg, W1, Wmu, Wp, z, m, u, dk = map(f32, (xk, deg, W1, Wmu, Wp, z, m, u, dk))
    N, h = xk.shape
    h2, h4 = W1.shape[0], Wmu.shape[0]
    dxk = np.empty_like(xk)
    dW1, db1 = np.empty_like(W1), np.empty((h2,), np.float32)
    dWmu, dbmu = np.empty_like(Wmu), np.empty((h4,), np.float32)
    dWp, dbp = np.empty((h4,), np.float32), np.empty((1,), np.float32)
    lib().ora_knet_x_bwd(_p(xk), C.c_int64(N), C.c_int(h), _p(deg), C.c_float(mu), C.c_float(sd), _p(W1), C.c_int(h2),
                         _p(Wmu), C.c_int(h4), _p(Wp), _p(z), _p(m), _p(u), _p(dk),
                         _p(dxk), _p(dW1), _p(db1), _p(dWmu), _p(dbmu), _p(dWp), _p(dbp))
    return dxk, dW1, db1, dWmu, dbmu, dWp, dbp


def knet_input_deg(deg, dmean, dstd, Wd, bd, Wmu, bmu, Wp, bp):
    deg, Wd, bd, Wmu, bmu, Wp, bp = map(f32, (deg, Wd, bd, Wmu, bmu, Wp, bp))
    N = deg.shape[0]
    k = np.empty((N,), np.float32)
    lib().ora_knet_input_deg(_p(deg), C.c_int64(N), C.c_float(dmean), C.c_float(dstd), _p(Wd), _p(bd), _p(Wmu), _p(bmu),
                             C.c_int(Wmu.shape[0]), _p(Wp), _p(bp), _p(k))
    return k


def allpairs_topk(xp, K=64, t=-0.05, noise_mode=NOISE_NONE, G=None, seed=(0, 0), rows=None):
    xp = f32(xp)
    N, h = xp.shape
    r0, r1 = (0, N) if rows is None else rows
    idx = np.empty((r1 - r0, K), np.int32)
    val = np.empty((r1 - r0, K), np.float32)
    Gc = f32(G) if G is not None else None
    lib().ora_allpairs_topk(_p(xp), C.c_int64(N), C.c_int(h), C.c_int64(r0), C.c_int64(r1), C.c_float(t),
                            C.c_int(noise_mode), _p(Gc), C.c_uint32(seed[0]), C.c_uint32(seed[1]), C.c_int(K), _p(idx), _p(val))
    return idx, val


def edgelist_topk(xp, rowptr, col, K=64, t=-0.05, noise_mode=NOISE_NONE, G=None, seed=(0, 0)):
    xp = f32(xp)
    N, h = xp.shape
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = i32(col)
    idx = np.empty((N, K), np.int32)
    val = np.empty((N, K), np.float32)
    Gc = f32(G) if G is not None else None
    lib().ora_edgelist_topk(_p(xp), C.c_int64(N), C.c_int(h), _p(rowptr), _p(col), C.c_float(t), C.c_int(noise_mode), _p(Gc),
                            C.c_uint32(seed[0]), C.c_uint32(seed[1]), C.c_int(K), _p(idx), _p(val))
    return idx, val


def select_scores(scores, K=64):
    scores = f32(scores)
    R, N = scores.shape
    idx = np.empty((R, K), np.int32)
    val = np.empty((R, K), np.float32)
    lib().ora_select_scores(_p(scores), C.c_int64(R), C.c_int64(N), C.c_int(K), _p(idx), _p(val))
    return idx, val


MODE_K_TIMES, MODE_K_ONLY = 0, 1


def softk(idx, val, k, mode=MODE_K_TIMES):
    idx, val, k = i32(idx), f32(val), f32(k)
    N, K = idx.shape
    w = np.empty((N, K), np.float32)
    rs = np.empty((N,), np.float32)
    lib().ora_softk(_p(idx), _p(val), _p(k), C.c_int64(N), C.c_int(K), C.c_int(mode), _p(w), _p(rs))
    return w, rs


def normalize(idx, w, rs):
    idx, w, rs = i32(idx), f32(w), f32(rs)
    N, K = idx.shape
    ahat = np.empty((N, K), np.float32)
    lib().ora_normalize(_p(idx), _p(w), _p(rs), C.c_int64(N), C.c_int(K), _p(ahat))
    return ahat


def spmm(idx, ahat, X):
    idx, ahat, X = i32(idx), f32(ahat), f32(X)
    N, K = idx.shape
    F = X.shape[1]
    Y = np.empty((N, F), np.float32)
    lib().ora_spmm(_p(idx), _p(ahat), _p(X), C.c_int64(N), C.c_int(K), C.c_int(F), _p(Y))
    return Y


def spmm_bwd(idx, ahat, X, dY, need_dx=True):
    idx, ahat, X, dY = i32(idx), f32(ahat), f32(X), f32(dY)
    N, K = idx.shape
    F = X.shape[1]
    dA = np.empty((N, K), np.float32)
    dX = np.empty_like(X) if need_dx else None
    lib().ora_spmm_bwd(_p(idx), _p(ahat), _p(X), _p(dY), C.c_int64(N), C.c_int(K), C.c_int(F), _p(dA), _p(dX))
    return dA, dX


def softk_norm_bwd(idx, val, k, w, rs, dA, mode=MODE_K_TIMES):
    idx, val, k, w, rs, dA = i32(idx), f32(val), f32(k), f32(w), f32(rs), f32(dA)
    N, K = idx.shape
    dval = np.empty((N, K), np.float32)
    dk = np.empty((N,), np.float32)
    lib().ora_softk_norm_bwd(_p(idx), _p(val), _p(k), _p(w), _p(rs), _p(dA), C.c_int64(N), C.c_int(K), C.c_int(mode),
                             _p(dval), _p(dk))
    return dval, dk


def softk_bwd(idx, val, k, dw, mode=MODE_K_TIMES):
    """ramp backward alone (no normalisation): dw -> dval, dk.  Pure numpy (float64)."""
    N, K = idx.shape
    r = np.arange(K, dtype=np.float32)[None, :]
    th = np.vectorize(tanh)((r - k[:, None]).astype(np.float32)).astype(np.float64)
    f = 1.0 - 0.5 * (1.0 + th)
    dfdk = 0.5 * (1.0 - th * th)
    valid = idx >= 0
    dw = dw.astype(np.float64) * valid
    if mode == MODE_K_TIMES:
        return (dw * f).astype(np.float32), (dw * val * dfdk).sum(1).astype(np.float32)
    return np.zeros_like(val), (dw * dfdk).sum(1).astype(np.float32)


def edge_bwd(xp, idx, val, dval, t=-0.05, perturb=False):
    xp, idx, val, dval = f32(xp), i32(idx), f32(val), f32(dval)
    N, h = xp.shape
    K = idx.shape[1]
    dxp = np.empty_like(xp)
    lib().ora_edge_bwd(_p(xp), C.c_int64(N), C.c_int(h), _p(idx), _p(val), _p(dval), C.c_int(K), C.c_float(t),
                       C.c_int(int(perturb)), _p(dxp))
    return dxp


# ---- edge-MLP scorers on a candidate edge list (dgm.py:1628-1719) ---------------------------
def edge_mlp_fwd(AB, xp, erow, col, deg, ex_in, ex_mode, t_ex, wdu, wdv, wex, b1, w2, b2, act=1):
    """-> p_edge [E], ex [E] (the per-edge extra that was used)"""
    AB, xp = f32(AB), f32(xp)
    N, h = xp.shape
    hw = AB.shape[1] // 2
    erow, col = i32(erow), i32(col)
    E = col.shape[0]
    o = lambda a: None if a is None else f32(a)  # noqa: E731
    deg, ex_in, wdu, wdv, wex, b1, w2 = o(deg), o(ex_in), o(wdu), o(wdv), o(wex), f32(b1), f32(w2)
    p, ex = np.empty(E, np.float32), np.empty(E, np.float32)
    lib().ora_edge_mlp_fwd(_p(AB), _p(xp), C.c_int64(N), C.c_int(h), C.c_int(hw), _p(erow), _p(col), C.c_int64(E), _p(deg),
                           _p(ex_in), C.c_int(ex_mode), C.c_float(t_ex), _p(wdu), _p(wdv), _p(wex), _p(b1), _p(w2),
                           C.c_float(float(b2)), C.c_int(act), _p(p), _p(ex))
    return p, ex


def edgelist_topk_p(p_edge, N, rowptr, col, K=64, noise_mode=NOISE_NONE, G=None, seed=(0, 0)):
    p_edge = f32(p_edge)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = i32(col)
    idx, val, eid = np.empty((N, K), np.int32), np.empty((N, K), np.float32), np.empty((N, K), np.int32)
    Gc = f32(G) if G is not None else None
    lib().ora_edgelist_topk_p(_p(p_edge), C.c_int64(N), _p(rowptr), _p(col), C.c_int(noise_mode), _p(Gc), C.c_uint32(seed[0]),
                              C.c_uint32(seed[1]), C.c_int(K), _p(idx), _p(val), _p(eid))
    return idx, val, eid


def edge_mlp_bwd(AB, idx, eid, val, dval, deg, ex, wdu, wdv, wex, b1, w2, b2, act=1, perturb=False):
    """-> dAB [N,2hw], dpar = [dwdu|dwdv|dwex|db1|dw2|db2], dex [N,K]"""
    AB, idx, eid, val, dval = f32(AB), i32(idx), i32(eid), f32(val), f32(dval)
    N, K = idx.shape
    hw = AB.shape[1] // 2
    o = lambda a: None if a is None else f32(a)  # noqa: E731
    deg, ex, wdu, wdv, wex = o(deg), o(ex), o(wdu), o(wdv), o(wex)
    dAB, dpar, dex = np.empty_like(AB), np.empty(5 * hw + 1, np.float32), np.empty((N, K), np.float32)
    lib().ora_edge_mlp_bwd(_p(AB), C.c_int64(N), C.c_int(hw), _p(idx), _p(eid), _p(val), _p(dval), C.c_int(K), _p(deg), _p(ex),
                           _p(wdu), _p(wdv), _p(wex), _p(f32(b1)), _p(f32(w2)), C.c_float(float(b2)), C.c_int(act),
                           C.c_int(int(perturb)), _p(dAB), _p(dpar), _p(dex))
    return dAB, dpar, dex


# ---- degree-only k-net modes (dgm.py:1492-1526) ---------------------------------------------------------------
def knet_deg(deg, mu, sd, eps, Wd, bd, Wmu, bmu, Wp, bp):
    """-> k [N], u [N] (pre-relu)"""
    deg, Wd, bd, Wmu, bmu, Wp, bp = map(f32, (deg, Wd, bd, Wmu, bmu, Wp, bp))
    N = deg.shape[0]
    k, u = np.empty((N,), np.float32), np.empty((N,), np.float32)
    lib().ora_knet_deg(_p(deg), C.c_int64(N), C.c_float(mu), C.c_float(sd), C.c_float(eps), _p(Wd), _p(bd), _p(Wmu), _p(bmu),
                       C.c_int(Wmu.shape[0]), _p(Wp), _p(bp), _p(k), _p(u))
    return k, u


def knet_deg_bwd(deg, mu, sd, eps, Wd, bd, Wmu, bmu, Wp, u, dk):
    """-> gradients of (Wd [3], bd [3], Wmu [h4,3], bmu [h4], Wp [h4], bp [1]) from the two sums S0, S1"""
    deg, u, dk = f32(deg), f32(u), f32(dk)
    S = np.empty(2, np.float32)
    lib().ora_knet_deg_bwd_sums(_p(deg), C.c_int64(deg.shape[0]), C.c_float(mu), C.c_float(sd), C.c_float(eps), _p(u), _p(dk), _p(S))
    return knet_deg_param_grads(S[0], S[1], *(np.asarray(a, np.float64) for a in (Wd, bd, Wmu, bmu, Wp)))


def knet_deg_param_grads(S0, S1, Wd, bd, Wmu, bmu, Wp):
    alpha, beta, gamma = Wmu @ Wd, Wmu @ bd + bmu, Wmu.T @ Wp        # m = alpha nd + beta; d in3 = gamma dkp
    f = lambda a: np.asarray(a, np.float32)  # noqa: E731
    return (f(gamma * S1), f(gamma * S0), f(np.outer(Wp, Wd * S1 + bd * S0)), f(Wp * S0), f(alpha * S1 + beta * S0),
            f([S0]))


# ---- CSR adjacency + the `DGG` class "for ICLR" (dgm.py:1730-1815; GCN_DGG_00 model.py:1314-1433) ---------------
def _rp(rowptr):
    return np.ascontiguousarray(rowptr, dtype=np.int64)


def csr_row_sum(vals, rowptr):
    vals, rowptr = f32(vals), _rp(rowptr)
    rs = np.empty(rowptr.shape[0] - 1, np.float32)
    lib().ora_csr_row_sum(_p(vals), _p(rowptr), C.c_int64(rs.shape[0]), _p(rs))
    return rs


def csr_normalize(rowptr, col, w, rs):
    rowptr, col, w, rs = _rp(rowptr), i32(col), f32(w), f32(rs)
    ahat = np.empty_like(w)
    lib().ora_csr_normalize(_p(rowptr), _p(col), _p(w), _p(rs), C.c_int64(rs.shape[0]), _p(ahat))
    return ahat


def csr_spmm(rowptr, col, a, X):
    rowptr, col, a, X = _rp(rowptr), i32(col), f32(a), f32(X)
    N, F = X.shape
    Y = np.empty((rowptr.shape[0] - 1, F), np.float32)
    lib().ora_csr_spmm(_p(rowptr), _p(col), _p(a), _p(X), C.c_int64(Y.shape[0]), C.c_int(F), _p(Y))
    return Y


def csr_spmm_bwd(rowptr, col, a, X, dY, need_dx=True):
    rowptr, col, a, X, dY = _rp(rowptr), i32(col), f32(a), f32(X), f32(dY)
    dA = np.empty_like(a)
    dX = np.empty_like(X) if need_dx else None
    lib().ora_csr_spmm_bwd(_p(rowptr), _p(col), _p(a), _p(X), _p(dY), C.c_int64(X.shape[0]), C.c_int(X.shape[1]), _p(dA), _p(dX))
    return dA, dX


def csr_norm_bwd(rowptr, col, w, rs, dA):
    rowptr, col, w, rs, dA = _rp(rowptr), i32(col), f32(w), f32(rs), f32(dA)
    dw = np.empty_like(w)
    lib().ora_csr_norm_bwd(_p(rowptr), _p(col), _p(w), _p(rs), _p(dA), C.c_int64(rs.shape[0]), _p(dw))
    return dw


def csr_rank_ramp(p, rowptr, col, w, b):
    """-> out [E], S [N], k [N], pos [E]"""
    p, rowptr, col = f32(p), _rp(rowptr), i32(col)
    N = rowptr.shape[0] - 1
    out, S, k, pos = np.empty_like(p), np.empty(N, np.float32), np.empty(N, np.float32), np.empty(p.shape[0], np.int32)
    lib().ora_csr_rank_ramp(_p(p), _p(rowptr), _p(col), C.c_int64(N), C.c_float(float(w)), C.c_float(float(b)), _p(out), _p(S),
                            _p(k), _p(pos))
    return out, S, k, pos


def csr_rank_ramp_bwd(p, rowptr, w, b, S, k, pos, g):
    """-> dp [E], dkz [N]"""
    p, rowptr, S, k, pos, g = f32(p), _rp(rowptr), f32(S), f32(k), i32(pos), f32(g)
    dp, dkz = np.empty_like(p), np.empty_like(S)
    lib().ora_csr_rank_ramp_bwd(_p(p), _p(rowptr), C.c_int64(S.shape[0]), C.c_float(float(w)), C.c_float(float(b)), _p(S), _p(k),
                                _p(pos), _p(g), _p(dp), _p(dkz))
    return dp, dkz


def csr_noisy_sigmoid(p, noise):
    p, noise = f32(p), f32(noise)
    out = np.empty_like(p)
    lib().ora_csr_noisy_sigmoid(_p(p), _p(noise), C.c_int64(p.shape[0]), _p(out))
    return out


def csr_rank_cut(p, rowptr, col, kcut):
    """-> out [E], pos [E]"""
    p, rowptr, col = f32(p), _rp(rowptr), i32(col)
    out, pos = np.empty_like(p), np.empty(p.shape[0], np.int32)
    lib().ora_csr_rank_cut(_p(p), _p(rowptr), _p(col), C.c_int64(rowptr.shape[0] - 1), C.c_int(int(kcut)), _p(out), _p(pos))
    return out, pos


def csr_uvdist(xp, rowptr, col, t=-0.05):
    xp, rowptr, col = f32(xp), _rp(rowptr), i32(col)
    p = np.empty(col.shape[0], np.float32)
    lib().ora_csr_uvdist(_p(xp), _p(rowptr), _p(col), C.c_int64(xp.shape[0]), C.c_int(xp.shape[1]), C.c_float(t), _p(p))
    return p


def csr_uvdist_bwd(xp, rowptr, col, p, dp, t=-0.05):
    xp, rowptr, col, p, dp = f32(xp), _rp(rowptr), i32(col), f32(p), f32(dp)
    dxp = np.empty_like(xp)
    lib().ora_csr_uvdist_bwd(_p(xp), _p(rowptr), _p(col), C.c_int64(xp.shape[0]), C.c_int(xp.shape[1]), C.c_float(t), _p(p), _p(dp),
                             _p(dxp))
    return dxp


def edge_mlp_bwd_csr(AB, rowptr, col, dval, b1, w2, b2, act=1):
    AB, rowptr, col, dval = f32(AB), _rp(rowptr), i32(col), f32(dval)
    hw = AB.shape[1] // 2
    dAB, dpar = np.empty_like(AB), np.empty(5 * hw + 1, np.float32)
    lib().ora_edge_mlp_bwd_csr(_p(AB), C.c_int64(AB.shape[0]), C.c_int(hw), _p(rowptr), _p(col), _p(dval), _p(f32(b1)), _p(f32(w2)),
                               C.c_float(float(b2)), C.c_int(act), _p(dAB), _p(dpar))
    return dAB, dpar


# ---- dense all-pairs alternates (DGG_LearnableK_SDD / DGG_StraightThrough, dgm.py:103-351) ----------------------------
def dense_rows_fwd(xq, t, temp, ramp, k=None, kfix=0, hs_start=2.0, interval=7.0, hard=False):
    """xq [B,N,h] -> out, y [B,N,N] fp32, pos [B,N,N] int32"""
    xq = f32(xq)
    B, N, h = xq.shape
    out, y, pos = np.empty((B, N, N), np.float32), np.empty((B, N, N), np.float32), np.empty((B, N, N), np.int32)
    kk = f32(k).reshape(-1) if k is not None else None
    lib().ora_dense_rows_fwd(_p(xq), C.c_int(B), C.c_int64(N), C.c_int(h), C.c_float(float(t)), C.c_float(float(temp)), C.c_int(ramp),
                             _p(kk) if kk is not None else None, C.c_int(int(kfix)), C.c_float(hs_start), C.c_float(interval),
                             C.c_int(int(hard)), _p(out), _p(y), _p(pos))
    return out, y, pos


def dense_rows_bwd(xq, t, temp, ramp, k, hs_start, interval, y, pos, g):
    """-> Cm [B,N,N], dk [B,N] (ramp 0) or None, dt (scalar, float64 sum of the row terms)"""
    xq, y, pos, g = f32(xq), f32(y), i32(pos), f32(g)
    B, N, h = xq.shape
    Cm, dt = np.empty((B, N, N), np.float32), np.empty(B * N, np.float32)
    kk = f32(k).reshape(-1) if k is not None else None
    dk = np.empty(B * N, np.float32) if ramp == 0 else None
    lib().ora_dense_rows_bwd(_p(xq), C.c_int(B), C.c_int64(N), C.c_int(h), C.c_float(float(t)), C.c_float(float(temp)), C.c_int(ramp),
                             _p(kk) if kk is not None else None, C.c_float(hs_start), C.c_float(interval), _p(y), _p(pos), _p(g),
                             _p(Cm), _p(dk) if dk is not None else None, _p(dt))
    return Cm, (dk.reshape(B, N) if dk is not None else None), float(dt.astype(np.float64).sum())


def dense_pairs_dx(xq, Cm):
    xq, Cm = f32(xq), f32(Cm)
    B, N, h = xq.shape
    dx = np.empty_like(xq)
    lib().ora_dense_pairs_dx(_p(xq), C.c_int(B), C.c_int64(N), C.c_int(h), _p(Cm), _p(dx))
    return dx


def feat_softmax(z):
    z = f32(z)
    out = np.empty_like(z)
    lib().ora_feat_softmax(_p(z), C.c_int64(z.size // z.shape[-1]), C.c_int(z.shape[-1]), _p(out))
    return out


def feat_softmax_bwd(out, g):
    out, g = f32(out), f32(g)
    dz = np.empty_like(out)
    lib().ora_feat_softmax_bwd(_p(out), _p(g), C.c_int64(out.size // out.shape[-1]), C.c_int(out.shape[-1]), _p(dz))
    return dz
