"""Differentiable Graph Generator modules with the reference's names, signatures and state_dict keys.

`DGG_LearnableK_debug(in_dim, latent_dim, args).forward(x, in_adj, noise, writer, epoch)` mirrors reference
dgm.py:1083, 1178 -- the live class behind GCN_DGG / SAGE_DGG / GCNII_DGG / GCNIIppi_DGG (model.py:133, 666, 910,
1198) -- but runs on the HIP kernels of libdgg_hip.so and returns an `EllAdjacency` (which still offers
`.to_dense()` / `.to_sparse()`) instead of a sparse COO built from dense [N,N] tensors.

Parameters are created in the reference's order with the reference's layer types, so `state_dict()` has the
same 34 keys/shapes (reference checkpoints load with strict=True) and the same default initialisation under
the same torch seed.  Parameters that the reference registers but never uses (t, k_W, ...) are kept for that reason only.

Edge scorers: `u-v-dist` (all-pairs or edge-list candidates) and the edge-MLP family on edge-list candidates
(`u-v-deg` -- the reference's default, train_small_graphs.py:184-191 -- `u-v-A_uv`, `u-v-deg-dist`, `edge_conv`, `A_uv`;
reference dgm.py:1628-1725).
"""
import math
import weakref

import torch
import torch.nn as nn

from . import ops
from .adjacency import AllPairs, CsrAdjacency, EllAdjacency, _cached, csr_candidates, csr_pattern

_EDGE_MLP_MODES = ("u-v-A_uv", "u-v-deg", "u-v-deg-dist", "edge_conv", "A_uv")


def _capturing():
    """a hipGraph is being captured on the current stream (nothing can be read back, no host decision can depend on device data)"""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


class LearnableKEncoder(nn.Module):
    """k_mu / k_logvar / k_project head (reference dgm.py:2029-2063); deterministic branch on the GPU path."""

    def __init__(self, in_dim, latent_dim, learn_k_bias=False, args=None):
        super().__init__()
        self.learn_k_bias = learn_k_bias
        self.k_mu = nn.Linear(in_dim, latent_dim)
        self.k_logvar = nn.Linear(in_dim, latent_dim)
        self.k_project = nn.Linear(latent_dim, 1)
        self.args = args


class _KnetFeatFn(torch.autograd.Function):
    """k_estimate_net after the node encoder (reference dgm.py:1566-1586, shared by modes "x" and "gcn-x-deg"):
    per-node features xk [N,h], prior degree -> learned k [N]."""

    @staticmethod
    def forward(ctx, xk, deg, W1, b1, Wmu, bmu, Wp, bp, bf16=False):
        mu_sd = ops.degree_stats(deg)
        ctx.h = xk.shape[1]
        ctx.bf16 = bool(bf16)                  # wide latents only: the two wide products on the bf16 matrix cores (module.gemm_dtype)
        ctx.mfma = ctx.h in ops.KNET_MFMA_WIDTHS and W1.shape[0] * 2 == ctx.h and Wmu.shape[0] * 4 == ctx.h
        if ctx.mfma:        # matrix-core k-net: only u is saved, the backward re-runs layer 1 from xk (same bits for k)
            k, u = ops.knet_x_fwd_slim(xk, deg, mu_sd, W1, b1, Wmu, bmu, Wp.reshape(-1), bp)
            ctx.save_for_backward(W1, Wmu, bmu, Wp, mu_sd, u, xk, deg, b1)
            return k
        k, z, u, feat = ops.knet_x_fwd(xk, deg, mu_sd, W1, b1, Wmu, bmu, Wp.reshape(-1), bp, bf16=ctx.bf16)
        ctx.save_for_backward(W1, Wmu, bmu, Wp, mu_sd, z, u, feat)
        return k

    @staticmethod
    def backward(ctx, dk):
        if ctx.mfma:
            W1, Wmu, bmu, Wp, mu_sd, u, xk, deg, b1 = ctx.saved_tensors
            dxk, dW1, db1, dWmu, dbmu, dWp, dbp = ops.knet_x_bwd_fused(xk, deg, mu_sd, W1, b1, Wmu, bmu, Wp.reshape(-1), u, dk.contiguous())
            return dxk, None, dW1, db1, dWmu, dbmu, dWp.reshape(Wp.shape), dbp, None
        W1, Wmu, bmu, Wp, mu_sd, z, u, feat = ctx.saved_tensors
        dxk, dW1, db1, dWmu, dbmu, dWp, dbp = ops.knet_x_bwd(ctx.h, mu_sd, W1, Wmu, bmu, Wp.reshape(-1), z, u, feat, dk.contiguous(), bf16=ctx.bf16)
        return dxk, None, dW1, db1, dWmu, dbmu, dWp.reshape(Wp.shape), dbp, None


class _KnetDegFn(torch.autograd.Function):
    """Degree-only k-net modes (reference dgm.py:1492-1526): "input_deg" (constants args.deg_mean / deg_std, eps 1e-5)
    and "learn_normalized_degree" (batch statistics, no eps).  The net is affine in the scalar nd_i, so the backward is
    two reductions (S0 = sum dkp, S1 = sum dkp nd; HIP) followed by parameter-sized algebra."""

    @staticmethod
    def forward(ctx, deg, Wd, bd, Wmu, bmu, Wp, bp, consts):
        mu_sd = ops.degree_stats(deg) if consts is None else None
        dmean, dstd, eps = (0.0, 0.0, 0.0) if consts is None else (consts[0], consts[1], 1e-5)
        k, u = ops.knet_deg_fwd(deg, mu_sd, dmean, dstd, eps, Wd.reshape(-1), bd, Wmu, bmu, Wp.reshape(-1), bp)
        ctx.consts = (dmean, dstd, eps)
        ctx.mu_sd = mu_sd
        ctx.save_for_backward(deg, Wd, bd, Wmu, bmu, Wp, u)
        return k

    @staticmethod
    def backward(ctx, dk):
        deg, Wd, bd, Wmu, bmu, Wp, u = ctx.saved_tensors
        dmean, dstd, eps = ctx.consts
        S = ops.knet_deg_bwd_sums(deg, ctx.mu_sd, dmean, dstd, eps, u, dk.contiguous())
        S0, S1 = S[0], S[1]
        wd, wp = Wd.reshape(-1), Wp.reshape(-1)
        alpha, beta, gamma = Wmu @ wd, Wmu @ bd + bmu, Wmu.t() @ wp       # m = alpha nd + beta; d in3 = gamma dkp
        return (None, (gamma * S1).reshape(Wd.shape), gamma * S0, torch.outer(wp, wd * S1 + bd * S0), wp * S0,
                (alpha * S1 + beta * S0).reshape(Wp.shape), S0.reshape(1), None)


class _DualProjFn(torch.autograd.Function):
    """The two input projections of the generator on ONE pass over x: xp = leaky(x We^T + be) (node_encode_for_edges, dgm.py:1097-1100)
    and xk = leaky(x Wk^T + bk) (node_encode_for_k, 1123-1126).  x is the widest tensor of the edge-list configurations (Pubmed: 500
    columns, Cora: 1433) and two separate layers read it twice forward and twice backward.  Forward: dgg_linear_fwd_multi (the same
    fmaf chains as one dgg_linear_fwd per layer: bit-identical); backward: the two activation derivatives, then ONE product
    [dxp' | dxk']^T x for both weight gradients."""

    @staticmethod
    def forward(ctx, x, We, be, Wk, bk):
        xp, xk = ops.linear_fwd_multi(x, [(We, be, ops.ACT_LEAKY, 0), (Wk, bk, ops.ACT_LEAKY, 0)])
        ctx.save_for_backward(x, We, Wk, xp, xk)
        return xp, xk

    @staticmethod
    def backward(ctx, dxp, dxk):
        x, We, Wk, xp, xk = ctx.saved_tensors
        h1, h2 = We.shape[0], Wk.shape[0]
        parts = [torch.zeros_like(y) if g is None else ops.act_bwd(y, g.contiguous(), ops.ACT_LEAKY) for y, g in ((xp, dxp), (xk, dxk))]
        dP = torch.cat(parts, 1)
        dW, db = ops.gemm_tn(dP, x, colsum=True)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_fwd(dP, torch.cat([We, Wk], 0), None, ops.ACT_NONE, 1)          # dP [N, h1+h2] @ [h1+h2, d]
        return dx, dW[:h1], db[:h1], dW[h1:], db[h1:]


class _DGGSoftAdjXpFn(torch.autograd.Function):
    """_DGGSoftAdjFn on an already projected xp (see _DualProjFn): u-v-dist scoring + perturbation + top-K + ramp; the gradient
    goes back to xp."""

    @staticmethod
    def forward(ctx, xp, k, cfg):
        if cfg["cand"] is None:
            idx, val = ops.allpairs_topk(xp, cfg["K"], cfg["t"], cfg["noise_mode"], cfg["G"], cfg["seed"], algo=cfg["algo"],
                                         k_limit=k, status=cfg)
        else:
            rowptr, col = cfg["cand"]
            idx, val = ops.edgelist_topk(xp, rowptr, col, cfg["K"], cfg["t"], cfg["noise_mode"], cfg["G"], cfg["seed"])
        w, rs = ops.softk_fwd(idx, val, k, cfg.get("fwd_mode", cfg["mode"]))
        ctx.cfg = cfg
        cfg["part"] = ops.part_build(idx, w, xp.shape[0]) if cfg.get("want_bwd", any(ctx.needs_input_grad)) else None
        ctx.save_for_backward(xp, k, idx, val)
        ctx.mark_non_differentiable(idx, val, rs)
        return w, idx, val, rs

    @staticmethod
    def backward(ctx, dw, *_):
        xp, k, idx, val = ctx.saved_tensors
        cfg = ctx.cfg
        fused = ops.softk_edge_bwd(xp, idx, val, k, dw.contiguous(), t=cfg["t"], perturb=cfg["noise_mode"] != ops.NOISE_NONE,
                                   mode=cfg["mode"], normalized=False, part=cfg.get("part"))
        if fused is not None:
            dxp, dk, _ = fused
        else:
            dval, dk = ops.softk_bwd(idx, val, k, dw.contiguous(), mode=cfg["mode"], normalized=False)
            dxp = ops.edge_bwd(xp, idx, val, dval, t=cfg["t"], perturb=cfg["noise_mode"] != ops.NOISE_NONE, part=cfg.get("part"))
        return dxp, dk, None


class _DGGWideAdjFn(torch.autograd.Function):
    """_DGGSoftAdjXpFn on CHUNKED rows: all-pairs candidates under the ranked noise generator whose learned degrees outgrew the 64-rank
    list.  The reference ramps over the whole dense row with an unbounded learned degree (dgm.py:1402-1421, 1580-1584); here row i
    keeps its ceil(k_i + 8.5) + 1 ranks in chunks of 64 (cfg["layout"], ops.chunk_layout).  Backward: the row-major cotangent is
    brought into the record order of the payload partition and the score backward runs by rows and by destination (no atomics)."""

    @staticmethod
    def forward(ctx, xp, k, cfg):
        lay = cfg["layout"]
        idx, val, w, rs = ops.allpairs_topk_wide(xp, k, lay, cfg.get("fwd_mode", cfg["mode"]), cfg["t"], cfg["seed"],
                                                 noise_mode=cfg.get("wide_noise", ops.NOISE_RANKED))
        ctx.cfg = cfg
        cfg["partp"] = ops.partp_build(idx, w, val, rs, xp.shape[0], layout=lay) if cfg.get("want_bwd", any(ctx.needs_input_grad)) else None
        ctx.save_for_backward(xp, k, idx, val)
        ctx.mark_non_differentiable(idx, val, rs)
        return w, idx, val, rs

    @staticmethod
    def backward(ctx, dw, *_):
        xp, k, idx, val = ctx.saved_tensors
        cfg = ctx.cfg
        partp = cfg["partp"]
        assert partp is not None, "chunked rows: no payload partition for this shape"
        dw = dw.contiguous()
        dxp, dk = ops.softk_edge_bwd_p(xp, idx, val, k, dw, ops.partp_gather(partp, dw), None, None, 0, cfg["t"],
                                       cfg.get("wide_noise", ops.NOISE_RANKED) != ops.NOISE_NONE, cfg["mode"], False, partp)
        return dxp, dk, None


class _DGGSoftAdjFn(torch.autograd.Function):
    """x, learned k, projection parameters -> soft (unnormalised) ELL adjacency values: projection, u-v-dist scoring +
    perturbation + top-K, ramp (reference dgm.py:1197-1292)."""

    @staticmethod
    def forward(ctx, x, k, We, be, cfg):
        xp = ops.linear_fwd(x, We, be, ops.ACT_LEAKY)
        if cfg["cand"] is None:
            idx, val = ops.allpairs_topk(xp, cfg["K"], cfg["t"], cfg["noise_mode"], cfg["G"], cfg["seed"], algo=cfg["algo"],
                                         k_limit=k, status=cfg)
        else:
            rowptr, col = cfg["cand"]
            idx, val = ops.edgelist_topk(xp, rowptr, col, cfg["K"], cfg["t"], cfg["noise_mode"], cfg["G"], cfg["seed"])
        w, rs = ops.softk_fwd(idx, val, k, cfg.get("fwd_mode", cfg["mode"]))
        ctx.cfg = cfg
        # destination-ordered partition of the active entries: the column-side terms of the backward run on it instead of
        # entry-wise float atomics (built only when a backward can follow)
        cfg["part"] = ops.part_build(idx, w, x.shape[0]) if cfg.get("want_bwd", any(ctx.needs_input_grad)) else None
        ctx.save_for_backward(x, We, xp, k, idx, val)
        ctx.mark_non_differentiable(idx, val, rs)
        return w, idx, val, rs

    @staticmethod
    def backward(ctx, dw, *_):
        x, We, xp, k, idx, val = ctx.saved_tensors
        cfg = ctx.cfg
        fused = ops.softk_edge_bwd(xp, idx, val, k, dw.contiguous(), t=cfg["t"], perturb=cfg["noise_mode"] != ops.NOISE_NONE,
                                   mode=cfg["mode"], normalized=False, part=cfg.get("part"))
        if fused is not None:
            dxp, dk, _ = fused
        else:
            dval, dk = ops.softk_bwd(idx, val, k, dw.contiguous(), mode=cfg["mode"], normalized=False)
            dxp = ops.edge_bwd(xp, idx, val, dval, t=cfg["t"], perturb=cfg["noise_mode"] != ops.NOISE_NONE, part=cfg.get("part"))
        dx, dWe, dbe = ops.linear_bwd(x, We, xp, dxp, ops.ACT_LEAKY, need_dx=ctx.needs_input_grad[0])
        return dx, dk, dWe, dbe, None


class _DGGEdgeMlpAdjFn(torch.autograd.Function):
    """Same generator with an edge-MLP scorer on edge-list candidates (reference dgm.py:1628-1725).  The first MLP layer
    is split into per-node products AB = xp [Wa | Wb]^T (MFMA GEMM) and per-edge terms (include/dgg_hip.h,
    dgg_edge_mlp_fwd); the mode-specific slicing of the reference's parameters into (Wcat, wdu, wdv, wex, b1, w2, b2)
    happens in differentiable torch ops on the (tiny) parameter tensors, outside this node."""

    @staticmethod
    def forward(ctx, x, k, deg, ex_in, We, be, Wcat, wdu, wdv, wex, eb1, w2, b2, cfg):
        xp = ops.linear_fwd(x, We, be, ops.ACT_LEAKY)
        rowptr, col, erow = cfg["cand"]
        AB = ops.linear_fwd(xp, Wcat, None, ops.ACT_NONE)
        sdeg = deg if wdu is not None else None
        p_edge, ex = ops.edge_mlp_fwd(AB, xp, erow, col, sdeg, ex_in, cfg["ex_mode"], cfg["t_ex"], wdu, wdv, wex, eb1, w2, b2,
                                      cfg["act"])
        idx, val, eid = ops.edgelist_topk_p(p_edge, x.shape[0], rowptr, col, cfg["K"], cfg["noise_mode"], cfg["G"], cfg["seed"])
        w, rs = ops.softk_fwd(idx, val, k, cfg.get("fwd_mode", cfg["mode"]))
        ctx.cfg = cfg
        # optional tensors (None allowed): kept outside save_for_backward, detached
        ctx.opt = tuple(None if t_ is None else t_.detach() for t_ in (sdeg, ex, wdu, wdv, wex))
        ctx.save_for_backward(x, We, xp, k, idx, val, eid, AB, Wcat, eb1, w2, b2)
        ctx.mark_non_differentiable(idx, val, rs)
        return w, idx, val, rs

    @staticmethod
    def backward(ctx, dw, *_):
        x, We, xp, k, idx, val, eid, AB, Wcat, eb1, w2, b2 = ctx.saved_tensors
        sdeg, ex, wdu, wdv, wex = ctx.opt
        cfg = ctx.cfg
        hw = Wcat.shape[0] // 2
        dval, dk = ops.softk_bwd(idx, val, k, dw.contiguous(), mode=cfg["mode"], normalized=False)
        dAB, dpar, dex = ops.edge_mlp_bwd(AB, idx, eid, val, dval, sdeg, ex, wdu, wdv, wex, eb1, w2, b2, cfg["act"],
                                          cfg["noise_mode"] != ops.NOISE_NONE, need_dex=cfg["ex_mode"] == 2)
        dxp, dWcat, _ = ops.linear_bwd(xp, Wcat, AB, dAB, ops.ACT_NONE, need_dx=True, need_db=False)
        if cfg["ex_mode"] == 2:                          # exp(t ||xp_u - xp_v||) also depends on the projection
            dxp = dxp + ops.edge_bwd(xp, idx, val, dex, t=cfg["t_ex"], perturb=False)
        dx, dWe, dbe = ops.linear_bwd(x, We, xp, dxp, ops.ACT_LEAKY, need_dx=ctx.needs_input_grad[0])
        g = lambda t_, a, b: None if t_ is None else dpar[a:b]  # noqa: E731
        return (dx, dk, None, None, dWe, dbe, dWcat if ctx.needs_input_grad[6] else None, g(wdu, 0, hw), g(wdv, hw, 2 * hw),
                g(wex, 2 * hw, 3 * hw), dpar[3 * hw:4 * hw], dpar[4 * hw:5 * hw], dpar[5 * hw:5 * hw + 1], None)


class _DGGScoresFn(torch.autograd.Function):
    """Raw edge probabilities on the stored entries of in_adj, CSR order (reference edge_prob_net, dgm.py:1607-1725), for the
    forward variants that return them as the adjacency: debug_step 0 / 1 (dgm.py:1202-1209, 1240-1246) and the k-select mode
    `edge_p-cdf`, which scatters the unsorted probabilities back (dgm.py:1400)."""

    @staticmethod
    def forward(ctx, x, deg, ex_in, We, be, Wcat, wdu, wdv, wex, eb1, w2, b2, cfg):
        xp = ops.linear_fwd(x, We, be, ops.ACT_LEAKY)
        rowptr, col, erow = cfg["cand"]
        ctx.cfg = cfg
        if Wcat is None:                                  # u-v-dist
            p = ops.csr_uvdist_fwd(xp, rowptr, col, cfg["t"])
            ctx.opt = ()
            ctx.save_for_backward(x, We, xp, p)
            return p
        AB = ops.linear_fwd(xp, Wcat, None, ops.ACT_NONE)
        sdeg = deg if wdu is not None else None
        p, ex = ops.edge_mlp_fwd(AB, xp, erow, col, sdeg, ex_in, cfg["ex_mode"], cfg["t_ex"], wdu, wdv, wex, eb1, w2, b2, cfg["act"])
        ctx.opt = tuple(None if t_ is None else t_.detach() for t_ in (sdeg, ex, wdu, wdv, wex))
        ctx.save_for_backward(x, We, xp, p, AB, Wcat, eb1, w2, b2)
        return p

    @staticmethod
    def backward(ctx, dp):
        cfg = ctx.cfg
        rowptr, col, _ = cfg["cand"]
        dp = dp.contiguous()
        if not ctx.opt:
            x, We, xp, p = ctx.saved_tensors
            dxp = ops.csr_uvdist_bwd(xp, rowptr, col, p, dp, cfg["t"])
            dx, dWe, dbe = ops.linear_bwd(x, We, xp, dxp, ops.ACT_LEAKY, need_dx=ctx.needs_input_grad[0])
            return (dx, None, None, dWe, dbe) + (None,) * 8
        x, We, xp, p, AB, Wcat, eb1, w2, b2 = ctx.saved_tensors
        sdeg, ex, wdu, wdv, wex = ctx.opt
        hw = Wcat.shape[0] // 2
        dAB, dpar, dex = ops.edge_mlp_bwd(AB, col, None, p, dp, sdeg, ex, wdu, wdv, wex, eb1, w2, b2, cfg["act"], False,
                                          need_dex=cfg["ex_mode"] == 2, rowptr=rowptr)
        dxp, dWcat, _ = ops.linear_bwd(xp, Wcat, AB, dAB, ops.ACT_NONE, need_dx=True, need_db=False)
        if cfg["ex_mode"] == 2:                          # exp(t ||xp_u - xp_v||) also depends on the projection
            dxp = dxp + ops.csr_uvdist_bwd(xp, rowptr, col, ex, dex, cfg["t_ex"])
        dx, dWe, dbe = ops.linear_bwd(x, We, xp, dxp, ops.ACT_LEAKY, need_dx=ctx.needs_input_grad[0])
        g = lambda t_, a, b: None if t_ is None else dpar[a:b]  # noqa: E731
        return (dx, None, None, dWe, dbe, dWcat if ctx.needs_input_grad[5] else None, g(wdu, 0, hw), g(wdv, hw, 2 * hw),
                g(wex, 2 * hw, 3 * hw), dpar[3 * hw:4 * hw], dpar[4 * hw:5 * hw], dpar[5 * hw:5 * hw + 1], None)


_PAD_CACHE = {}
_EDGE_MLP_FUSED = ("u-v-deg", "u-v-A_uv", "u-v-deg-dist", "edge_conv", "A_uv")     # edge-MLP scorers the fused layer covers


def _pad_features(x, params, keys):
    """Feature counts that are not a multiple of 4 (Cora 1 433, Citeseer 3 703) keep the 16-byte loads and the one-pass weight-gradient
    kernel away from x: the fused layer then works on a zero-padded copy of x (the features are data: cached per tensor object and
    version) and zero-padded input weights -- the extra columns contribute exact zeros at the END of every k-ordered chain, so every
    result keeps its bits -- and slices the gradients back.  -> (x_eff, params_eff, d) with d = 0 when nothing was padded."""
    d = x.shape[1]
    pad = (-d) % 4
    if pad == 0 or not PAD_ODD_FEATURES:
        return x, params, 0
    key = id(x)
    ent = _PAD_CACHE.get(key)
    if ent is None or ent[0]() is not x or ent[1] != x._version or x.requires_grad:
        xpad = torch.nn.functional.pad(x.detach(), (0, pad))
        if not x.requires_grad and not _capturing():
            for k_ in [k_ for k_, v in _PAD_CACHE.items() if v[0]() is None]:
                del _PAD_CACHE[k_]
            _PAD_CACHE[key] = (weakref.ref(x), x._version, xpad)
    else:
        xpad = ent[2]
    out = []
    for k_, p_ in zip(keys, params):
        if k_ in ("We", "Wk"):
            p_ = torch.nn.functional.pad(p_.detach(), (0, pad))             # [h, d] -> [h, d + pad]
        elif k_ == "Wc":
            p_ = torch.nn.functional.pad(p_.detach(), (0, 0, 0, pad))       # [d, out] -> [d + pad, out]
        out.append(p_)
    return xpad, tuple(out), d


def _unpad_grads(g, keys, d):
    if d:
        for k_ in ("We", "Wk"):
            g[k_] = g[k_].reshape(-1, g[k_].shape[-1])[:, :d]
        g["Wc"] = g["Wc"][:d]
        if g.get("x") is not None:
            g["x"] = g["x"][:, :d]
    return g


PAD_ODD_FEATURES = __import__("os").environ.get("DGG_PAD_ODD_FEATURES", "1") != "0"


class _FusedDGGConvFn(torch.autograd.Function):
    """generator -> normalize_adj -> relu(A (x Wc)) as ONE autograd node on the hand-scheduled step of dgg_amd.parallel.ShardedDGGConv
    (reference dgm.py:1178-1292, model.py:1205-1219, 580-599): x is read ONCE for the three projections [xp | xk | x Wc] and once for
    their weight gradients, the ramp is applied inside the search, normalisation is folded into the partition of the active entries,
    and none of the intermediate adjacency tensors is an autograd edge (no per-edge tensors are allocated, filled or concatenated
    by torch between the kernels).  Same kernels and bits as the separate modules wherever those run the same kernels."""

    @staticmethod
    def forward(ctx, x, deg, layer, *params):
        x, params, ctx.d_orig = _pad_features(x, params, layer.PARAM_KEYS)
        P = dict(zip(layer.PARAM_KEYS, params))
        # (layer.want_backward was set by forward_conv BEFORE .apply(): grad mode is always off in here, and ctx.needs_input_grad says
        #  (True, ...) under torch.no_grad() too)
        Z = layer.forward(x, deg, P)
        ctx.layer, ctx.state = layer, layer.saved
        ctx.save_for_backward(x, *params)
        ctx.set_materialize_grads(False)
        # second output: the NORMALISED adjacency values [N,K] -- differentiable, for the layers after this one that read the same
        # adjacency (model.py:1266-1290); its cotangent joins the aggregation's own inside the backward.
        # Both outputs leave as fresh VIEWS: autograd hangs this node on the returned objects, and the tensors kept in `saved` (which the
        # node's ctx holds) must not be those objects -- output -> grad_fn -> ctx -> saved -> output is a reference cycle that only the
        # cyclic collector frees, i.e. a step's whole state (gigabytes on chunked rows) stays allocated for an unbounded number of steps
        return Z.view(Z.shape), layer.saved["ahat"].view(layer.saved["ahat"].shape)

    @staticmethod
    def backward(ctx, dZ, dahat):
        x, *params = ctx.saved_tensors
        layer = ctx.layer
        P = dict(zip(layer.PARAM_KEYS, params))
        layer.saved, layer._fwd_gen = ctx.state, ctx.state["gen"]       # (another forward of the same module may have run since)
        layer.x_grad = bool(ctx.needs_input_grad[0])
        if dZ is None:
            dZ = torch.zeros_like(ctx.state["Z"])
        g = layer.backward(dZ.contiguous(), x, P, dA_ext=dahat)
        for k_ in layer.PARAM_KEYS:
            g[k_] = g[k_].reshape(P[k_].shape)
        g = _unpad_grads(g, layer.PARAM_KEYS, ctx.d_orig)
        return (g.get("x"), None, None) + tuple(g[k_] for k_ in layer.PARAM_KEYS)


class _FusedDGGMlpConvFn(torch.autograd.Function):
    """_FusedDGGConvFn with an edge-MLP scorer (u-v-deg, u-v-A_uv, u-v-deg-dist, edge_conv; reference dgm.py:1628-1719) on edge-list
    candidates: the scorer's terms arrive in the per-node / per-edge form of DGG_LearnableK_debug._edge_mlp_terms (sliced from the
    reference's parameters by differentiable torch ops outside this node) and get their gradients from the same backward."""
    SC_KEYS = ("Wcat", "wdu", "wdv", "wex", "b1", "w2", "b2")

    @staticmethod
    def forward(ctx, x, deg, layer, sc_static, Wcat, wdu, wdv, wex, b1, w2, b2, *params):
        x, params, ctx.d_orig = _pad_features(x, params, layer.PARAM_KEYS)
        P = dict(zip(layer.PARAM_KEYS, params))
        det = lambda t_: None if t_ is None else t_.detach()  # noqa: E731
        ctx.packed = packed = sc_static.get("packed")
        if packed is not None:
            # `Wcat` is edge_encode.0.weight [h, 2h + extras] itself (columns [u | v | extras], dgm.py:1101-1105): sliced here, outside
            # autograd (the slices of the separate-modules path cost a dozen tiny copy / zero-fill launches per step in their backward)
            h_, cols = packed
            W0 = Wcat.detach()
            pick = lambda c_: None if c_ is None else W0[:, c_].contiguous()  # noqa: E731
            Wcat, wdu, wdv, wex = torch.cat([W0[:, :h_], W0[:, h_:2 * h_]], 0), pick(cols[0]), pick(cols[1]), pick(cols[2])
            w2 = w2.detach().reshape(-1)
        layer.scorer = dict(sc_static, Wcat=det(Wcat), wdu=det(wdu), wdv=det(wdv), wex=det(wex), b1=det(b1), w2=det(w2), b2=det(b2))
        Z = layer.forward(x, deg, P)
        ctx.layer, ctx.state, ctx.scorer = layer, layer.saved, layer.scorer
        ctx.save_for_backward(x, *params)
        ctx.set_materialize_grads(False)
        return Z.view(Z.shape), layer.saved["ahat"].view(layer.saved["ahat"].shape)       # (fresh views: see _FusedDGGConvFn.forward)

    @staticmethod
    def backward(ctx, dZ, dahat):
        x, *params = ctx.saved_tensors
        layer = ctx.layer
        P = dict(zip(layer.PARAM_KEYS, params))
        layer.saved, layer._fwd_gen, layer.scorer = ctx.state, ctx.state["gen"], ctx.scorer
        layer.x_grad = bool(ctx.needs_input_grad[0])
        if dZ is None:
            dZ = torch.zeros_like(ctx.state["Z"])
        g = layer.backward(dZ.contiguous(), x, P, dA_ext=dahat)
        gs = g["scorer"]
        if ctx.packed is not None:                       # d loss / d edge_encode.0.weight in ONE concatenation
            hw = gs["Wcat"].shape[0] // 2
            extras = [gs[k_][:, None] for k_, c_ in zip(("wdu", "wdv", "wex"), ctx.packed[1]) if c_ is not None]
            order = sorted(range(len(extras)), key=[c_ for c_ in ctx.packed[1] if c_ is not None].__getitem__)
            dW0 = torch.cat([gs["Wcat"][:hw], gs["Wcat"][hw:]] + [extras[o_] for o_ in order], 1)
            sc_grads = (dW0, None, None, None, gs["b1"], gs["w2"].reshape(1, -1), gs["b2"])
        else:
            sc_grads = tuple(gs[k_] for k_ in _FusedDGGMlpConvFn.SC_KEYS)
        for k_ in layer.PARAM_KEYS:
            g[k_] = g[k_].reshape(P[k_].shape)
        g = _unpad_grads(g, layer.PARAM_KEYS, ctx.d_orig)
        return (g.get("x"), None, None, None) + sc_grads + tuple(g[k_] for k_ in layer.PARAM_KEYS)


class DGG_LearnableK_debug(nn.Module):
    """Drop-in for reference dgm.py:1077-1727 (modes u-v-dist / x / {k_times_edge_prob, k_only}, soft output)."""

    def __init__(self, in_dim=32, latent_dim=64, args=None):
        super().__init__()
        h = latent_dim
        self.in_dim, self.latent_dim = in_dim, h
        self.extra_edge_dim, self.extra_k_dim = args.extra_edge_dim, args.extra_k_dim
        self.hard = args.dgg_hard
        self.deg_mean, self.deg_std = args.deg_mean, args.deg_std
        # --- same registration order / layer types as the reference (dgm.py:1097-1143) -> same keys, same init
        self.node_encode_for_edges = nn.Sequential(nn.Linear(in_dim, h), nn.LeakyReLU())
        self.edge_encode = nn.Sequential(nn.Linear(2 * h + self.extra_edge_dim, h), nn.LeakyReLU(), nn.Linear(h, 1))
        self.t = nn.Parameter(torch.tensor(-0.1))
        self.edge_conv_phi = nn.Linear(h, h // 2)
        self.edge_conv_theta = nn.Linear(h, h // 2)
        self.edge_conv_encode = nn.Linear(h // 2, 1)
        self.edge_prob_net_mode = args.dgg_mode_edge_net
        self.input_degree_decode = nn.Linear(3, 1, bias=True)
        self.combine_input_degree = nn.Sequential(nn.Linear(h + 3, h), nn.LeakyReLU())
        self.adj_project = nn.Linear(1, 1)
        self.k_net_mode = args.dgg_mode_k_net
        self.signal_project = nn.Linear(256, 1, bias=True)
        self.input_degree_project = nn.Linear(1, 3, bias=True)
        self.node_encode_for_k = nn.Sequential(nn.Linear(in_dim, h), nn.LeakyReLU())
        self.k_embed = nn.Sequential(nn.Linear(h + self.extra_k_dim, h // 2), nn.LeakyReLU())
        self.k_W = nn.Parameter(torch.rand(h, h, requires_grad=True))
        k_in = 3 if self.k_net_mode in ("input_deg", "learn_normalized_degree") else h // 2
        self.k_net = LearnableKEncoder(in_dim=k_in, latent_dim=h // 4, args=args)
        self.k_select_mode = args.dgg_mode_k_select
        self.args = args
        # --- GPU-path controls (not in the reference) ---------------------------------------------------------
        self.ell_width = getattr(args, "dgg_ell_width", ops.DEFAULT_K)
        self.topk_algo = getattr(args, "dgg_topk_algo", 0)
        self._explicit_noise = None          # test hook: the G that gumbel_sample(log_p, G) would receive
        self._seed = None
        self._overflow = None                # device bool: a row's ramp support k_i + 8.5 exceeded the ELL width (see check_ell_bound)

    # test / reproducibility hooks ---------------------------------------------------------------------------
    def set_noise(self, G):
        """Use an explicit noise matrix [N,N] (what dgm.py:1226 samples) instead of the counter-based generator."""
        self._explicit_noise = G

    def set_seed(self, s0, s1=0):
        self._seed = (int(s0) & 0xFFFFFFFF, int(s1) & 0xFFFFFFFF)

    def check_ell_bound(self):
        """The sparse formulation keeps `ell_width` (64) entries per row, which is EXACT while every row's ramp support fits:
        k_i + 8.5 <= ell_width, or the row has no more candidates than the width.  The learned k is
        unbounded (k = relu(kp sd + mu) + 1, dgm.py:1580-1584), so every forward ORs the violation into a device flag (no
        sync on the hot path); this method reads it (one sync) and raises.  Called by `EllAdjacency.to_dense()/to_sparse()`,
        by the training harness once per epoch, and on every forward when DGG_STRICT_BOUND=1.
        The same call reports the ranked symmetric noise generator running out of workspace for its dense tier (rows that could
        not be settled come back empty; dgg_topk_rsym.hip)."""
        err = self.__dict__.get("_rsym_err")
        if err is not None:
            t3, depth = self.__dict__.get("_rsym_t3"), self.__dict__.get("_rsym_depth")
            self._rsym_err = self._rsym_t3 = self._rsym_depth = None
            if bool(err.any()):
                raise RuntimeError("DGG_LearnableK_debug: the ranked symmetric noise generator could not settle every row inside its "
                                   "workspace (too many rows far from everything else); set args.dgg_sym_generator = 'hash'")
            deep = depth is not None and float(depth) > 0.3 * math.log(8.0)       # tier 2 walked > 8x the pairs of tier 1 (all owners)
            nt3 = int(t3) if t3 is not None else 0
            if (nt3 > 0 or deep) and self._sym_generator_now() == "ranked":
                # nodes so far from everything else that their noise rows had to be written out in full: exact, but every such row
                # costs a complete walk of all owners' sequences -- on this data the per-pair hash generator (same law) is cheaper.
                # The decision belongs to THIS module (args is usually shared by every DGG of a model) and is taken only when the
                # caller asked for it (args.dgg_sym_generator = "auto"): a health check must not silently change the noise stream
                # of a seeded run.  Captured hipGraphs keep replaying the generator they were captured with.
                import warnings
                auto = getattr(self.args, "dgg_sym_generator", "ranked") == "auto"
                many = nt3 > max(4, self.__dict__.get("_rsym_rows", 0) // 4000) or deep
                warnings.warn(f"DGG_LearnableK_debug: {nt3} rows needed the dense tier of the ranked symmetric noise generator"
                              + (" and its second tier walked more than 8x deeper than the first" if deep else "") +
                              ("; this module switches to the per-pair hash generator (same law, N^2 sweep) for the following forwards"
                               if auto and many else "; the per-pair hash generator (args.dgg_sym_generator = 'hash', or 'auto' to let "
                               "the module switch by itself) is cheaper on such data"))
                if auto and many:
                    self._sym_generator = "hash"
                    self._sym_spread = True             # (rows far from everything else: the hash noise's per-row front end, _fused_configure)
        fl = self.__dict__.get("_fused_layer")
        if fl is not None and fl.wide_sticky is not None and not _capturing():
            # (a device word the layout kernel ORs into on every forward -- every replay of a captured step -- under a fixed capacity)
            try:
                fl.check_wide()
            except RuntimeError as e:
                raise RuntimeError("DGG_LearnableK_debug: the chunked rows of a captured step outgrew the capacity they were captured with: run "
                                   f"one eager forward and capture again ({e})") from None
        flag = self.__dict__.get("_overflow_dev")
        if flag is not None and not _capturing() and bool(flag.item()):
            flag.zero_()
            self._overflow = torch.ones((), dtype=torch.bool, device=flag.device)
        if self._overflow is not None and bool(self._overflow):
            self._overflow = None
            raise RuntimeError(
                f"DGG_LearnableK_debug: a row's learned degree satisfies k + 8.5 > ell_width = {self.ell_width} while it has more "
                "candidates than that: ranks the reference still weights were dropped (row sums, normalisation and gradients "
                "differ from the reference from here on).  Edge-list candidates: set args.dgg_wide_rows = 'csr' (rows of any width; "
                "'auto' picks it whenever a row would lose weight, except inside a hipGraph capture).  All-pairs candidates: "
                "args.dgg_wide_rows = 'auto' keeps every weighted rank in chunked rows (ranked noise generator; one eager forward before a capture).")

    def _sym_generator_now(self):
        """generator for symmetric noise in the next forward: "ranked" | "hash".  args.dgg_sym_generator: "ranked" (default), "hash", or
        "auto" = ranked until check_ell_bound() finds this module's data in the ranked generator's slow regime"""
        own = self.__dict__.get("_sym_generator")
        if own is not None:
            return own
        return "hash" if getattr(self.args, "dgg_sym_generator", "ranked") == "hash" else "ranked"

    def _asym_generator_now(self, x, seed):
        """NOISE_RANKED or NOISE_HASH for this forward's asymmetric noise on all-pairs candidates, and the watch on the ranked search's
        DATA-DEPENDENT cost.  The search visits about L exp(D / 0.3) ranks of a row, D = the spread of 0.05 ||xp_i - xp_j|| the row
        sees: ~80 on unit-scale features (0.25 ms at N = 100 000), thousands once learned / unnormalised latents spread the
        distances several times wider -- on ONE wavefront per row.  Measured at N = 100 000 (tools/time_topk.py, ms; feature scale
        x1 / x4 / x16): ranked 0.35 / 9.0 / 170; the per-pair hash evaluators depend on the regime just as much and are SLOWER where
        the ranked search is slow (guess-and-verify 3.5 / 198 / 318, adaptive 13.7 / 143 / 435, MFMA-bounded 41 / 123 / 226) --
        when distances and noise both matter, every pair has to be scored, which is what the exhaustive kernel costs (~200 ms).
        So there was no cheaper exact evaluator to route to (rounds 3-5; since round 6 the ranked search itself takes the rows'
        nearest-neighbour bound on such data -- ShardedDGGConv.tight_bound -- and the hash generators the chunked rows' per-row front
        end -- _hash_spread_now); args.dgg_asym_generator:
          "auto" (default)  the ranked generator, WATCHED: every `args.dgg_pilot_every` (16) forwards -- and on the first -- a pilot
                            walks ~1000 sampled rows with a budget of 64 blocks (ops.ranked_probe: one small launch, one readback);
                            when its estimate exceeds `args.dgg_ranked_warn_us` (5000) the module warns, once per regime change,
                            with the measured depth (graphs below 8192 nodes and hipGraph captures are not probed);
          "ranked" / "hash" no pilot; "hash" = the per-pair hash generator (same law, another realisation)."""
        policy = getattr(self.args, "dgg_asym_generator", "auto")
        if policy == "hash":
            return ops.NOISE_HASH
        N = x.shape[0]
        if policy != "auto" or N < 8192:
            return ops.NOISE_RANKED
        st = self.__dict__.setdefault("_asym_state", {"n": 0, "slow": False, "probe": None})
        every = max(1, int(getattr(self.args, "dgg_pilot_every", 16)))
        if st["n"] % every == 0 and not _capturing():
            with torch.no_grad():
                xp = ops.linear_fwd(x.detach(), self.node_encode_for_edges[0].weight.detach(), self.node_encode_for_edges[0].bias.detach(),
                                    ops.ACT_LEAKY)
                pr = ops.ranked_probe(xp, None, ops.T_DIST, seed, stride=max(1, N // 1024), max_blocks=64)
            est = ops.ranked_cost_estimate(pr, N)
            slow = est > float(getattr(self.args, "dgg_ranked_warn_us", 5000.0))
            if slow and not st["slow"]:
                import warnings
                warnings.warn(f"DGG_LearnableK_debug: the ranked noise search walks deep on this data: {pr['blocks_per_row']:.1f} blocks of 64 "
                              f"ranks per sampled row ({100 * pr['budget_hit_frac']:.1f} % of them beyond the 64-block budget), estimated "
                              f"{est / 1e3:.1f} ms per forward at N = {N} (0.25 ms on unit-scale features).  The latent distances "
                              "0.05 ||xp_i - xp_j|| spread over more than the noise scale 0.3; results stay exact, and no evaluator of this "
                              "build is faster in that regime (see _asym_generator_now)")
            st.update(slow=slow, probe=dict(pr, ranked_us_estimate=est))
        st["n"] += 1
        return ops.NOISE_RANKED

    def _hash_spread_now(self, x, seed):
        """True while the latents of this module are SPREAD over several noise scales -- the regime in which the 64-rank entry of the per-pair
        hash generators (one threshold for the whole graph, guessed from the noise law alone) loses every row to its exhaustive fallback
        (N = 100 000, features x4: 145-170 ms) and the chunked rows' front end (threshold per row on G + the row's nearest-neighbour bound)
        does not (7-9 ms; 2-3x the 64-rank entry on unit-scale data, hence not always).  Measured with the ranked search's pilot, which
        walks deep on exactly that data: ~1000 sampled rows, 16-block budget, every args.dgg_pilot_every (16) forwards; more than 8 blocks
        per row = spread (unit scale: 3, features x2: 5, x4: 16+).  Graphs below 8192 nodes and hipGraph captures are not probed."""
        N = x.shape[0]
        if N < 8192:
            return False
        st = self.__dict__.setdefault("_hash_state", {"n": 0, "spread": False})
        every = max(1, int(getattr(self.args, "dgg_pilot_every", 16)))
        if st["n"] % every == 0 and not _capturing():
            with torch.no_grad():
                xp = ops.linear_fwd(x.detach(), self.node_encode_for_edges[0].weight.detach(), self.node_encode_for_edges[0].bias.detach(),
                                    ops.ACT_LEAKY)
                pr = ops.ranked_probe(xp, None, ops.T_DIST, seed if seed is not None else (0, 0), stride=max(1, N // 1024), max_blocks=16)
            st["spread"] = pr["blocks_per_row"] > 8.0
            st["blocks"] = pr["blocks_per_row"]
        st["n"] += 1
        return st["spread"]

    def forward_conv(self, x, in_adj, conv_weight, want_norm=False):
        """`GCNConv(x, normalize_adj(self(x, in_adj)))` with conv_weight = GCNConv.W [in, out] as one fused autograd node
        (_FusedDGGConvFn) -> (Z, unnormalised EllAdjacency, DETACHED: its values carry no autograd edge -- the loss of the
        reference's training scripts reads the class scores only, train_small_graphs.py:226-230), or None when this configuration
        is outside the fused step (the caller then runs the modules one after the other): scorer u-v-dist (any candidates) or an
        edge-MLP scorer (u-v-deg, u-v-A_uv, u-v-deg-dist, edge_conv, A_uv; edge-list candidates), k-net "x", soft
        k_times_edge_prob / k_only output, widths the partitioned backward covers, rows that fit the ELL width.
        want_norm: additionally return the NORMALISED adjacency as a differentiable EllAdjacency for the layers that read the same
        graph after this one (GCN_DGG's second layer): its gradient flows back into the generator through the same node.
        The steps: _fused_outside (configurations the node does not cover) -> candidates -> _fused_scorer (edge-MLP terms) ->
        _fused_configure (noise generator, what rows wider than 64 ranks do, flags) -> the node -> _fused_result (what the forward's
        learned degrees decide after the fact, the returned adjacencies)."""
        from .parallel import ShardedDGGConv
        why = self._fused_outside(x, in_adj, conv_weight)
        if why is not None:
            return self._fused_fallback(why)
        a, N = self.args, x.shape[0]
        mlp_mode = self.edge_prob_net_mode in _EDGE_MLP_FUSED
        wide_state = None
        if isinstance(in_adj, AllPairs):
            cand, deg, rowptr = None, in_adj.prior_degree, None
            if self.__dict__.get("_ap_wide", {}).get("on") or (getattr(a, "dgg_wide_rows", "auto") == "csr" and
                                                               N <= int(getattr(a, "dgg_allpairs_csr_max", 8192))):
                return None                                   # (policy "csr": every column ranked, the modules' CSR form)
        else:
            if isinstance(in_adj, EllAdjacency):
                in_adj = in_adj.to_sparse().detach()
            rowptr, col, deg = csr_candidates(in_adj)
            cand = (rowptr, col)
            wide_state = self._wide_rows_state(in_adj, rowptr)
            if wide_state is True:                            # (known before any kernel runs: this graph takes the CSR form -- no discarded forward)
                return self._fused_fallback("rows wider than the list with learned degrees beyond it (CSR form)")
        mlp, sc_static = self._fused_scorer(in_adj) if mlp_mode else (None, None)
        layer = self.__dict__.get("_fused_layer")
        if layer is None or layer.N != N:
            layer = self.__dict__["_fused_layer"] = ShardedDGGConv(ops, N, K=64, t=ops.T_DIST)
        noise_mode, chunk_active = self._fused_configure(layer, x, cand, wide_state, mlp_mode)
        kn = self.k_net
        params = (self.node_encode_for_edges[0].weight, self.node_encode_for_edges[0].bias, self.node_encode_for_k[0].weight,
                  self.node_encode_for_k[0].bias, self.k_embed[0].weight, self.k_embed[0].bias, kn.k_mu.weight, kn.k_mu.bias,
                  kn.k_project.weight, kn.k_project.bias, conv_weight)
        # no backward can follow (torch.no_grad(), frozen parameters): the partition's sort -- read by the backward only -- is skipped
        layer.want_backward = ops.backward_will_follow(x, *params, *([mlp["Wcat"], mlp["b1"], mlp["w2"], mlp["b2"]] if mlp_mode else []))
        if mlp_mode:
            Z, ahat = _FusedDGGMlpConvFn.apply(x, deg, layer, sc_static, mlp["Wcat"], mlp["wdu"], mlp["wdv"], mlp["wex"], mlp["b1"],
                                               mlp["w2"], mlp["b2"], *params)
        else:
            Z, ahat = _FusedDGGConvFn.apply(x, deg, layer, *params)       # (ops.ChunkCapacityError: a learned degree is NaN)
        return self._fused_result(layer, Z, ahat, in_adj, cand, rowptr, noise_mode, chunk_active, mlp_mode, want_norm)

    def _fused_outside(self, x, in_adj, conv_weight):
        """-> the reason this configuration is outside the fused layer, or None.  Which clause sends a forward to the separate modules
        is LOGGED, once per module and clause (logger "dgg_amd", level INFO), and counted in self.fused_fallback: the separate modules
        are ~1.7x slower at Pubmed size, and a silent fall-back looks like a performance bug of the fused layer"""
        a, h = self.args, self.latent_dim
        fin, fout = conv_weight.shape
        mlp_mode = self.edge_prob_net_mode in _EDGE_MLP_FUSED
        clauses = (
            ("scorer outside u-v-dist / the edge-MLP family", self.edge_prob_net_mode != "u-v-dist" and not mlp_mode),
            ("edge-MLP scorer on all-pairs candidates", mlp_mode and isinstance(in_adj, AllPairs)),
            ("k-net mode other than 'x'", self.k_net_mode != "x"),
            ("k-select mode other than k_times_edge_prob / k_only", self.k_select_mode not in ("k_times_edge_prob", "k_only")),
            ("dgg_hard", bool(self.hard)),
            ("debug_step 0 / 1", a.debug_step in (0, 1)),
            ("stochastic_k in training mode", bool(getattr(a, "stochastic_k", False) and self.training)),
            ("explicit noise tensor", self._explicit_noise is not None),
            ("args.dgg_fused_layer = False", not getattr(a, "dgg_fused_layer", True)),
            ("input not on the GPU", not x.is_cuda),
            ("latent width outside {16, 32, 64, 128}", h not in (16, 32, 64, 128) or h not in ops.KNET_MFMA_WIDTHS),
            ("conv width outside {16, 32, 64, 128}", fout not in (16, 32, 64, 128)),
            ("conv wider than its input (fout > fin)", fout > fin),
            ("ell_width other than 64", self.ell_width != 64),
            ("input dtype other than float32", x.dtype != torch.float32),
        )
        return next((why for why, hit in clauses if hit), None)

    def _fused_scorer(self, in_adj):
        """the edge-MLP scorer's parameters and per-edge inputs for the fused node, in the CSR order of the candidates -> (mlp, static)"""
        h = self.latent_dim
        avals = _cached("values_f32", in_adj, lambda: in_adj.coalesce().values().to(torch.float32).contiguous())
        if self.edge_prob_net_mode in ("edge_conv", "A_uv"):
            mlp, ex_in = self._edge_mlp_terms(avals)
            packed = None
        else:                                                 # edge_encode.0.weight goes into the node whole (sliced inside it)
            need = {"u-v-deg": 2, "u-v-A_uv": 1, "u-v-deg-dist": 3}[self.edge_prob_net_mode]
            W0 = self.edge_encode[0].weight
            assert W0.shape[1] == 2 * h + need, f"edge mode {self.edge_prob_net_mode!r} needs extra_edge_dim={need} (edge_encode.0 is {tuple(W0.shape)})"
            cols = {"u-v-deg": (2 * h, 2 * h + 1, None), "u-v-A_uv": (None, None, 2 * h), "u-v-deg-dist": (2 * h, 2 * h + 1, 2 * h + 2)}
            packed = (h, cols[self.edge_prob_net_mode])
            mlp = dict(Wcat=W0, wdu=None, wdv=None, wex=None, b1=self.edge_encode[0].bias, w2=self.edge_encode[2].weight,
                       b2=self.edge_encode[2].bias, act=ops.ACT_LEAKY, ex_mode={"u-v-deg": 0, "u-v-A_uv": 1, "u-v-deg-dist": 2}[self.edge_prob_net_mode],
                       t_ex=-1.0 if self.edge_prob_net_mode == "u-v-deg-dist" else 0.0)
            ex_in = avals if self.edge_prob_net_mode == "u-v-A_uv" else None
        return mlp, dict(erow=csr_pattern(in_adj)[2], ex_in=ex_in, ex_mode=mlp["ex_mode"], t_ex=mlp["t_ex"], act=mlp["act"], packed=packed)

    def _fused_configure(self, layer, x, cand, wide_state, mlp_mode):
        """sets the engine up for this forward: noise generator, selection mode, what rows wider than 64 ranks do, device flags
        -> (noise_mode, chunk_active = this forward reads its chunk layout back)"""
        a, N = self.args, x.shape[0]
        noise_mode, _, seed = self._noise_cfg()
        if cand is None and noise_mode == ops.NOISE_RANKED:
            noise_mode = self._asym_generator_now(x, seed)
        elif cand is not None:
            noise_mode = {ops.NOISE_RANKED: ops.NOISE_HASH, ops.NOISE_RANKED_SYM: ops.NOISE_HASH_SYM}.get(noise_mode, noise_mode)
        mode = ops.MODE_K_TIMES_EDGE_PROB if self.k_select_mode == "k_times_edge_prob" else ops.MODE_K_ONLY
        layer.cand, layer.noise_mode, layer.seed, layer.mode, layer.scorer = cand, noise_mode, seed, mode, None
        layer.sym_fallback, layer.sym_hash = getattr(a, "dgg_sym_generator", "auto") != "ranked", False
        # the module fell back from the ranked symmetric generator on THIS data (spread latents): the hash noise's forwards take the
        # chunked rows' per-row front end, whatever the learned degrees (parallel.py, force_chunked)
        # ... and so do forwards under a per-pair hash generator the CALLER chose (args.dgg_sym_generator / dgg_asym_generator = "hash")
        # while a pilot finds the latents spread (_hash_spread_now)
        layer.force_chunked = cand is None and noise_mode in (ops.NOISE_HASH, ops.NOISE_HASH_SYM) and \
            ((noise_mode == ops.NOISE_HASH_SYM and bool(self.__dict__.get("_sym_spread"))) or self._hash_spread_now(x, seed))
        layer.tight_bound = getattr(a, "dgg_tight_bound", "auto")     # ranked search: nearest-neighbour bound in its stop tests when the walk is deep
        layer.x_grad = bool(x.requires_grad)
        # all-pairs rows wider than the 64-rank list (learned degrees k_i + 9.5 > 64): chunked rows inside the engine, from the forward
        # that first needs them (one readback of the chunk count per forward; a hipGraph capture replays the last eager layout)
        chunked = cand is None and self._chunk_policy(noise_mode)
        layer.wide_rows = "auto" if chunked else "off"
        layer.wide_cap = None
        if chunked and _capturing():
            # nothing can be read back under capture: the layout of the last eager forward on this module, with some slack, becomes a
            # FIXED capacity whose overflow flags check_ell_bound() reads; no wide row then: the list, with its enforced bound
            last = getattr(layer, "last_layout", None)
            layer.wide_cap = None if last is None or last[0] == N else (last[0] + last[0] // 8 + 64, min(ops.chunk_maxm_for(N), last[1] + max(1, last[1] // 8)))
            if layer.wide_cap is None:
                layer.wide_rows = "off"
        if cand is not None and not mlp_mode:                 # the ELL-width bound is tested inside the search kernel (no extra launches)
            # (a forward whose fate is decided from its own learned degrees -- wide_state None -- raises a SCRATCH flag: if it is
            #  discarded for the CSR form its flag must not reach check_ell_bound, and flags of earlier forwards must survive it)
            name = "_overflow_scratch" if (wide_state is None and not _capturing()) else "_overflow_dev"
            flag = self.__dict__.get(name)
            if flag is None or flag.device != x.device:
                flag = self.__dict__[name] = torch.zeros((1,), device=x.device, dtype=torch.int32)
            layer.overflow = flag
        return noise_mode, chunked and layer.wide_rows == "auto" and layer.wide_cap is None

    def _fused_result(self, layer, Z, ahat, in_adj, cand, rowptr, noise_mode, chunk_active, mlp_mode, want_norm):
        """after the node ran: generator status, what this forward's learned degrees decide (CSR form from here on / overflow flags),
        and the returned adjacencies"""
        st, N = layer.saved, layer.N
        if noise_mode == ops.NOISE_RANKED_SYM:                # the reference's DEFAULT noise (symmetric_noise=True, dgm.py:1216-1223): the
            self._note_rsym(getattr(layer, "rsym_last", None), N)     # generator's status words, checked by check_ell_bound as for the modules
            if layer.sym_hash:
                self._sym_switch()
        if st.get("partp") is None:                           # (shape outside the partitioned backward: the separate modules)
            return self._fused_fallback("shape outside the partitioned backward")
        k = st["k"]
        if cand is not None and self._wide_rows(in_adj, rowptr, k):
            # rows wider than the ELL with learned degrees beyond it: the CSR form from here on (this forward is discarded, once per graph)
            if not mlp_mode and "_overflow_scratch" in self.__dict__:
                self._overflow_scratch.zero_()
            return None
        lay = st.get("layout")
        if cand is None and lay is None and not chunk_active and self._allpairs_wide(N, k):
            return None                                       # learned degrees beyond the list: the modules' CSR form (every column ranked)
        if cand is None and lay is None and not chunk_active:
            self._track_overflow(k, None)
        elif cand is None and layer.wide_cap is not None:     # fixed capacity (capture): overflow flags in layer.wide_sticky (check_ell_bound)
            pass
        elif mlp_mode:
            ent = self.__dict__.get("_wide_cache", {}).get(id(in_adj))
            if not (ent is not None and ent[0]() is in_adj and ent[1] <= self.ell_width):     # (no row can outgrow the list otherwise)
                self._track_overflow(k, rowptr[1:] - rowptr[:-1])
        elif __import__("os").environ.get("DGG_STRICT_BOUND") == "1":
            self.check_ell_bound()
        unnorm = EllAdjacency(st["idx"], st["w"], N, rs=st["rs"], k=k, score=st["val"], owner=self, layout=lay)
        if not want_norm:
            return Z, unnorm
        if st.get("side_join"):                               # the partition's sort ran on the layer's side stream: a later layer's
            torch.cuda.current_stream().wait_stream(layer._side_stream())     # backward reads it BEFORE this node's own backward joins
            st["side_join"] = False
        # (the partition's records reach a later layer's backward only when the sort ran: a forward without a backward skips it)
        pp = (st["partp"], st["rs"]) if st.get("partp_sorted", True) else None
        if lay is not None and pp is None:                    # (chunked rows as a separate module: the CSR kernels)
            return Z, unnorm, EllAdjacency(st["idx"], ahat.detach(), N, k=k, score=st["val"], normalized=True, owner=self, layout=lay)
        return Z, unnorm, EllAdjacency(st["idx"], ahat, N, k=k, score=st["val"], normalized=True, owner=self, partp=pp, layout=lay)

    def _fused_fallback(self, why):
        """forward_conv leaves for the separate modules: counted per reason, logged once per module and reason -> None"""
        fb = self.__dict__.setdefault("fused_fallback", {})
        if why not in fb:
            import logging
            logging.getLogger("dgg_amd").info("DGG_LearnableK_debug.forward_conv: the fused layer does not cover this configuration (%s): "
                                              "the generator, normalize_adj and the layer run as separate modules", why)
        fb[why] = fb.get(why, 0) + 1
        return None

    def _sym_switch(self):
        """the ranked symmetric generator could not settle a forward on this module's data (redone under the symmetric per-pair hash):
        the following forwards use the per-pair hash generator directly (same law, another realisation); said once"""
        if self.__dict__.get("_sym_generator") != "hash":
            import warnings
            warnings.warn("DGG_LearnableK_debug: the ranked symmetric noise generator could not settle every row of this forward inside its "
                          "workspace (rows far from everything else: a property of the latent features); the forward was evaluated under the "
                          "symmetric per-pair hash generator instead (same law) and this module stays with it "
                          "(args.dgg_sym_generator = 'ranked' keeps the ranked generator and raises instead)")
            self._sym_generator = "hash"
        self._sym_spread = True

    def _note_rsym(self, st, N):
        """status words of the ranked symmetric generator (noise_mode 5) of one forward, ORed / maxed into the module's (read by
        check_ell_bound: rows it could not settle, rows of its dense tier, depth of its second tier)"""
        if not st or st.get("rsym_err") is None:
            return
        prev = self.__dict__.get("_rsym_err")
        self._rsym_err = st["rsym_err"] if prev is None else (prev | st["rsym_err"])
        if N > 1024:                                 # (smaller graphs take the dense tier by design: dgg_topk_rsym.hip, SMALL_N)
            self._rsym_rows = int(N)
            prev3 = self.__dict__.get("_rsym_t3")
            self._rsym_t3 = st["rsym_tier3"] if prev3 is None else torch.maximum(prev3, st["rsym_tier3"])
            prevd = self.__dict__.get("_rsym_depth")
            self._rsym_depth = st["rsym_depth"] if prevd is None else torch.maximum(prevd, st["rsym_depth"])

    def _chunk_policy(self, noise_mode):
        """All-pairs rows wider than the 64-rank list as CHUNKED rows (ops.chunk_layout; any learned degree, any graph size, every
        counter-based noise generator and unperturbed scores)?  The ranked generator settles rows of up to 32 chunks in register lists and
        wider ones through threshold buffers; unperturbed scores and the per-pair hash generators take the threshold buffers for every
        row; the ranked SYMMETRIC generator has no wide-row form of its own -- wide rows are evaluated under the symmetric per-pair hash
        (the same law, another realisation: ops.allpairs_topk_wide).  Latent widths 16-128.
        args.dgg_wide_rows: "auto" (default) / "chunked" yes; "csr": the complete candidate pattern in CSR form instead (graphs of at
        most args.dgg_allpairs_csr_max nodes, every column ranked; "csr_auto": from the forward that first needs it); "ell": the
        64-rank list with the enforced bound.  Explicit noise tensors keep the CSR form."""
        policy = getattr(self.args, "dgg_wide_rows", "auto")
        return (policy in ("auto", "chunked") and noise_mode in (ops.NOISE_NONE, ops.NOISE_HASH, ops.NOISE_HASH_SYM, ops.NOISE_RANKED, ops.NOISE_RANKED_SYM)
                and self.ell_width == 64
                and self.latent_dim in (16, 32, 64, 128) and self.edge_prob_net_mode == "u-v-dist")

    def wide_row_plan(self, N, all_pairs, noise_mode):
        """What rows that need more than 64 ranks do for a graph of N nodes under args.dgg_wide_rows -- the ONE summary of the policy
        (the predicates the forwards use: _chunk_policy, _wide_rows_state, _allpairs_wide):
          "chunked"          all-pairs candidates: ceil(k_i + 8.5) + 1 ranks of every row in chunks of 64, any width (auto / chunked)
          "csr"              the CSR form of select_top_k from the first forward (csr)
          "csr_when_needed"  the CSR form from the forward whose learned degrees first need it, and from then on (auto on edge lists
                             and for explicit noise tensors on small all-pairs graphs; csr_auto)
          "list"             64 ranks per row, the bound enforced by check_ell_bound (ell; graphs beyond dgg_allpairs_csr_max without a
                             chunked form)"""
        policy = getattr(self.args, "dgg_wide_rows", "auto")
        if not all_pairs:
            return {"ell": "list", "csr": "csr"}.get(policy, "csr_when_needed")
        if self._chunk_policy(noise_mode):
            return "chunked"
        if policy in ("ell", "chunked") or N > int(getattr(self.args, "dgg_allpairs_csr_max", 8192)):
            return "list"
        return "csr" if policy == "csr" else "csr_when_needed"

    def _track_overflow(self, k, ncand):
        over = k.detach() + 8.5 > float(self.ell_width)
        if ncand is not None:
            over = over & (ncand > self.ell_width)
        flag = over.any()
        self._overflow = flag if self._overflow is None else (self._overflow | flag)
        if __import__("os").environ.get("DGG_STRICT_BOUND") == "1":
            self.check_ell_bound()

    def _wide_rows_state(self, in_adj, rowptr):
        """What is known BEFORE this forward's learned degrees: True = the CSR form of select_top_k (rows of any width), False = the
        64-wide ELL is exact whatever k is (or is forced), None = it depends on this forward's k (_wide_rows decides).
        args.dgg_wide_rows: "csr" always (edge-list candidates), "ell" never, "auto" (default): exactly when the ELL would lose a
        non-zero weight, i.e. some row has more candidates than the ELL width AND a learned degree k_i + 8.5 above it.  The widest row
        is read back once per GRAPH OBJECT (cached by identity); graphs whose rows all fit never synchronise again.  A graph that once
        needed the CSR form keeps it (exact for every degree; the adjacency type does not flip from one forward to the next)."""
        policy = getattr(self.args, "dgg_wide_rows", "auto")
        if policy == "ell":
            return False
        if policy == "csr":
            return True
        cache = self.__dict__.setdefault("_wide_cache", {})
        ent = cache.get(id(in_adj))
        if ent is None or ent[0]() is not in_adj:
            if _capturing():
                return False
            for key in [k_ for k_, v in cache.items() if v[0]() is None]:
                del cache[key]
            lens = rowptr[1:] - rowptr[:-1]
            ent = cache[id(in_adj)] = [weakref.ref(in_adj), int(lens.max().item()) if lens.numel() else 0, False]
        if ent[1] <= self.ell_width:
            return False
        return True if ent[2] else None

    def _wide_rows(self, in_adj, rowptr, k):
        """Should this forward go through the CSR form of select_top_k instead of the 64-wide ELL?  While a graph with rows wider than
        the ELL is UNDECIDED the learned degrees are tested on the device and one flag is read back on EVERY forward (one host
        synchronisation; round 4 read it every 16th forward only, and the forwards in between could truncate a row -- ADVICE round 4);
        the first forward that needs the CSR form makes the decision sticky.  While a hipGraph is being captured nothing can be read
        back: the list is used and its bound enforced (check_ell_bound)."""
        st = self._wide_rows_state(in_adj, rowptr)
        if st is not None:
            return st
        if _capturing():
            return False
        ent = self._wide_cache[id(in_adj)]
        lens = rowptr[1:] - rowptr[:-1]
        ent[2] = bool(((k.detach() + 8.5 > float(self.ell_width)) & (lens > self.ell_width)).any().item())
        return ent[2]

    # ---- all-pairs candidates whose learned degrees outgrow the 64-wide list --------------------------------------------------
    def _allpairs_pattern(self, N, device):
        """the complete candidate set as a CSR pattern (rowptr, col, erow): what the reference's dense [N,N] rows are"""
        ent = self.__dict__.get("_ap_pattern")
        if ent is None or ent[0] != (N, device):
            ar = torch.arange(N, device=device, dtype=torch.int32)
            pat = (torch.arange(N + 1, device=device, dtype=torch.int64) * N, ar.repeat(N), ar.repeat_interleave(N))
            ent = self.__dict__["_ap_pattern"] = ((N, device), pat)
        return ent[1]

    def _allpairs_wide(self, N, k):
        """All-pairs candidates keep 64 ranks per row, exact while k_i + 8.5 <= 64; the learned degree is unbounded (dgm.py:1580-1584)
        and training moves it past that within a few steps (test_learned_degrees_of_a_trained_model_and_the_all_pairs_list).  Graphs
        of at most `args.dgg_allpairs_csr_max` (8192) nodes -- the sizes at which the reference's dense [N,N] formulation runs at
        all -- then take select_top_k on the COMPLETE candidate pattern in CSR form (every column ranked, any degree: N^2 entries,
        a few milliseconds at N = 3000), as rows wider than the list do for edge-list candidates.  args.dgg_wide_rows: "auto"
        (default: from the forward in which some k_i + 8.5 first exceeds the width, and from then on -- one flag read back per forward
        while the list is still in use; a hipGraph capture replays the last decision), "csr" (always), "ell" (never: the bound is
        enforced by check_ell_bound).  Larger graphs keep the list and the enforced bound."""
        policy = getattr(self.args, "dgg_wide_rows", "auto")
        if policy in ("ell", "chunked") or N > int(getattr(self.args, "dgg_allpairs_csr_max", 8192)):
            return False
        if policy == "csr":
            return True
        st = self.__dict__.setdefault("_ap_wide", {"on": False})
        if not st["on"] and not _capturing():
            st["on"] = bool((k.detach() + 8.5 > float(self.ell_width)).any().item())
        return st["on"]

    def _csr_soft_adjacency(self, x, in_adj, k, noise_mode, G, seed, mode, pattern=None, deg=None):
        """select_top_k on the CSR pattern of in_adj (ops.CsrSoftkFn: rows of any width, exact for any learned degree): edge
        probabilities as in _scores_adjacency, then perturbation + rank + ramp per row.  pattern / deg given (all-pairs candidates):
        the complete pattern, no in_adj."""
        if pattern is None:
            in_adj = in_adj.coalesce()
            pattern = csr_pattern(in_adj)
            _, _, deg = csr_candidates(in_adj)
        We, be = self.node_encode_for_edges[0].weight, self.node_encode_for_edges[0].bias
        cfg = dict(cand=pattern, t=ops.T_DIST)
        if self.edge_prob_net_mode == "u-v-dist":
            p = _DGGScoresFn.apply(x, None, None, We, be, None, None, None, None, None, None, None, cfg)
        else:
            mlp, ex_in = self._edge_mlp_terms(in_adj.values().to(torch.float32))
            cfg.update(ex_mode=mlp["ex_mode"], t_ex=mlp["t_ex"], act=mlp["act"])
            p = _DGGScoresFn.apply(x, deg, ex_in, We, be, mlp["Wcat"], mlp["wdu"], mlp["wdv"], mlp["wex"], mlp["b1"], mlp["w2"], mlp["b2"], cfg)
        w = ops.CsrSoftkFn.apply(p, k, pattern[0], pattern[1], noise_mode, G, seed, mode)
        if self.hard and mode == ops.MODE_K_TIMES_EDGE_PROB:      # straight-through (see forward): ramp mask forward, soft gradient
            ramp = ops.CsrSoftkFn.apply(p.detach(), k.detach(), pattern[0], pattern[1], noise_mode, G, seed, ops.MODE_K_ONLY)
            w = (ramp - w).detach() + w
        return CsrAdjacency(pattern[0], pattern[1], pattern[2], w, x.shape[0], k=k.detach())

    def _noise_cfg(self):
        if not self.args.perturb_edge_prob:
            return ops.NOISE_NONE, None, (0, 0)
        if self._explicit_noise is not None:
            return ops.NOISE_EXPLICIT, self._explicit_noise, (0, 0)
        seed = self._seed
        if seed is None:   # fresh noise per forward, reproducible under torch.manual_seed (CPU generator: no sync)
            s = torch.randint(0, 2 ** 31 - 1, (2,))
            seed = (int(s[0]), int(s[1]))
        # asymmetric noise: the ranked generator (rows produced in decreasing order -> early-stopping top-k search);
        # symmetric noise (dgm.py:1216-1223) is keyed on the unordered pair: the ranked symmetric generator (every pair owned by
        # one endpoint, which lists its largest noises first; args.dgg_sym_generator = "hash" selects the per-pair hash + N^2 sweep)
        if self.args.symmetric_noise:
            ranked = self.ell_width == 64 and self.latent_dim in ops.RSYM_WIDTHS and self._sym_generator_now() == "ranked"
            return (ops.NOISE_RANKED_SYM if ranked else ops.NOISE_HASH_SYM), None, seed
        # the ranked generator's row search is written for the full 64-wide list
        return (ops.NOISE_RANKED if self.ell_width == 64 else ops.NOISE_HASH), None, seed

    def _edge_mlp_terms(self, avals):
        """The reference's scorer parameters in the per-node / per-edge form of dgg_edge_mlp_fwd (differentiable slicing).

            u-v-deg       edge_encode.0 [h, 2h+2] on [u, v, deg_u, deg_v]       (dgm.py:1645-1670; raw degrees)
            u-v-A_uv      edge_encode.0 [h, 2h+1] on [u, v, a_uv]               (dgm.py:1628-1644)
            u-v-deg-dist  edge_encode.0 [h, 2h+3] on [u, v, deg_u, deg_v, exp(-||u-v||)]   (dgm.py:1671-1702)
            edge_conv     theta(v-u) + phi(u) = (phi-theta) u + theta v, bias b_theta + b_phi, no activation (1703-1719)
            A_uv          adj_project(a_uv): hidden width 1, A = B = 0          (dgm.py:1720-1725)"""
        h, mode = self.latent_dim, self.edge_prob_net_mode
        d = dict(wdu=None, wdv=None, wex=None, ex_mode=0, t_ex=0.0, act=ops.ACT_LEAKY)
        ex_in = None
        if mode in ("u-v-deg", "u-v-A_uv", "u-v-deg-dist"):
            W0 = self.edge_encode[0].weight
            need = {"u-v-deg": 2, "u-v-A_uv": 1, "u-v-deg-dist": 3}[mode]
            assert W0.shape[1] == 2 * h + need, f"edge mode {mode!r} needs extra_edge_dim={need} (edge_encode.0 is {tuple(W0.shape)})"
            d.update(Wcat=torch.cat([W0[:, :h], W0[:, h:2 * h]], 0), b1=self.edge_encode[0].bias,
                     w2=self.edge_encode[2].weight.reshape(-1), b2=self.edge_encode[2].bias)
            if mode != "u-v-A_uv":
                d.update(wdu=W0[:, 2 * h], wdv=W0[:, 2 * h + 1])
            if mode == "u-v-A_uv":
                d.update(wex=W0[:, 2 * h], ex_mode=1)
                ex_in = avals
            if mode == "u-v-deg-dist":
                d.update(wex=W0[:, 2 * h + 2], ex_mode=2, t_ex=-1.0)
        elif mode == "edge_conv":
            Th, Ph = self.edge_conv_theta, self.edge_conv_phi
            d.update(Wcat=torch.cat([Ph.weight - Th.weight, Th.weight], 0), b1=Th.bias + Ph.bias,
                     w2=self.edge_conv_encode.weight.reshape(-1), b2=self.edge_conv_encode.bias, act=ops.ACT_NONE)
        else:
            w = self.adj_project.weight
            d.update(Wcat=torch.zeros((2, h), device=w.device, dtype=w.dtype), b1=self.adj_project.bias, w2=torch.ones_like(w).reshape(-1),
                     b2=torch.zeros_like(self.adj_project.bias), wex=w.reshape(-1), ex_mode=1, act=ops.ACT_NONE)
            ex_in = avals
        return d, ex_in

    @staticmethod
    def _norm_deg(deg, consts, eps):
        """normalised degree and the (mean, std) it was normalised with: batch statistics (unbiased std) or constants"""
        mu, sd = (deg.mean(), deg.std()) if consts is None else (deg.new_tensor(consts[0]), deg.new_tensor(consts[1]))
        return (deg - mu) / (sd + eps), mu, sd

    def _stochastic_k(self, feat, deg, consts, embed):
        """stochastic_k in TRAINING mode (reference dgm.py:2041-2056): the latent of the k-net is sampled with the
        reparameterisation trick, latent = k_mu(z) + eps * exp(k_logvar(z) / 2), eps ~ N(0, 1).  The dense layers run on the
        MFMA linear kernel (autograd through ops.LinearFn); the sampling and the [N]-sized tail are elementwise torch ops."""
        kn = self.k_net
        z = ops.LinearFn.apply(feat.contiguous(), self.k_embed[0].weight, self.k_embed[0].bias, ops.ACT_LEAKY, 0) if embed else feat.contiguous()
        mu_l = ops.LinearFn.apply(z, kn.k_mu.weight, kn.k_mu.bias, ops.ACT_NONE, 0)
        logvar = ops.LinearFn.apply(z, kn.k_logvar.weight, kn.k_logvar.bias, ops.ACT_NONE, 0)
        latent = mu_l + torch.randn_like(mu_l) * torch.exp(0.5 * logvar)
        kp = ops.LinearFn.apply(latent.contiguous(), kn.k_project.weight, kn.k_project.bias, ops.ACT_NONE, 0).reshape(-1)
        _, mu, sd = self._norm_deg(deg, consts, 0.0)
        return torch.relu(kp * sd + mu) + 1.0

    def _scores_adjacency(self, x, in_adj):
        """debug_step 0 (dgm.py:1202-1209), debug_step 1 (dgm.py:1240-1246) and k-select `edge_p-cdf` (dgm.py:1368-1401, whose
        scatter puts the UNSORTED probabilities back and whose learned k never reaches the output): the adjacency IS the edge
        probability of every stored entry of in_adj -> CsrAdjacency on its pattern.  Perturbed probabilities (debug_step 1 /
        edge_p-cdf with perturb_edge_prob) are dense in the reference (every non-edge becomes 1e-8 exp(G) > 0) and are not
        produced; dgg_hard turns these outputs into all-ones matrices there (dgm.py:1301-1306 with idxs=None)."""
        if self.hard:
            raise NotImplementedError("dgg_hard with debug_step 0/1 or edge_p-cdf is a dense all-ones matrix in the reference")
        if self.args.perturb_edge_prob and self.args.debug_step != 0:
            raise NotImplementedError("perturbed edge probabilities are returned as a DENSE [N,N] matrix by the reference "
                                      "(debug_step 1 / edge_p-cdf with perturb_edge_prob=True)")
        if isinstance(in_adj, AllPairs):
            raise NotImplementedError("all-pairs candidates: the raw probability matrix is dense [N,N]")
        if isinstance(in_adj, (EllAdjacency, CsrAdjacency)):
            in_adj = in_adj.to_sparse().detach()
        in_adj = in_adj.coalesce()
        pattern = csr_pattern(in_adj)
        _, _, deg = csr_candidates(in_adj)
        We, be = self.node_encode_for_edges[0].weight, self.node_encode_for_edges[0].bias
        cfg = dict(cand=pattern, t=ops.T_DIST)
        if self.edge_prob_net_mode == "u-v-dist":
            p = _DGGScoresFn.apply(x, None, None, We, be, None, None, None, None, None, None, None, cfg)
        else:
            mlp, ex_in = self._edge_mlp_terms(in_adj.values().to(torch.float32))
            cfg.update(ex_mode=mlp["ex_mode"], t_ex=mlp["t_ex"], act=mlp["act"])
            p = _DGGScoresFn.apply(x, deg, ex_in, We, be, mlp["Wcat"], mlp["wdu"], mlp["wdv"], mlp["wex"], mlp["b1"], mlp["w2"],
                                   mlp["b2"], cfg)
        return CsrAdjacency(pattern[0], pattern[1], pattern[2], p, x.shape[0])

    def forward(self, x, in_adj, noise=True, writer=None, epoch=None):
        """x [N,dim] fp32 on the GPU; in_adj: sparse COO [N,N] (coalesced, self loops added by the caller) whose
        stored entries are the candidate edges, or `AllPairs(prior_degree)`.  `noise` is accepted and ignored
        exactly like the reference (perturbation is gated by args.perturb_edge_prob only, dgm.py:1211)."""
        assert x.ndim == 2 and len(in_adj.shape) == 2
        if self.edge_prob_net_mode != "u-v-dist" and self.edge_prob_net_mode not in _EDGE_MLP_MODES:
            raise Exception("mode not found")
        if self.k_net_mode not in ("x", "gcn-x-deg", "input_deg", "learn_normalized_degree"):
            # "pass" returns k = None, which the reference's own select_top_k cannot consume (dgm.py:1485, 1412)
            raise NotImplementedError(f"k-net mode {self.k_net_mode!r}: the HIP path implements 'x', 'gcn-x-deg', 'input_deg' "
                                      "and 'learn_normalized_degree'")
        if self.k_select_mode not in ("k_times_edge_prob", "k_only", "edge_p-cdf"):
            raise Exception("mode not found")
        if self.args.debug_step in (0, 1) or self.k_select_mode == "edge_p-cdf":
            return self._scores_adjacency(x, in_adj)
        avals = erow = None
        if isinstance(in_adj, AllPairs):
            if self.edge_prob_net_mode != "u-v-dist":
                raise NotImplementedError("all-pairs candidates are defined for 'u-v-dist' only (the other scorers read "
                                          "per-edge features of in_adj, dgm.py:1628-1725)")
            cand, deg = None, in_adj.prior_degree
        else:
            if isinstance(in_adj, EllAdjacency):     # dgg_adj_input != "input_adj": previous learned graph
                in_adj = in_adj.to_sparse().detach()
            rowptr, col, deg = csr_candidates(in_adj)
            cand = (rowptr, col)
            if self.edge_prob_net_mode != "u-v-dist":
                in_adj = in_adj.coalesce()
                erow, avals = in_adj.indices()[0].to(torch.int32), in_adj.values().to(torch.float32)
        noise_mode, G, seed = self._noise_cfg()
        literal = bool(self.hard and getattr(self.args, "dgg_hard_literal", False))
        if noise_mode == ops.NOISE_RANKED and cand is None and not literal:
            noise_mode = self._asym_generator_now(x, seed)       # the ranked search's depth is a property of the data: guarded
        if noise_mode == ops.NOISE_RANKED_SYM and (cand is not None or literal):
            noise_mode = ops.NOISE_HASH_SYM
        if noise_mode == ops.NOISE_RANKED and (cand is not None or literal):
            noise_mode = ops.NOISE_HASH              # edge-list candidates: every candidate is scored, per-pair hash noise
                                                     # (literal dgg_hard: the full ranking of a row needs per-pair noise as well)
        cfg = dict(cand=cand, K=self.ell_width, t=ops.T_DIST, noise_mode=noise_mode, G=G, seed=seed, algo=self.topk_algo,
                   sym_fallback=getattr(self.args, "dgg_sym_generator", "auto") != "ranked",
                   mode=ops.MODE_K_TIMES_EDGE_PROB if self.k_select_mode == "k_times_edge_prob" else ops.MODE_K_ONLY)
        if literal and (self.edge_prob_net_mode != "u-v-dist" or x.shape[0] > 8192):
            raise NotImplementedError("dgg_hard_literal: the literal return_hard_or_soft needs the full ranking of every row's N scores "
                                      "(u-v-dist scorer, N <= 8192); use the default straight-through dgg_hard otherwise")
        if self.hard and cfg["mode"] == ops.MODE_K_TIMES_EDGE_PROB and not literal:
            # dgg_hard: straight-through adjacency `(hard - soft).detach() + soft` with hard = the ramp mask at the selected
            # columns (the SDD class's definition, dgm.py:343-346; for k_only hard == soft).  The debug class's own
            # return_hard_or_soft (dgm.py:1294-1311) scatters an already-unsorted matrix through the sort permutation, which is
            # not a function of the graph; args.dgg_hard_literal=True reproduces it (small graphs).
            cfg["fwd_mode"] = ops.MODE_HARD_ST
        We, be = self.node_encode_for_edges[0].weight, self.node_encode_for_edges[0].bias
        kn = self.k_net
        xp_dual = None
        if self.k_net_mode in ("x", "gcn-x-deg"):
            Wk, bk = self.node_encode_for_k[0].weight, self.node_encode_for_k[0].bias
            if (self.edge_prob_net_mode == "u-v-dist" and not literal and We.shape[0] % 32 == 0 and Wk.shape[0] % 32 == 0
                    and We.shape[0] + Wk.shape[0] <= 256 and getattr(self.args, "dgg_fused_projections", True)):
                xp_dual, xk = _DualProjFn.apply(x, We, be, Wk, bk)       # both projections on one pass over x
            else:
                xk = ops.LinearFn.apply(x, Wk, bk, ops.ACT_LEAKY, 0)
            if self.k_net_mode == "gcn-x-deg":       # relu(normalize_adj(in_adj) @ xk @ k_W)   (dgm.py:1528-1540)
                if cand is None:
                    raise NotImplementedError("k-net mode 'gcn-x-deg' aggregates over the stored entries of in_adj")
                pat = csr_pattern(in_adj)
                nadj = CsrAdjacency(pat[0], pat[1], pat[2], in_adj.coalesce().values().float(), x.shape[0]).normalize()
                xk = ops.LinearFn.apply(nadj.matmul(xk), self.k_W, None, ops.ACT_RELU, 1)
            if getattr(self.args, "stochastic_k", False) and self.training:
                k = self._stochastic_k(torch.cat([xk, self._norm_deg(deg, None, 1e-5)[0].unsqueeze(1)], 1), deg, None, embed=True)
            else:
                k = _KnetFeatFn.apply(xk, deg, self.k_embed[0].weight, self.k_embed[0].bias, kn.k_mu.weight, kn.k_mu.bias,
                                      kn.k_project.weight, kn.k_project.bias, getattr(self, "gemm_dtype", None) == torch.bfloat16)
        elif getattr(self.args, "stochastic_k", False) and self.training:
            consts = (float(self.deg_mean), float(self.deg_std)) if self.k_net_mode == "input_deg" else None
            nd, _, _ = self._norm_deg(deg, consts, 1e-5 if consts is not None else 0.0)
            in3 = nd.unsqueeze(1) * self.input_degree_project.weight.reshape(1, -1) + self.input_degree_project.bias
            k = self._stochastic_k(in3, deg, consts, embed=False)
        else:
            consts = (float(self.deg_mean), float(self.deg_std)) if self.k_net_mode == "input_deg" else None
            k = _KnetDegFn.apply(deg, self.input_degree_project.weight, self.input_degree_project.bias, kn.k_mu.weight,
                                 kn.k_mu.bias, kn.k_project.weight, kn.k_project.bias, consts)
        if cand is not None and not literal and self._wide_rows(in_adj, rowptr, k):
            # rows wider than the ELL and learned degrees that may exceed it: the CSR form (no width limit)
            return self._csr_soft_adjacency(x, in_adj, k, noise_mode, G, seed, cfg["mode"])
        if cand is None and not literal and self._chunk_policy(noise_mode) and not _capturing():
            # learned degrees beyond the 64-rank list: chunked rows (same generator, same search, ceil(k_i + 8.5) + 1 ranks per row)
            lay = ops.chunk_layout(k.detach(), ncols=x.shape[0])          # (one readback: the chunk count sizes the arrays)
            if lay is not None and lay.wide:
                cfg["layout"] = lay
                cfg["wide_noise"] = {ops.NOISE_RANKED_SYM: ops.NOISE_HASH_SYM}.get(noise_mode, noise_mode)
                xp = xp_dual if xp_dual is not None else ops.LinearFn.apply(x, We, be, ops.ACT_LEAKY, 0)
                cfg["want_bwd"] = ops.backward_will_follow(xp, k)
                w, idx, val, rs = _DGGWideAdjFn.apply(xp, k, cfg)
                if writer is not None:
                    f = w.detach() if (cfg["mode"] == ops.MODE_K_ONLY or "fwd_mode" in cfg) else (w.detach() / val.clamp(min=1e-30))
                    fs = torch.zeros(x.shape[0], device=x.device).index_add_(0, lay.cnode.long(), f.sum(-1))
                    writer.add_scalar("values/first_k_std", fs.std(), epoch)
                    writer.add_scalar("values/first_k_mean", fs.mean(), epoch)
                return EllAdjacency(idx, w, x.shape[0], rs=rs, k=k.detach(), score=val, owner=self, layout=lay)
            chunk_checked = lay is not None
        else:
            chunk_checked = False
        if cand is None and not literal and self.edge_prob_net_mode == "u-v-dist" and not chunk_checked and self._allpairs_wide(x.shape[0], k):
            # learned degrees beyond the list on all-pairs candidates: every column ranked, CSR form (per-pair hash noise: the ranked
            # generators produce a row's noise in decreasing order for a search that stops early -- here nothing stops early)
            nm = {ops.NOISE_RANKED: ops.NOISE_HASH, ops.NOISE_RANKED_SYM: ops.NOISE_HASH_SYM}.get(noise_mode, noise_mode)
            return self._csr_soft_adjacency(x, None, k, nm, G, seed, cfg["mode"], pattern=self._allpairs_pattern(x.shape[0], x.device), deg=deg)
        if self.edge_prob_net_mode == "u-v-dist" and xp_dual is not None:
            cfg["want_bwd"] = ops.backward_will_follow(xp_dual, k)
            w, idx, val, rs = _DGGSoftAdjXpFn.apply(xp_dual, k, cfg)
        elif self.edge_prob_net_mode == "u-v-dist":
            cfg["want_bwd"] = ops.backward_will_follow(x, k, We, be)
            w, idx, val, rs = _DGGSoftAdjFn.apply(x, k, We, be, cfg)
        else:
            mlp, ex_in = self._edge_mlp_terms(avals)
            cfg.update(cand=(rowptr, col, erow), ex_mode=mlp["ex_mode"], t_ex=mlp["t_ex"], act=mlp["act"])
            w, idx, val, rs = _DGGEdgeMlpAdjFn.apply(x, k, deg, ex_in, We, be, mlp["Wcat"], mlp["wdu"], mlp["wdv"], mlp["wex"],
                                                     mlp["b1"], mlp["w2"], mlp["b2"], cfg)
        k = k.detach()
        if not chunk_checked:                    # (chunk_checked: the layout just read back says every row fits the list)
            self._track_overflow(k, None if cand is None else (rowptr[1:] - rowptr[:-1]))
        if cfg.get("rsym_fell_back"):
            self._sym_switch()
        self._note_rsym(cfg, x.shape[0])
        if writer is not None:   # the two scalars the reference logs from inside the DGG (dgm.py:1259-1261)
            f = w.detach() if (cfg["mode"] == ops.MODE_K_ONLY or "fwd_mode" in cfg) else (w.detach() / val.clamp(min=1e-30))
            writer.add_scalar("values/first_k_std", f.sum(-1).std(), epoch)
            writer.add_scalar("values/first_k_mean", f.sum(-1).mean(), epoch)
        if literal:
            # reference dgm.py:1294-1311 to the letter: ones where the RANK of a column equals the column INDEX of a strong neighbour
            xp = ops.linear_fwd(x.detach(), We.detach(), be.detach(), ops.ACT_LEAKY)
            hval, hidx = ops.LiteralHardFn.apply(w, idx, xp, cand, cfg["t"], noise_mode, G, seed)
            return EllAdjacency(hidx, hval, x.shape[0], owner=self)
        return EllAdjacency(idx, w, x.shape[0], rs=rs, k=k, score=val, part=cfg.get("part"), owner=self)


class _DGGClassFn(torch.autograd.Function):
    """Edge ranks + degree + ramp of the `DGG` class (reference dgm.py:1781-1812) on the CSR pattern of `adj`.
    W_e (x_u - x_v) = Q_u - Q_v with Q = xe W_e^T, so the scorer is dgg_edge_mlp_fwd with AB = xe [W_e | -W_e]^T,
    w2 = 1, b2 = 0 (edge_feat.sum(-1), dgm.py:1786)."""

    @staticmethod
    def forward(ctx, xe, We, be, wdd, bdd, pattern, noise=None, kcut=None):
        rowptr, col, erow = pattern
        h = We.shape[0]
        Wcat = torch.cat([We, -We], 0)
        AB = ops.linear_fwd(xe, Wcat, None, ops.ACT_NONE)
        ones, zero = torch.ones(h, device=xe.device), torch.zeros(1, device=xe.device)
        p, _ = ops.edge_mlp_fwd(AB, xe, erow, col, None, None, 0, 0.0, None, None, None, be, ones, zero, ops.ACT_LEAKY)
        # DGG_Ablations (dgm.py:1930-1933): a second sigmoid over rank + U(-1,1) noise
        p2 = p if noise is None else ops.csr_noisy_sigmoid_fwd(p, noise)
        if kcut is None:
            out, S, k, pos = ops.csr_rank_ramp_fwd(p2, rowptr, col, wdd.reshape(-1), bdd)
        else:                                                       # fixed k (dgm.py:1940-1942): no degree estimator
            out, pos = ops.csr_rank_cut_fwd(p2, rowptr, col, kcut)
            S = k = torch.full((xe.shape[0],), float(kcut), device=xe.device)
        ctx.noisy, ctx.kcut = noise is not None, kcut
        ctx.save_for_backward(xe, We, be, wdd, bdd, rowptr, col, Wcat, AB, ones, zero, p, p2, S, k, pos)
        ctx.mark_non_differentiable(k)
        return out, k

    @staticmethod
    def backward(ctx, g, _dk):
        xe, We, be, wdd, bdd, rowptr, col, Wcat, AB, ones, zero, p, p2, S, k, pos = ctx.saved_tensors
        h = We.shape[0]
        if ctx.kcut is None:
            dp, dkz = ops.csr_rank_ramp_bwd(p2, rowptr, wdd.reshape(-1), bdd, S, k, pos, g.contiguous())
            dwdd, dbdd = torch.dot(dkz, S).reshape(wdd.shape), dkz.sum().reshape(bdd.shape)
        else:
            dp = ops.csr_rank_cut_bwd(pos, g.contiguous(), ctx.kcut)
            dwdd, dbdd = torch.zeros_like(wdd), torch.zeros_like(bdd)
        if ctx.noisy:
            dp = ops.csr_noisy_sigmoid_bwd(p2, dp)
        dAB, dpar, _ = ops.edge_mlp_bwd(AB, col, None, p, dp, None, None, None, None, None, be, ones, zero, ops.ACT_LEAKY, False,
                                        rowptr=rowptr)
        dxe, dWcat, _ = ops.linear_bwd(xe, Wcat, AB, dAB, ops.ACT_NONE, need_dx=True, need_db=False)
        return dxe, dWcat[:h] - dWcat[h:], dpar[3 * h:4 * h], dwdd, dbdd, None, None, None


class DGG(nn.Module):
    """Drop-in for the reference's `DGG` "for ICLR" (dgm.py:1730-1815), the generator behind the *_DGG_00 wrappers
    (model.py:1314-1433): x' = leaky(x W); rank_uv = sigmoid(sum_h leaky(W_e (x'_u - x'_v) + b_e)) on the stored entries of
    `adj`; k = leaky(Linear(1,1)(sum_v rank_uv)); every edge kept with weight rank * (ramp(position - k) + 1).
    Returns (CsrAdjacency with the pattern of `adj`, x')."""

    def __init__(self, in_dim=32, latent_dim=64, args=None):
        super().__init__()
        self.args = args
        self.node_encoder = nn.Sequential(nn.Linear(in_dim, latent_dim), nn.LeakyReLU())
        self.edge_encoder = nn.Sequential(nn.Linear(latent_dim + self.args.extra_edge_dim, latent_dim), nn.LeakyReLU())
        self.degree_decoder = nn.Sequential(nn.Linear(1, 1, bias=True), nn.LeakyReLU())

    def forward(self, x, adj, noise=True, writer=None, epoch=None):
        assert x.ndim == 2 and len(adj.shape) == 2
        assert self.edge_encoder[0].weight.shape[1] == self.node_encoder[0].weight.shape[0], \
            "DGG feeds x'_u - x'_v (latent_dim features) to edge_encoder: extra_edge_dim must be 0 (dgm.py:1784-1785)"
        if isinstance(adj, (CsrAdjacency, EllAdjacency)):
            adj = adj.to_sparse().detach()
        pattern = csr_pattern(adj)
        xe = ops.LinearFn.apply(x, self.node_encoder[0].weight, self.node_encoder[0].bias, ops.ACT_LEAKY, 0)
        out, k = _DGGClassFn.apply(xe, self.edge_encoder[0].weight, self.edge_encoder[0].bias, self.degree_decoder[0].weight,
                                   self.degree_decoder[0].bias, pattern)
        return CsrAdjacency(pattern[0], pattern[1], pattern[2], out, x.shape[0], k=k), xe


class DGG_Ablations(DGG):
    """Drop-in for the reference's `DGG_Ablations` (dgm.py:1876-1968; behind GCN_DGG_Ablations / GAT_DGG_Ablations,
    model.py:406-486, 1436-1561): the `DGG` generator with edge_rank = sigmoid(sigmoid(score) + noise), noise ~ U(-1,1) drawn per
    stored edge on every call (dgm.py:1930-1933), and an optional FIXED k: `k=int` keeps the k best-ranked edges of each row
    with their rank and zeroes the others (dgm.py:1940-1942) instead of the learned degree + ramp.
    `set_noise(t)` pins the noise tensor of the next call (parity tests); otherwise it comes from torch's generator like the
    reference's `torch.rand`."""

    _noise = None

    def set_noise(self, noise):
        self._noise = noise

    def forward(self, x, adj, k=None, writer=None, epoch=None):
        assert x.ndim == 2 and len(adj.shape) == 2
        assert self.edge_encoder[0].weight.shape[1] == self.node_encoder[0].weight.shape[0], \
            "DGG_Ablations feeds x'_u - x'_v (latent_dim features) to edge_encoder: extra_edge_dim must be 0 (dgm.py:1924-1927)"
        if isinstance(adj, (CsrAdjacency, EllAdjacency)):
            adj = adj.to_sparse().detach()
        pattern = csr_pattern(adj)
        E = pattern[1].shape[0]
        noise, self._noise = self._noise, None
        if noise is None:
            noise = torch.rand(E, device=x.device) * 2 - 1
        assert noise.shape == (E,)
        xe = ops.LinearFn.apply(x, self.node_encoder[0].weight, self.node_encoder[0].bias, ops.ACT_LEAKY, 0)
        out, kk = _DGGClassFn.apply(xe, self.edge_encoder[0].weight, self.edge_encoder[0].bias, self.degree_decoder[0].weight,
                                    self.degree_decoder[0].bias, pattern, noise.to(torch.float32).contiguous(),
                                    None if k is None else int(k))
        return CsrAdjacency(pattern[0], pattern[1], pattern[2], out, x.shape[0], k=kk), xe


# ---------------------------------------------------------------------------------------------------------------------------
# dense all-pairs alternates.  Dense [B,N,N] in and out as in the reference; the softmax over all N
# columns makes them O(N^2) by definition -- batches of small graphs, N <= 8192.
# ---------------------------------------------------------------------------------------------------------------------------
class _KMuProject(nn.Module):
    """LearnableKEncoder as the SDD class builds it (dgm.py:248-251, 2024-2063): k_mu / k_logvar Linear(in_dim, latent_dim),
    k_project Linear(latent_dim, 1); deterministic path k = k_project(k_mu(x))"""

    def __init__(self, in_dim, latent_dim):
        super().__init__()
        self.k_mu = nn.Linear(in_dim, latent_dim)
        self.k_logvar = nn.Linear(in_dim, latent_dim)
        self.k_project = nn.Linear(latent_dim, 1)

    def forward(self, x2):
        lat = ops.LinearFn.apply(x2, self.k_mu.weight, self.k_mu.bias, ops.ACT_NONE, 0)
        return ops.LinearFn.apply(lat, self.k_project.weight, self.k_project.bias, ops.ACT_NONE, 0)


class DGG_LearnableK_SDD(nn.Module):
    """Drop-in for the reference's `DGG_LearnableK_SDD` (dgm.py:185-351) on its runnable configuration: dist_fn="metric",
    k_net_input="raw", noise=False (`noise=True` calls gumbel_sample with three arguments and raises in the reference,
    SURVEY 2.2; dist_fn="mlp" ends in nn.Softmax over a size-1 dimension, i.e. a constant 1).
    forward(x [B,N,in_dim], temp, noise=False) -> (adj [B,N,N], k [B,N,1]):
        xq = softmax(leaky(x W + b)); y = softmax_j(log(exp(-t |xq_i - xq_j|)) / temp); rows sorted descending;
        k = k_net(x) + k_bias; first_k = sigmoid((hs_start - interval r) + interval (k - 1)); adj = y first_k at the sorted
        columns; hard: (first_k - adj).detach() + adj (dgm.py:343-346).
    Distances are direct differences (torch.cdist's matmul form above 25 rows adds ~1e-4 of cancellation noise to near-zero
    distances in fp32; parity is pinned on the reference evaluated in float64)."""

    def __init__(self, in_dim=32, latent_dim=64, k_bias=1.0, hard=False, self_loops_noise=False, dist_fn="metric", k_net_input="raw",
                 hs_start=2, hs_end=-5, n_agents=None, learn_k_bias=None):
        super().__init__()
        if dist_fn != "metric" or k_net_input != "raw":
            raise Exception("DGG_LearnableK_SDD: only dist_fn='metric', k_net_input='raw' run in the reference (SURVEY 2.2)")
        torch.manual_seed(0)                                                       # dgm.py:207
        self.in_dim, self.latent_dim, self.hard, self.self_loops_noise = in_dim, latent_dim, hard, self_loops_noise
        self.dist_fn, self.k_net_input = dist_fn, k_net_input
        self.input_project = nn.Sequential(nn.Linear(in_dim, latent_dim), nn.LeakyReLU(), nn.Softmax(dim=-1))
        self.t = nn.Parameter(torch.ones(1))
        interval = hs_start - hs_end
        self.register_buffer("interval", torch.tensor(interval))
        self.register_buffer("k_bias", torch.tensor(k_bias))
        self.register_buffer("hs_start", torch.tensor(hs_start))
        self.register_buffer("hs_end", torch.tensor(hs_end))
        self.k_net = _KMuProject(in_dim, latent_dim)
        self._consts = (float(hs_start), float(interval))           # host copies of the ramp buffers (no device sync per forward)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._consts = (float(self.hs_start), float(self.interval))

    def forward(self, x, temp, noise=False):
        if noise:
            raise Exception("DGG_LearnableK_SDD: noise=True is not runnable in the reference (gumbel_sample arity, dgm.py:295)")
        assert x.ndim == 3
        B, N, d = x.shape
        x2 = x.reshape(B * N, d)
        z = ops.LinearFn.apply(x2, self.input_project[0].weight, self.input_project[0].bias, ops.ACT_LEAKY, 0)
        xq = ops.FeatSoftmaxFn.apply(z).reshape(B, N, self.latent_dim)
        k = self.k_net(x2).reshape(B, N) + self.k_bias
        adj = ops.DenseRowsFn.apply(xq, self.t, k, float(temp), ops.RAMP_SDD, 0, self._consts[0], self._consts[1], bool(self.hard))
        return adj, k.unsqueeze(-1)


class DGG_StraightThrough(nn.Module):
    """Drop-in for the reference's `DGG_StraightThrough` (dgm.py:103-182) with dist_fn="metric", noise=False:
    y = softmax_j(log(exp(-t |x_i - x_j|)) / temp) on the RAW inputs (dgm.py:157); hard: ones at the k largest entries of each
    row, gradient of y (dgm.py:83-98); soft: y.  `project` is constructed (state_dict parity) but, as in the reference, its
    output does not reach the metric distance."""

    def __init__(self, in_dim=32, latent_dim=64, k=3, hard=True, self_loops_noise=False, dist_fn="mlp"):
        super().__init__()
        if dist_fn != "metric":
            raise Exception("DGG_StraightThrough: dist_fn='mlp' ends in nn.Softmax over a size-1 dimension (constant 1); use 'metric'")
        self.in_dim, self.latent_dim, self.k, self.hard, self.self_loops_noise, self.dist_fn = in_dim, latent_dim, k, hard, self_loops_noise, dist_fn
        self.project = nn.Sequential(nn.Linear(in_dim, latent_dim), nn.LeakyReLU(), nn.Softmax(dim=-1))
        self.t = nn.Parameter(torch.ones(1))

    def forward(self, x, temp, noise=False):
        if noise:
            raise Exception("DGG_StraightThrough: noise=True is not runnable in the reference (gumbel_sample arity, dgm.py:80)")
        assert x.ndim == 3
        return ops.DenseRowsFn.apply(x, self.t, None, float(temp), ops.RAMP_TOPK, int(self.k), 0.0, 0.0, bool(self.hard))
