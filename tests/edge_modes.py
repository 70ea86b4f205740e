"""Test-side decomposition of the reference's edge-MLP scorers (dgm.py:1628-1725) into the per-node / per-edge form
of the oracle (oracle/dgg_oracle.c, "edge-MLP scorers"):  z = A_u + B_v + deg terms + ex * wex + b1.

    mode          first layer                                   extras                         hidden  act
    u-v-deg       edge_encode.0 [h, 2h+2] on [u, v, du, dv]     raw degrees (folded per node)  h       leaky
    u-v-A_uv      edge_encode.0 [h, 2h+1] on [u, v, a_uv]       ex = in_adj value              h       leaky
    u-v-deg-dist  edge_encode.0 [h, 2h+3] on [u,v,du,dv,e^-d]   degrees + ex = exp(-||u-v||)   h       leaky
    edge_conv     theta(v-u) + phi(u)  ->  (phi-theta) u + theta v, bias b_theta + b_phi       h/2     none
    A_uv          adj_project(a_uv)    ->  hidden width 1, A = B = 0, wex = w, w2 = 1          1       none
"""
import numpy as np

EDGE_MLP_MODES = ("u-v-deg", "u-v-A_uv", "u-v-deg-dist", "edge_conv", "A_uv")


def decompose(P, mode, h):
    """P(name) -> numpy parameter.  Returns the oracle's argument set."""
    f = np.float32
    if mode in ("u-v-deg", "u-v-A_uv", "u-v-deg-dist"):
        W0, b1 = P("edge_encode.0.weight"), P("edge_encode.0.bias")
        d = dict(hw=h, Wcat=np.concatenate([W0[:, :h], W0[:, h:2 * h]], 0), b1=b1, w2=P("edge_encode.2.weight")[0],
                 b2=float(P("edge_encode.2.bias")[0]), act=1, wdu=None, wdv=None, wex=None, ex_mode=0, t_ex=0.0)
        if mode != "u-v-A_uv":
            d["wdu"], d["wdv"] = W0[:, 2 * h].copy(), W0[:, 2 * h + 1].copy()
        if mode == "u-v-A_uv":
            d["wex"], d["ex_mode"] = W0[:, 2 * h].copy(), 1
        if mode == "u-v-deg-dist":
            d["wex"], d["ex_mode"], d["t_ex"] = W0[:, 2 * h + 2].copy(), 2, -1.0
        return d
    if mode == "edge_conv":
        Th, Ph = P("edge_conv_theta.weight"), P("edge_conv_phi.weight")
        return dict(hw=h // 2, Wcat=np.concatenate([(Ph - Th).astype(f), Th], 0),
                    b1=(P("edge_conv_theta.bias") + P("edge_conv_phi.bias")).astype(f), w2=P("edge_conv_encode.weight")[0],
                    b2=float(P("edge_conv_encode.bias")[0]), act=0, wdu=None, wdv=None, wex=None, ex_mode=0, t_ex=0.0)
    if mode == "A_uv":
        return dict(hw=1, Wcat=np.zeros((2, h), f), b1=P("adj_project.bias").astype(f), w2=np.ones(1, f), b2=0.0, act=0,
                    wdu=None, wdv=None, wex=P("adj_project.weight")[0].astype(f), ex_mode=1, t_ex=0.0)
    raise ValueError(mode)


def assemble_grads(mode, h, dWcat, dpar):
    """oracle-form gradients -> gradients of the reference's parameters (by state_dict name)."""
    hw = dWcat.shape[0] // 2
    dwdu, dwdv, dwex, db1, dw2, db2 = (dpar[0:hw], dpar[hw:2 * hw], dpar[2 * hw:3 * hw], dpar[3 * hw:4 * hw],
                                       dpar[4 * hw:5 * hw], dpar[5 * hw:5 * hw + 1])
    g = {}
    if mode in ("u-v-deg", "u-v-A_uv", "u-v-deg-dist"):
        cols = [dWcat[:hw], dWcat[hw:]]
        if mode != "u-v-A_uv":
            cols += [dwdu[:, None], dwdv[:, None]]
        if mode != "u-v-deg":
            cols += [dwex[:, None]]
        g["edge_encode.0.weight"] = np.concatenate(cols, 1)
        g["edge_encode.0.bias"], g["edge_encode.2.weight"], g["edge_encode.2.bias"] = db1, dw2[None, :], db2
    elif mode == "edge_conv":
        g["edge_conv_phi.weight"] = dWcat[:hw]
        g["edge_conv_theta.weight"] = dWcat[hw:] - dWcat[:hw]
        g["edge_conv_phi.bias"] = g["edge_conv_theta.bias"] = db1
        g["edge_conv_encode.weight"], g["edge_conv_encode.bias"] = dw2[None, :], db2
    else:
        g["adj_project.weight"], g["adj_project.bias"] = dwex.reshape(1, 1), db1
    return g
