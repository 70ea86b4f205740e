"""TEST / BASELINE INFRASTRUCTURE -- never imported by the product package.

Dense, reference-SHAPED restatement of the hot path in torch on the CPU: the same sequence of [N,N] tensor operations the
reference performs (one dense matrix per stage), written from the reference's behaviour, not copied from it.  It exists for
two things only:

  * `bench.py`'s `cpu_baseline` leg at the sizes BASELINE.md section 3 names (N = 2 708, 4 000 all-pairs, 19 717), where the
    dense formulation can still be allocated -- "the reference CPU path timed beside" the GPU number;
  * pinning: `tests/test_oracle_golden.py::test_dense_restatement_matches_reference_goldens` checks it against the goldens
    that `tests/golden/make_golden.py` produced by importing /root/reference (forward 1e-6, gradients 1e-5).

Stages and the reference lines they restate:
  edge probabilities  exp(-0.05 ||xp_u - xp_v||) on the candidate entries, 0 elsewhere        dgm.py:1607-1627
  perturbation        exp(log(p + 1e-8) + G), G ~ Gumbel(0, 0.3) (asymmetric / mirrored)        dgm.py:1211-1229, 14-29
  learned degree      k = relu(k_project(k_mu(leaky(k_embed([leaky(xW_k+b), nd])))) sd + mu) + 1  dgm.py:1562-1586, 2051-2063
  soft top-k          sort descending, ramp 1 - 0.5(1 + tanh(r - k)), multiply, scatter back     dgm.py:1402-1421
  normalisation       rs^-1/2 A rs^-1/2 with ROW sums on both sides (elementwise form: bit-identical to the reference's
                      diag @ A @ diag)                              model.py:1205-1219
  graph conv          relu((A x) W)                                                              model.py:594-598
"""
import time

import torch
import torch.nn.functional as F


def gumbel_noise(shape, symmetric=False, generator=None):
    """Gumbel(0, 0.3) = -0.3 log(-log U); symmetric: upper triangle mirrored, zero diagonal (dgm.py:1216-1223)"""
    U = torch.rand(shape, generator=generator).clamp_(min=torch.finfo(torch.float32).tiny, max=1 - 2 ** -24)
    G = -0.3 * torch.log(-torch.log(U))
    if symmetric:
        G = torch.triu(G, 1)
        G = G + G.t()
    return G


def learned_degree(x, deg, P):
    """k-net mode "x": per-node features + normalised prior degree -> k [N]"""
    mu, sd = deg.mean(), deg.std()
    nd = (deg - mu) / (sd + 1e-5)
    xk = F.leaky_relu(F.linear(x, P["Wk"], P["bk"]))
    z = F.leaky_relu(F.linear(torch.cat([xk, nd[:, None]], 1), P["W1"], P["b1"]))
    kp = F.linear(F.linear(z, P["Wmu"], P["bmu"]), P["Wp"].reshape(1, -1), P["bp"]).reshape(-1)
    return F.relu(kp * sd + mu) + 1.0


def dgg_dense(x, rows, cols, deg, P, G=None):
    """Soft adjacency [N,N] (dense) and k [N].  rows / cols (int64) list the candidate entries (None: every ordered pair);
    deg [N] is the prior degree (row sums of in_adj); G [N,N] the Gumbel sample or None (perturb_edge_prob = False)."""
    N = x.shape[0]
    xp = F.leaky_relu(F.linear(x, P["We"], P["be"]))
    if rows is None:
        dist = torch.cdist(xp, xp, compute_mode="donot_use_mm_for_euclid_dist")      # ||xp_u - xp_v|| for every pair
        p = torch.exp(-0.05 * dist)
    else:
        dist = torch.linalg.vector_norm(xp[rows] - xp[cols], dim=-1)
        p = torch.zeros((N, N), dtype=x.dtype).index_put((rows, cols), torch.exp(-0.05 * dist))
    if G is not None:
        p = torch.exp(torch.log(p + 1e-8) + G)
    k = learned_degree(x, deg, P)
    s, order = torch.sort(p, dim=-1, descending=True)
    ramp = 1 - 0.5 * (1 + torch.tanh(torch.arange(N, dtype=x.dtype)[None, :] - k[:, None]))
    A = torch.zeros_like(p).scatter(1, order, s * ramp)
    return A, k


def normalize_dense(A):
    d = A.sum(-1) ** -0.5
    return d[:, None] * A * d[None, :]


def gcn_dense(Ahat, x, W):
    return torch.relu((Ahat @ x) @ W)


def step(x, rows, cols, deg, P, G=None):
    """one forward + backward of DGG -> normalize -> GCNConv with a ones cotangent; returns (Z, k, grads dict)"""
    Pg = {k_: v.detach().clone().requires_grad_(True) for k_, v in P.items()}
    A, k = dgg_dense(x, rows, cols, deg, Pg, G)
    Z = gcn_dense(normalize_dense(A), x, Pg["Wc"])
    Z.sum().backward()
    return Z.detach(), k.detach(), {k_: v.grad for k_, v in Pg.items()}


def timed_step(x, rows, cols, deg, P, perturb=True, symmetric=False, iters=1):
    """wall time of `iters` steps, Gumbel sampling included (the reference samples [N,N] noise per forward)"""
    N = x.shape[0]
    t0 = time.perf_counter()
    for _ in range(iters):
        G = gumbel_noise((N, N), symmetric) if perturb else None
        Z, k, _ = step(x, rows, cols, deg, P, G)
    return (time.perf_counter() - t0) / iters, float(k.mean())
