#!/usr/bin/env python3
"""Randomised cross-check of the fused layer against the separate modules (diagnostic, one GPU): 36 GCN_DGG models over the six scorers,
both k-select modes, perturbation on / off, symmetric / asymmetric noise and random shapes; outputs must be identical, every parameter
gradient within 5e-4 of max:  python tools/stress_fused_layer.py"""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from argparse import Namespace
import dgg_amd
from test_parallel_gloo import random_candidates
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
modes = ["u-v-dist", "u-v-deg", "u-v-A_uv", "u-v-deg-dist", "edge_conv", "A_uv"]
bad = 0
for trial in range(36):
    N = int(rng.integers(40, 700)); h = int(rng.choice([16, 32, 64])); d = int(rng.integers(h, 3 * h)); C = int(rng.integers(2, 9))
    mode = modes[trial % len(modes)]
    perturb = bool(rng.integers(0, 2)); sym = bool(rng.integers(0, 2)); ksel = ["k_times_edge_prob", "k_only"][int(rng.integers(0, 2))]
    args = Namespace(extra_edge_dim={"u-v-deg": 2, "u-v-deg-dist": 3, "u-v-A_uv": 1}.get(mode, 0), extra_k_dim=1, dgg_hard=False, deg_mean=3.899,
                     deg_std=5.288, dgg_mode_edge_net=mode, dgg_mode_k_net="x", dgg_mode_k_select=ksel, debug_step=3, perturb_edge_prob=perturb,
                     symmetric_noise=sym, stochastic_k=False, dgg_adj_input="input_adj", n_dgg_layers=1)
    torch.manual_seed(trial)
    m1 = dgg_amd.GCN_DGG(nfeat=d, nhidden=h, nclass=C, args=args).to(dev).eval()
    with torch.no_grad():
        m1.dggs[0].k_net.k_project.weight.mul_(0.1); m1.conv2.W.mul_(0.2)
    m2 = copy.deepcopy(m1)
    m2.dggs[0].args = Namespace(**dict(vars(args), dgg_fused_layer=False))
    for m in (m1, m2):
        m.dggs[0].set_seed(77 + trial, 5)
    x = torch.rand(N, d, generator=torch.Generator().manual_seed(trial)).to(dev)
    rowptr, col = random_candidates(N, seed=trial, hi=int(rng.integers(2, 30)))
    rows = torch.repeat_interleave(torch.arange(N), rowptr[1:] - rowptr[:-1])
    keep = rows != col
    A = torch.sparse_coo_tensor(torch.stack([rows[keep], col[keep].long()]), torch.rand(int(keep.sum())) + 0.5, (N, N)).coalesce().to(dev)
    y = torch.randint(0, C, (N,), generator=torch.Generator().manual_seed(2)).to(dev)
    outs = []
    for m in (m1, m2):
        logp, adj, _ = m(x, A)
        torch.nn.functional.nll_loss(logp, y).backward()
        outs.append((logp, adj))
    fused_used = m1.dggs[0].__dict__.get("_fused_layer") is not None
    e_out = float((outs[0][0] - outs[1][0]).detach().abs().max())
    worst = 0.0
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        if p2.grad is None:
            continue
        ref = p2.grad
        g = p1.grad if p1.grad is not None else torch.zeros_like(ref)
        worst = max(worst, float((g - ref).abs().max() / ref.abs().max().clamp_min(1e-30)))
    ok = e_out < 2e-5 and worst < 5e-4
    bad += not ok
    print(f"{trial:2d} N={N:3d} d={d:3d} h={h:2d} C={C} {mode:12s} {ksel:18s} pert={int(perturb)} sym={int(sym)} fused={int(fused_used)} out {e_out:.1e} grad {worst:.1e} {'ok' if ok else 'MISMATCH'}")
print("mismatches:", bad)
sys.exit(1 if bad else 0)
