#!/usr/bin/env python3
"""GCN_DGG on all-pairs candidates at N = 100 000 with the small-graph script's optimiser groups (train_small_graphs.py:399-418), any of
the reference's noise settings: how the learned degrees (and the step time) evolve over `--steps` Adam steps.  Prints one line every
`--every` steps: step, loss, k mean / max, chunks of the widest row, seconds per step so far."""
import argparse
import os
import sys
import time
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgg_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=100_000)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--sym", type=int, default=0)
    ap.add_argument("--perturb", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, d, h, C = a.nodes, 128, 64, 7
    g = torch.Generator().manual_seed(5)
    deg = 24 + 16 * torch.rand(N, generator=g)
    x = torch.randn(N, d, generator=g)
    y = (x[:, :C] + 0.3 * torch.randn(N, C, generator=g)).argmax(1).to(dev)
    x = x.to(dev)
    A = dgg_amd.AllPairs(deg.to(dev))
    args = Namespace(extra_edge_dim=0, extra_k_dim=1, dgg_hard=False, deg_mean=3.899, deg_std=5.288, dgg_mode_edge_net="u-v-dist",
                     dgg_mode_k_net="x", dgg_mode_k_select="k_times_edge_prob", debug_step=3, perturb_edge_prob=bool(a.perturb),
                     symmetric_noise=bool(a.sym), stochastic_k=False, dgg_adj_input="input_adj", n_dgg_layers=1)
    torch.manual_seed(11)
    m = dgg_amd.GCN_DGG(nfeat=d, nhidden=h, nclass=C, args=args).to(dev).train()
    with torch.no_grad():
        m.dggs[0].k_net.k_project.weight.mul_(0.1)
    opt = torch.optim.Adam([{"params": m.params1, "weight_decay": 0.01}, {"params": m.params2, "weight_decay": 5e-4}], lr=0.01)
    torch.cuda.synchronize()
    t0 = tl = time.perf_counter()
    for step in range(a.steps):
        opt.zero_grad()
        logp, adj, _ = m(x, A)
        loss = torch.nn.functional.nll_loss(logp, y)
        loss.backward()
        m.dggs[0].check_ell_bound()
        opt.step()
        if step % a.every == 0 or step == a.steps - 1:
            torch.cuda.synchronize()
            now = time.perf_counter()
            k = adj.k
            print(f"step {step:4d} loss {float(loss):.4f} k mean {float(k.mean()):9.1f} max {float(k.max()):10.1f} "
                  f"widest row {None if adj.layout is None else adj.layout.maxm} chunks {None if adj.layout is None else adj.layout.chunks} "
                  f"type {type(adj).__name__}  {(now - tl) / max(1, a.every if step else 1):.3f} s/step, {now - t0:.1f} s total, "
                  f"mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
            tl = now


if __name__ == "__main__":
    main()
