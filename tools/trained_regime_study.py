#!/usr/bin/env python3
"""Which regime of the ranked search do TRAINED generators live in?  

The asymmetric ranked search visits about L exp(D / 0.3) ranks of a row, D = the spread of 0.05 ||xp_i - xp_j|| the row sees
(DESIGN.md section 6): 2 blocks of 64 ranks on unit-scale random features, hundreds once the latent distances spread over several
noise scales.  Here GCN_DGG (all-pairs candidates, the small-graph script's optimiser groups: Adam lr 0.01, weight decay 0.01 / 5e-4,
train_small_graphs.py:399-418) is trained for 200 epochs on two synthetic node-classification sets and the search's walk is
measured with dgg_allpairs_ranked_probe on the model's CURRENT latent features after every 10th epoch:

  cora-shaped   2 708 nodes, 1 433 sparse bag-of-words features (18 words per node from class-dependent vocabularies), row-
                normalised as T.NormalizeFeatures does (train_small_graphs.py:345), 7 classes, 140 training labels
  randn         8 192 nodes, 128 standard-normal features, 7 classes read off the first features plus noise, 5 % training labels

Output: a table (epoch, loss, train / validation accuracy, learned k mean / max, blocks / gathered / scored candidates per row, the
spread of 0.05 ||xp_i - xp_j|| over random pairs in units of the noise scale 0.3) on stdout and as JSON under gpurun_out/."""
import json
import os
import sys
from argparse import Namespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgg_amd  # noqa: E402
from dgg_amd import ops  # noqa: E402


def cora_shaped(g):
    N, d, C, words = 2708, 1433, 7, 18
    y = torch.randint(0, C, (N,), generator=g)
    base = torch.rand(C, d, generator=g) ** 6                      # every class has its own heavy-tailed vocabulary
    common = torch.rand(d, generator=g) ** 3
    p = 0.6 * base / base.sum(1, keepdim=True) + 0.4 * (common / common.sum())[None, :]
    x = torch.zeros(N, d)
    for i in range(N):
        w_ = torch.multinomial(p[y[i]], words, replacement=False, generator=g)
        x[i, w_] = 1.0
    x = x / x.sum(1, keepdim=True)
    train = torch.zeros(N, dtype=torch.bool)
    for c in range(C):
        train[(y == c).nonzero()[:20, 0]] = True
    prior = torch.exp(torch.randn(N, generator=g) * 0.9 + 0.95).clamp(1.0, 168.0)     # Cora's degree statistics (mean 3.9, heavy tail)
    return x, y, train, prior


def randn_set(g):
    N, d, C = 8192, 128, 7
    x = torch.randn(N, d, generator=g)
    y = (x[:, :C] + 0.3 * torch.randn(N, C, generator=g)).argmax(1)
    train = torch.rand(N, generator=g) < 0.05
    prior = 24 + 16 * torch.rand(N, generator=g)
    return x, y, train, prior


def run(name, make, dev, epochs=200, every=10):
    g = torch.Generator().manual_seed(5)
    x, y, train, prior = make(g)
    N, d = x.shape
    h = 64
    args = Namespace(extra_edge_dim=0, extra_k_dim=1, dgg_hard=False, deg_mean=float(prior.mean()), deg_std=float(prior.std()),
                     dgg_mode_edge_net="u-v-dist", dgg_mode_k_net="x", dgg_mode_k_select="k_times_edge_prob", debug_step=3,
                     perturb_edge_prob=True, symmetric_noise=False, stochastic_k=False, dgg_adj_input="input_adj", n_dgg_layers=1,
                     dgg_asym_generator="ranked")
    torch.manual_seed(11)
    m = dgg_amd.GCN_DGG(nfeat=d, nhidden=h, nclass=7, args=args).to(dev)
    opt = torch.optim.Adam([{"params": m.params1, "weight_decay": 0.01}, {"params": m.params2, "weight_decay": 5e-4}], lr=0.01)
    x, y, train = x.to(dev), y.to(dev), train.to(dev)
    A = dgg_amd.AllPairs(prior.to(dev))
    val = ~train
    rows = []
    for ep in range(epochs + 1):
        m.train()
        opt.zero_grad()
        logp, adj, _ = m(x, A)
        if not torch.isfinite(adj.k).all():
            print(f"{name}: learned degrees not finite at epoch {ep}")
            break
        loss = torch.nn.functional.nll_loss(logp[train], y[train])
        loss.backward()
        m.dggs[0].check_ell_bound()
        opt.step()
        if ep % every == 0:
            with torch.no_grad():
                dg = m.dggs[0]
                xp = ops.linear_fwd(x, dg.node_encode_for_edges[0].weight.detach(), dg.node_encode_for_edges[0].bias.detach(), ops.ACT_LEAKY)
                pr = ops.ranked_probe(xp, None, ops.T_DIST, (1234, ep))
                i_ = torch.randint(0, N, (20000,), device=dev)
                j_ = torch.randint(0, N, (20000,), device=dev)
                sd = 0.05 * (xp[i_] - xp[j_]).norm(dim=1)
                pred = logp.argmax(1)
                rows.append(dict(epoch=ep, loss=float(loss), acc_train=float((pred[train] == y[train]).float().mean()),
                                 acc_val=float((pred[val] == y[val]).float().mean()), k_mean=float(adj.k.mean()), k_max=float(adj.k.max()),
                                 chunked=type(adj).__name__ + ("" if getattr(adj, "layout", None) is None else " (chunked)"), blocks=pr["blocks_per_row"], gathered=pr["gathered_per_row"],
                                 scored=pr["scored_per_row"], max_blocks=pr["max_blocks"],
                                 spread_over_noise_scale=float((sd.quantile(0.95) - sd.quantile(0.05)) / 0.3)))
    print(f"\n### {name}: N = {N}, d = {d}")
    print("| epoch | loss | acc train / val | k mean / max | adjacency | blocks / gathered / scored per row (max blocks) | spread of 0.05·dist (5-95 %) / 0.3 |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['epoch']} | {r['loss']:.3f} | {r['acc_train']:.2f} / {r['acc_val']:.2f} | {r['k_mean']:.1f} / {r['k_max']:.0f} | "
              f"{r['chunked']} | {r['blocks']:.2f} / {r['gathered']:.0f} / {r['scored']:.0f} ({r['max_blocks']}) | "
              f"{r['spread_over_noise_scale']:.2f} |")
    return rows


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    out = {"cora_shaped": run("cora-shaped", cora_shaped, dev), "randn": run("randn", randn_set, dev)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r05_trained_regime_study.json"), "w") as f:
        json.dump(out, f, indent=1)
