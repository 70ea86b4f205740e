#!/usr/bin/env python3
"""Times DGG_LearnableK_debug on all-pairs candidates in its two forms (diagnostic): the 64-wide list (learned degrees inside it) and
the complete pattern in CSR form (degrees beyond it): python tools/time_allpairs_csr.py [N ...]"""
import os
import sys
import time
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dgg_amd  # noqa: E402

dev = torch.device("cuda", 0)
for N in [int(a) for a in sys.argv[1:]] or [1000, 3000, 8192]:
    d = h = 64
    for policy, prior in (("ell", 30.0), ("csr", 90.0)):
        args = Namespace(extra_edge_dim=0, extra_k_dim=1, dgg_hard=False, deg_mean=3.899, deg_std=5.288, dgg_mode_edge_net="u-v-dist",
                         dgg_mode_k_net="x", dgg_mode_k_select="k_times_edge_prob", debug_step=3, perturb_edge_prob=True,
                         symmetric_noise=False, stochastic_k=False, dgg_adj_input="input_adj", n_dgg_layers=1, dgg_wide_rows=policy)
        torch.manual_seed(0)
        m = dgg_amd.GCN_DGG(nfeat=d, nhidden=h, nclass=7, args=args).to(dev).train()
        with torch.no_grad():
            m.dggs[0].k_net.k_project.weight.mul_(0.1)
        x = torch.randn(N, d, device=dev)
        A = dgg_amd.AllPairs(torch.full((N,), prior, device=dev))
        y = torch.randint(0, 7, (N,), device=dev)

        def step():
            for p_ in m.parameters():
                p_.grad = None
            logp, adj, _ = m(x, A)
            torch.nn.functional.nll_loss(logp, y).backward()
            return adj

        for _ in range(3):
            adj = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            adj = step()
        torch.cuda.synchronize()
        print(f"N={N} {type(adj).__name__:13s} k~{float(adj.k.mean()):.0f}: GCN_DGG forward + backward {1e3 * (time.perf_counter() - t0) / 5:.2f} ms "
              f"(peak memory {torch.cuda.max_memory_allocated() / 2**20:.0f} MiB)")
        torch.cuda.reset_peak_memory_stats()
