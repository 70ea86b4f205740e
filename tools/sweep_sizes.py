#!/usr/bin/env python3
"""The unperturbed sweep (noise_mode 0) across node counts: time per call, rows handed to the exhaustive fallback, candidate sums.
`DGG_SWEEP_STATS=1 python tools/sweep_sizes.py [--klimit]`"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgg_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--klimit", action="store_true")
ap.add_argument("--sizes", default="8192,10000,12000,16000,19717,30000,50000,100000")
ap.add_argument("--h", type=int, default=64)
a = ap.parse_args()
dev = torch.device("cuda:0")
for N in [int(v) for v in a.sizes.split(",")]:
    g = torch.Generator().manual_seed(N)
    xp = (torch.randn(N, a.h, generator=g) * 0.7).to(dev)
    k = (24 + 17 * torch.rand(N, generator=g)).to(dev) if a.klimit else None
    ts, tx = [], []
    for r in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        idx, val, ws = ops.allpairs_topk(xp, 64, noise_mode=ops.NOISE_NONE, k_limit=k, return_ws=True, algo=2)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    for r in range(3 if N <= 30000 else 0):                  # every pair scored (algo 1): the alternative below the sweep's threshold
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        xi, xv = ops.allpairs_topk(xp, 64, noise_mode=ops.NOISE_NONE, k_limit=k, algo=1)
        e1.record()
        torch.cuda.synchronize()
        tx.append(e0.elapsed_time(e1))
        assert torch.equal(xi, idx) and torch.equal(xv, val)
    nfail, st = ops.fast_path_failed_rows(ws, N, a.h, stats=True)
    print(f"N {N:7d} h {a.h}: {np.median(ts[1:]):8.3f} ms (exhaustive {min(tx) if tx else float('nan'):8.3f})   fallback rows {nfail:6d} ({100.0 * nfail / N:5.1f} %)   A hits/row {st[0] / N:7.1f} kept {st[1] / N:6.1f} B hits/row {st[2] / N:7.1f}  why {st[3:]}", flush=True)
