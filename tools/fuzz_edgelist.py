#!/usr/bin/env python3
"""Differential fuzz of the edge-list candidate entries (ops.edgelist_topk: candidates = the entries of in_adj, reference dgm.py:1607-1627 on
the graph's own edges) against the oracle: node counts from tiny to Pubmed's, every latent width the kernels specialise, degree laws
with empty rows, hubs beyond the 64-entry list, duplicate edges and self loops, awkward features, every counter-based noise setting.
Whole result bit for bit.  `python tools/fuzz_edgelist.py --minutes 3`"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgg_amd import ops  # noqa: E402
from oracle import oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=3.0)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
NM = {"none": (ops.NOISE_NONE, O.NOISE_NONE), "hash": (ops.NOISE_HASH, O.NOISE_HASH), "hash_sym": (ops.NOISE_HASH_SYM, O.NOISE_HASH_SYM)}
t_end = time.time() + 60.0 * a.minutes
case = fails = 0
while time.time() < t_end:
    case += 1
    N = int(rng.choice([1, 2, 63, 64, 65, 300, 1000, 2708, 5000, 19717]))
    h = int(rng.choice([8, 16, 24, 32, 64, 128, 200, 256, 2048] if N <= 5000 else [16, 32, 64, 128]))
    law = str(rng.choice(["sparse", "mixed", "hubs", "dense"]))
    if law == "sparse":
        deg = rng.poisson(3, N)
    elif law == "mixed":
        deg = rng.integers(0, 40, N)
    elif law == "hubs":
        deg = rng.poisson(4, N)
        hub = rng.random(N) < 0.01
        deg[hub] = rng.integers(65, min(max(N, 66), 900), int(hub.sum()))
    else:
        deg = rng.integers(50, 80, N)
    deg = np.minimum(deg, 4 * N + 4)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.integers(0, N, int(rowptr[-1])).astype(np.int32)            # duplicates and self loops occur by themselves
    fk = str(rng.choice(["randn", "zeros", "huge", "dups"]))
    if fk == "randn":
        xp = rng.standard_normal((N, h)).astype(np.float32) * float(rng.choice([0.1, 1.0, 3.0]))
    elif fk == "zeros":
        xp = np.zeros((N, h), np.float32)
    elif fk == "huge":
        xp = rng.standard_normal((N, h)).astype(np.float32) * 300.0
    else:
        xp = rng.standard_normal((max(N // 20, 1), h)).astype(np.float32)[rng.integers(0, max(N // 20, 1), N)]
    noise = str(rng.choice(list(NM)))
    seed = (int(rng.integers(0, 2 ** 31)), int(rng.integers(0, 2 ** 31)))
    desc = f"case {case}: N {N} h {h} degrees {law} (max {int(deg.max()) if N else 0}) features {fk} noise {noise} seed {seed}"
    try:
        gi, gv = ops.edgelist_topk(torch.from_numpy(xp).to(dev), torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev), 64, noise_mode=NM[noise][0], seed=seed)
        ri, rv = O.edgelist_topk(xp, rowptr, col, K=64, noise_mode=NM[noise][1], seed=seed)
        assert np.array_equal(gi.cpu().numpy(), ri), "indices differ (first row %d)" % int(np.argwhere((gi.cpu().numpy() != ri).any(1))[0][0])
        assert np.array_equal(gv.cpu().numpy(), rv), "scores differ"
        print(f"ok   {desc}", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL {desc}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{case} cases, {fails} failures")
sys.exit(1 if fails else 0)
