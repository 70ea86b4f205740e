#!/usr/bin/env python3
"""Time the any-width chunked-row evaluators (ops.allpairs_topk_wide) per noise generator at N = 100 000, h = 64 for a given degree prior.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgg_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=100_000)
ap.add_argument("--k", type=float, default=130.0)
ap.add_argument("--tail", type=float, default=0.0, help="fraction of rows with k up to 30x the mean")
ap.add_argument("--modes", default="ranked,hash,hash_sym,none")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
N, h = a.nodes, 64
g = torch.Generator().manual_seed(1)
xp = (torch.randn(N, h, generator=g) * 0.8).to(dev)
k = a.k * (0.8 + 0.4 * torch.rand(N, generator=g))
if a.tail > 0:
    sel = torch.rand(N, generator=g) < a.tail
    k[sel] = a.k * (1 + 29 * torch.rand(int(sel.sum()), generator=g) ** 2)
k = k.to(dev)
lay = ops.chunk_layout(k, ncols=N)
print(f"N {N} k mean {float(k.mean()):.1f} max {float(k.max()):.1f}: {lay.chunks} chunks, widest row {lay.maxm}")
NM = {"ranked": ops.NOISE_RANKED, "hash": ops.NOISE_HASH, "hash_sym": ops.NOISE_HASH_SYM, "none": ops.NOISE_NONE}
for name in a.modes.split(","):
    ts = []
    for r in range(a.reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.allpairs_topk_wide(xp, k, lay, seed=(1, r), noise_mode=NM[name])
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{name:9s} {min(ts[1:]):9.3f} ms (first call {ts[0]:.3f})", flush=True)
