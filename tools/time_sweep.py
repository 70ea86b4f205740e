#!/usr/bin/env python3
"""Times the unperturbed all-pairs path (noise_mode 0) on one GPU and prints its candidate statistics (diagnostic):
   DGG_SWEEP_STATS=1 python tools/time_sweep.py [N] [h] [algos, e.g. 2,5]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgg_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
h = int(sys.argv[2]) if len(sys.argv) > 2 else 64
algos = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2]
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = torch.randn(N, 128, generator=g).to(dev)
W = (torch.randn(h, 128, generator=g) * 0.1).to(dev)
b = (torch.randn(h, generator=g) * 0.1).to(dev)
xp = ops.linear_fwd(x, W, b, ops.ACT_LEAKY)
res = {}
for algo in algos:
    for _ in range(2):
        idx, val, ws = ops.allpairs_topk(xp, 64, noise_mode=ops.NOISE_NONE, algo=algo, return_ws=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        idx, val, ws = ops.allpairs_topk(xp, 64, noise_mode=ops.NOISE_NONE, algo=algo, return_ws=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    line = f"algo {algo}: {ms:8.3f} ms per call (N={N}, h={h}; includes workspace allocation)"
    if algo == 2:
        nfail, st = ops.fast_path_failed_rows(ws, N, h, stats=True)
        line += f"  fallback rows {nfail}  per row: phase-A hits {st[0] / N:.1f}, kept {st[1] / N:.1f}, phase-B hits {st[2] / N:.1f}"
    print(line)
    res[algo] = (idx.clone(), val.clone())
if len(res) > 1:
    a, b2 = list(res.values())[:2]
    print("identical lists:", bool(torch.equal(a[0], b2[0]) and torch.equal(a[1], b2[1])))
