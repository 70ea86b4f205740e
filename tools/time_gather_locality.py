#!/usr/bin/env python3
"""How much of the aggregation's time is the gather's distance from the L2?  relu(A H) for 100 000 rows x 32 active entries with
the gathered table H shrunk from 100 000 rows (25.6 MB: Infinity Cache) to 12 500 (3.2 MB: every XCD's L2 holds it) -- same
instruction stream, same bytes per wavefront (diagnostic; host launch overhead can dominate on a slow box: read the kernel durations
from `rocprofv3 --kernel-trace`): python tools/time_gather_locality.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgg_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
N, K, F = 100_000, 64, 64
for ncols in (100_000, 50_000, 25_000, 12_500, 6_250):
    idx = torch.randint(0, ncols, (N, K), generator=g, dtype=torch.int32)
    idx[:, 32:] = -1
    ahat = torch.rand(N, K, generator=g)
    ahat[:, 32:] = 0
    H = torch.randn(ncols, F, generator=g)
    idx, ahat, H = idx.to(dev), ahat.to(dev), H.to(dev)
    for _ in range(3):
        ops.spmm_fwd(idx, ahat, H, 2)
    best = 1e9
    for _w in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ops.spmm_fwd(idx, ahat, H, 2)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    t0 = time.perf_counter() - best
    print(f"table {ncols:7d} rows ({ncols * F * 4 / 1e6:5.1f} MB): {(time.perf_counter() - t0) / 20 * 1e6:7.1f} us per aggregation of {N} rows x 32 entries")
