#!/usr/bin/env python3
"""time dgg_allpairs_rowmin_bound at N = 100 000 (h from argv, default 64)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgg_amd import ops
h = int(sys.argv[1]) if len(sys.argv) > 1 else 64
xp = torch.randn(100000, h, device="cuda")
ops.rowmin_logp_bound(xp); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    lp = ops.rowmin_logp_bound(xp)
torch.cuda.synchronize()
print(f"rowmin N=100k h={h}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms; bound mean {float(lp.mean()):.4f} min {float(lp.min()):.4f}")
