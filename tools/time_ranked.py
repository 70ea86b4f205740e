#!/usr/bin/env python3
"""Times the ranked search + ramp (dgg_allpairs_topk_ranked_softk) on the benchmark's shape (N = 100 000, h = 64, k ~ 24..41) with HIP events:
median / min of 40 calls with fresh seeds.  DGG_HIP_SO selects a diagnostic build (tools/build_variant.sh)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dgg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N, d, h = 100_000, 128, 64
P = bench.make_params(d, h, dev)
x = torch.randn(N, d, generator=torch.Generator().manual_seed(1000)).to(dev)
xp = ops.linear_fwd(x, P["We"], P["be"], ops.ACT_LEAKY)
k = (24 + 17 * torch.rand(N, generator=torch.Generator().manual_seed(7))).to(dev)
ref = ops.allpairs_topk_softk(xp, k, seed=(1234, 0))
ts = []
for s in range(45):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.allpairs_topk_softk(xp, k, seed=(1234, s))
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts = ts[5:]
chk = int(ref[0].long().sum().item()) ^ int(ref[1].view(torch.int32).long().sum().item())
print(f"ranked search: median {np.median(ts):7.1f} us  min {min(ts):7.1f} us   checksum {chk}")
