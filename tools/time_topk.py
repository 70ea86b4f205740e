#!/usr/bin/env python3
"""Times the all-pairs top-64 evaluators on one GPU (diagnostic): python tools/time_topk.py [N] [feature scale]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgg_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = (torch.randn(N, 128, generator=g) * scale).to(dev)
W = (torch.randn(64, 128, generator=g) * 0.1).to(dev)
b = (torch.randn(64, generator=g) * 0.1).to(dev)
xp = ops.linear_fwd(x, W, b, ops.ACT_LEAKY)
k = (24 + 16 * torch.rand(N, generator=g)).to(dev)
for name, mode, algo, kl in [("none/sweep+klimit", ops.NOISE_NONE, 2, k), 
                             ("hash/gv(4)", ops.NOISE_HASH, 4, k), ("ranked+klimit", ops.NOISE_RANKED, 0, k)]:
    try:
        ops.allpairs_topk(xp, 64, noise_mode=mode, seed=(1, 2), algo=algo, k_limit=kl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            ops.allpairs_topk(xp, 64, noise_mode=mode, seed=(1, 2), algo=algo, k_limit=kl)
        torch.cuda.synchronize()
        print(f"scale {scale:5.1f} {name:18s} {(time.perf_counter() - t0) / 2 * 1e3:9.3f} ms", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"scale {scale:5.1f} {name:18s} failed: {e!r}", flush=True)
