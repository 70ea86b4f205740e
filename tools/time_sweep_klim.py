"""Unperturbed all-pairs stage with the learned-degree limit: time per call, fallback rows and candidate counts (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgg_amd import ops
N, h = 100_000, 64
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = torch.randn(N, 128, generator=g).to(dev)
W = (torch.randn(h, 128, generator=g) * 0.1).to(dev)
b = (torch.randn(h, generator=g) * 0.1).to(dev)
xp = ops.linear_fwd(x, W, b, ops.ACT_LEAKY)
for lo, hi in ((24, 40), (4, 44)):
    k = (lo + (hi - lo) * torch.rand(N, generator=g)).to(dev)
    for kl in (None, k):
        for _ in range(2):
            idx, val, ws = ops.allpairs_topk(xp, 64, noise_mode=ops.NOISE_NONE, k_limit=kl, return_ws=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); idx, val, ws = ops.allpairs_topk(xp, 64, noise_mode=ops.NOISE_NONE, k_limit=kl, return_ws=True); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        nfail, st = ops.fast_path_failed_rows(ws, N, h, stats=True)
        print(f"k in [{lo},{hi}] klim={'yes' if kl is not None else 'no '}: {min(ts):.3f} ms  fallback rows {nfail}  A {st[0]/N:.1f} kept {st[1]/N:.1f} B {st[2]/N:.1f}", flush=True)
