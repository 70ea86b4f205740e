"""Timing of the ranked symmetric all-pairs path (noise_mode 5) at N=100k, h=64: per-kernel durations via torch profiler-free
event timing of the whole call, status counters, and a comparison with the hash-symmetric guess-and-verify path."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgg_amd
from dgg_amd import ops
dev = torch.device("cuda:0")
N, h = int(os.environ.get("N", 100000)), 64
g = torch.Generator().manual_seed(1000)
x = torch.randn(N, 128, generator=g).to(dev)
W = (torch.rand(h, 128, generator=g) * 2 - 1).mul_(1 / np.sqrt(128)).to(dev)
b = torch.zeros(h, device=dev)
xp = ops.linear_fwd(x, W, b, ops.ACT_LEAKY)
k = (24 + 16 * torch.rand(N, generator=g)).to(dev)
for name, nm in [("rsym", ops.NOISE_RANKED_SYM), ("hash_sym", ops.NOISE_HASH_SYM)]:
    for kl in (None, k):
        os.environ["DGG_RSYM_STATS"] = "1"
        idx, val, ws = ops.allpairs_topk(xp, 64, noise_mode=nm, seed=(1234, 0), return_ws=True, k_limit=kl)
        st = ops.rsym_status(ws, N) if nm == ops.NOISE_RANKED_SYM else None
        os.environ["DGG_RSYM_STATS"] = "0"
        torch.cuda.synchronize()
        ts = []
        for s in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.allpairs_topk(xp, 64, noise_mode=nm, seed=(1234 + s, 0), k_limit=kl)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(name, "klimit" if kl is not None else "full", "ms:", [round(t, 3) for t in ts], st, flush=True)
