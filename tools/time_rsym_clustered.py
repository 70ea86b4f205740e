"""Ranked symmetric vs hash-symmetric all-pairs stage on clustered data with far outliers (the case that drives the ranked
symmetric generator into its dense tier): time per call and tier counters."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dgg_amd  # noqa: F401
from dgg_amd import ops
dev = torch.device("cuda:0")
N, d, h = 100_000, 128, 64
g = torch.Generator().manual_seed(5)
x = torch.randn(N, d, generator=g)
x[20_000:23_000] *= 0.05
nout = int(os.environ.get("OUTLIERS", 40))
x[23_000:23_000 + nout] = x[23_000:23_000 + nout] * 0.01 + 3.0
x = x.to(dev)
W = (torch.randn(h, d, generator=g) * 0.1).to(dev)
b = (torch.randn(h, generator=g) * 0.1).to(dev)
xp = ops.linear_fwd(x, W, b, ops.ACT_LEAKY)
k = (24 + 16 * torch.rand(N, generator=g)).to(dev)
for name, nm in [("rsym", ops.NOISE_RANKED_SYM), ("hash_sym", ops.NOISE_HASH_SYM)]:
    idx, val, ws = ops.allpairs_topk(xp, 64, noise_mode=nm, seed=(7, 1), return_ws=True, k_limit=k)
    st = ops.rsym_status(ws, N) if nm == ops.NOISE_RANKED_SYM else None
    torch.cuda.synchronize()
    ts = []
    for s in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.allpairs_topk(xp, 64, noise_mode=nm, seed=(8 + s, 1), k_limit=k)
        e1.record()
        torch.cuda.synchronize()
        ts.append(round(e0.elapsed_time(e1), 3))
    print(name, ts, st, flush=True)
