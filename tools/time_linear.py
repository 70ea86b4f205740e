#!/usr/bin/env python3
"""Times the fused projection (linear_fwd_multi: xp | xk | H) and the fused weight gradients (linear_bwd_multi) of the headline
shape (diagnostic): python tools/time_linear.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgg_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d, h = 128, 64
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = torch.randn(N, d, generator=g).to(dev)
We, Wk = (torch.randn(h, d, generator=g) * 0.1).to(dev), (torch.randn(h, d, generator=g) * 0.1).to(dev)
be, bk = (torch.randn(h, generator=g) * 0.1).to(dev), (torch.randn(h, generator=g) * 0.1).to(dev)
Wc = torch.rand(d, 64, generator=g).to(dev)
dxp, dxk, dH = (torch.randn(N, 64, generator=g).to(dev) for _ in range(3))


def timed(fn, R=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R * 1e3


tf = timed(lambda: ops.linear_fwd_multi(x, [(We, be, 1, 0), (Wk, bk, 1, 0), (Wc, None, 0, 1)]))
tb = timed(lambda: ops.linear_bwd_multi(x, [(We, None, dxp, 0, 0, True), (Wk, None, dxk, 0, 0, True), (Wc, None, dH, 0, 1, False)]))
print(f"N={N}: fused projection {tf:.1f} us, fused weight gradients {tb:.1f} us (eager, incl. pack / reduce / allocations)")
