#!/usr/bin/env python3
"""Times the k-net forward / backward (diagnostic): python tools/time_knet.py [N].  DGG_KNET_LDS=1 selects the round-3 kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgg_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
h = 64
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
xk = torch.randn(N, h, generator=g).to(dev)
deg = (24 + 16 * torch.rand(N, generator=g)).to(dev)
W1 = (torch.randn(h // 2, h + 1, generator=g) * 0.2).to(dev)
b1 = (torch.randn(h // 2, generator=g) * 0.1).to(dev)
Wmu = (torch.randn(h // 4, h // 2, generator=g) * 0.3).to(dev)
bmu = (torch.randn(h // 4, generator=g) * 0.1).to(dev)
Wp = (torch.randn(h // 4, generator=g) * 0.05).to(dev)
bp = torch.tensor([0.05], device=dev)
dk = torch.randn(N, generator=g).to(dev)
mu_sd = ops.degree_stats(deg)


def timed(fn, R=30):
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


k, u = ops.knet_x_fwd_slim(xk, deg, mu_sd, W1, b1, Wmu, bmu, Wp, bp)
tf = timed(lambda: ops.knet_x_fwd_slim(xk, deg, mu_sd, W1, b1, Wmu, bmu, Wp, bp))
tb = timed(lambda: ops.knet_x_bwd_fused(xk, deg, mu_sd, W1, b1, Wmu, bmu, Wp, u, dk))
print(f"N={N} lds_form={os.environ.get('DGG_KNET_LDS', '0')}: fwd {tf:.1f} us, bwd {tb:.1f} us (eager, incl. allocations)")
