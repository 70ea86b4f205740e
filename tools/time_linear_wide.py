#!/usr/bin/env python3
"""Times the fused projection and the fused weight gradients on a WIDE input (Pubmed shape by default: N = 19 717, d = 500; diagnostic):
python tools/time_linear_wide.py [N] [d]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgg_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 19_717
d = int(sys.argv[2]) if len(sys.argv) > 2 else 500
h = 64
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
ref = [(dxp.double().T @ x.double()), (dxk.double().T @ x.double()), (x.double().T @ dH.double())]
got = ops.linear_bwd_multi(x, [(We, None, dxp, 0, 0, True), (Wk, None, dxk, 0, 0, True), (Wc, None, dH, 0, 1, False)])
err = max(float((g_[0].double() - r_).abs().max() / r_.abs().max()) for g_, r_ in zip(got, ref))
print(f"N={N} d={d} DGG_LIN_NACC={os.environ.get('DGG_LIN_NACC')}: fused projection {tf:.1f} us, fused weight gradients {tb:.1f} us "
      f"(eager, incl. pack / reduce / allocations); weight-gradient error {err:.1e} of max")
