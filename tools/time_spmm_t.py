#!/usr/bin/env python3
"""Times the gather kernels of the fused GCNII stack on one PPI-size graph (diagnostic): python tools/time_spmm_t.py [n] [deg] [skew]
   skew > 1 draws the neighbours as n * u^skew (hub columns, as the bench's PPI-shaped graphs have: kernels that walk destination-ordered
   records behave differently there -- the default uniform graph flattered a variant that lost in the step)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgg_amd import _lib, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1783
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 29
F, K = 2048, 32
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
skew = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
idx = (n * torch.rand(n, K, generator=g) ** skew).to(torch.int32).clamp_(0, n - 1)
idx[:, deg:] = -1
ahat = torch.rand(n, K, generator=g)
ahat[:, deg:] = 0
idx, ahat = idx.to(dev), ahat.to(dev)
xb = torch.randn(n, F, generator=g).to(dev).to(torch.bfloat16)
dyb = torch.randn(n, F, generator=g).to(dev).to(torch.bfloat16)
part = ops.part_build(idx, ahat, n)
L = _lib.lib()
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


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


dX = torch.zeros(n, F, device=dev)
Y = torch.empty(n, F, device=dev)
Yb = torch.empty(n, F, device=dev, dtype=torch.bfloat16)
dA = torch.empty(n, K, device=dev)
ws = torch.empty(int(L.dgg_ell_sddmm_b16_ws_floats(n, K, F)), device=dev)
gb = n * deg * F * 2 / 1e9
print(f"n = {n}, {deg} neighbours per row, F = {F}: {gb:.3f} GB of gathered rows per kernel")
t = timed(lambda: L.dgg_ell_spmm_fwd_b16(p(idx), p(ahat), p(xb), n, K, F, p(Y), p(Yb), F, st))
print("aggregation (spmm_fwd_b16)            %.1f us  %.1f TB/s" % (t, gb / t * 1e3))
t = timed(lambda: L.dgg_ell_sddmm_b16_sliced(p(idx), p(ahat), p(xb), p(dyb), n, K, F, 1, p(ws), p(dA), 0, st))
print("SDDMM, sliced + sum                   %.1f us  %.1f TB/s" % (t, gb / t * 1e3))
t = timed(lambda: L.dgg_ell_spmm_t_part_b16(p(ahat), p(dyb), n, K, F, p(part), n, p(dX), st))
print("transposed aggregation (spmm_t_cols)  %.1f us  %.1f TB/s" % (t, gb / t * 1e3))
# the wide score backward's row pass (latent 2048, fp32 rows of 8 KB): whole rows against 256-feature slices
xp = torch.randn(n, F, generator=g).to(dev) * 0.3
val = torch.rand(n, K, generator=g).to(dev)
dval = (torch.randn(n, K, generator=g) * (idx.cpu() >= 0)).to(dev)
own, dd = torch.empty(n, F, device=dev), torch.empty(n, K, device=dev)
ws2 = torch.empty(int(L.dgg_edge_bwd_wide_rows_ws_floats(n, K, F)), device=dev)
gb2 = 2 * n * deg * F * 4 / 1e9
t = timed(lambda: L.dgg_edge_bwd_wide_rows(p(xp), n, F, p(idx), p(val), p(dval), K, 0, C.c_float(1.0), 1, p(own), p(dd), st))
print("wide score backward, row pass         %.1f us  %.1f TB/s" % (t, gb2 / t * 1e3))
t = timed(lambda: L.dgg_edge_bwd_wide_rows_sliced(p(xp), n, F, p(idx), p(val), p(dval), K, 0, C.c_float(1.0), 1, p(ws2), p(own), p(dd), st))
print("  the same by 256-feature slices      %.1f us  %.1f TB/s" % (t, gb2 / t * 1e3))
