#!/usr/bin/env python3
"""Times the bf16 layer products of a PPI-size graph in their fused and unfused forms (diagnostic): python tools/time_bf16.py [n]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgg_amd import _lib, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2250
F = 2048
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
hi, h0, x, gr = (torch.randn(n, F, generator=g).to(dev) for _ in range(4))
W = (torch.randn(2 * F, F, generator=g) / 64).to(dev)
L = _lib.lib()
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


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


S1, S2, Scat = ops.pack_bf16(hi), ops.pack_bf16(h0), ops.pack_bf16(torch.cat([hi, h0], 1))
Wt, Wp = ops.pack_bf16(W, transpose=True), ops.pack_bf16(W)
out = torch.empty(n, F, device=dev)
out2 = torch.empty(n, 2 * F, device=dev)
Gp = ops.pack_bf16(gr)
dhi, dh0 = torch.empty(n, F, device=dev), torch.empty(n, F, device=dev)
n64 = (n + 63) // 64 * 64
hiT, h0T, GT = ops.pack_bf16(hi, transpose=True), ops.pack_bf16(h0, transpose=True), ops.pack_bf16(gr, transpose=True)
dW = torch.empty_like(W)
print(f"n = {n}, F = {F} (37.7 GFLOP per product at n = 2250)")
print("forward, fused epilogue (split A)      %.1f us" % timed(lambda: L.dgg_gcnii_gemm_bf16_split(p(S1), p(S2), p(Wt), n, F, 2 * F, F, p(hi), p(h0), p(x), C.c_float(0.4), C.c_float(0.5), p(out), st)))
print("forward, fused + relu + dropout        %.1f us" % timed(lambda: L.dgg_gcnii_gemm_bf16_split_act(p(S1), p(S2), p(Wt), n, F, 2 * F, F, p(hi), p(h0), p(x), C.c_float(0.4), C.c_float(0.5), 1, C.c_float(0.2), 1, 2, p(out), None, st)))
print("forward product alone (plain, cat A)   %.1f us" % timed(lambda: L.dgg_gemm_nt_bf16(p(Scat), p(Wt), n, F, 2 * F, C.c_float(1.0), p(out), st)))
print("  + separate epilogue pass             %.1f us" % timed(lambda: L.dgg_gcnii_epilogue_fwd(p(out), p(hi), p(h0), p(x), n * F, C.c_float(0.4), C.c_float(0.5), p(dhi), st)))
print("[d hi | d h0], fused epilogue          %.1f us" % timed(lambda: L.dgg_gcnii_dsupport_bf16(p(Gp), p(Wp), n, F, p(gr), C.c_float(0.4), C.c_float(0.5), p(dhi), p(dh0), st)))
dhib = torch.empty(n, F, device=dev, dtype=torch.bfloat16)
for ob, ac in ((0, 0), (1, 0), (0, 1), (1, 1), (2, 1)):
    print("[d hi | d h0] fused, bf16 copy %d (2: no fp32 d hi), accumulate %d  %.1f us" % (ob, ac, timed(lambda: L.dgg_gcnii_dsupport_bf16_b(p(Gp), p(Wp), n, F, p(gr), C.c_float(0.4), C.c_float(0.5), None if ob == 2 else p(dhi), p(dh0), p(dhib) if ob else None, ac, st))))
print("[d hi | d h0] product alone (plain)    %.1f us" % timed(lambda: L.dgg_gemm_nt_bf16(p(Gp), p(Wp), n, 2 * F, F, C.c_float(0.4), p(out2), st)))
print("weight gradient (rows2, plain)         %.1f us" % timed(lambda: L.dgg_gemm_nt_bf16_rows2(p(hiT), p(h0T), F, p(GT), 2 * F, F, n64, C.c_float(0.4), p(dW), st)))
