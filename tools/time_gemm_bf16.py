#!/usr/bin/env python3
"""Times ops.gemm_nt_bf16 (the GCNII stack's plain product) against torch.matmul (hipBLASLt) on the PPI stack's three product shapes for a
graph of n nodes: forward [n,4096] x [2048,4096]^T, d support [n,2048] x [2048,2048]^T (one half), weight gradient [4096,n64] x [2048,n64]^T."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def t_us(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for n in (591, 1100, 1300, 1500, 2000, 2560, 3480):
    n64 = (n + 63) // 64 * 64
    for name, (M, N, K) in {"fwd": (n, 2048, 4096), "dsup": (n, 2048, 2048), "dW": (4096, 2048, n64)}.items():
        A = torch.randn(M, K, device=dev).bfloat16()
        B = torch.randn(N, K, device=dev).bfloat16()
        ours = t_us(lambda: ops.gemm_nt_bf16(A, B))
        lib = t_us(lambda: torch.matmul(A, B.t()))
        fl = 2.0 * M * N * K
        print(f"n {n:5d} {name:5s} M {M:5d} N {N:5d} K {K:5d}: ours {ours:7.1f} us ({fl / ours / 1e6:6.0f} TF/s)   torch {lib:7.1f} us ({fl / lib / 1e6:6.0f} TF/s)", flush=True)
