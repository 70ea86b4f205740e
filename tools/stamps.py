#!/usr/bin/env python3
"""Diagnostic: per-section cycle shares of the pruned all-pairs kernel (build: make -C .../csrc stamps).
Run on the GPU box:  DGG_HIP_SO=.../libdgg_hip_stamps.so python tools/stamps.py [N]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import dgg_amd  # noqa: E402
from dgg_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
noise = int(sys.argv[2]) if len(sys.argv) > 2 else ops.NOISE_HASH
dev = torch.device("cuda:0")
P = bench.make_params(128, 64, dev)
g = torch.Generator(device="cpu").manual_seed(1000)
x = torch.randn(N, 128, generator=g).to(dev)
xp = ops.linear_fwd(x, P["We"], P["be"], ops.ACT_LEAKY)
L = dgg_amd._lib.lib()
out = (C.c_ulonglong * 8)()
ops.allpairs_topk(xp, 64, noise_mode=noise, seed=(1234, 0), algo=2)
torch.cuda.synchronize()
L.dgg_debug_read_stamps(out, 1)
ops.allpairs_topk(xp, 64, noise_mode=noise, seed=(1234, 0), algo=2)
torch.cuda.synchronize()
L.dgg_debug_read_stamps(out, 1)
v = list(out)
tot = sum(v[:4])
names = ["tile stage + MFMA", "stage A (hash + compare)", "stage B (bound)", "flush (exact + merge)"]
for n, c in zip(names, v[:4]):
    print(f"{n:28s} {c:16d} cycles  {100.0 * c / tot:5.1f} %")
print(f"stage-B reg executions {v[4]}  ({v[4] / (N * (N / 32) / 32 * 16 / 1):.4f} of reg-tiles)   flushes {v[5]} ({v[5] / N:.2f} per row)")
