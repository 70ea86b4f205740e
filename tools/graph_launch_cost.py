#!/usr/bin/env python3
"""Is the captured headline step bound by the HOST (hipGraphLaunch) or by the GPU?  Host time to enqueue K replays (no sync) against
the time until they have all finished, and the sum of the step's kernel durations from events around single replays."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from argparse import Namespace
from dgg_amd import ops
dev = torch.device("cuda:0")
a = Namespace(algo=0, prior="24,40", data="randn", feat_scale=1.0)
run = bench.SyntheticRun(a, dev, 1, 0, False, 100_000, 128, 64, ops.NOISE_RANKED)
for s in range(5):
    run.step(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run.step(0)
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
for K in (1, 5, 20, 100):
    res = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append(((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
    res.sort(key=lambda r: r[1])
    print(f"K={K:4d}: host enqueue {res[2][0]:.4f} ms per replay, until finished {res[2][1]:.4f} ms per replay")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"one replay between events (GPU time of an isolated replay): median {ts[10]:.4f} ms, min {ts[0]:.4f} ms")
