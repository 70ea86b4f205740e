#!/usr/bin/env python3
"""Times the payload-partition build (dgg_partp_build_norm) on the adjacency of one real step of the synthetic workload
(diagnostic): python tools/time_part.py [N].  DGG_PP_THREADS / DGG_PP_SHIFT select the workgroup size / bucket width."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from dgg_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dev = torch.device("cuda", 0)
a = argparse.Namespace(algo=0)
run = bench.SyntheticRun(a, dev, 1, 0, False, N, 128, 64, ops.NOISE_RANKED)
run.step(0)
s = run.layer.saved
idx, w, val, rs = s["idx"], s["w"], s["val"], s["rs"]
for _ in range(3):
    ops.partp_build(idx, w, val, rs, N, rs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
R = 20
e0.record()
for _ in range(R):
    ops.partp_build(idx, w, val, rs, N, rs)
e1.record()
torch.cuda.synchronize()
act = int(((idx >= 0) & (w != 0)).sum().item())
print(f"N={N} active={act} threads={os.environ.get('DGG_PP_THREADS', 'default')} width={os.environ.get('DGG_PP_WIDTH', 'auto')}: "
      f"{e0.elapsed_time(e1) / R * 1e3:.1f} us per build (incl. workspace + ahat allocation)")
