#!/usr/bin/env python3
"""The per-pair hash generators' 64-rank entry (guess-and-verify, noise_mode 2 / 3) and the ranked symmetric generator (5) across node
counts: time per call and rows redone by the fallback / deeper tiers.  `python tools/gv_sizes.py`"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for h in (64,):
    for N in (1024, 2048, 4096, 8192, 12000, 19717, 30000, 100000):
        g = torch.Generator().manual_seed(N)
        xp = (torch.randn(N, h, generator=g) * 0.7).to(dev)
        k = (24 + 17 * torch.rand(N, generator=g)).to(dev)
        line = f"N {N:7d} h {h}:"
        for name, nm in (("hash", ops.NOISE_HASH), ("hash_sym", ops.NOISE_HASH_SYM)):
            for kl in (None, k):
                ts = []
                for r in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    idx, val, ws = ops.allpairs_topk(xp, 64, noise_mode=nm, seed=(3, r), k_limit=kl, return_ws=True)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                nfail = int(ws[4:8].view(torch.int32).item())
                line += f"  {name}{'+klim' if kl is not None else ''} {np.median(ts[1:]):7.3f} ms fail {nfail:5d}"
        st = {"sym_fallback": False}
        ts = []
        for r in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.allpairs_topk(xp, 64, noise_mode=ops.NOISE_RANKED_SYM, seed=(3, r), k_limit=k, status=st)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        line += f"  rsym+klim {np.median(ts[1:]):7.3f} ms err {int(st['rsym_err'].item())} tier3 {int(st['rsym_tier3'].item())}"
        print(line, flush=True)
