#!/usr/bin/env python3
"""Differential fuzz of the chunked-row evaluators (ops.allpairs_topk_wide, every generator) against the oracle on awkward inputs:
node counts around the tile sizes, duplicate / all-zero / clustered / huge-magnitude features, degree laws from narrow to 'more than
half the columns'.  Sampled rows (plus the widest and the special ones) bit for bit.  Not a test: a GPU soak (`python tools/fuzz_anywide.py
--minutes 5`); every failure prints the case so that it can be pinned in tests/test_chunked_rows.py."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dgg_amd import ops  # noqa: E402
from oracle import oracle as O  # noqa: E402
import test_chunked_rows as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=5.0)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--rows", type=int, default=24)
ap.add_argument("--sizes", default="1023,1024,1025,2047,3000,4097,8191,12000,20000", help="node counts to draw from")
ap.add_argument("--narrow", action="store_true", help="the 64-rank entry (ops.allpairs_topk, every generator incl. ranked symmetric) instead of the chunked rows")
a = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
K = 64
NM = {"none": (ops.NOISE_NONE, O.NOISE_NONE), "hash": (ops.NOISE_HASH, O.NOISE_HASH), "hash_sym": (ops.NOISE_HASH_SYM, O.NOISE_HASH_SYM),
      "ranked": (ops.NOISE_RANKED, O.NOISE_RANKED)}


def features(kind, N, h, g):
    if kind == "randn":
        return torch.randn(N, h, generator=g) * float(rng.choice([0.1, 0.7, 3.0]))
    if kind == "clustered":
        c = torch.randn(int(rng.integers(2, 12)), h, generator=g) * 2.0
        return c[torch.randint(0, c.shape[0], (N,), generator=g)] + 1e-3 * torch.randn(N, h, generator=g)
    if kind == "duplicates":
        base = torch.randn(max(N // 50, 3), h, generator=g)
        return base[torch.randint(0, base.shape[0], (N,), generator=g)]
    if kind == "zeros":
        x = torch.zeros(N, h)
        x[: N // 3] = torch.randn(N // 3, h, generator=g) * 0.5
        return x
    if kind == "huge":
        return torch.randn(N, h, generator=g) * 300.0
    if kind == "onehot":
        x = torch.zeros(N, h)
        x[torch.arange(N), torch.randint(0, h, (N,), generator=g)] = 1.0 + torch.rand(N, generator=g)
        return x
    raise ValueError(kind)


def degrees(kind, N, g):
    if kind == "narrow":
        return 1.0 + 50.0 * torch.rand(N, generator=g)
    if kind == "mixed":
        return 1.0 + 300.0 * torch.rand(N, generator=g) ** 2
    if kind == "tail":
        k = 20.0 + 60.0 * torch.rand(N, generator=g)
        sel = torch.rand(N, generator=g) < 0.01
        k[sel] = 100.0 + (0.9 * N) * torch.rand(int(sel.sum()), generator=g) ** 3
        return k
    if kind == "half":
        return (0.2 + 0.5 * torch.rand(N, generator=g)) * N
    raise ValueError(kind)


t_end = time.time() + 60.0 * a.minutes
case = fails = 0
while time.time() < t_end:
    case += 1
    N = int(rng.choice([int(v) for v in a.sizes.split(",")]))
    h = int(rng.choice([16, 32, 64, 128]))          # (the chunked rows exist for latent 16 / 32 / 64 / 128)
    fk = str(rng.choice(["randn", "clustered", "duplicates", "zeros", "huge", "onehot"]))
    dk = str(rng.choice(["narrow", "mixed", "tail", "half"] if N <= 4097 else ["narrow", "mixed", "tail"]))
    noise = str(rng.choice(list(NM)))
    mode = int(rng.choice([0, 1]))
    seed = (int(rng.integers(0, 2 ** 31)), int(rng.integers(0, 2 ** 31)))
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    xp = features(fk, N, h, g).to(dev)
    k = degrees(dk, N, g).to(dev)
    desc = f"case {case}: N {N} h {h} features {fk} degrees {dk} noise {noise} mode {mode} seed {seed}"
    if a.narrow:
        noise = str(rng.choice(list(NM) + ["ranked_sym"]))
        om = {"ranked_sym": (ops.NOISE_RANKED_SYM, O.NOISE_RANKED_SYM)}.get(noise) or NM[noise]
        kl = None if rng.random() < 0.3 else (1.0 + 70.0 * torch.rand(N, generator=g)).to(dev)
        desc = f"case {case}: narrow N {N} h {h} features {fk} noise {noise} k_limit {kl is not None} seed {seed}"
        try:
            st = {"sym_fallback": False}
            idx, val = ops.allpairs_topk(xp, K, noise_mode=om[0], seed=seed, k_limit=kl, status=st)
            torch.cuda.synchronize()
            if noise == "ranked_sym" and int(st["rsym_err"].item()) != 0:
                print(f"skip {desc}: the ranked symmetric generator reported 'not settled' (callers redo under hash noise)", flush=True)
                continue
            rows = sorted(set(rng.choice(N, size=min(a.rows, N), replace=False).tolist()) | {0, N - 1})
            xp_c = xp.cpu().numpy()
            gi, gv = idx.cpu().numpy(), val.cpu().numpy()
            L = np.full(N, K) if kl is None else np.minimum(np.ceil(kl.cpu().numpy().astype(np.float32) + np.float32(8.5)) + 1, K).astype(np.int64)
            if noise == "ranked_sym":                                # (the oracle walks every owner's sequence: whole matrix at once)
                ri_all, rv_all = O.allpairs_topk(xp_c, K=K, noise_mode=om[1], seed=seed) if N <= 4097 else (None, None)
                if ri_all is None:
                    print(f"skip {desc}: oracle too slow at this size", flush=True)
                    continue
            for r in rows:
                ri, rv = (ri_all[r], rv_all[r]) if noise == "ranked_sym" else [v[0] for v in O.allpairs_topk(xp_c, K=K, noise_mode=om[1], seed=seed, rows=(r, r + 1))]
                keep = np.arange(K) < L[r]
                assert np.array_equal(gi[r], np.where(keep, ri, -1)), f"row {r}: ranks differ"
                assert np.array_equal(gv[r], np.where(keep, rv, np.float32(0))), f"row {r}: scores differ"
            print(f"ok   {desc}: {len(rows)} rows checked", flush=True)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print(f"FAIL {desc}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        continue
    try:
        lay = ops.chunk_layout(k, ncols=N)
        idx, val, w, rs = ops.allpairs_topk_wide(xp, k, lay, mode=mode, seed=seed, noise_mode=NM[noise][0])
        torch.cuda.synchronize()
        cptr = lay.cptr.cpu().numpy().astype(np.int64)
        width = cptr[1:] - cptr[:-1]
        cand = np.where(width <= (40 if N > 4097 else 1 << 30))[0]          # (the oracle's insertion list is O(K) per column)
        rows = sorted(set(rng.choice(cand, size=min(a.rows, len(cand)), replace=False).tolist()) | {int(cand[np.argmax(width[cand])]), 0, N - 1} & set(cand.tolist()))
        T._check_rows_against_oracle(lay, xp, k, idx, val, w, rs, NM[noise][1], seed, mode, rows=rows)
        print(f"ok   {desc}: {lay.chunks} chunks, widest {int(width.max())}, {len(rows)} rows checked", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL {desc}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"{case} cases, {fails} failures")
sys.exit(1 if fails else 0)
