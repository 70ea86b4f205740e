"""Caller-side input pipeline of the small-graph script (SURVEY.md 8b' / 8f rank 4): Planetoid citation graphs and the
noisy-edge augmentation, as sparse O(E) host code (numpy / scipy; nothing here touches the GPU).

The reference goes through `torch_geometric.datasets.Planetoid(..., transform=NormalizeFeatures())`
(train_small_graphs.py:339-348) or its own `utils.load_citation` (utils.py:122-196); both read the public
`ind.<name>.{x,tx,allx,y,ty,ally,graph,test.index}` files of Kipf & Welling's GCN release.  This reader parses the same
files and returns what the training step consumes: row-normalised features, the symmetric edge list without self
loops (both directions, sorted row-major = a coalesced COO), labels and the public split.
"""
import os
import pickle

import numpy as np
import scipy.sparse as sp


def _pkl(path):
    with open(path, "rb") as f:
        return pickle.load(f, encoding="latin1")


def _cache_key(paths):
    """cache validity: size and mtime of every source file"""
    return "|".join(f"{os.path.basename(p_)}:{os.path.getsize(p_)}:{int(os.path.getmtime(p_))}" for p_ in paths)


def load_planetoid(name, data_dir, cache_dir=None):
    """-> dict(x float32 [N,d] row-normalised (dense), rows/cols int32 [E] (symmetric, no self loops, row-major),
    y int64 [N], train_idx / val_idx / test_idx int64).
    cache_dir: the parsed arrays are kept there as `<name>.planetoid.npz` (the pickled scipy / networkx-style sources take
    seconds to parse for Pubmed) and reused while size and mtime of every source file are unchanged."""
    name = name.lower()
    if cache_dir is not None:
        srcs = [os.path.join(data_dir, f"ind.{name}.{s_}") for s_ in ("x", "y", "tx", "ty", "allx", "ally", "graph", "test.index")]
        key, cfile = _cache_key(srcs), os.path.join(cache_dir, f"{name}.planetoid.npz")
        if os.path.exists(cfile):
            with np.load(cfile, allow_pickle=False) as z:
                if str(z["cache_key"]) == key:
                    out = {k_: z[k_] for k_ in z.files if k_ != "cache_key"}
                    out["num_classes"] = int(out["num_classes"])
                    return out
        out = load_planetoid(name, data_dir)
        os.makedirs(cache_dir, exist_ok=True)
        tmp = cfile + f".tmp{os.getpid()}.npz"
        np.savez(tmp, cache_key=np.array(key), **out)
        os.replace(tmp, cfile)
        return out
    p = lambda s: os.path.join(data_dir, f"ind.{name}.{s}")  # noqa: E731
    x, y, tx, ty, allx, ally, graph = (_pkl(p(s)) for s in ("x", "y", "tx", "ty", "allx", "ally", "graph"))
    test_idx = np.array([int(line.strip()) for line in open(p("test.index"))], dtype=np.int64)
    test_sorted = np.sort(test_idx)
    if name == "citeseer":
        # isolated test nodes are missing from tx/ty: insert zero rows at their positions
        full = np.arange(test_sorted.min(), test_sorted.max() + 1)
        tx_ext = sp.lil_matrix((len(full), x.shape[1]))
        tx_ext[test_sorted - test_sorted.min(), :] = tx
        ty_ext = np.zeros((len(full), y.shape[1]))
        ty_ext[test_sorted - test_sorted.min(), :] = ty
        tx, ty = tx_ext, ty_ext
    feats = sp.vstack((allx, tx)).tolil()
    feats[test_idx, :] = feats[test_sorted, :]
    labels = np.vstack((ally, ty))
    labels[test_idx, :] = labels[test_sorted, :]
    N = feats.shape[0]
    # undirected edge set without self loops, both directions
    src = np.fromiter((u for u, nb in graph.items() for _ in nb), dtype=np.int64)
    dst = np.fromiter((v for nb in graph.values() for v in nb), dtype=np.int64)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = np.unique(np.concatenate([src * N + dst, dst * N + src]))
    rows, cols = (key // N).astype(np.int32), (key % N).astype(np.int32)
    # NormalizeFeatures / utils.normalize: rows sum to one, empty rows stay zero
    feats = sp.csr_matrix(feats, dtype=np.float32)
    rs = np.asarray(feats.sum(1)).reshape(-1)
    inv = np.where(rs > 0, 1.0 / np.where(rs > 0, rs, 1.0), 0.0).astype(np.float32)
    feats = sp.diags(inv).dot(feats)
    n_train = y.shape[0]
    return dict(x=np.asarray(feats.todense(), dtype=np.float32), rows=rows, cols=cols, y=labels.argmax(1).astype(np.int64),
                train_idx=np.arange(n_train, dtype=np.int64), val_idx=np.arange(n_train, n_train + 500, dtype=np.int64),
                test_idx=test_sorted, num_classes=int(labels.shape[1]))


def add_noisy_edges(rows, cols, N, noise_level, reference_stream=None, seed=0):
    """Adds random off-diagonal, not-yet-present DIRECTED entries of value 1, each with probability 10 * noise_level
    (reference utils.py:92-110).  Returns the augmented (rows, cols, vals), row-major sorted.

    reference_stream=True reproduces the reference's realisation exactly (np.random.seed(0); rand(N, N): an O(N^2)
    host array, fine for the citation graphs); False draws the same distribution in O(E + noise) -- the number of noisy
    entries is binomial, their positions uniform without replacement -- for graphs where N^2 does not fit.  Default:
    exact stream up to N = 20 000."""
    p = noise_level * 10
    if reference_stream is None:
        reference_stream = N <= 20_000
    have = rows.astype(np.int64) * N + cols.astype(np.int64)
    if reference_stream:
        np.random.seed(0)
        hit = np.flatnonzero(np.random.rand(N, N).reshape(-1) < p)
    else:
        rng = np.random.default_rng(seed)
        n_noise = int(rng.binomial(N * N, p))
        # uniform positions WITHOUT replacement: draw until n_noise distinct ones exist, then drop the excess at random (np.unique
        # sorts: cutting its tail would always remove the largest flat indices, i.e. thin out the last rows)
        hit = np.unique(rng.integers(0, N * N, size=int(n_noise * 1.02) + 16))
        while hit.shape[0] < n_noise:
            hit = np.unique(np.concatenate([hit, rng.integers(0, N * N, size=n_noise - hit.shape[0] + 16)]))
        if hit.shape[0] > n_noise:
            hit = np.sort(rng.permutation(hit)[:n_noise])
    hit = hit[(hit // N) != (hit % N)]
    hit = np.setdiff1d(hit, have, assume_unique=False)
    key = np.concatenate([have, hit])
    order = np.argsort(key, kind="stable")
    key = key[order]
    return (key // N).astype(np.int32), (key % N).astype(np.int32), np.ones(key.shape[0], np.float32)
