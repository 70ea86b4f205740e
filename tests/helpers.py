"""Shared helpers for the oracle / parity tests."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    fx = {k: z[k] for k in z.files}
    fx["meta"] = json.loads(bytes(fx["meta"]).decode())
    return fx


def csr_from_coo(rows, cols, N):
    """rows sorted ascending (coalesced COO)."""
    rowptr = np.zeros(N + 1, np.int64)
    np.add.at(rowptr, rows.astype(np.int64) + 1, 1)
    return np.cumsum(rowptr), cols.astype(np.int32)


def ell_to_dense(idx, w, N):
    out = np.zeros((idx.shape[0], N), np.float32)
    r = np.repeat(np.arange(idx.shape[0]), idx.shape[1])
    c = idx.reshape(-1)
    m = c >= 0
    out[r[m], c[m]] = w.reshape(-1)[m]
    return out


def ulp_diff(a, b):
    """distance in units of float32 ulps between positive floats"""
    ia = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(ia - ib)
