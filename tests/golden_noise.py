"""Seeded, machine-independent test noise shared by the golden generator and the tests.

The reference draws its Gumbel(0, 0.3) noise from torch's global RNG
(dgm.py:1149-1151, 1226), which cannot be re-created outside the process that
drew it.  For fixtures that are too large to store the noise matrix we draw it
from numpy's PCG64 (stable across numpy versions) in float64 and snap it to a
2^-16 grid, so that the float32 values are bit-identical on every machine
(a float64 libm difference of 1 ulp cannot move a value across a grid cell
except with probability ~1e-11 per element).  The crc32 of the array is stored
in the fixture and re-checked at test time.
"""
import zlib

import numpy as np

GUMBEL_SCALE = 0.3  # dgm.py:1149-1151


def grid_gumbel(seed: int, shape, scale: float = GUMBEL_SCALE) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    u = rng.random(size=shape)
    u = np.clip(u, 2.0 ** -24, 1.0 - 2.0 ** -24)
    g = -scale * np.log(-np.log(u))
    g = np.round(g * 65536.0) / 65536.0
    return g.astype(np.float32)


def grid_normal(seed: int, shape, scale: float = 1.0) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    g = rng.standard_normal(size=shape) * scale
    g = np.round(g * 4096.0) / 4096.0
    return g.astype(np.float32)


def crc(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF
