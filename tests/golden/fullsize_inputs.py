"""Seeded inputs of the BASELINE-size parity fixtures (tests/golden/fullsize.npz).

The fixture stores only seeds + the reference's integer outputs; the embeddings are regenerated on
the GPU box.  Everything here is bit-reproducible across machines: RandomState's legacy
standard_normal stream, float64 element-wise arithmetic, and row sums of squares taken with cumsum
(strictly sequential -- np.sum / np.linalg.norm pick a SIMD-width-dependent summation order).
`digest()` is stored in the fixture and re-checked by the tests, so a machine that generated other
bits fails loudly instead of reporting spurious rank differences.
"""
import hashlib

import numpy as np


def unit_rows64(x):
    ss = np.cumsum(x * x, axis=1)[:, -1]
    return x / np.sqrt(ss)[:, None]


def eval_pair(n, d, noise, seed):
    """e1: unit rows; e2[i] = unit(e1[i] + noise * unit gaussian): gold of row i is column i."""
    rng = np.random.RandomState(seed)
    a = unit_rows64(rng.standard_normal((n, d)))
    b = unit_rows64(a + noise * unit_rows64(rng.standard_normal((n, d))))
    return a.astype(np.float32), b.astype(np.float32)


def knn_table(n, d, seed, clusters=0):
    """unit rows; clusters > 0: rows scatter round `clusters` centres (neighbourhoods with structure, like a
    trained table) instead of being isotropic."""
    rng = np.random.RandomState(seed)
    x = rng.standard_normal((n, d))
    if clusters:
        c = rng.standard_normal((clusters, d))
        x = c[rng.randint(0, clusters, n)] + 0.7 * x
    return unit_rows64(x).astype(np.float32)


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:32]


def row_set_hash(idx):
    """order-independent 64-bit hash of every row's id set: sum of splitmix64(id) mod 2^64."""
    x = np.asarray(idx).astype(np.uint64)
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
        return x.sum(axis=1, dtype=np.uint64)


# (name, n, d, noise, seed): the 15K test split (10,500 pairs) at BASELINE.json's dim 75 and the shipped dim 100
EVAL_15K = [("eval15k_d75", 10500, 75, 2.0, 11), ("eval15k_d100", 10500, 100, 2.4, 12)]
EVAL_100K = ("eval100k_d100", 70000, 100, 2.2, 13, 256)          # + number of sampled query rows
KNN_15K = ("knn15k", 15000, 100, 1499, 14)                        # int((1 - 0.9) * 15000) = 1499
KNN_100K = ("knn100k", 100000, 100, 2000, 15, 32)                 # int((1 - 0.98) * 100000) = 2000; sampled rows
