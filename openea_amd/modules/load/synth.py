"""Seeded synthetic KG pairs with the shapes of the OpenEA benchmark datasets.

The datasets are not vendored (figshare download, reference README.md:187-189) and there is
no network, so every measurement uses these.  Entities get Zipf-like
degrees; every entity occurs in at least one triple; entity i of KG1 is aligned with entity i
of KG2; links are split 20% / 10% / 70% (README.md:252-255).
"""
import os

import numpy as np

from .kg import KG
from .kgs import KGs

# name -> (entities per KG, (#rel1, #rel2), (#triples1, #triples2)); docs/Dataset_Statistics.png
SHAPES = {
    "EN-FR-15K-V1": (15000, (267, 210), (47334, 40864)),
    "D-W-15K-V2": (15000, (167, 121), (73983, 83365)),
    "EN-FR-100K-V1": (100000, (400, 300), (309607, 258285)),
    "EN-DE-100K-V1": (100000, (381, 196), (335359, 336240)),
    "EN-FR-100K-V2": (100000, (379, 287), (649902, 561391)),
    "tiny": (400, (12, 9), (1500, 1300)),
    "small": (3000, (40, 31), (10000, 9000)),
}


def _zipf_choice(rng, n, size, a=0.9):
    """indices in [0,n) with P(i) ~ (i+1)^-a (hubs at low indices)."""
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), a)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    return np.minimum(np.searchsorted(cdf, rng.rand(size)), n - 1).astype(np.int64)


def _triples(rng, prefix, n_ent, n_rel, n_tri):
    tri = set()
    # every entity occurs once as a head
    for h, r, t in zip(np.arange(n_ent), _zipf_choice(rng, n_rel, n_ent, 1.1), rng.permutation(n_ent)):
        tri.add((int(h), int(r), int(t)))
    while len(tri) < n_tri:
        m = int((n_tri - len(tri)) * 1.2) + 16
        hs = _zipf_choice(rng, n_ent, m)
        ts = rng.randint(0, n_ent, m)
        rs = _zipf_choice(rng, n_rel, m, 1.1)
        for h, r, t in zip(hs, rs, ts):
            if h != t:
                tri.add((int(h), int(r), int(t)))
                if len(tri) >= n_tri:
                    break
    return {("%s/e%d" % (prefix, h), "%s/r%d" % (prefix, r), "%s/e%d" % (prefix, t)) for h, r, t in tri}


def generate_uri_dataset(shape="EN-FR-15K-V1", seed=0):
    n_ent, (r1, r2), (t1, t2) = SHAPES[shape]
    rng = np.random.RandomState(seed)
    tri1 = _triples(rng, "kg1", n_ent, r1, t1)
    tri2 = _triples(rng, "kg2", n_ent, r2, t2)
    order = rng.permutation(n_ent)
    links = [("kg1/e%d" % i, "kg2/e%d" % i) for i in order]
    n_train, n_valid = int(0.2 * n_ent), int(0.1 * n_ent)
    return tri1, tri2, links[:n_train], links[n_train:n_train + n_valid], links[n_train + n_valid:]


def make_kgs(shape="EN-FR-15K-V1", mode="mapping", ordered=True, seed=0, verbose=False):
    """-> KGs built through the same id-assignment code as a real dataset."""
    tri1, tri2, train, valid, test = generate_uri_dataset(shape, seed)
    return KGs(KG(tri1, set(), verbose=verbose), KG(tri2, set(), verbose=verbose), train, test, valid_links=valid,
               mode=mode, ordered=ordered, verbose=verbose)


def write_dataset(folder, shape="tiny", seed=0, division="721_5fold/1/"):
    """write the reference's folder layout (README.md:204-220) for end-to-end runs."""
    tri1, tri2, train, valid, test = generate_uri_dataset(shape, seed)
    os.makedirs(os.path.join(folder, division), exist_ok=True)

    def dump(path, rows):
        with open(path, "w", encoding="utf8") as f:
            for row in rows:
                f.write("\t".join(row) + "\n")
    dump(os.path.join(folder, "rel_triples_1"), sorted(tri1))
    dump(os.path.join(folder, "rel_triples_2"), sorted(tri2))
    dump(os.path.join(folder, "attr_triples_1"), [])
    dump(os.path.join(folder, "attr_triples_2"), [])
    dump(os.path.join(folder, "ent_links"), train + valid + test)
    dump(os.path.join(folder, division, "train_links"), train)
    dump(os.path.join(folder, division, "valid_links"), valid)
    dump(os.path.join(folder, division, "test_links"), test)
    return folder
