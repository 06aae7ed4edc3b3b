"""Statistics of the REFERENCE negative sampler over many runs, for the distributional comparison with the Philox
restatement (the two cannot agree draw by draw: the reference uses python's `random`).

Inputs = tests/golden/neg_sampling.npz (triples, entity list, 256 positives, 40-entry neighbour lists).  The reference's
generate_neg_triples_fast (modules/train/batch.py:89-119) is run RUNS times with neighbour lists and RUNS times without;
stored: how often a positive's family of k negatives corrupts the head, how often a family mixes both sides (retries
after a true triple was drawn), and how often each POSITION of the candidate list was drawn.
Run in the build container only:  python tests/golden/make_neg_stats_golden.py
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402

RUNS, K = 300, 10


def family_stats(neg, pos, cand_pos_head, cand_pos_tail, n_cand):
    """neg [n_pos * K, 3] -> (#families with >= 1 head corruption that are pure head, #mixed, position histogram)"""
    neg = np.asarray(neg).reshape(len(pos), K, 3)
    head_side = neg[:, :, 0] != pos[:, None, 0]
    tail_side = neg[:, :, 2] != pos[:, None, 2]
    pure_head = int((head_side.all(1)).sum())
    mixed = int((head_side.any(1) & tail_side.any(1)).sum())
    hist = np.zeros(n_cand, np.int64)
    for p in range(len(pos)):
        for j in range(K):
            if head_side[p, j]:
                hist[cand_pos_head[p][int(neg[p, j, 0])]] += 1
            elif tail_side[p, j]:
                hist[cand_pos_tail[p][int(neg[p, j, 2])]] += 1
    return pure_head, mixed, hist


def main():
    ref = import_reference()
    g = np.load(os.path.join(HERE, 'neg_sampling.npz'))
    tri = set(map(tuple, g['triples'].tolist()))
    ents = g['entity_list'].tolist()
    pos = g['pos']
    nbr = {e: g['nbr'][i].tolist() for i, e in enumerate(ents)}
    out = {}
    for name, neighbor in (('truncated', nbr), ('uniform', None)):
        random.seed(123)
        np.random.seed(123)
        lists_h = [nbr[int(h)] if neighbor else ents for h in pos[:, 0]]
        lists_t = [nbr[int(t)] if neighbor else ents for t in pos[:, 2]]
        cph = [{e: i for i, e in enumerate(lst)} for lst in lists_h]
        cpt = [{e: i for i, e in enumerate(lst)} for lst in lists_t]
        n_cand = len(lists_h[0])
        pure, mixed, hist = 0, 0, np.zeros(n_cand, np.int64)
        for _ in range(RUNS):
            neg = ref.bat.generate_neg_triples_fast([tuple(map(int, p)) for p in pos], tri, ents, K, neighbor=neighbor)
            a, b, h = family_stats(neg, pos, cph, cpt, n_cand)
            pure, mixed, hist = pure + a, mixed + b, hist + h
        out[name + '_pure_head'] = np.array([pure])
        out[name + '_mixed'] = np.array([mixed])
        out[name + '_hist'] = hist
        print(name, 'pure head families', pure, 'of', RUNS * len(pos), 'mixed', mixed, 'hist min/max', hist.min(), hist.max())
    out['runs'] = np.array([RUNS])
    np.savez_compressed(os.path.join(HERE, 'neg_stats.npz'), **out)


if __name__ == '__main__':
    main()
