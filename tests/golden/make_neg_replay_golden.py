"""A RECORDED run of the reference negative sampler, for the replay tests of the device sampler and its oracle.

The reference's generate_neg_triples_fast (modules/train/batch.py:89-119) draws with python's `random.sample` and
`np.random.binomial`; the Philox restatement cannot agree with it draw by draw.  Here the two functions are wrapped while the
reference runs, and every value they return is stored: per positive and round the Bernoulli value and the POSITIONS (in the candidate
list) of the sampled entities.  Fed with this record, the oracle (np_oracle.sample_negatives_replay) and the device kernel
(oea_sample_negatives_replay) must produce the reference's negatives exactly -- which pins the algorithm itself (rounds, one side
per round, distinct draws, removal of true triples except in the last round) against the reference rather than against a
restatement.  Three cases: the inputs of neg_sampling.npz with 40-entry neighbour lists (truncated sampling) and without (uniform),
and a small DENSE graph (12 entities, most triples true) with max_try = 3, where most positives need several rounds and the last
round accepts true triples.
Run in the build container only:  python tests/golden/make_neg_replay_golden.py
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402


def record(ref, pos, tri, ents, k, neighbor, max_try):
    """-> (negatives [n_pos * k, 3] in the reference's order, replay int32 [n_pos, max_try, 1 + k])"""
    calls = []                                    # ('b', value) / ('s', population, sample)
    real_sample, real_binomial = random.sample, np.random.binomial

    def sample(population, n):
        out = real_sample(population, n)
        calls.append(("s", population, out))
        return out

    def binomial(*a, **kw):
        v = real_binomial(*a, **kw)
        calls.append(("b", int(v)))
        return v
    random.sample, np.random.binomial = sample, binomial
    try:
        neg = ref.bat.generate_neg_triples_fast([tuple(map(int, p)) for p in pos], tri, ents, k, neighbor=neighbor, max_try=max_try)
    finally:
        random.sample, np.random.binomial = real_sample, real_binomial
    replay = np.full((len(pos), max_try, 1 + k), -1, np.int32)
    neg = np.asarray(neg, np.int32)
    # the calls come in (binomial, sample) pairs; a positive's rounds end when its k negatives are complete
    ci = 0
    for p in range(len(pos)):
        have = 0
        for i in range(max_try):
            kind, side = calls[ci]
            assert kind == "b"
            _, population, drawn = calls[ci + 1]
            ci += 2
            where = {e: j for j, e in enumerate(population)}
            assert len(where) == len(population) and len(drawn) == k - have
            replay[p, i, 0] = side
            replay[p, i, 1:1 + len(drawn)] = [where[e] for e in drawn]
            h, r, t = map(int, pos[p])
            cand = [(e, r, t) if side else (h, r, e) for e in drawn]
            have += len(cand) if i == max_try - 1 else sum(1 for x in cand if x not in tri)
            if have == k:
                break
        assert have == k
    assert ci == len(calls)
    return neg, replay


def main():
    ref = import_reference()
    g = np.load(os.path.join(HERE, "neg_sampling.npz"))
    out = {}
    tri = set(map(tuple, g["triples"].tolist()))
    ents = g["entity_list"].tolist()
    nbr = {e: g["nbr"][i].tolist() for i, e in enumerate(ents)}
    random.seed(2024)
    np.random.seed(2024)
    for name, neighbor in (("truncated", nbr), ("uniform", None)):
        neg, replay = record(ref, g["pos"], tri, ents, 10, neighbor, 10)
        out["neg_" + name], out["replay_" + name] = neg, replay
    # dense graph: 12 entities, 2 relations, ~70 % of all (h, r, t) are true triples
    rng = np.random.RandomState(7)
    d_ents = list(range(12))
    allt = [(h, r, t) for h in d_ents for r in range(2) for t in d_ents]
    d_tri = [allt[i] for i in np.nonzero(rng.rand(len(allt)) < 0.7)[0]]
    d_pos = np.asarray([d_tri[i] for i in rng.choice(len(d_tri), 64, replace=False)], np.int32)
    neg, replay = record(ref, d_pos, set(d_tri), d_ents, 5, None, 3)
    out.update(dense_triples=np.asarray(d_tri, np.int32), dense_entities=np.asarray(d_ents, np.int32), dense_pos=d_pos,
               neg_dense=neg, replay_dense=replay)
    rounds = [int((out["replay_" + n][:, :, 0] >= 0).sum(1).max()) for n in ("truncated", "uniform", "dense")]
    print("rounds used (max per case):", rounds, " positives with more than one round:",
          [int(((out["replay_" + n][:, :, 0] >= 0).sum(1) > 1).sum()) for n in ("truncated", "uniform", "dense")])
    np.savez_compressed(os.path.join(HERE, "neg_replay.npz"), **out)


if __name__ == "__main__":
    main()
