"""where the epoch sampler's 0.37 ms goes (100K shape, 793 K positives x 10): the launch as it is | max_try = 1 (the last try accepts
without probing the membership set: no probes) | uniform candidates (the entity list, 400 KB, instead of the 800 MB neighbour tables)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from openea_amd import ops
from openea_amd.models.trainer import RelationTripleEpochs, refresh_neighbours, EmbeddingTable
from openea_amd.modules.base.initializers import truncated_normal_host
dev = torch.device("cuda", 0)
kgs = bench.cached_kgs("EN-FR-100K-V1", "swapping")
rng = np.random.RandomState(1)
ent = EmbeddingTable(truncated_normal_host(rng, (kgs.entities_num, 100), 0.1), True, "e", dev)
ep = RelationTripleEpochs(kgs, 20000, 10, seed=2, dev=dev)
k1 = int(0.02 * kgs.kg1.entities_num)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
b = ep.batches
S = len(b.splits)
for mode in ("truncated", "uniform"):
    if mode == "truncated":
        ep.set_neighbours(refresh_neighbours(ent, kgs.kg1.entities_list, k1), refresh_neighbours(ent, kgs.kg2.entities_list, k1))
    else:
        ep.set_neighbours(None, None)
    ep._sides = (ep.s1.side(), ep.s2.side())
    neg = ep._epoch_neg_buf()
    for mt in (10, 1):
        us = t(lambda: ops.sample_negatives_epoch(b.dall, ep._off_dev, ep._spl_dev, S, 10, ep._sides[0], ep._sides[1], 2, 0, neg, ep.err, max_try=mt))
        print("%-9s candidates, max_try %2d: %.0f us" % (mode, mt, us), flush=True)
