"""standalone cost of what the side stream does per epoch at the 100K shape: layout (shuffle), sampler, plan build"""
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
ep.set_neighbours(refresh_neighbours(ent, kgs.kg1.entities_list, k1), refresh_neighbours(ent, kgs.kg2.entities_list, k1))
ep._sides = (ep.s1.side(), ep.s2.side())
b = ep.batches
neg = ep._epoch_neg_buf()
S = len(b.splits)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
gen = torch.Generator(device=dev); gen.manual_seed(1)
print("positives %d, steps %d" % (int(b.offsets[-1]), S))
print("layout (oea_epoch_layout): %.0f us" % t(lambda: b.shuffle(gen)))
print("sampler (oea_sample_negatives_epoch): %.0f us" % t(lambda: ops.sample_negatives_epoch(b.dall, ep._off_dev, ep._spl_dev, S, 10, ep._sides[0], ep._sides[1], 2, 0, neg, ep.err)))
dims = (int(b.offsets[-1]), S, int(np.diff(b.offsets).max()), kgs.entities_num, 100)
plan = ops.step_plan_buffer(*dims, dev=dev)
print("plan build (oea_step_plan_build): %.0f us" % t(lambda: ops.step_plan_build(b.dall, neg, 10, ep._off_dev, *dims, plan)))
