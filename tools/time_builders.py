#!/usr/bin/env python
"""Host (numpy restatements) vs device (csrc/graph_build.hip) adjacency builders at the EN-FR-100K-V1 shape.
Run on the GPU box: python tools/time_builders.py [shape]"""
import contextlib
import io
import os
import sys
import time
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from openea_amd import ops  # noqa: E402
from openea_amd.approaches import alinet, rdgcn  # noqa: E402
from openea_amd.approaches.gcn_align import GCN_Utils  # noqa: E402
from openea_amd.modules.load.synth import make_kgs  # noqa: E402


def timed(label, fn, *a, **kw):
    torch.cuda.synchronize()
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        out = fn(*a, **kw)
    torch.cuda.synchronize()
    print("%-58s %8.3f s" % (label, time.time() - t), flush=True)
    return out


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "EN-FR-100K-V1"
    ops.lib()
    kgs = make_kgs(shape, mode="mapping", seed=0)
    triples = kgs.kg1.relation_triples_list + kgs.kg2.relation_triples_list
    n_ent, n_rel = kgs.entities_num, kgs.relations_num
    print("shape %s: %d entities, %d relations, %d triples" % (shape, n_ent, n_rel, len(triples)))
    tri = np.asarray(triples, np.int64)
    ops.build_primal_adj(tri[:1000], n_ent)                     # first-use costs (module load, pool) out of the timings
    u = GCN_Utils(types.SimpleNamespace(), kgs)
    timed("GCN-Align host: get_weighted_adj + preprocess_adj", lambda: u.preprocess_adj(u.get_weighted_adj(n_ent, triples)))
    timed("GCN-Align device: oea_build_weighted_adj (from list)", ops.build_weighted_adj, triples, n_ent, n_rel)
    timed("GCN-Align device: oea_build_weighted_adj (from array)", ops.build_weighted_adj, tri, n_ent, n_rel)
    kg1 = timed("AliNet AKG(kg1) (python sets, both paths)", alinet.AKG, kgs.kg1.relation_triples_set)
    linked = set(kgs.train_entities1 + kgs.valid_entities1 + kgs.test_entities1)
    a = timed("AliNet host: generate_2hop_triples(kg1)", alinet.generate_2hop_triples, kg1, linked, as_array=True)
    d = timed("AliNet device: generate_2hop_triples_device(kg1)", alinet.generate_2hop_triples_device, kg1, linked)
    print("   2-hop triples equal:", a.shape == d.shape and bool(np.array_equal(a, d)), a.shape)
    timed("AliNet host: no_weighted_adj (1-hop)", alinet.no_weighted_adj, n_ent, tri)
    timed("AliNet device: no_weighted_adj_device (1-hop)", alinet.no_weighted_adj_device, n_ent, tri)
    timed("AliNet host: no_weighted_adj (2-hop)", alinet.no_weighted_adj, n_ent, a)
    timed("AliNet device: no_weighted_adj_device (2-hop)", alinet.no_weighted_adj_device, n_ent, a)
    timed("RDGCN host: get_sparse_tensor", rdgcn.get_sparse_tensor, triples, n_ent)
    timed("RDGCN device: oea_build_primal_adj", ops.build_primal_adj, tri, n_ent)
    head, tail, _, _ = timed("RDGCN host: rfunc (python sets)", rdgcn.rfunc, triples, n_ent, n_rel)
    hd = timed("RDGCN host: dual_adjacency", rdgcn.dual_adjacency, head, tail, len(head))
    dd = timed("RDGCN device: oea_build_dual_adj", ops.build_dual_adj, tri, len(head))
    print("   dual adjacency equal:", bool(np.array_equal(hd, dd.cpu().numpy())))


if __name__ == "__main__":
    main()
