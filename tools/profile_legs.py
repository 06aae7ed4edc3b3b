#!/usr/bin/env python
"""Workload for rocprofv3 runs of the non-bench legs: alignment eval (inner / CSLS / manhattan) at the
15K and 100K test sizes, neighbour search at 15K / 100K, one GCN-Align epoch (D-W-15K-V2 shape),
one AliNet sparse-attention layer forward+backward.  Prints wall-clock per leg (device-synchronised)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openea_amd import ops  # noqa: E402
from openea_amd.modules.finding.alignment import greedy_alignment_device  # noqa: E402


def unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def timed(name, fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%-48s %10.3f ms" % (name, dt * 1e3))
    return dt


def main():
    ops.lib()
    rng = np.random.RandomState(0)
    only = os.environ.get("LEGS", "")          # LEGS=eval70k: just the 70,000^2 inner eval (PMC passes)
    if only == "csls":
        n, d = 10500, 100
        e1 = unit(rng, n, d)
        t1 = ops.to_table(e1)
        t2 = ops.to_table(e1 + 0.4 * unit(rng, n, d))
        timed("eval csls10  %6d^2 x %d" % (n, d), lambda: greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, 10), reps=5)
        return
    if only == "eval70k":
        n, d = 70000, 100
        e1 = unit(rng, n, d)
        t1 = ops.to_table(e1)
        t2 = ops.to_table(e1 + 0.4 * unit(rng, n, d))
        timed("eval inner   %6d^2 x %d" % (n, d), lambda: greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, 0), reps=1)
        return
    for n, d in (() if only == "gnn" else ((10500, 100), (70000, 100), (10500, 300))):
        e1 = unit(rng, n, d)
        t1 = ops.to_table(e1)
        t2 = ops.to_table(e1 + 0.4 * unit(rng, n, d))
        dt = timed("eval inner   %6d^2 x %d" % (n, d), lambda: greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, 0))
        print("   -> %.1f TFLOP/s (2*N1*N2*d), %.0f pairs/s" % (2.0 * n * n * d / dt / 1e12, n / dt))
        if n <= 10500:
            timed("eval csls10  %6d^2 x %d" % (n, d), lambda: greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "inner", False, 10))
            timed("eval manhattan %6d^2 x %d" % (n, d), lambda: greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "manhattan", False, 0), reps=2)
    for n, d, k in (() if only == "gnn" else ((15000, 100, 1499), (100000, 100, 2000))):
        t = ops.to_table(unit(rng, n, d))
        dt = timed("neighbours   %6d x %d, k=%d" % (n, d, k), lambda: ops.topk_inner(t, t, d, k), reps=2)
        print("   -> %.0f query rows/s" % (n / dt))
    # GCN-Align SE epoch, D-W-15K-V2 shape
    from openea_amd.approaches.gcn_align import GCN_Align
    from openea_amd.modules.load.synth import make_kgs
    from openea_amd.run.default_args import get_args
    kgs = make_kgs("D-W-15K-V2", mode="mapping", seed=0)
    m = GCN_Align()
    m.set_args(get_args("GCN_Align", output="/tmp/oea_out/", training_data="synthetic/dw15k/", dataset_division="f/"))
    m.set_kgs(kgs)
    m.init()
    se = m.model_se
    train = np.asarray(kgs.train_links, np.int32)
    k, t = m.args.neg_triple_num, len(train)
    negs = tuple(ops.to_ids(x.astype(np.int32)) for x in (np.repeat(train[:, 0], k), rng.choice(kgs.entities_num, t * k),
                                                           rng.choice(kgs.entities_num, t * k), np.repeat(train[:, 1], k)))
    dt = timed("GCN-Align SE epoch (E=%d, nnz=%d, d=100)" % (kgs.entities_num, se.adj.nnz), lambda: se.train_step(negs), reps=20)
    nnz, n, d = se.adj.nnz, kgs.entities_num, 100
    alg = 4 * (nnz * (8 + 4 * d) + 4 * n * d) + 4 * (2 * t + 4 * t * k) * d + 12 * n * d
    print("   -> algorithmic %.1f MB/epoch, %.0f GB/s" % (alg / 1e6, alg / dt / 1e9))
    # sparse attention layer (AliNet-like): E=30000, 2-hop nnz ~ 1e6, d=400
    from openea_amd.models.graph_ops import EdgeGraph, sparse_attention
    n, nnz, d = 30000, 1000000, 400
    rows = np.minimum(rng.zipf(1.5, nnz) - 1, n - 1)
    g = EdgeGraph(rows, rng.randint(0, n, nnz), rng.rand(nnz).astype(np.float32), (n, n), ops.device())
    z = torch.randn(g.nnz, device=g.dev, requires_grad=True)
    v = torch.randn(n, d, device=g.dev, requires_grad=True)

    def attn():
        out = sparse_attention(g, z, v)
        out.sum().backward()
    dt = timed("sparse attention fwd+bwd (N=%d, nnz=%d, d=%d)" % (n, g.nnz, d), attn, reps=5)


if __name__ == "__main__":
    main()
