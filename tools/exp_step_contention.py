#!/usr/bin/env python
"""Experiment: how much of the fused step's time is same-row atomic contention?  Times GRAD / APPLY phases of
one step (5,000 positives, k = 10 grouped negatives, d = 75) for uniform vs Zipf-distributed heads."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openea_amd import ops  # noqa: E402


def run(name, heads, n_ent=30000, n_rel=500, d=75, n=5000, k=10, rel_zipf=True):
    rng = np.random.RandomState(0)
    ent = ops.to_table(rng.standard_normal((n_ent, d)).astype(np.float32))
    rel = ops.to_table(rng.standard_normal((n_rel, d)).astype(np.float32))
    ea, ra = torch.full_like(ent, 0.1), torch.full_like(rel, 0.1)
    r = np.minimum(rng.zipf(1.5, n) - 1, n_rel - 1) if rel_zipf else rng.randint(0, n_rel, n)
    pos = np.stack([heads, r, rng.randint(0, n_ent, n)], 1).astype(np.int32)
    neg = np.repeat(pos, k, axis=0)
    corrupt_h = rng.rand(n * k) < 0.5
    rnd = rng.randint(0, n_ent, n * k)
    neg[corrupt_h, 0] = rnd[corrupt_h]
    neg[~corrupt_h, 2] = rnd[~corrupt_h]
    cfg = ops.make_step_cfg(loss='limited', pos_margin=0.01, neg_margin=2.0, balance=0.2, neg_group_k=k)
    ws = ops.step_workspace(n_ent, n_rel, ent.shape[1])
    loss = torch.zeros(1, dtype=torch.float64, device=ent.device)
    p, q = ops.to_ids(pos), ops.to_ids(neg)
    for phase, label in ((ops.PHASE_GRAD, "grad"), (ops.PHASE_APPLY, "apply")):
        pass
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tg = ta = 0.0
    reps = 50
    for it in range(reps + 5):
        ev[0].record()
        ops.triple_step(ent, ea, rel, ra, d, p, q, cfg, ws, loss, phase=ops.PHASE_GRAD)
        ev[1].record()
        ops.triple_step(ent, ea, rel, ra, d, p, q, cfg, ws, loss, phase=ops.PHASE_APPLY)
        ev[2].record()
        torch.cuda.synchronize()
        if it >= 5:
            tg += ev[0].elapsed_time(ev[1])
            ta += ev[1].elapsed_time(ev[2])
    print("%-28s grad %7.1f us   apply %7.1f us" % (name, tg / reps * 1e3, ta / reps * 1e3))


def main():
    ops.lib()
    rng = np.random.RandomState(1)
    n, n_ent = 5000, 30000
    run("uniform heads", rng.randint(0, n_ent, n))
    run("uniform heads+rels", rng.randint(0, n_ent, n), rel_zipf=False)
    w = 1.0 / np.arange(1, n_ent + 1) ** 0.9
    run("zipf(0.9) heads", rng.choice(n_ent, n, p=w / w.sum()))
    w = 1.0 / np.arange(1, n_ent + 1) ** 1.2
    run("zipf(1.2) heads", rng.choice(n_ent, n, p=w / w.sum()))
    run("one head", np.zeros(n, np.int64))


if __name__ == "__main__":
    main()
