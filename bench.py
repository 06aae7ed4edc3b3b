#!/usr/bin/env python
"""bench.py -- BASELINE.json metric on MI355X.

Workload (BASELINE.json configs[1]): BootEA/AlignE-style translational step -- truncated
negative sampling (eps = 0.9 -> 1,499 neighbours, k = 10 negatives per positive) + limited
loss + Adagrad -- on a synthetic KG pair with the EN-FR-15K-V1 shape (no dataset on disk),
batch 5,000 positives per GPU, dim = 75 (BASELINE.json; the shipped bootea_args_15K.json
uses 100: pass --dim 100).

A "step" = sample the negatives of one batch on the device + one fused optimiser step.
value = positives (training triples) consumed per second, whole job.

  python bench.py [--gpus N --steps K --warmup W]          (N>1: launched by torch.distributed.run)

One JSON line on stdout (rank 0).  Extra legs, outside the timed region:
  * roofline   -- dominant kernel (triple_fwd_bwd) timed with HIP events on its stream
  * cpu_baseline -- the C oracle port of the same step on ONE host core, bounded sample
  * extra      -- alignment-eval pairs/s (10,500 test pairs, inner + CSLS) and neighbour-search rows/s
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PROFILE_STRIDE = 8        # HIP-event marks on every 8th step of the timed region (a record costs ~3 us of enqueue)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dim", type=int, default=75)
    ap.add_argument("--shape", default="EN-FR-15K-V1")
    ap.add_argument("--batch", type=int, default=5000, help="positives per GPU per step")
    ap.add_argument("--neg", type=int, default=10)
    ap.add_argument("--eps", type=float, default=0.9, help="truncated_epsilon")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the eval / neighbour legs")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    from openea_amd import ops
    from openea_amd.models.trainer import (EmbeddingTable, RelationTripleEpochs, TripleTrainer,
                                            refresh_neighbours)
    from openea_amd.modules.load.synth import make_kgs

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if os.environ.get("OEA_BENCH_ONE_GPU"):       # test hook: all ranks share GPU 0 (with OEA_BENCH_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    ops.lib()
    dev = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("OEA_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        group = dist.group.WORLD

    # ---- data + model state (identical on every rank: same seeds) ---------------------------------
    kgs = make_kgs(args.shape, mode="swapping", seed=0)
    d = args.dim
    rng = np.random.RandomState(1)
    from openea_amd.modules.base.initializers import truncated_normal_host
    ent = EmbeddingTable(truncated_normal_host(rng, (kgs.entities_num, d), 1.0 / np.sqrt(d)), True, "ent_embeds", dev)
    rel = EmbeddingTable(truncated_normal_host(rng, (kgs.relations_num, d), 1.0 / np.sqrt(d)), True, "rel_embeds", dev)
    cfg = ops.make_step_cfg(loss="limited", loss_norm="L2", pos_margin=0.01, neg_margin=2.0, balance=0.2,
                            ent_l2_norm=True, rel_l2_norm=True, optimizer="Adagrad", lr=0.01, neg_group_k=args.neg)
    trainer = TripleTrainer(ent, rel, cfg, "Adagrad", dist_group=group)
    epochs = RelationTripleEpochs(kgs, args.batch * world, args.neg, seed=2, dev=dev, rank=rank, world=world)
    k1 = int((1 - args.eps) * kgs.kg1.entities_num)      # basic_model.py:270-271 (1499 at eps=0.9, N=15000)
    k2 = int((1 - args.eps) * kgs.kg2.entities_num)
    t0 = time.time()
    nbr1 = refresh_neighbours(ent, kgs.kg1.entities_list, k1)
    nbr2 = refresh_neighbours(ent, kgs.kg2.entities_list, k2)
    torch.cuda.synchronize()
    nbr_first_s = time.time() - t0
    epochs.set_neighbours(nbr1, nbr2)

    steps_per_epoch = len(epochs.batches.splits)

    def run_steps(n_steps):
        """n_steps optimiser steps = whole epochs (one enqueue call each on a single GPU) plus a
        per-step tail; returns (positives consumed, triples scored) on this rank."""
        npos = nscored = 0
        done = 0
        while done < n_steps:
            s = epochs.global_step % steps_per_epoch
            if s == 0 and n_steps - done >= steps_per_epoch:
                n = epochs.run_epoch(trainer)
                npos += n
                nscored += n * (1 + args.neg)
                done += steps_per_epoch
                continue
            pos, neg = epochs.batch(s)
            if pos.shape[0] or world > 1:                     # DP: every rank joins every exchange, rows or not
                trainer.step(pos, neg)
            npos += pos.shape[0]
            nscored += pos.shape[0] * (1 + args.neg)
            done += 1
            if epochs.global_step % steps_per_epoch == 0:
                epochs.end_epoch()
        return npos, nscored

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # priming (untimed, before the W warm-up steps): one whole epoch + its end-of-epoch shuffle, so that the
    # first-use costs of both enqueue paths and of the permutation kernels are not inside the timed region
    run_steps(steps_per_epoch - epochs.global_step % steps_per_epoch)
    run_steps(args.warmup)
    barrier()
    ops.profile_begin(stride=PROFILE_STRIDE)
    t0 = time.perf_counter()
    n_pos_total, n_scored = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    (fwd_ms, _gap_ms, apply_ms), n_calls = ops.profile_end(4)
    epoch_loss = trainer.pop_loss()
    epochs.check()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    cnt = torch.tensor([n_pos_total], dtype=torch.float64, device=dev)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed_max = float(t.item())
    total_pos = float(cnt.item())

    if rank != 0:
        return
    value = total_pos / elapsed_max

    # ---- roofline of the dominant kernel (per launch, this rank) ---------------------------------
    launches = max(n_calls, 1)
    alg_bytes_per_launch = 24.0 * d * (n_scored / max(args.steps, 1))     # 24*d B per scored triple
    fwd_avg_s = fwd_ms / 1e3 / launches
    achieved = alg_bytes_per_launch / fwd_avg_s / 1e9 if fwd_avg_s > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_triple_fwd_bwd.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            # PMC traffic was collected on the default workload; it says nothing about another shape
            same = (args.shape, args.dim, args.batch, args.neg) == ("EN-FR-15K-V1", 75, 5000, 10) and world == 1
            traffic = tj.get("hbm_bytes_per_launch") if same else None
        except Exception:
            traffic = None
    roofline = {"kernel": "triple_fwd_bwd", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "avg_kernel_us": round(fwd_avg_s * 1e6, 2), "apply_rows_avg_us": round(apply_ms / launches * 1e3, 2),
                "algorithmic_bytes_per_launch": int(alg_bytes_per_launch),
                "launches_timed": int(n_calls), "launches_total": int(args.steps)}

    extra = {"neighbour_refresh_first_call_s": round(nbr_first_s, 3), "epoch_loss_sum": epoch_loss,
             "triple_steps_per_epoch": steps_per_epoch, "neighbours_k": [k1, k2]}
    if not args.no_extra and world == 1:
        extra.update(extra_legs(torch, ops, ent, kgs, d, k1))
    cpu = None
    if not args.no_cpu and world == 1:
        cpu = cpu_baseline(kgs, d, args, k1, k2)

    out = {
        "metric": "training triples/sec (positives consumed; truncated negative sampling k=%d + limited loss + Adagrad)" % args.neg,
        "value": round(value, 1), "unit": "triples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BootEA/AlignE translational step, %s shape (synthetic), dim=%d, batch=%d positives/GPU, "
                               "k=%d, truncated eps=%.2f" % (args.shape, d, args.batch, args.neg, args.eps),
                   "entities": kgs.entities_num, "relations": kgs.relations_num,
                   "parallelism": "dp%d" % world if world > 1 else "single"},
        "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
    }
    print(json.dumps(out))


def extra_legs(torch, ops, ent, kgs, d, k1):
    """alignment-eval pairs/s and neighbour-search rows/s (second half of BASELINE.json's metric)."""
    from openea_amd.modules.finding.alignment import greedy_alignment_device
    e1 = ent.lookup(kgs.test_entities1)
    e2 = ent.lookup(kgs.test_entities2)
    out = {}
    for name, csls in (("eval_pairs_per_s_inner", 0), ("eval_pairs_per_s_inner_csls10", 10)):
        greedy_alignment_device(e1, e2, d, [1, 5, 10, 50], "inner", False, csls)       # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            greedy_alignment_device(e1, e2, d, [1, 5, 10, 50], "inner", False, csls)
        torch.cuda.synchronize()
        out[name] = round(e1.shape[0] * reps / (time.perf_counter() - t0), 1)
    greedy_alignment_device(e1, e2, d, [1, 5, 10, 50], "manhattan", False, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    greedy_alignment_device(e1, e2, d, [1, 5, 10, 50], "manhattan", False, 0)
    torch.cuda.synchronize()
    out["eval_pairs_per_s_manhattan"] = round(e1.shape[0] / (time.perf_counter() - t0), 1)
    from openea_amd.models.trainer import refresh_neighbours
    refresh_neighbours(ent, kgs.kg1.entities_list, k1)                                 # warm (first-use allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        refresh_neighbours(ent, kgs.kg1.entities_list, k1)
    torch.cuda.synchronize()
    out["neighbour_rows_per_s"] = round(len(kgs.kg1.entities_list) * reps / (time.perf_counter() - t0), 1)
    out["eval_pairs"] = int(e1.shape[0])
    return out


def cpu_baseline(kgs, d, args, k1, k2):
    """The C oracle port of the same step (sampler + fused step), ONE host thread, a bounded
    sample of the same workload.  Reported baseline, not the target."""
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import cport
    rng = np.random.RandomState(1)
    from openea_amd.modules.base.initializers import truncated_normal_host
    ent = truncated_normal_host(rng, (kgs.entities_num, d), 1.0 / np.sqrt(d))
    rel = truncated_normal_host(rng, (kgs.relations_num, d), 1.0 / np.sqrt(d))
    ent_acc, rel_acc = np.full_like(ent, 0.1), np.full_like(rel, 0.1)
    t1 = np.asarray(kgs.kg1.relation_triples_list, np.int32)
    t2 = np.asarray(kgs.kg2.relation_triples_list, np.int32)
    b1 = int(len(t1) / (len(t1) + len(t2)) * args.batch)
    b2 = args.batch - b1
    # candidate lists: uniform-random neighbour lists of the right length (the cost of sampling does
    # not depend on which ids are in the list; computing the real top-k on one core would blow the budget)
    tabs, ents, eposs, nbrs = [], [], [], []
    for kg, k in ((kgs.kg1, k1), (kgs.kg2, k2)):
        e = np.asarray(kg.entities_list, np.int32)
        ep = np.full(kgs.entities_num, -1, np.int32)
        ep[e] = np.arange(len(e), dtype=np.int32)
        tabs.append(cport.tripleset_build(np.asarray(sorted(kg.relation_triples_set), np.int32)))
        ents.append(e)
        eposs.append(ep)
        nbrs.append(e[rng.randint(0, len(e), (len(e), k))].astype(np.int32))
    steps, t0 = 0, time.perf_counter()
    per_epoch = max(min(len(t1) // b1, len(t2) // b2), 1)
    while True:
        s_ = steps % per_epoch
        p1 = t1[s_ * b1:(s_ + 1) * b1]
        p2 = t2[s_ * b2:(s_ + 1) * b2]
        n1 = cport.sample_negatives(p1, args.neg, tabs[0], ents[0], eposs[0], nbrs[0], seed=2, step=steps)
        n2 = cport.sample_negatives(p2, args.neg, tabs[1], ents[1], eposs[1], nbrs[1], seed=2, step=steps)
        cport.triple_step(ent, ent_acc, rel, rel_acc, np.concatenate([p1, p2]), np.concatenate([n1, n2]),
                          loss="limited", loss_norm="L2", pos_margin=0.01, neg_margin=2.0, balance=0.2,
                          optimizer="Adagrad", lr=0.01)
        steps += 1
        el = time.perf_counter() - t0
        if el > 12.0:
            break
    return {"value": round(steps * args.batch / el, 1), "unit": "triples/s", "cores": 1, "kind": "port",
            "sample": "%d steps of the same workload (batch %d, k=%d, dim=%d): oracle/c/oracle.c sampler + step, "
                      "fp64 internals, 1 thread, %.1f s" % (steps, args.batch, args.neg, d, el)}


if __name__ == "__main__":
    main()
