#!/usr/bin/env python
"""bench.py -- BASELINE.json metric on MI355X.

Workload, THE SAME AT EVERY N: the BootEA/AlignE-style translational step -- truncated negative sampling (eps = 0.98 -> 2,000
neighbours, k = 10 negatives per positive) + limited loss + Adagrad -- on a synthetic KG pair with the EN-FR-100K-V1 shape (no
dataset on disk), dim = 100, batch 20,000 positives (bootea_args_100K.json): the larger of the two shapes BASELINE.json's
metric names, HBM-resident (2 x 80 MB of tables + state + 80 MB of gradient scratch), and the configuration north_star shards.
BASELINE config 2 (EN-FR-15K-V1, dim 75, batch 5,000: 9 MB tables, cache-resident) is measured the same way in the same run
as the side block extra.shape_15k.

A "step" = the negatives of one batch drawn on the device + one fused optimiser step, enqueued the way the product
does it (RelationTripleEpochs.run_steps -> oea_triple_epoch_range: ONE C call per epoch touched, next epoch's shuffle
and negatives on a side stream).  value = positives (training triples) consumed per second, whole job.

  python bench.py [--gpus N --steps K --warmup W] [--scaling weak|strong] [--exchange step|epoch|allreduce]

N > 1: `python bench.py --gpus N` launches its own N ranks (torch.distributed.run on 127.0.0.1, one process per GPU,
RCCL); started under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.  The batch of 20,000 stays GLOBAL (STRONG
scaling: every rank scores 20,000 / N positives of every batch) with the PARITY-PRESERVING exchange (--exchange step: entity
rows owned by id mod N; the N-rank job equals the single-GPU job) moving only the BOUNDARY rows a step's batch refers to
(--exchange halo: all-to-all of the gradient rows to their owners, all-to-all of the rows the next step reads; --exchange step
= the dense reduce-scatter / all-gather of every owned row, a side leg), one C call per epoch over the C ABI's RCCL
communicator (oea_triple_epoch_range_halo), with HIP-event phase times.  The same configuration on ONE GPU is timed in the same run
(extra.single_gpu_same_config).  --exchange epoch (local SGD, one exchange per epoch: drifts from the single-GPU job,
tests/test_dist_gpu.py) and --scaling weak are named side legs.

Timing: W untimed warm-up steps, then the K-step region -- barrier + synchronize on both sides, MAX over ranks --
is timed --repeats times (default 50) and the MEDIAN region is reported (a single region is a few ms at this shape: one
sample of it says little).

OUTPUT.  The LAST line of stdout (rank 0) is ONE compact JSON object of at most 4 KB: the contract's keys, `roofline`
(triple_wave, the scoring kernel of the step: HIP events on its dispatches, design bytes, counter traffic), `roofline_eval`, `cpu_baseline` and a few dozen
scalars under `extra`.  Everything else -- per-kernel counter dictionaries, region times, formulas, notes, provenance -- goes
to bench_detail.json (repo root; also gpurun_out/ when that directory exists), named by the line's `detail` key:
  roofline      dominant kernel (triple_wave, fwd + bwd; triple_grouped under OEA_STEP_WAVE=0), timed live by HIP events attached to its dispatches.
                `frac` is priced on the bytes the kernel is DESIGNED to move -- it shares the positive's three rows across
                its k negatives: 8*d*(3+k) B per positive (read 3+k rows, accumulate 3+k gradient rows); SURVEY 8d's
                24*d B per scored triple (which counts those rows once per triple) is kept as `frac_sec8d` in the detail.
                `traffic` = HBM bytes per launch from FETCH_SIZE / WRITE_SIZE, collected IN THIS RUN by two rocprofv3 --pmc
                passes over a short child run of the same workload
  cpu_baseline  the C oracle port of the same step on 1 host thread and on more host cores (OpenMP), bounded sample
  extra         alignment-eval pairs/s (device-resident tables AND the reference-signature call: numpy in, pairs out),
                neighbour rows/s, the EN-FR-15K-V1 shape (extra.shape_15k), and the GNN legs of BASELINE.json configs 3-5
                (extra.gnn): GCN-Align SE epoch at the D-W-15K-V2 shape, AliNet at the EN-DE-100K-V1 shape (epoch + sparse
                attention forward + backward under both softmax groupings) and its evaluation at 70,000^2 x 1,200 (inner,
                CSLS 10), RDGCN's evaluation metric (manhattan, d = 300, 70,000 pairs).  bf16-split sweeps are priced
                against 2.5 PFLOP/s / 3, fp32 sweeps against 157.3 TFLOP/s.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TOPS = 39.3   # fp64 vector ADD/SUB rate = half of the 78.6 TFLOP/s FMA peak (256 CUs x 64 lanes x 2.4 GHz)
PROFILE_STRIDE = int(os.environ.get("OEA_BENCH_PROFILE_STRIDE", "1"))   # steps of the kernel-timing regions that carry HIP events
KERNEL_TIMING_REGIONS = 2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=50, help="timed K-step regions (median reported)")
    ap.add_argument("--dim", type=int, default=None, help="default: 75 (15K shapes, BASELINE config 2) / 100 (100K shapes)")
    ap.add_argument("--shape", default=None, help="default: EN-FR-100K-V1 at every N (EN-FR-15K-V1 is the side block extra.shape_15k)")
    ap.add_argument("--batch", type=int, default=None,
                    help="positives per GPU per step (weak) / per job (strong); default 5,000 (15K shapes) / 20,000 (100K shapes)")
    ap.add_argument("--neg", type=int, default=10)
    ap.add_argument("--eps", type=float, default=None, help="truncated_epsilon; default 0.9 (15K) / 0.98 (100K)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None, help="default: strong at N > 1 (the BASELINE batch)")
    ap.add_argument("--exchange", choices=("halo", "epoch", "step", "allreduce"), default="halo",
                    help="N > 1: how the ranks exchange (models/trainer.py:TripleTrainer).  halo (default): the entity-id partition "
                         "moving only the BOUNDARY rows a step's batch refers to (all-to-all of gradient rows to their owners, "
                         "all-to-all of the rows the next step reads; the N-rank job EQUALS the single-GPU job); step: the same "
                         "partition with a dense reduce-scatter / all-gather of every owned row per step; epoch: local steps on "
                         "each rank's share of the batch, one exchange per epoch (local SGD: drifts).  step and epoch are measured "
                         "too (extra.other_exchange, extra.local_sgd)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the eval / neighbour / 100K-shape legs")
    ap.add_argument("--no-gnn", action="store_true", help="skip the GNN legs (BASELINE configs 3-5)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run rocprofv3 FETCH_SIZE / WRITE_SIZE passes")
    ap.add_argument("--leg", default=None, help="internal: child mode of a rocprofv3 counter pass (pmc_gnn)")
    return ap.parse_args()


def resolve_defaults(args, world):
    """ONE workload at every N: the EN-FR-100K-V1 shape, dim 100, the shipped batch of 20,000
    (bootea_args_100K.json) -- the larger of the two shapes BASELINE.json's metric names ("EN-FR 15K/100K"); it fits one GPU,
    its tables (80 MB + state) are HBM-resident so the roofline fraction is a bandwidth statement, and it is the configuration
    north_star shards, so the N = 1 and N = 8 values are the same problem.  N > 1 keeps the batch GLOBAL (strong scaling).  The
    EN-FR-15K-V1 shape (BASELINE config 2: dim 75, batch 5,000) is measured the same way in the same run as the named side
    block `shape_15k` (N = 1)."""
    if args.shape is None:
        args.shape = "EN-FR-100K-V1"
    big = "100K" in args.shape
    if args.dim is None:
        args.dim = 100 if big else 75
    if args.batch is None:
        args.batch = 20000 if big else 5000
    if args.eps is None:
        args.eps = 0.98 if big else 0.9
    if args.scaling is None:
        args.scaling = "strong" if world > 1 else "weak"
    return args


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, rendezvous on
    127.0.0.1) and pass rank 0's JSON line through."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


KGS_CACHE_VERSION = 2        # bump when modules/load/synth.py:make_kgs changes what it generates


def _cache_dir():
    """a directory only this user can write (0700, owned by us, not a symlink): the synthetic-KG cache is a pickle, and a
    pickle from a path another user could have pre-created is code execution"""
    d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "oea_bench_cache_%d" % os.getuid())
    try:
        os.makedirs(d, mode=0o700, exist_ok=True)
        st = os.lstat(d)
        import stat
        if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022):
            return None
        return d
    except OSError:
        return None


def cached_kgs(shape, mode):
    """synthetic KG pair of a BASELINE shape (seed 0); pickled in a private directory so that the child runs of this bench
    (the rocprofv3 counter passes) and the other ranks do not rebuild it"""
    import pickle
    from openea_amd.modules.load.synth import make_kgs
    d = _cache_dir()
    p = os.path.join(d, "kgs_v%d_%s_%s.pkl" % (KGS_CACHE_VERSION, shape, mode)) if d else None
    if p and os.path.exists(p):
        try:
            if os.lstat(p).st_uid == os.getuid():
                with open(p, "rb") as f:
                    return pickle.load(f)
        except Exception:
            pass
    kgs = make_kgs(shape, mode=mode, seed=0)
    if p:
        try:
            tmp = p + ".%d" % os.getpid()
            with open(tmp, "wb") as f:
                pickle.dump(kgs, f, protocol=pickle.HIGHEST_PROTOCOL)
            os.replace(tmp, p)
        except Exception:
            pass
    return kgs


class Workload:
    """tables + samplers + trainer of one shape, and the timed K-step regions on it"""

    def __init__(self, torch, ops, shape, dim, batch, neg, eps, dev, rank=0, world=1, group=None, scaling="weak", exchange=None):
        from openea_amd.models.trainer import EmbeddingTable, RelationTripleEpochs, TripleTrainer, refresh_neighbours
        from openea_amd.modules.base.initializers import truncated_normal_host
        self.torch, self.ops, self.shape, self.d, self.batch, self.neg, self.eps = torch, ops, shape, dim, batch, neg, eps
        self.world, self.rank = world, rank
        self.global_batch = batch * world if scaling == "weak" else batch
        # ---- data + model state (identical on every rank: same seeds) ---------------------------------
        self.kgs = kgs = cached_kgs(shape, "swapping")
        rng = np.random.RandomState(1)
        self.ent = EmbeddingTable(truncated_normal_host(rng, (kgs.entities_num, dim), 1.0 / np.sqrt(dim)), True, "ent_embeds", dev)
        self.rel = EmbeddingTable(truncated_normal_host(rng, (kgs.relations_num, dim), 1.0 / np.sqrt(dim)), True, "rel_embeds", dev)
        cfg = ops.make_step_cfg(loss="limited", loss_norm="L2", pos_margin=0.01, neg_margin=2.0, balance=0.2,
                                ent_l2_norm=True, rel_l2_norm=True, optimizer="Adagrad", lr=0.01, neg_group_k=neg)
        self.trainer = TripleTrainer(self.ent, self.rel, cfg, "Adagrad", dist_group=group, exchange=exchange)
        self.epochs = RelationTripleEpochs(kgs, self.global_batch, neg, seed=2, dev=dev, rank=rank, world=world)
        self.k1 = int((1 - eps) * kgs.kg1.entities_num)      # basic_model.py:270-271 (1499 at eps=0.9, N=15000)
        self.k2 = int((1 - eps) * kgs.kg2.entities_num)
        t0 = time.time()
        nbr1 = refresh_neighbours(self.ent, kgs.kg1.entities_list, self.k1)
        nbr2 = refresh_neighbours(self.ent, kgs.kg2.entities_list, self.k2)
        torch.cuda.synchronize()
        self.nbr_first_s = time.time() - t0
        self.epochs.set_neighbours(nbr1, nbr2)
        self.steps_per_epoch = len(self.epochs.batches.splits)

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        self.torch.cuda.synchronize()

    def measure(self, steps, warmup, repeats):
        """-> dict(times_s [repeats] (max over ranks), pos [repeats] (all ranks), kernel marks)."""
        torch, ops, ep, tr = self.torch, self.ops, self.epochs, self.trainer
        # priming (untimed, before the W warm-up steps): one whole epoch + its end-of-epoch shuffle, so that the
        # first-use costs of the enqueue path and of the permutation kernels are not inside the timed regions
        ep.run_steps(tr, self.steps_per_epoch - ep.in_epoch)
        ep.run_steps(tr, warmup)
        self.barrier()
        times, pos = [], []
        prof = None

        def kernel_timing():
            # the same K-step region, every PROFILE_STRIDE-th step carrying HIP events attached to its kernels' dispatches.  Kept out
            # of the throughput regions: a dispatch with events is not pipelined behind its predecessor (42.8 instead of 33.7 us per
            # step with events on every step at the 15K shape).  Run in the MIDDLE of the repeats: the step's cost grows with training
            # (DESIGN 4.1b), and the kernel's average over the timed regions is what `roofline` is about -- behind the last region it
            # read the heaviest state of the run (54 us where the kernel trace of the same command averages 49)
            ops.profile_begin(stride=PROFILE_STRIDE)
            for _ in range(KERNEL_TIMING_REGIONS):
                ep.run_steps(tr, steps)
            self.barrier()
            return ops.profile_end(4)
        for rep in range(repeats):
            if rep == repeats // 2:
                prof = kernel_timing()
            self.barrier()
            t0 = time.perf_counter()
            n = ep.run_steps(tr, steps)                 # exactly K optimiser steps
            self.barrier()
            times.append(time.perf_counter() - t0)
            pos.append(n)
        if prof is None:
            prof = kernel_timing()
        (fwd_ms, gap_ms, apply_ms), n_calls = prof
        loss = tr.pop_loss()
        ep.check()
        touched = self.touched_rows()
        plan_stats = None
        try:
            if getattr(ep, "_plan", None) is not None and getattr(ep, "_plan_ready", False):
                plan_stats = ops.step_plan_stats(ep._plan, *ep._plan_dims)
        except Exception as e:      # noqa: BLE001 -- a diagnostic, never the bench line
            plan_stats = {"error": str(e)[:200]}
        t = torch.tensor(times, dtype=torch.float64, device=self.ent.var.device)
        c = torch.tensor(pos, dtype=torch.float64, device=self.ent.var.device)
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
        return dict(times=t.cpu().numpy(), pos=c.cpu().numpy(), pos_local=np.asarray(pos, np.float64),
                    fwd_ms=fwd_ms, gap_ms=gap_ms, apply_ms=apply_ms, n_calls=n_calls, loss=loss, touched_rows=touched, plan_stats=plan_stats)

    def touched_rows(self, n_steps=3):
        """entity + relation rows that RECEIVE gradient per step on this rank, counted: after the timed regions a few further steps
        of the current epoch are run through the per-step API in two phases (GRAD | APPLY) and the touched flags of the atomic
        scratch are counted in between -- what the optimiser has to visit, and what `step_frac` prices at 20*d B per row (VERDICT
        r05: the whole table over-counted by a third; the rows a batch merely REFERS to -- inactive negatives included -- would
        still over-count by 3x at the bench's state)"""
        torch, ops, ep, tr = self.torch, self.ops, self.epochs, self.trainer
        b, k = ep.batches, self.neg
        neg_all = getattr(ep, "_neg_all", None)
        if self.world > 1 or neg_all is None or not getattr(ep, "_epoch_negs_ready", False) or getattr(tr, "ent_acc", None) is None:
            return None
        ent, rel = tr.ent, tr.rel
        flags = ops.step_entity_flags(tr.ws, ent.rows, rel.rows, ent.ld)
        out = []
        for s in range(min(n_steps, len(b.splits))):
            o0, o1 = int(b.offsets[s]), int(b.offsets[s + 1])
            if o1 <= o0:
                continue
            p, n = b.dall[o0:o1], neg_all[o0 * k:o1 * k]
            ops.triple_step(ent.var, tr.ent_acc, rel.var, tr.rel_acc, ent.dim, p, n, tr.cfg, tr.ws, tr.loss, phase=ops.PHASE_GRAD)
            rows = int((flags != 0).sum().item())
            ops.triple_step(ent.var, tr.ent_acc, rel.var, tr.rel_acc, ent.dim, p, n, tr.cfg, tr.ws, tr.loss, phase=ops.PHASE_APPLY)
            out.append(rows + int(torch.unique(p[:, 1]).numel()))
        tr.pop_loss()
        return float(np.mean(out)) if out else None

    def phase_times(self, steps):
        """N > 1, one C call per epoch (oea_triple_epoch_range_comm): HIP events at the phase boundaries of `steps` further
        steps -> microseconds per step and phase on this rank (the collectives' events include the wait for the slowest rank)"""
        comm = getattr(self.trainer, "comm", None)
        if comm is None or self.trainer.part is None:
            return {"note": "exchange mode '%s' is not driven by the one-call partitioned epoch: no phase events" % self.trainer.exchange}
        comm.profile_begin()
        self.epochs.run_steps(self.trainer, steps)
        self.barrier()
        ms, n = comm.profile_end()
        out = {"%s_us" % k: round(v / max(n, 1) * 1e3, 2) for k, v in ms.items()}
        out.update({"steps_timed": n, "sum_us": round(sum(ms.values()) / max(n, 1) * 1e3, 2), "communicator": comm.description(),
                    "note": "HIP events on the launch stream at the phase boundaries of oea_triple_epoch_range_comm, rank 0; "
                            "reduce_scatter includes the small all-reduce of the relation rows; events between the phases stop "
                            "the kernels from pipelining, so sum_us is a little above ms_per_step"})
        return out

    def summarize(self, m, steps, traffic=None):
        """median region -> (value, ms_per_step, roofline dict)"""
        order = np.argsort(m["times"])
        i_med = int(order[len(order) // 2])
        t_med = float(m["times"][i_med])
        value = float(m["pos"][i_med]) / t_med
        ms_per_step = t_med / steps * 1e3
        d, neg = self.d, self.neg
        launches = max(m["n_calls"], 1)
        pos_per_launch = float(m["pos_local"].mean()) / steps                      # this rank's positives per launch
        scored_per_launch = pos_per_launch * (1 + neg)
        sec8d_bytes = 24.0 * d * scored_per_launch                                 # 24*d B per scored triple
        design_bytes = 8.0 * d * (3 + neg) * pos_per_launch                        # rows the grouped kernel reads + accumulates
        fwd_s = m["fwd_ms"] / 1e3 / launches
        apply_s = m["apply_ms"] / 1e3 / launches
        achieved = design_bytes / fwd_s / 1e9 if fwd_s > 0 else 0.0
        sec8d = sec8d_bytes / fwd_s / 1e9 if fwd_s > 0 else 0.0
        traffic = traffic or {}
        tb = traffic.get("hbm_bytes_per_launch")
        # whole step: scoring + Adagrad (20*d B per touched row: value and accumulator read + written, gradient read).  Touched rows
        # are COUNTED (touched_rows: the flags of the atomic scratch after a GRAD phase on the epoch's own batches)
        counted = m.get("touched_rows")
        touched = counted if counted else min(self.kgs.entities_num + self.kgs.relations_num, scored_per_launch * 2)
        step_design = design_bytes + 20.0 * d * touched
        roofline = {"kernel": "triple_wave (fwd + bwd; triple_grouped where OEA_STEP_WAVE=0 or the shape is outside its rule)", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": tb, "traffic_source": traffic.get("source"),
                    "hbm_frac": round(tb / fwd_s / 1e9 / HBM_PEAK_GBS, 4) if (tb and fwd_s > 0) else None,
                    "design_bytes_per_launch": int(design_bytes),
                    "achieved_sec8d": round(sec8d, 1), "frac_sec8d": round(min(sec8d / HBM_PEAK_GBS, 1.0), 4),
                    "sec8d_bytes_per_launch": int(sec8d_bytes),
                    "step_frac": round(step_design / (ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                    "design_bytes_per_step": int(step_design), "touched_rows_per_step": int(touched),
                    "touched_rows_source": "counted: touched flags after a GRAD phase on the epoch's batches" if counted else "assumed (min(table rows, 2 * scored triples))",
                    "avg_kernel_us": round(fwd_s * 1e6, 2), "apply_rows_avg_us": round(apply_s * 1e6, 2),
                    "gap_between_kernels_us": round(m["gap_ms"] * 1e3 / launches, 2),
                    "launches_timed": int(m["n_calls"]), "step_plan": m.get("plan_stats"),
                    "timing": "HIP start/stop events attached to the kernel's dispatch (hipExtLaunchKernelGGL) on every "
                              "%d-th step of %d further K-step regions run half-way through the throughput regions (events on a "
                              "dispatch stop it from being pipelined behind its predecessor, so they stay out of `value`; a "
                              "pipelined dispatch in a rocprofv3 kernel trace measures 1-3 us less)"
                              % (PROFILE_STRIDE, KERNEL_TIMING_REGIONS),
                    "note": "frac = design bytes (8*d*(3+k) per positive: the positive's rows are shared by its k negatives) / "
                            "kernel time / HBM peak; frac_sec8d = SURVEY 8d's 24*d per scored triple, which counts the shared "
                            "rows once per triple (capped at 1: it is not a bandwidth); hbm_frac = counter bytes "
                            "(2*FETCH_SIZE + WRITE_SIZE, gfx950 correction) / kernel time / peak; step_frac = design bytes of "
                            "scoring + Adagrad (20*d per touched row) / wall time of a step / peak.  At the 15K shape the "
                            "tables (9 MB) are cache-resident and the kernel is latency-bound"}
        if traffic.get("apply_rows_hbm_bytes_per_launch") and apply_s > 0:
            roofline["apply_rows_traffic"] = traffic["apply_rows_hbm_bytes_per_launch"]
            roofline["apply_kernel"] = traffic.get("apply_kernel")
            roofline["apply_rows_hbm_frac"] = round(traffic["apply_rows_hbm_bytes_per_launch"] / apply_s / 1e9 / HBM_PEAK_GBS, 4)
        return value, ms_per_step, roofline


# ---- HBM traffic of the step kernels, measured in this run ---------------------------------------------------------------
def _pmc_pass(counter, child_args, timeout_s):
    """one rocprofv3 --pmc pass over a short child run of this bench -> {kernel name: average counter value per launch}"""
    import collections
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        raise RuntimeError("rocprofv3 not found")
    tmp = os.environ.get("TMPDIR", "/tmp")
    out = tempfile.mkdtemp(prefix="oea_pmc_", dir=tmp)
    try:
        cmd = [rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
               sys.executable, os.path.abspath(__file__)] + child_args
        env = dict(os.environ, TMPDIR=tmp)
        p = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
        acc, cnt = collections.defaultdict(float), collections.Counter()
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == counter:
                    acc[r["Kernel_Name"]] += float(r["Counter_Value"])
                    cnt[r["Kernel_Name"]] += 1
        if not acc:
            raise RuntimeError("no %s rows (rc %d): %s" % (counter, p.returncode, p.stderr.decode(errors="replace")[-300:]))
        return {k: acc[k] / cnt[k] for k in acc}, cnt
    finally:
        shutil.rmtree(out, ignore_errors=True)


def measure_traffic(shape, d, batch, neg, eps, timeout_s=240):
    """FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (MI355X_MICROARCH.md, HBM section: both count KB; on gfx950
    FETCH_SIZE reports half of the bytes read) over a short run of the same workload: bytes = (2 FETCH + WRITE) * 1024 per
    launch.  Falls back to the committed profile of the same workload (profiles/traffic_<shape>.json)."""
    child = ["--shape", shape, "--dim", str(d), "--batch", str(batch), "--neg", str(neg), "--eps", str(eps), "--steps", "40",
             "--warmup", "5", "--repeats", "2", "--no-cpu", "--no-extra", "--no-gnn", "--no-traffic"]
    try:
        t0 = time.time()
        fetch, n = _pmc_pass("FETCH_SIZE", child, timeout_s)
        write, _ = _pmc_pass("WRITE_SIZE", child, timeout_s)

        def pick(sub):
            ks = [k for k in fetch if sub in k]
            if not ks:
                return None, 0
            k = max(ks, key=lambda kk: n[kk])
            return int((2.0 * fetch[k] + write.get(k, 0.0)) * 1024), int(n[k])
        tb, calls = pick("triple_wave")
        kname = "triple_wave"
        if tb is None:
            tb, calls = pick("triple_grouped")
            kname = "triple_grouped"
        # the optimiser kernel the timed steps run: apply_step_plan where the epoch is planned (its first epoch still runs apply_rows:
        # the kernel with the most launches of the child run is the one)
        ab, an = pick("apply_step_plan")
        aname = "apply_step_plan"
        ab2, an2 = pick("apply_rows")
        if ab is None or an2 > an:
            ab, aname = ab2, "apply_rows"
        if tb is None:
            raise RuntimeError("neither triple_wave nor triple_grouped in the counter output")
        return {"hbm_bytes_per_launch": tb, "apply_rows_hbm_bytes_per_launch": ab, "apply_kernel": aname, "launches": calls, "kernel": kname,
                "source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes over a "
                          "40-step child run of the same workload, (2*FETCH + WRITE) KB averaged over %d launches (%.0f s)"
                          % (calls, time.time() - t0)}
    except Exception as e:          # noqa: BLE001 -- any failure of the profiler must not take the bench line down
        p = os.path.join(ROOT, "profiles", "traffic_%s.json" % shape)
        try:
            tj = json.load(open(p))
            if tj.get("workload") == [shape, d, batch, neg]:
                return {"hbm_bytes_per_launch": tj.get("hbm_bytes_per_launch"),
                        "source": "committed profile %s (in-run rocprofv3 passes failed: %s)" % (os.path.basename(p), str(e)[:200])}
        except Exception:
            pass
        return {"hbm_bytes_per_launch": None, "source": "unavailable (%s)" % str(e)[:200]}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    import torch
    from openea_amd import ops

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    resolve_defaults(args, world)
    if os.environ.get("OEA_BENCH_ONE_GPU"):       # test hook: all ranks share GPU 0 (with OEA_BENCH_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    ops.lib()
    dev = torch.device("cuda", local_rank)
    group = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("OEA_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        group = dist.group.WORLD
        if rank == 0:
            cached_kgs(args.shape, "swapping")     # one rank builds the synthetic KGs, the others read the pickle
        dist.barrier()

    if args.leg == "pmc_gnn":                   # child of measure_gnn_traffic's rocprofv3 passes: launches only, no line
        pmc_child(torch, ops, dev)
        return
    wl = Workload(torch, ops, args.shape, args.dim, args.batch, args.neg, args.eps, dev, rank, world, group, args.scaling,
                  args.exchange)
    m = wl.measure(args.steps, args.warmup, args.repeats)
    extra = {}
    if world > 1:
        try:
            extra["exchange_phases"] = wl.phase_times(args.steps)                  # every rank takes part
        except Exception as e:      # noqa: BLE001 -- a side measurement must not take the line down
            extra["exchange_phases"] = {"error": repr(e)[:300]}
    if not args.no_extra:                   # eval / neighbour legs: row-sharded over the ranks (every rank takes part)
        extra.update(extra_legs(torch, ops, wl.ent, wl.kgs, args.dim, wl.k1))
    multi = None
    if world > 1 and not os.environ.get("OEA_BENCH_NO_SIDE"):                     # (OEA_BENCH_NO_SIDE=1: the headline exchange only)
        multi = multi_gpu_legs(torch, ops, dev, args, rank, world, group, wl)      # every rank takes part
    if rank != 0:
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        return
    traffic = None
    if world == 1 and not args.no_traffic:
        traffic = measure_traffic(args.shape, args.dim, args.batch, args.neg, args.eps)
    value, ms_per_step, roofline = wl.summarize(m, args.steps, traffic)
    extra.update({"neighbour_refresh_first_call_s": round(wl.nbr_first_s, 3), "epoch_loss_sum": m["loss"],
                  "triple_steps_per_epoch": wl.steps_per_epoch, "neighbours_k": [wl.k1, wl.k2],
                  "regions_ms": [round(float(x) * 1e3, 4) for x in m["times"]],
                  "ms_per_step_min": round(float(m["times"].min()) / args.steps * 1e3, 4),
                  "ms_per_step_max": round(float(m["times"].max()) / args.steps * 1e3, 4)})
    if world > 1:
        import torch.distributed as dist
        xb = wl.trainer.exchange_bytes_per_step()
        ep_b = wl.trainer.epoch_exchange_bytes()
        if wl.trainer.local_epochs:
            xb = int(ep_b / max(wl.steps_per_epoch, 1))
        extra["exchange_mode"] = wl.trainer.exchange
        extra["exchange_bytes_per_step_per_rank"] = xb
        extra["exchange_bytes_per_epoch_per_rank"] = ep_b if wl.trainer.local_epochs else xb * wl.steps_per_epoch
        extra["exchange"] = wl.trainer.exchange_description()
        # what the exchange alone would cost on the ring: per-link xGMI 153 GB/s, 7 links, realistic RCCL bus bandwidth ~350 GB/s
        extra["exchange_predicted_us_per_step_at_350GBs"] = round(xb / 350e9 * 1e6, 1)
        extra.update(multi or {})
        extra["collective_backend"] = "RCCL (torch.distributed nccl)" if backend == "nccl" else backend
        extra["collective_world_size"] = dist.get_world_size()
        one = (multi or {}).get("single_gpu_same_config")
        if one:
            extra["speedup_vs_single_gpu_same_config"] = round(value / one["value"], 4)
    cpu = None
    if not args.no_cpu and world == 1:
        cpu = cpu_baseline(wl.kgs, args.dim, args, wl.k1, wl.k2)
    big = "100K" in args.shape
    eval_dim = args.dim
    if not args.no_extra and world == 1 and big and args.dim == 100:
        del wl
        torch.cuda.empty_cache()
        try:
            extra["shape_15k"] = side_shape(torch, ops, dev, args, "EN-FR-15K-V1", 75, 5000, 0.9)
        except Exception as e:      # noqa: BLE001 -- a failing side leg must not take the headline line down
            extra["shape_15k"] = {"error": repr(e)[:300]}
    if not args.no_gnn and world == 1:
        torch.cuda.empty_cache()
        gtraffic = None if args.no_traffic else measure_gnn_traffic()
        try:
            extra["gnn"] = gnn_legs(torch, ops, dev, gtraffic)
        except Exception as e:      # noqa: BLE001 -- a failing side leg must not take the headline line down
            extra["gnn"] = {"error": repr(e)[:500]}
        # counter traffic of the CSLS evaluation (70,000^2 x 100) and of the neighbour search (100,000^2, k = 2,000), priced
        # with the wall times of the headline shape's legs
        if gtraffic and big and "error" not in gtraffic:
            for leg_name, key, rate_key, units in (("csls_eval_70k", "csls_eval_hbm", "eval_pairs_per_s_inner_csls10", 70000),
                                                   ("knn_100k", "neighbour_search_hbm", "neighbour_rows_per_s", 100000)):
                t = gtraffic.get(leg_name)
                if t and extra.get(rate_key):
                    ms = units / extra[rate_key] * 1e3
                    extra[key] = {"hbm_bytes_per_call": t["hbm_bytes_per_call"], "ms_per_call": round(ms, 3),
                                  "hbm_frac": round(t["hbm_bytes_per_call"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "kernels": t["kernels"], "traffic_source": gtraffic.get("source")}
        elif gtraffic and big:
            extra["csls_eval_hbm"] = extra["neighbour_search_hbm"] = gtraffic

    roofline_eval = eval_roofline(extra, eval_dim)
    if roofline_eval:
        roofline["eval"] = roofline_eval
    per_gpu = args.batch if args.scaling == "weak" else args.batch / world
    gb = args.batch * world if args.scaling == "weak" else args.batch
    out = {
        "metric": "training triples/sec (positives consumed; truncated negative sampling k=%d + limited loss + Adagrad)" % args.neg,
        "value": round(value, 1), "unit": "triples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "repeats": args.repeats, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BootEA/AlignE translational step, %s shape (synthetic), dim=%d, GLOBAL batch=%d positives (%s "
                               "scaling: %g per GPU x %d GPU%s%s), k=%d, truncated eps=%.2f; the same workload at every N%s"
                               % (args.shape, args.dim, gb, args.scaling, per_gpu, world, "s" if world > 1 else "",
                                  "; the shipped configuration is batch=%d: N>1 weak scaling trains with a LARGER global batch "
                                  "than any BASELINE configuration" % args.batch if (world > 1 and args.scaling == "weak") else "",
                                  args.neg, args.eps,
                                  " (bootea_args_100K.json; BASELINE config 2 = EN-FR-15K-V1, dim 75, batch 5,000 is extra.shape_15k)"
                                  if (big and world == 1) else ""),
                   "global_batch": gb, "entities": extra_entities(args.shape),
                   "parallelism": ("dp%d, exchange = %s: %s" % (world, extra.get("exchange_mode"), extra.get("exchange")))
                   if world > 1 else "single",
                   "timing": "median of %d regions of %d steps, each bracketed by barrier + synchronize" % (args.repeats, args.steps),
                   "parity_note": "TF1 op semantics / optimiser arithmetic are restated, not executed (no TensorFlow "
                                  "here): SURVEY H1/H3/H4, DESIGN.md section 5"},
        "roofline": roofline, "roofline_eval": roofline_eval, "cpu_baseline": cpu, "extra": extra,
    }
    emit(out)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


# bf16 prefilter: three bf16 products per exact product -> the dense bf16 MFMA peak (2.5 PFLOP/s) / 3 per algorithmic flop;
# a bf16 leg is NEVER priced against the fp32 peak (a frac above 1 is a wrong peak)
BF16_SPLIT_PEAK_TF = 2500.0 / 3.0
FP32_MFMA_PEAK_TF = 157.3


def eval_roofline(x, d):
    """roofline block of the alignment evaluation of the headline shape (second half of BASELINE.json's metric)"""
    if not x.get("eval_pairs_per_s_inner"):
        return None
    n1 = x["eval_pairs"]
    fl = 2.0 * n1 * n1 * d
    tf = fl * x["eval_pairs_per_s_inner"] / n1 / 1e12
    tf32 = fl * x["eval_pairs_per_s_inner_fp32_sweep"] / n1 / 1e12
    bf = x.get("eval_bf16_prefilter")
    peak = BF16_SPLIT_PEAK_TF if bf else FP32_MFMA_PEAK_TF
    return {"kernel": ("rank_bf16_kernel + prologue / fix-up / finish (oea_rank_eval_metrics_bf16: bf16 hi / lo split on "
                       "v_mfma_f32_32x32x16_bf16, three products per exact product, exact decisions for the recorded "
                       "pairs -- identical ranks)" if bf else "rank_inner_kernel + prologue / tail (oea_rank_eval_metrics)")
                      + ": greedy_alignment of the %d test pairs, inner product, whole call incl. the copy of the metrics "
                        "to the host" % n1,
            "bound": "mfma", "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s (algorithmic: 2*N1*N2*d)",
            "frac": round(tf / peak, 4), "flops_per_call": fl, "pairs_per_s": x["eval_pairs_per_s_inner"],
            "fp32_sweep": {"pairs_per_s": x["eval_pairs_per_s_inner_fp32_sweep"], "achieved": round(tf32, 2), "peak": FP32_MFMA_PEAK_TF,
                           "frac": round(tf32 / FP32_MFMA_PEAK_TF, 4), "kernel": "rank_inner_kernel (v_mfma_f32_32x32x2_f32), OEA_EVAL_BF16=0"},
            "speedup_vs_fp32_sweep": round(x["eval_pairs_per_s_inner"] / x["eval_pairs_per_s_inner_fp32_sweep"], 3),
            "csls10_pairs_per_s": x.get("eval_pairs_per_s_inner_csls10"),
            "csls10_fp32_sweep_pairs_per_s": x.get("eval_pairs_per_s_inner_csls10_fp32_sweep"),
            "reference_signature_pairs_per_s": x.get("eval_pairs_per_s_inner_reference_signature"),
            "note": "second half of BASELINE.json's metric (alignment-eval pairs/s), device-resident tables.  peak = the dense bf16 "
                    "MFMA peak (2.5 PFLOP/s) / 3 products per exact product when the prefilter runs, else the fp32 MFMA peak; "
                    "reference_signature_pairs_per_s = modules.finding.alignment.greedy_alignment(numpy, numpy, ...) -> pairs, i.e. "
                    "host arrays in (PCIe copy included) and the reference's return value out"}


# ---- the one stdout line ------------------------------------------------------------------------------------------------
COMPACT_LIMIT = 4096          # bytes; the driver keeps an 8 KB tail of stdout (BENCH_r04: a 21.8 KB line was not parsed)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _r(v, n=4):
    return round(v, n) if isinstance(v, float) else v


def compact_line(out, detail_path=None):
    """the contract's keys + roofline + roofline_eval + cpu_baseline + a few dozen scalars of `extra`; everything else (per-kernel
    counter dictionaries, region times, notes, provenance strings) lives in bench_detail.json"""
    rl = out.get("roofline") or {}
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "repeats", "ms_per_step", "higher_is_better",
                             "scaling", "vs_baseline", "dtype", "data") if k in out}
    c["metric"] = "training triples/sec (k=10 truncated negatives + limited loss + Adagrad)"
    cfg = out.get("config") or {}
    wl = cfg.get("workload", "")
    c["config"] = {"workload": wl[:200], "global_batch": cfg.get("global_batch"), "entities": cfg.get("entities"),
                   "parallelism": (cfg.get("parallelism") or "")[:60]}
    r = _pick(rl, ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_frac", "avg_kernel_us", "design_bytes_per_launch",
                   "step_frac", "apply_rows_avg_us", "apply_rows_hbm_frac", "launches_timed"))
    r["kernel"] = "triple_grouped" if os.environ.get("OEA_STEP_WAVE", "1") == "0" else "triple_wave"
    r.setdefault("traffic", None)
    c["roofline"] = r
    re_ = out.get("roofline_eval")
    if re_:
        e = _pick(re_, ("bound", "achieved", "peak", "frac", "pairs_per_s", "speedup_vs_fp32_sweep", "csls10_pairs_per_s",
                        "reference_signature_pairs_per_s"))
        e["unit"] = "TFLOP/s"
        e["kernel"] = "rank_bf16_kernel" if "bf16" in re_.get("kernel", "")[:20] else "rank_inner_kernel"
        e["fp32_sweep_frac"] = (re_.get("fp32_sweep") or {}).get("frac")
        c["roofline_eval"] = e
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = dict(_pick(cb, ("value", "unit", "cores", "kind", "host_cores")), sample=(cb.get("sample") or "")[:120])
    else:
        c["cpu_baseline"] = None
    x = out.get("extra") or {}
    e = _pick(x, ("eval_pairs", "eval_pairs_per_s_inner", "eval_pairs_per_s_inner_csls10", "eval_pairs_per_s_inner_reference_signature",
                  "eval_pairs_per_s_manhattan", "neighbour_rows_per_s", "neighbour_bf16_frac", "ms_per_step_min", "ms_per_step_max",
                  "exchange_mode", "exchange_bytes_per_step_per_rank", "collective_backend", "collective_world_size",
                  "speedup_vs_single_gpu_same_config"))
    ph = x.get("exchange_phases")
    if isinstance(ph, dict):
        e["exchange_phases"] = {k: v for k, v in ph.items() if k.endswith("_us") or k in ("steps_timed", "error")}
    one = x.get("single_gpu_same_config")
    if isinstance(one, dict):
        e["single_gpu_same_config"] = _pick(one, ("value", "ms_per_step"))
    for k in ("other_exchange", "local_sgd", "other_scaling"):
        if isinstance(x.get(k), dict):
            e[k] = _pick(x[k], ("value", "ms_per_step", "exchange_mode", "scaling", "global_batch"))
    s15 = x.get("shape_15k")
    if isinstance(s15, dict):
        b = _pick(s15, ("value", "ms_per_step", "eval_pairs_per_s_inner", "eval_pairs_per_s_inner_csls10", "neighbour_rows_per_s", "error"))
        r15 = s15.get("roofline") or {}
        b.update({"frac": r15.get("frac"), "hbm_frac": r15.get("hbm_frac"), "avg_kernel_us": r15.get("avg_kernel_us"),
                  "step_frac": r15.get("step_frac"), "workload": "EN-FR-15K-V1 dim 75 batch 5000 (BASELINE config 2)"})
        e["shape_15k"] = b
    g = x.get("gnn")
    if isinstance(g, dict):
        if "error" in g:
            e["gnn"] = {"error": g["error"][:200]}
        else:
            gg = {}
            gc = g.get("gcn_align_se_epoch_DW15K") or {}
            gg["gcn_align_DW15K"] = {"ms_per_epoch": gc.get("ms_per_epoch"), "spmm_frac": (gc.get("roofline") or {}).get("frac"),
                                     "spmm_hbm_frac": (gc.get("roofline") or {}).get("hbm_frac")}
            al = g.get("alinet_EN-DE-100K") or {}
            gg["alinet_EN-DE-100K"] = {"ms_per_epoch": al.get("ms_per_epoch"), "ms_per_epoch_row": al.get("ms_per_epoch_grouping_row"),
                                       "ms_per_epoch_reorder": al.get("ms_per_epoch_grouping_reorder"),
                                       "attn_fwd_ms": al.get("attention_fwd_ms"), "attn_bwd_ms": al.get("attention_bwd_ms"),
                                       "attn_frac": (al.get("roofline") or {}).get("frac"),
                                       "attn_row_fwd_ms": (al.get("grouping_row") or {}).get("attention_fwd_ms"),
                                       "attn_row_bwd_ms": (al.get("grouping_row") or {}).get("attention_bwd_ms"),
                                       "attn_row_frac": ((al.get("grouping_row") or {}).get("roofline") or {}).get("frac"),
                                       "spmm_1hop_frac": (al.get("roofline_1hop_aggregate") or {}).get("frac"),
                                       "spmm_1hop_hbm_frac": (al.get("roofline_1hop_aggregate") or {}).get("hbm_frac")}
            for k_ in ("attn_reorder_fwd_ms", "attn_reorder_bwd_ms"):
                if al.get(k_) is not None:
                    gg["alinet_EN-DE-100K"][k_] = al[k_]
            rd = g.get("rdgcn_eval_70000x300") or {}
            gg["rdgcn_eval_70000x300"] = {"manhattan_ms": rd.get("manhattan_ms"), "manhattan_csls10_ms": rd.get("manhattan_csls10_ms"),
                                          "inner_ms": rd.get("inner_ms"), "manhattan_frac": (rd.get("roofline") or {}).get("frac"),
                                          "inner_frac": (rd.get("roofline_inner") or {}).get("frac")}
            ae = g.get("alinet_eval_70000x1200")
            if isinstance(ae, dict):
                gg["alinet_eval_70000x1200"] = _pick(ae, ("inner_ms", "inner_csls10_ms", "frac", "csls_frac", "peak", "bf16_prefilter",
                                                          "records_per_row", "fallback", "fp32_sweep_ms", "error"))
            e["gnn"] = gg
    c["extra"] = e
    if detail_path:
        c["detail"] = detail_path

    def clean(o):
        if isinstance(o, dict):
            return {k: clean(v) for k, v in o.items()}
        if isinstance(o, float):
            return round(o, 4) if abs(o) < 1e6 else round(o, 1)
        return o
    c = clean(c)
    line = json.dumps(c, separators=(",", ":"))
    # never let the line grow past the limit: drop the optional blocks, least important first
    for k in ("gnn", "shape_15k", "other_scaling", "local_sgd", "other_exchange", "exchange_phases"):
        if len(line) <= COMPACT_LIMIT:
            break
        c["extra"].pop(k, None)
        c["extra"]["dropped_for_length"] = c["extra"].get("dropped_for_length", []) + [k]
        line = json.dumps(c, separators=(",", ":"))
    return line


def emit(out):
    """everything measured -> bench_detail.json (repo root, and gpurun_out/ when it exists: that directory travels back from
    the GPU box); ONE compact JSON line (<= 4 KB) as the LAST line of stdout"""
    rel = None
    txt = json.dumps(out, indent=1)
    targets = [os.environ.get("OEA_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")) and not os.environ.get("OEA_BENCH_DETAIL"):
        targets.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    for t in targets:
        try:
            with open(t, "w") as f:
                f.write(txt)
            rel = rel or os.path.relpath(t, ROOT)
        except OSError:
            pass
    sys.stdout.flush()
    print(compact_line(out, rel), flush=True)


def multi_gpu_legs(torch, ops, dev, args, rank, world, group, main_wl):
    """N > 1 only, every rank.  (1) the SAME configuration on one GPU, timed in this run (every rank runs it on its own GPU,
    no collective; rank 0's number is reported) -- the reference point of the speed-up; (2) the side legs: the other exchange
    mode on the same shape and the other scaling mode (one workload at every N: no other shape here)."""
    import torch.distributed as dist
    out = {}

    def run(shape, dim, batch, eps, exchange, scaling, steps, warmup, repeats, single=False):
        if rank == 0:
            cached_kgs(shape, "swapping")
        dist.barrier()
        if single:
            wl = Workload(torch, ops, shape, dim, batch, args.neg, eps, dev)
        else:
            wl = Workload(torch, ops, shape, dim, batch, args.neg, eps, dev, rank, world, group, scaling, exchange)
        m = wl.measure(steps, warmup, repeats)
        value, ms, _ = wl.summarize(m, steps)
        res = {"shape": shape, "value": round(value, 1), "unit": "triples/s", "ms_per_step": round(ms, 4), "steps": steps,
               "repeats": repeats, "global_batch": wl.global_batch, "triple_steps_per_epoch": wl.steps_per_epoch}
        if not single:
            ep_b = wl.trainer.epoch_exchange_bytes()
            xb = int(ep_b / max(wl.steps_per_epoch, 1)) if wl.trainer.local_epochs else wl.trainer.exchange_bytes_per_step()
            res.update({"exchange_mode": wl.trainer.exchange, "scaling": scaling, "exchange_bytes_per_step_per_rank": xb})
            if wl.trainer.local_epochs:
                res["parity"] = ("local SGD with one exchange per epoch: NOT the single-GPU job (2.9e-2 relative L2 drift of the "
                                 "entity table after 6 epochs at 2 ranks, tests/test_dist_gpu.py); reported as a side leg only")
        dist.barrier()
        del wl
        torch.cuda.empty_cache()
        return res
    big = "100K" in args.shape
    steps = min(args.steps, 58) if big else args.steps
    wu = min(args.warmup, 5)
    one = run(args.shape, args.dim, args.batch, args.eps, None, None, steps, wu, 5, single=True)
    one["note"] = "the same shape / dim / batch / k on ONE GPU (rank 0's; every rank ran it on its own GPU at the same time)"
    out["single_gpu_same_config"] = one
    other = "step" if main_wl.trainer.exchange != "step" else "halo"
    out["other_exchange"] = run(args.shape, args.dim, args.batch, args.eps, other, args.scaling, steps, wu, 5)
    if main_wl.trainer.exchange != "epoch":
        out["local_sgd"] = run(args.shape, args.dim, args.batch, args.eps, "epoch", args.scaling, steps, wu, 5)
    if not args.no_extra and not os.environ.get("OEA_BENCH_ONE_GPU"):       # (the one-GPU wiring test stops here: host-staged collectives)
        out["other_scaling"] = run(args.shape, args.dim, args.batch, args.eps, args.exchange, "weak" if args.scaling == "strong" else "strong",
                                   steps, wu, 3)
    return out


def extra_entities(shape):
    from openea_amd.modules.load.synth import SHAPES
    return 2 * SHAPES[shape][0]


def side_shape(torch, ops, dev, args, shape, dim, batch, eps):
    """the other shape of BASELINE.json's metric, measured like the headline (N = 1): step throughput + kernel events + counter
    traffic, alignment evaluation over its test pairs, neighbour refresh"""
    wl = Workload(torch, ops, shape, dim, batch, args.neg, eps, dev)
    steps = args.steps
    m = wl.measure(steps, min(args.warmup, 10), max(5, min(args.repeats, 20)))
    out = {}
    out.update(extra_legs(torch, ops, wl.ent, wl.kgs, dim, wl.k1))
    traffic = None if args.no_traffic else measure_traffic(shape, dim, batch, args.neg, eps)
    value, ms_per_step, roofline = wl.summarize(m, steps, traffic)
    ev = eval_roofline(out, dim)
    if ev:
        roofline["eval"] = ev
    out.update({"workload": "%s shape (synthetic), dim=%d, batch=%d, k=%d, truncated eps=%.2f (k_nbr=%d)" % (shape, dim, batch, args.neg, eps, wl.k1),
                "value": round(value, 1), "unit": "triples/s", "ms_per_step": round(ms_per_step, 4), "steps": steps,
                "repeats": len(m["times"]), "roofline": roofline, "neighbour_refresh_first_call_s": round(wl.nbr_first_s, 3),
                "triple_steps_per_epoch": wl.steps_per_epoch})
    return out


def extra_legs(torch, ops, ent, kgs, d, k1):
    """alignment-eval pairs/s and neighbour-search rows/s (second half of BASELINE.json's metric).  Under torch.distributed
    the query rows are sharded over the ranks (models/dist.py) -- every rank calls this."""
    from openea_amd.modules.finding.alignment import greedy_alignment_device
    from openea_amd.models import dist as mdist
    world = mdist.world()[1]

    def sync():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
    e1 = ent.lookup(kgs.test_entities1)
    e2 = ent.lookup(kgs.test_entities2)
    out = {}
    saved_env = os.environ.get("OEA_EVAL_BF16")
    # the product path (certified bf16 prefilter from 3e8 pairs (~17,000^2) on: identical ranks) and, beside it, the exact fp32 sweep
    for name, csls, bf16 in (("eval_pairs_per_s_inner", 0, "1"), ("eval_pairs_per_s_inner_csls10", 10, "1"),
                             ("eval_pairs_per_s_inner_fp32_sweep", 0, "0"), ("eval_pairs_per_s_inner_csls10_fp32_sweep", 10, "0")):
        os.environ["OEA_EVAL_BF16"] = bf16
        greedy_alignment_device(e1, e2, d, [1, 5, 10, 50], "inner", False, csls)       # warm
        sync()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            greedy_alignment_device(e1, e2, d, [1, 5, 10, 50], "inner", False, csls)
        sync()
        out[name] = round(e1.shape[0] * reps / (time.perf_counter() - t0), 1)
    if saved_env is None:
        os.environ.pop("OEA_EVAL_BF16", None)
    else:
        os.environ["OEA_EVAL_BF16"] = saved_env
    out["eval_bf16_prefilter"] = bool(ops.eval_bf16_enabled(e1.shape[0], e2.shape[0]))
    if world == 1:
        # the REFERENCE-SIGNATURE call beside the device-resident one: numpy arrays in (host -> device copy
        # inside the call), (alignment_rest, hits1, mr, mrr) out -- alignment.py:13-84's arguments and return value
        from openea_amd.modules.finding.alignment import greedy_alignment
        h1 = e1[:, :d].cpu().numpy()
        h2 = e2[:, :d].cpu().numpy()
        _quiet(lambda: greedy_alignment(h1, h2, [1, 5, 10, 50], 1, "inner", False, 0, True))
        sync()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            rest = _quiet(lambda: greedy_alignment(h1, h2, [1, 5, 10, 50], 1, "inner", False, 0, True))[0]
            assert len(rest) == h1.shape[0]
        out["eval_pairs_per_s_inner_reference_signature"] = round(h1.shape[0] * reps / (time.perf_counter() - t0), 1)
    greedy_alignment_device(e1, e2, d, [1, 5, 10, 50], "manhattan", False, 0)
    sync()
    t0 = time.perf_counter()
    greedy_alignment_device(e1, e2, d, [1, 5, 10, 50], "manhattan", False, 0)
    sync()
    out["eval_pairs_per_s_manhattan"] = round(e1.shape[0] / (time.perf_counter() - t0), 1)
    from openea_amd.models.trainer import refresh_neighbours
    refresh_neighbours(ent, kgs.kg1.entities_list, k1)                                 # warm (first-use allocations)
    sync()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        refresh_neighbours(ent, kgs.kg1.entities_list, k1)
    sync()
    out["neighbour_rows_per_s"] = round(len(kgs.kg1.entities_list) * reps / (time.perf_counter() - t0), 1)
    out["eval_pairs"] = int(e1.shape[0])
    n1, dd, nn = e1.shape[0], d, len(kgs.kg1.entities_list)
    out["eval_inner_fp32_sweep_mfma_frac"] = round(2.0 * n1 * n1 * dd * out["eval_pairs_per_s_inner_fp32_sweep"] / n1 / (FP32_MFMA_PEAK_TF * 1e12), 4)
    ms_n = nn / out["neighbour_rows_per_s"] * 1e3
    out["neighbour_ms_per_refresh"] = round(ms_n, 3)
    stream = nn >= int(os.environ.get("OEA_TOPK_SYM_MIN", "12288")) and os.environ.get("OEA_TOPK_BF16", "1")[:1] != "0"
    if stream:      # symmetric bf16 sweep: the upper triangle, 2 flop per pair and dimension, three bf16 products per exact one
        out["neighbour_bf16_frac"] = round(float(nn) * nn * dd / (ms_n * 1e-3) / 1e12 / BF16_SPLIT_PEAK_TF, 4)
    else:
        out["neighbour_fp32_frac"] = round(2.0 * nn * nn * dd / (ms_n * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4)
    out["mfma_frac_note"] = ("neighbour_bf16_frac = N^2*d flop (the upper triangle of the symmetric sweep) / wall time of the whole "
                             "refresh / (2.5 PFLOP/s / 3): the sweep runs on the bf16 hi / lo split, priced against the bf16 peak; "
                             "eval_inner_fp32_sweep_mfma_frac is the exact fp32 sweep (OEA_EVAL_BF16=0) against 157.3 TFLOP/s")
    return out


# ---- GNN legs: BASELINE.json configs 3-5 -----------------------------------------------------------------------------
def _timed_events(torch, fn, reps):
    """device time of `reps` calls of fn on torch's current stream (the stream the library launches on), ms per call"""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _wall(torch, fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def _mfma_block(ops, n1, n2, d, ms, **kw):
    """inner-product evaluation: 2*N1*N2*d algorithmic flop / wall time, against the peak of the pipe the product path runs on
    (certified bf16 prefilter from 3e8 pairs on: 2.5 PFLOP/s / 3 products; fp32 MFMA below)"""
    bf = bool(ops.eval_bf16_enabled(n1, n2))
    peak = BF16_SPLIT_PEAK_TF if bf else FP32_MFMA_PEAK_TF
    tf = 2.0 * n1 * n2 * d / (ms * 1e-3) / 1e12
    return dict({"kernel": "rank_bf16_kernel (bf16 hi / lo split, 3 products per exact product)" if bf else "rank_inner_kernel (fp32 MFMA)",
                 "bound": "mfma", "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s (algorithmic: 2*N1*N2*d)",
                 "frac": round(tf / peak, 4), "ms": round(ms, 3)}, **kw)


def _hbm_block(kernel, alg_bytes, ms, **kw):
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return dict({"kernel": kernel, "bound": "hbm", "algorithmic_bytes": int(alg_bytes), "ms": round(ms, 4),
                 "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}, **kw)


PMC_LEGS = ("gcn_spmm_15k", "alinet_1hop_100k", "attn_runs_fwd", "attn_runs_bwd", "attn_row_fwd", "attn_row_bwd", "csls_eval_70k",
            "knn_100k")
PMC_MARK0 = 1000        # marker launch of leg i: fill_kernel with (PMC_MARK0 + i) workgroups of 256


def _quiet(fn):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn()


def _build_gcn(torch, ops, dev, rng):
    from openea_amd.approaches.gcn_align import GCN_Align
    from openea_amd.run.default_args import get_args
    kgs = cached_kgs("D-W-15K-V2", "mapping")

    def mk():
        m = GCN_Align()
        m.set_args(get_args("GCN_Align", output="/tmp/oea_out/", training_data="synthetic/dw15k/", dataset_division="f/"))
        m.set_kgs(kgs)
        m.init()
        return m
    m = _quiet(mk)
    se = m.model_se
    d = m.args.se_dim
    x = ops.gather_rows(se.W, d, se.row_ids, normalize=True)
    return m, kgs, se, d, x


def _build_alinet(torch, ops, dev, grouping=None, epochs=0):
    """AliNet at the EN-DE-100K-V1 shape -> (model, kgs, init seconds, ms per epoch or None)"""
    from openea_amd.approaches import AliNet
    from openea_amd.run.default_args import get_args
    kgs = cached_kgs("EN-DE-100K-V1", "mapping")
    kw = dict(scale="100K", output="/tmp/oea_out/", training_data="synthetic/EN-DE-100K-V1/", dataset_division="f/", max_epoch=1,
              start_valid=10 ** 6, eval_freq=10 ** 6)
    if grouping:
        kw["attn_grouping"] = grouping
    res = {}

    def mk():
        a = AliNet()
        a.set_args(get_args("AliNet", **kw))
        a.set_kgs(kgs)
        t0 = time.perf_counter()
        a.init()
        torch.cuda.synchronize()
        res["init_s"] = time.perf_counter() - t0
        if epochs:
            a.run()
            torch.cuda.synchronize()
            a.args.max_epoch = epochs
            t0 = time.perf_counter()
            a.run()
            torch.cuda.synchronize()
            res["ms_epoch"] = (time.perf_counter() - t0) / epochs * 1e3
        return a
    a = _quiet(mk)
    return a, kgs, res["init_s"], res.get("ms_epoch")


def _attn_closures(torch, ops, dev, g2, n, d):
    z = torch.randn(g2.nnz, device=dev)
    v = torch.randn(n, ops.pad4(d), device=dev)
    dout = torch.randn(n, ops.pad4(d), device=dev)
    res = {}

    def fwd():
        res["o"], res["a"] = ops.sparse_attn_fwd(g2.attn, z, v, d, 0.2, n)

    def bwd():
        ops.sparse_attn_bwd(g2.attn, z, v, res["a"], dout, d, 0.2)
    return fwd, bwd


def _row_grouped(g2, dev):
    """the same ordered edge list under the per-row softmax grouping (SURVEY H3's other reading)"""
    from openea_amd.models.graph_ops import EdgeGraph
    return EdgeGraph(g2._rows_o, g2._cols_o, g2.e_vals.cpu().numpy(), g2.shape, dev, grouping="row")


def _eval_tables(torch, ops, dev, n_e, d_e, rng):
    e1 = rng.standard_normal((n_e, d_e)).astype(np.float32)
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    t1 = ops.to_table(e1, dev=dev)
    t2 = ops.to_table((e1 + 0.4 * rng.standard_normal((n_e, d_e)).astype(np.float32) / np.sqrt(d_e)).astype(np.float32), dev=dev)
    return t1, t2


def pmc_child(torch, ops, dev):
    """child of a rocprofv3 --pmc pass (--leg pmc_gnn): a few launches of every kernel whose HBM traffic the parent prices,
    each group behind a MARKER launch (fill_kernel with PMC_MARK0 + i workgroups) so that the parent can tell the aggregate of
    the 1-hop graph from the attention's (same kernel, same grid) in the counter file"""
    from openea_amd.modules.finding.alignment import greedy_alignment_device
    from openea_amd.models.trainer import EmbeddingTable, refresh_neighbours
    from openea_amd.modules.base.initializers import truncated_normal_host
    rng = np.random.RandomState(0)
    mark_buf = torch.empty(256 * (PMC_MARK0 + len(PMC_LEGS) + 1), dtype=torch.float32, device=dev)

    def leg(name, fn, reps):
        i = PMC_LEGS.index(name)
        torch.cuda.synchronize()
        ops.check(ops.lib().oea_fill_f32(mark_buf.data_ptr(), 256 * (PMC_MARK0 + i), 0.0, torch.cuda.current_stream().cuda_stream))
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        # END marker: what follows (building the next operator, its warm-up calls) belongs to no leg
        ops.check(ops.lib().oea_fill_f32(mark_buf.data_ptr(), 256 * (PMC_MARK0 + len(PMC_LEGS)), 0.0, torch.cuda.current_stream().cuda_stream))
    m, kgs, se, d, x = _build_gcn(torch, ops, dev, rng)
    se.adj.mm(x, d, act=1)
    leg("gcn_spmm_15k", lambda: se.adj.mm(x, d, act=1), 3)
    del m, se, x
    a, kgs, _, _ = _build_alinet(torch, ops, dev)
    g2, g1 = a.adj[1], a.adj[0]
    n, d = kgs.entities_num, a.args.layer_dims[1]
    x1 = torch.randn(n, ops.pad4(d), device=dev)
    g1.fwd.apply(x1, d)
    leg("alinet_1hop_100k", lambda: g1.fwd.apply(x1, d), 3)
    for tag, g in (("runs", g2), ("row", _row_grouped(g2, dev))):
        fwd, bwd = _attn_closures(torch, ops, dev, g, n, d)
        fwd(), bwd()
        leg("attn_%s_fwd" % tag, fwd, 2)
        leg("attn_%s_bwd" % tag, bwd, 2)
    del a, g1, g2, x1
    torch.cuda.empty_cache()
    t1, t2 = _eval_tables(torch, ops, dev, 70000, 100, rng)
    greedy_alignment_device(t1, t2, 100, [1, 5, 10, 50], "inner", False, 10)
    leg("csls_eval_70k", lambda: greedy_alignment_device(t1, t2, 100, [1, 5, 10, 50], "inner", False, 10), 2)
    del t1, t2
    torch.cuda.empty_cache()
    kgs = cached_kgs("EN-FR-100K-V1", "swapping")
    ent = EmbeddingTable(truncated_normal_host(np.random.RandomState(1), (kgs.entities_num, 100), 0.1), True, "ent_embeds", dev)
    refresh_neighbours(ent, kgs.kg1.entities_list, 2000)
    leg("knn_100k", lambda: refresh_neighbours(ent, kgs.kg1.entities_list, 2000), 1)


def _pmc_pass_legs(counter, timeout_s):
    """one rocprofv3 --pmc pass over `bench.py --leg pmc_gnn` -> {leg: {kernel name: (sum of the counter, launches)}}"""
    import collections
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        raise RuntimeError("rocprofv3 not found")
    tmp = os.environ.get("TMPDIR", "/tmp")
    out = tempfile.mkdtemp(prefix="oea_pmc_", dir=tmp)
    try:
        cmd = [rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
               sys.executable, os.path.abspath(__file__), "--leg", "pmc_gnn"]
        p = subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
        rows = []
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == counter:
                    rows.append((int(r.get("Dispatch_Id") or 0), r["Kernel_Name"], int(r.get("Grid_Size") or 0), float(r["Counter_Value"])))
        if not rows:
            raise RuntimeError("no %s rows (rc %d): %s" % (counter, p.returncode, p.stderr.decode(errors="replace")[-300:]))
        rows.sort()
        legs = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        cur = None
        for _, name, grid, val in rows:
            if "fill_kernel" in name and grid % 256 == 0 and PMC_MARK0 <= grid // 256 <= PMC_MARK0 + len(PMC_LEGS):
                i = grid // 256 - PMC_MARK0
                cur = PMC_LEGS[i] if i < len(PMC_LEGS) else None          # the end marker closes the leg
                continue
            if cur is not None:
                legs[cur][name][0] += val
                legs[cur][name][1] += 1
        if not legs:
            raise RuntimeError("no marker launches in the counter file (columns: Grid_Size / Dispatch_Id missing?)")
        return legs
    finally:
        shutil.rmtree(out, ignore_errors=True)


PMC_REPS = {"gcn_spmm_15k": 3, "alinet_1hop_100k": 3, "attn_runs_fwd": 2, "attn_runs_bwd": 2, "attn_row_fwd": 2, "attn_row_bwd": 2,
            "csls_eval_70k": 2, "knn_100k": 1}


def measure_gnn_traffic(timeout_s=300):
    """FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 correction: bytes = (2 FETCH + WRITE) KB) of the GNN / CSLS / neighbour
    kernels, per CALL of the operator -> {leg: {"hbm_bytes_per_call", "kernels": {name: bytes per launch}}} or {"error"}"""
    try:
        t0 = time.time()
        fetch = _pmc_pass_legs("FETCH_SIZE", timeout_s)
        write = _pmc_pass_legs("WRITE_SIZE", timeout_s)
        out = {}
        for leg_name in PMC_LEGS:
            if leg_name not in fetch:
                continue
            ks = {}
            total = 0.0
            for name, (fv, n) in fetch[leg_name].items():
                wv = write.get(leg_name, {}).get(name, [0.0, 0])[0]
                b = (2.0 * fv + wv) * 1024
                total += b
                ks[name[:90]] = {"hbm_bytes_per_launch": int(b / max(n, 1)), "launches_per_call": round(n / PMC_REPS[leg_name], 2)}
            out[leg_name] = {"hbm_bytes_per_call": int(total / PMC_REPS[leg_name]), "kernels": ks}
        out["source"] = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes over "
                         "`bench.py --leg pmc_gnn` (marker launches separate the operators), (2*FETCH + WRITE) KB, %.0f s" % (time.time() - t0))
        return out
    except Exception as e:          # noqa: BLE001 -- the profiler must not take the bench line down
        return {"error": "unavailable (%s)" % str(e)[:300]}


def _with_hbm(block, traffic, leg_name, ms, kernel_sub=None):
    """attach the counter traffic of one operator call to its roofline block: hbm_frac = counter bytes / time / HBM peak.
    FETCH_SIZE / WRITE_SIZE count what crosses the L2 <-> fabric boundary, i.e. L2 misses INCLUDING those the 256 MB Infinity
    Cache (MALL) serves: hbm_frac can exceed what HBM delivers (6.3 TB/s achievable = 0.79), and counter bytes above the
    formula bytes mean gathered rows are fetched more than once per nonzero's share of L2"""
    t = (traffic or {}).get(leg_name)
    if not t:
        block["traffic"] = None
        block["traffic_source"] = (traffic or {}).get("error", "not collected")
        return block
    b = t["hbm_bytes_per_call"]
    if kernel_sub:
        b = sum(int(v["hbm_bytes_per_launch"] * v["launches_per_call"]) for k, v in t["kernels"].items() if kernel_sub in k)
    block["traffic"] = int(b)
    block["hbm_frac"] = round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    block["traffic_source"] = traffic.get("source")
    return block


def gnn_legs(torch, ops, dev, traffic=None):
    out = {}
    rng = np.random.RandomState(0)
    # ---- config 3: GCN-Align 2-layer CSR aggregate, D-W-15K-V2 shape --------------------------------------------
    m, kgs, se, d, x = _build_gcn(torch, ops, dev, rng)
    train = np.asarray(kgs.train_links, np.int32)
    k, t = m.args.neg_triple_num, len(train)
    negs = tuple(ops.to_ids(x_.astype(np.int32)) for x_ in (np.repeat(train[:, 0], k), rng.choice(kgs.entities_num, t * k),
                                                             rng.choice(kgs.entities_num, t * k), np.repeat(train[:, 1], k)))
    nnz, n = se.adj.nnz, kgs.entities_num
    ms_epoch = _wall(torch, lambda: se.train_step(negs), 50)
    spmm_bytes = nnz * (8 + 4 * d) + 4 * n * d
    epoch_bytes = 4 * spmm_bytes + 4 * (2 * t + 4 * t * k) * d + 12 * n * d
    ms_spmm = _timed_events(torch, lambda: se.adj.mm(x, d, act=1), 50)
    out["gcn_align_se_epoch_DW15K"] = {
        "workload": "GCN-Align structure model, full-batch epoch (2 aggregates fwd, 2 bwd, L1 hinge over %d links x %d negatives, "
                    "SGD through the row normalisation), D-W-15K-V2 shape: E=%d, nnz=%d, d=%d" % (t, k, n, nnz, d),
        "ms_per_epoch": round(ms_epoch, 4), "epochs_per_s": round(1e3 / ms_epoch, 1),
        "roofline_epoch": _hbm_block("whole epoch (wall, synchronised)", epoch_bytes, ms_epoch,
                                     formula="4*(nnz*(8+4d)+4Nd) + 4*(2t+4tk)*d + 12*N*d (SURVEY 8d)"),
        "roofline": _with_hbm(_hbm_block("spmm_csr_kernel (one aggregate, relu fused; HIP events on the launch stream, 50 launches)",
                                         spmm_bytes, ms_spmm, formula="nnz*(8+4d) + 4*N*d (SURVEY 8d)",
                                         perfect_reuse_bytes=int(nnz * 8 + 8 * n * d),
                                         note="E=30,000: X (12 MB) and the CSR (4 MB) are cache-resident"),
                              traffic, "gcn_spmm_15k", ms_spmm)}
    del m, se
    torch.cuda.empty_cache()
    # ---- config 5 (evaluation half): RDGCN's metric -- manhattan similarity + rank, d = 300, 70,000 test pairs ----
    from openea_amd.modules.finding.alignment import greedy_alignment_device
    n_e, d_e = 70000, 300
    t1, t2 = _eval_tables(torch, ops, dev, n_e, d_e, rng)
    tk_e = [1, 5, 10, 50]
    os.environ["OEA_L1_EVAL"] = "f64"                                   # every pair in fp64 (round 2's path, kept beside the default)
    f64 = greedy_alignment_device(t1, t2, d_e, tk_e, "manhattan", False, 0)
    ms_l1_f64 = _wall(torch, lambda: greedy_alignment_device(t1, t2, d_e, tk_e, "manhattan", False, 0), 1)
    f64_c = greedy_alignment_device(t1, t2, d_e, tk_e, "manhattan", False, 10)
    os.environ["OEA_L1_EVAL"] = "grid"
    g16 = greedy_alignment_device(t1, t2, d_e, tk_e, "manhattan", False, 0)
    ms_l1 = _wall(torch, lambda: greedy_alignment_device(t1, t2, d_e, tk_e, "manhattan", False, 0), 2)
    g16_c = greedy_alignment_device(t1, t2, d_e, tk_e, "manhattan", False, 10)
    ms_l1_csls = _wall(torch, lambda: greedy_alignment_device(t1, t2, d_e, tk_e, "manhattan", False, 10), 1)
    # the timed grid path gives the all-pairs fp64 path's ranks and nearest candidates, plain and with CSLS (asserted like the AliNet leg)
    l1_same = bool(torch.equal(g16[0], f64[0]) and torch.equal(g16[1], f64[1]) and torch.equal(g16_c[0], f64_c[0])
                   and torch.equal(g16_c[1], f64_c[1]))
    assert l1_same, "manhattan evaluation: grid path != all-pairs fp64 path at 70,000 x 300"
    l1_hits1 = int(g16[2][0])
    del f64, f64_c, g16, g16_c
    ms_in = _wall(torch, lambda: greedy_alignment_device(t1, t2, d_e, [1, 5, 10, 50], "inner", False, 0), 2)
    sad_ops = float(n_e) * n_e * ((d_e + 7) // 8 * 8) / 2.0                # one v_sad_u16 per pair and two columns
    out["rdgcn_eval_70000x300"] = {
        "workload": "greedy_alignment over 70,000 x 70,000 pairs at d = 300 (RDGCN's test(): eval_metric manhattan, then the same "
                    "with csls = 10 -- basic_model.py:132-135; inner beside it)",
        "manhattan_ms": round(ms_l1, 2), "manhattan_pairs_per_s": round(n_e / ms_l1 * 1e3, 1),
        "manhattan_csls10_ms": round(ms_l1_csls, 2), "manhattan_csls10_pairs_per_s": round(n_e / ms_l1_csls * 1e3, 1),
        "manhattan_all_pairs_fp64_ms": round(ms_l1_f64, 2), "identical_to_all_pairs_fp64": l1_same, "hits1": l1_hits1,
        "inner_ms": round(ms_in, 2), "inner_pairs_per_s": round(n_e / ms_in * 1e3, 1),
        "roofline": {"kernel": "l1_u16_strip_kernel (16-bit grid distances of every pair; exact fp64 similarities only where the "
                               "grid's error bound leaves the comparison with the gold one open: same ranks as the all-pairs kernel)",
                     "bound": "valu_int", "achieved": round(sad_ops / (ms_l1 * 1e-3) / 1e12, 2), "peak": FP64_VALU_PEAK_TOPS,
                     "unit": "T lane-instructions/s (v_sad_u16: two columns each; peak = one vector instruction per lane and clock)",
                     "frac": round(sad_ops / (ms_l1 * 1e-3) / 1e12 / FP64_VALU_PEAK_TOPS, 4),
                     "note": "wall time of the whole call (quantisation, strips written and read back, row kernel, metrics)"},
        "roofline_all_pairs_fp64": {"kernel": "rank_valu_kernel (fp64 |a-b| sums, bit-exact with scipy cdist cityblock)", "bound": "valu_fp64",
                                    "achieved": round(2.0 * n_e * n_e * d_e / (ms_l1_f64 * 1e-3) / 1e12, 2), "peak": FP64_VALU_PEAK_TOPS,
                                    "unit": "Tops/s (one fp64 sub + one fp64 add per pair and dimension)",
                                    "frac": round(2.0 * n_e * n_e * d_e / (ms_l1_f64 * 1e-3) / 1e12 / FP64_VALU_PEAK_TOPS, 4)},
        "roofline_inner": _mfma_block(ops, n_e, n_e, d_e, ms_in)}
    del t1, t2
    torch.cuda.empty_cache()
    # ---- config 4 (evaluation half): AliNet's test() -- [init, out0, out1] concatenated = 1,200-d rows (alinet.py:948-966),
    # eval_metric inner, then csls = 10 (alinet_args_100K.json:36-37), 70,000 test pairs: 11.8 TFLOP per pass ---------------------
    try:
        out["alinet_eval_70000x1200"] = alinet_eval_leg(torch, ops, dev, rng)
    except Exception as e:      # noqa: BLE001
        out["alinet_eval_70000x1200"] = {"error": repr(e)[:300]}
    torch.cuda.empty_cache()
    # ---- config 4: AliNet at the EN-DE-100K-V1 shape, under BOTH readings of tf.sparse_softmax -------------
    a, kgs, init_s, ms_epoch = _build_alinet(torch, ops, dev, epochs=3)
    g2, g1 = a.adj[1], a.adj[0]
    n, d = kgs.entities_num, a.args.layer_dims[1]
    nnz2 = g2.nnz
    agg = nnz2 * (8 + 4 * d) + 4 * n * d
    x1 = torch.randn(n, ops.pad4(d), device=dev)
    ms_1hop = _timed_events(torch, lambda: g1.fwd.apply(x1, d), 10)
    att = {}
    for tag, g in (("runs", g2), ("row", _row_grouped(g2, dev))):
        fwd, bwd = _attn_closures(torch, ops, dev, g, n, d)
        ms_f = _timed_events(torch, fwd, 10)
        ms_b = _timed_events(torch, bwd, 10)
        single = len(g.seg_row_host) == nnz2
        fwd_bytes = agg + 4 * nnz2 + 8 * n
        bwd_bytes = agg + (0 if single else agg)       # dV = transposed aggregate; d alpha dots only for multi-edge segments
        blk = _hbm_block("sparse attention operator fwd + bwd (softmax statistics, alpha, aggregate; d alpha, d z, d V) at d=%d, "
                         "grouping '%s': %d softmax groups over %d edges%s; HIP events on the launch stream"
                         % (d, tag, len(g.seg_row_host), nnz2, " (every group ONE edge: no softmax kernel runs, alpha = 1)" if single else ""),
                         fwd_bytes + bwd_bytes, ms_f + ms_b,
                         formula="fwd nnz*(12+4d)+4Nd+8N; bwd the transposed aggregate (+ one more pass of gathers for d alpha "
                                 "when groups have several edges)", perfect_reuse_bytes=int(2 * (nnz2 * 8 + 8 * n * d)))
        t_f, t_b = (traffic or {}).get("attn_%s_fwd" % tag), (traffic or {}).get("attn_%s_bwd" % tag)
        if t_f and t_b:
            tb = t_f["hbm_bytes_per_call"] + t_b["hbm_bytes_per_call"]
            blk.update(traffic=int(tb), hbm_frac=round(tb / ((ms_f + ms_b) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), traffic_source=traffic.get("source"))
        else:
            blk.update(traffic=None, traffic_source=(traffic or {}).get("error", "not collected"))
        att[tag] = {"attention_fwd_ms": round(ms_f, 4), "attention_bwd_ms": round(ms_b, 4), "softmax_groups": int(len(g.seg_row_host)),
                    "roofline": blk}
    del a
    torch.cuda.empty_cache()
    a_row, _, _, ms_epoch_row = _build_alinet(torch, ops, dev, grouping="row", epochs=3)
    del a_row
    torch.cuda.empty_cache()
    try:        # the third reading of tf.sparse_softmax (canonical re-ordering before the row softmax, values re-attached as fed)
        a_reo, _, _, ms_epoch_reo = _build_alinet(torch, ops, dev, grouping="reorder", epochs=3)
        del a_reo
    except Exception as e:      # noqa: BLE001
        ms_epoch_reo = None
        out["alinet_reorder_error"] = repr(e)[:200]
    out["alinet_EN-DE-100K"] = {
        "workload": "AliNet, layer_dims [500, 400, 300], EN-DE-100K-V1 shape: E=%d, 1-hop nnz=%d, 2-hop nnz=%d; one epoch = one "
                    "full-graph step + Adam; default attn_grouping 'runs' (TF1's run grouping on the column-major adjacency: %d "
                    "groups), 'row' (per-row softmax: %d groups) beside it" % (n, g1.nnz, nnz2, att["runs"]["softmax_groups"],
                                                                             att["row"]["softmax_groups"]),
        "init_s": round(init_s, 2), "ms_per_epoch": round(ms_epoch, 2), "ms_per_epoch_grouping_row": round(ms_epoch_row, 2),
        "ms_per_epoch_grouping_reorder": None if ms_epoch_reo is None else round(ms_epoch_reo, 2),
        "attention_fwd_ms": att["runs"]["attention_fwd_ms"], "attention_bwd_ms": att["runs"]["attention_bwd_ms"],
        "roofline": att["runs"]["roofline"], "grouping_row": att["row"],
        "roofline_1hop_aggregate": _with_hbm(_hbm_block("spmm_csr_kernel (1-hop aggregate, d=%d)" % d, g1.nnz * (8 + 4 * d) + 4 * n * d,
                                                        ms_1hop, formula="nnz*(8+4d) + 4*N*d",
                                                        perfect_reuse_bytes=int(g1.nnz * 8 + 8 * n * d)),
                                             traffic, "alinet_1hop_100k", ms_1hop)}
    out["note"] = ("frac = SURVEY 8d's formula bytes (every gathered row counted once per nonzero) / time / 8 TB/s; traffic = counter "
                   "bytes (2*FETCH_SIZE + WRITE_SIZE) crossing the L2 <-> fabric boundary per call, hbm_frac = traffic / time / 8 TB/s "
                   "-- the counters include L2 misses served by the 256 MB Infinity Cache, so hbm_frac above 0.79 (6.3 TB/s "
                   "achievable HBM) means MALL hits, and traffic below the formula bytes means rows re-used out of L2; "
                   "perfect_reuse_bytes = nnz*8 + 8*N*d is the floor with every row read once")
    return out


def alinet_eval_leg(torch, ops, dev, rng, n_e=70000, dims=(500, 400, 300)):
    """AliNet's evaluation shape: rows = three L2-normalised blocks (500 + 400 + 300 columns, norm sqrt(3)), inner product, plain
    and with CSLS 10; the product path (certified bf16 prefilter) beside the exact fp32 sweep, with the prefilter's record
    count and whether its fallback fired"""
    from openea_amd.modules.finding.alignment import greedy_alignment_device
    d_e = sum(dims)
    blocks1, blocks2 = [], []
    for d_b in dims:
        b1 = rng.standard_normal((n_e, d_b)).astype(np.float32)
        b2 = (b1 + 8.0 * rng.standard_normal((n_e, d_b)).astype(np.float32)).astype(np.float32)     # Hits@1 ~ 0.47 (0.3-0.7)
        blocks1.append(b1 / np.linalg.norm(b1, axis=1, keepdims=True))
        blocks2.append(b2 / np.linalg.norm(b2, axis=1, keepdims=True))
    t1 = ops.to_table(np.concatenate(blocks1, axis=1), dev=dev)
    t2 = ops.to_table(np.concatenate(blocks2, axis=1), dev=dev)
    del blocks1, blocks2
    tk = [1, 5, 10, 50]
    saved = os.environ.get("OEA_EVAL_BF16")
    res = {}
    try:
        os.environ["OEA_EVAL_BF16"] = "1"
        ref = greedy_alignment_device(t1, t2, d_e, tk, "inner", False, 0)
        ms_in = _wall(torch, lambda: greedy_alignment_device(t1, t2, d_e, tk, "inner", False, 0), 2)
        ref_c = greedy_alignment_device(t1, t2, d_e, tk, "inner", False, 10)
        ms_cs = _wall(torch, lambda: greedy_alignment_device(t1, t2, d_e, tk, "inner", False, 10), 1)
        stats = {}
        bf = bool(ops.eval_bf16_enabled(n_e, n_e))
        if bf:
            ops.rank_eval_metrics_bf16(t1, t2, d_e, tk, stats=stats)
        os.environ["OEA_EVAL_BF16"] = "0"
        os.environ["OEA_CSLS_BF16"] = "0"
        f32 = greedy_alignment_device(t1, t2, d_e, tk, "inner", False, 0)
        ms_f32 = _wall(torch, lambda: greedy_alignment_device(t1, t2, d_e, tk, "inner", False, 0), 1)
        f32_c = greedy_alignment_device(t1, t2, d_e, tk, "inner", False, 10)
        ms_f32_c = _wall(torch, lambda: greedy_alignment_device(t1, t2, d_e, tk, "inner", False, 10), 1)
        same = bool(torch.equal(ref[0], f32[0]) and torch.equal(ref[1], f32[1]) and torch.equal(ref_c[0], f32_c[0])
                    and torch.equal(ref_c[1], f32_c[1]))
    finally:
        os.environ.pop("OEA_CSLS_BF16", None)
        if saved is None:
            os.environ.pop("OEA_EVAL_BF16", None)
        else:
            os.environ["OEA_EVAL_BF16"] = saved
    fl = 2.0 * n_e * n_e * d_e
    peak = BF16_SPLIT_PEAK_TF if (bf and not stats.get("fallback")) else FP32_MFMA_PEAK_TF
    res.update({"workload": "greedy_alignment over %d x %d pairs at d = %d (AliNet's test(): inner, then csls = 10; "
                            "alinet.py:948-966, alinet_args_100K.json)" % (n_e, n_e, d_e),
                "inner_ms": round(ms_in, 2), "inner_csls10_ms": round(ms_cs, 2), "fp32_sweep_ms": round(ms_f32, 2),
                "fp32_sweep_csls10_ms": round(ms_f32_c, 2), "bf16_prefilter": bf, "records_per_row": round(stats.get("records", 0) / n_e, 2),
                "fallback": bool(stats.get("fallback", False)), "identical_to_fp32_sweep": same,
                "bound": "mfma", "peak": round(peak, 1), "unit": "TFLOP/s (algorithmic: 2*N1*N2*d)",
                "achieved": round(fl / (ms_in * 1e-3) / 1e12, 2), "frac": round(fl / (ms_in * 1e-3) / 1e12 / peak, 4),
                "csls_frac": round(2 * fl / (ms_cs * 1e-3) / 1e12 / peak, 4),
                "fp32_sweep_frac": round(fl / (ms_f32 * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4),
                "hits1": int(ref[2][0]),
                "note": "frac = 2*N1*N2*d / wall time of the whole call / peak (2.5 PFLOP/s / 3 bf16 products per exact product when the "
                        "prefilter ran, else 157.3 TFLOP/s fp32 MFMA); csls_frac counts the two sweeps of the CSLS evaluation"})
    return res


# the reference's OWN numpy functions (imported in place) timed in the build
# container (8 host cores, numpy 2.2.6 / OpenBLAS); they cannot run on the GPU box (/root/reference is not there)
REFERENCE_TIMINGS = {
    "provenance": "BASELINE.md section 3: reference functions imported from /root/reference, build container, 8 cores",
    "generate_neg_triples_fast_truncated_neg_per_s_per_core": 6.3e4,
    "generate_neighbours_single_thread_rows_per_s_15000x100_k1500": 2062,
    "find_neighbours_rows_per_s_100000x100_k2000": 618,
    "greedy_alignment_pairs_per_s_10500_inner_1thread": 4548,
    "greedy_alignment_pairs_per_s_10500_inner_csls10": 813,
    "greedy_alignment_pairs_per_s_10500_manhattan": 551,
}


def cpu_baseline(kgs, d, args, k1, k2):
    """The C oracle port of the same step (sampler + fused step) on ONE host thread and on more host cores (OpenMP), each
    on a bounded sample of the same workload.  Reported baseline, not the target."""
    from oracle import cport
    from openea_amd.modules.base.initializers import truncated_normal_host
    t1 = np.asarray(kgs.kg1.relation_triples_list, np.int32)
    t2 = np.asarray(kgs.kg2.relation_triples_list, np.int32)
    b1 = int(len(t1) / (len(t1) + len(t2)) * args.batch)
    b2 = args.batch - b1
    rng = np.random.RandomState(1)
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
    per_epoch = max(min(len(t1) // b1, len(t2) // b2), 1)

    def run(threads, budget_s):
        cport.set_num_threads(threads)
        ent = truncated_normal_host(np.random.RandomState(1), (kgs.entities_num, d), 1.0 / np.sqrt(d))
        rel = truncated_normal_host(np.random.RandomState(2), (kgs.relations_num, d), 1.0 / np.sqrt(d))
        ent_acc, rel_acc = np.full_like(ent, 0.1), np.full_like(rel, 0.1)
        steps, t0 = 0, time.perf_counter()
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
            if el > budget_s:
                return steps, el
    host_cores = os.cpu_count()
    # 1 thread and a few larger counts (the step's scatter is atomic adds on shared rows: past a few dozen threads it gets
    # slower, 256 threads measured 30x slower than one in round 2 and are no longer tried) -- the best is the baseline
    counts = sorted({1, min(8, host_cores), min(16, host_cores), min(32, host_cores)})
    runs = {}
    for c in counts:
        s_, e_ = run(c, 8.0 if c == 1 else 4.0)
        runs[c] = (s_, e_, s_ * args.batch / e_)
    best = max(runs, key=lambda c: runs[c][2])
    # the other two legs of the metric on the same host cores: alignment evaluation (the oracle's greedy_alignment:
    # similarity matrix + ranks, alignment.py:13-84) over the 10,500 test pairs, and the neighbour search
    # (find_neighbours, batch.py:157-165) on a sample of 1,500 query rows -- C loops under OpenMP, `best` threads
    from oracle import np_oracle as orc
    cport.set_num_threads(best)
    e_rng = np.random.RandomState(3)
    n_pairs = min(len(kgs.test_entities1), 10500)          # bounded sample: the 15K datasets' test set (70,000^2 on host cores takes minutes)
    e1 = e_rng.standard_normal((n_pairs, d)).astype(np.float32)
    e2 = e_rng.standard_normal((n_pairs, d)).astype(np.float32)
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 /= np.linalg.norm(e2, axis=1, keepdims=True)
    legs = {}
    for name, csls in (("eval_pairs_per_s_inner", 0), ("eval_pairs_per_s_inner_csls10", 10)):
        t0 = time.perf_counter()
        orc.greedy_alignment(e1, e2, [1, 5, 10, 50], best, "inner", False, csls, True)
        legs[name] = round(n_pairs / (time.perf_counter() - t0), 1)
    n_ent = len(kgs.kg1.entities_list)
    emb = e_rng.standard_normal((n_ent, d)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    t0 = time.perf_counter()
    cport.topk_inner(emb[:1500], emb, k1)
    legs["neighbour_rows_per_s"] = round(1500 / (time.perf_counter() - t0), 1)
    legs["threads"] = best
    legs["sample"] = "%d x %d x %d evaluation, 1,500 of %d query rows of the neighbour search (k = %d)" % (n_pairs, n_pairs, d, n_ent, k1)
    return {"value": round(runs[best][2], 1), "unit": "triples/s", "cores": best, "kind": "port", "host_cores": host_cores,
            "other_legs": legs,
            "value_by_threads": {str(c): round(runs[c][2], 1) for c in counts},
            "sample": "the same workload (batch %d, k=%d, dim=%d), oracle/c/oracle.c sampler + step (fp64 internals, OpenMP): "
                      % (args.batch, args.neg, d)
                      + ", ".join("%d steps on %d thread(s) in %.1f s" % (runs[c][0], c, runs[c][1]) for c in counts)
                      + "; the box has %d host cores (the scatter-add on shared rows stops scaling past a few dozen threads)" % host_cores,
            "reference_functions": REFERENCE_TIMINGS}


if __name__ == "__main__":
    main()
