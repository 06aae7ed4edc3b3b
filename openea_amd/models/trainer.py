"""Device-resident epoch engine for the translational path.

This is what replaces the reference's producer processes + queue + feed_dict +
``session.run([triple_loss, triple_optimizer])`` loop
(``BasicModel.launch_triple_training_1epo``, models/basic_model.py:211-236): positives,
triple membership set, neighbour lists, tables, optimiser state and the loss accumulator all
live in HBM; an epoch = one sampler launch for all its batches (both KGs, batch.py:36-45) + two kernels per step,
enqueued by one C call (oea_triple_epoch).

Multi-GPU (torch.distributed, backend nccl = RCCL): tables are replicated, each rank scores
its own slice of every batch, the gradient scratch is summed with ONE all-reduce per step and
every rank applies the identical optimiser update (include/openea_hip.h: OEA_PHASE_*), which
equals the single-GPU step on the concatenated batch.
"""
import os

import numpy as np
import torch

from .. import ops
from ..modules.train.batch import EpochBatches, TripleSampler, neighbours_device


class EmbeddingTable:
    """A trainable table: raw variable [rows, ld] on the device + the l2_norm flag that the
    reference bakes into the tensor returned by init_embeddings (initializers.py:26)."""

    def __init__(self, host_init, is_l2_norm, name, dev=None, visible_rows=None):
        """visible_rows: the rows that ARE the reference's variable when further rows are stacked below them
        (TransD keeps ent_transfer / rel_transfer in rows [visible_rows, rows) of the same table)."""
        host_init = np.asarray(host_init, np.float32)
        self.name = name
        self.rows, self.dim = host_init.shape
        self.visible_rows = self.rows if visible_rows is None else int(visible_rows)
        self.is_l2_norm = bool(is_l2_norm)
        self.var = ops.to_table(host_init, dev=dev)          # [rows, ld], pad columns zero
        self.ld = self.var.shape[1]

    def lookup(self, ids):
        """tf.nn.embedding_lookup(self, ids) -> device [n, ld] (normalised when l2_norm)."""
        if not hasattr(ids, "is_cuda"):
            ids = ops.to_ids(np.asarray(ids, np.int32), self.var.device)
        return ops.gather_rows(self.var, self.dim, ids, normalize=self.is_l2_norm)

    def eval(self, session=None):
        """`.eval(session=...)` of the reference: the (normalised) table on the host [rows, dim]."""
        ids = torch.arange(self.visible_rows, dtype=torch.int32, device=self.var.device)
        return self.lookup(ids)[:, :self.dim].cpu().numpy()

    def raw(self):
        return self.var[:self.visible_rows, :self.dim].cpu().numpy()


def cfg_allows_c_epoch(cfg):
    """the one-call partitioned epoch covers what the partition covers (SGD / Adagrad, TransE / TransH / TransD scores)"""
    return (cfg.score_kind in (ops.SCORE_TRANSE, ops.SCORE_TRANSH, ops.SCORE_TRANSD)
            and cfg.opt_kind in (ops.OPT_KIND["SGD"], ops.OPT_KIND["Adagrad"]))


class TripleTrainer:
    def __init__(self, ent, rel, cfg, optimizer='Adagrad', dist_group=None, replicated=False, exchange=None):
        """replicated=True (with a dist_group): every rank feeds the SAME full batch (small steps that are not worth
        sharding: MTransE's mapping step, BootEA's alignment step); the gradient scratch is then averaged instead of
        summed -- its only purpose is to give every replica the same bits (fp32 atomics reorder per process).

        exchange (with a dist_group, not replicated; default: OEA_DP_EXCHANGE or 'step'):
          'step'      the G-rank job EQUALS the single-GPU job: per step, gradients reduce-scattered to the owners of the
                      entity rows (owner = id mod G), owners run the optimiser, updated rows all-gathered;
          'halo'      the same partition and the same result as 'step' (bit for bit in the fixed-point build), but a step moves only
                      the BOUNDARY rows -- the rows its batch refers to (BASELINE.json north_star: "all-gather of boundary
                      embeddings"): every rank derives the row lists of every rank's share of every step from the epoch's
                      positives and negatives, gradient rows go to their owners by all-to-all, the rows the next step's readers
                      refer to come back by all-to-all, one dense all-gather ends each call (oea_triple_epoch_range_halo:
                      ~19 MB per rank and step instead of 161 MB at the 100K shape with 8 ranks).  TransE / TransH scores with the
                      one-call epoch; anything else runs 'step';
          'allreduce' the same with one dense all-reduce and a replicated update (round 1);
          'epoch'     BASELINE.json north_star: "RCCL all-gather of boundary embeddings ... each epoch" -- every rank trains
                      its share of every batch on its LOCAL copy of the tables with the fused single-GPU epoch call (no
                      collective, no host work per step); at the epoch's end the copies are reconciled: the changes of
                      tables and optimiser state since the last exchange are summed over the ranks (reduce-scatter to
                      the owner of each row + all-gather of the owned rows = one all-reduce of the deltas).  Inside an
                      epoch a rank reads rows other ranks are training with a delay of at most one epoch (local SGD): NOT
                      equal to the single-GPU job -- drift measured in tests/test_dist_gpu.py, DESIGN.md section 6."""
        self.ent, self.rel, self.cfg = ent, rel, cfg
        self.replicated = bool(replicated)
        import os as _os
        self.exchange = exchange or {"partition": "step"}.get(_os.environ.get("OEA_DP_EXCHANGE", "step"),
                                                               _os.environ.get("OEA_DP_EXCHANGE", "step"))
        if self.exchange not in ("step", "halo", "allreduce", "epoch"):
            raise ValueError("dp_exchange: 'step', 'halo', 'allreduce' or 'epoch'")
        if self.exchange == "halo" and (cfg.score_kind not in (ops.SCORE_TRANSE, ops.SCORE_TRANSH) or optimizer not in ('Adagrad', 'SGD')):
            self.exchange = "step"        # TransD's stacked transfer rows are not in the boundary lists; dense optimisers move every row
        if dist_group is None or self.replicated:
            self.exchange = "step"
        if self.exchange == "epoch" and cfg.score_kind not in (ops.SCORE_TRANSE, ops.SCORE_TRANSD):
            self.exchange = "step"        # TransH keeps a third trained table outside (ent, rel): per-step exchange only
        if self.exchange == "epoch" and optimizer not in ('Adagrad', 'SGD'):
            # epoch_sync sums the ranks' CHANGES of tables + state: sound only when a rank leaves the rows it did not touch
            # alone (SGD / Adagrad).  Adam and Adadelta (with l2_norm) move every row every step (m, v and the table decay
            # where the gradient is zero), so the sum would count that decay G times (m flips sign, v goes negative):
            # those optimisers keep the per-step exchange (dense all-reduce of the gradients, replicated update).
            self.exchange = "step"
        self._snap = None
        dev = ent.var.device
        self.dev = dev
        if optimizer == 'Adagrad':        # tf.train.AdagradOptimizer: initial_accumulator_value = 0.1
            self.ent_acc = torch.full_like(ent.var, 0.1)
            self.rel_acc = torch.full_like(rel.var, 0.1)
        elif optimizer in ('Adam', 'Adadelta'):       # (m, v) / (accum, accum_update): zeros, [2, rows, ld]
            self.ent_acc = torch.zeros((2,) + tuple(ent.var.shape), dtype=torch.float32, device=dev)
            self.rel_acc = torch.zeros((2,) + tuple(rel.var.shape), dtype=torch.float32, device=dev)
        else:
            self.ent_acc = self.rel_acc = None
        self.t = 0                        # optimiser steps taken (Adam's bias correction counts them)
        self.ws = ops.step_workspace(ent.rows, rel.rows, ent.ld, dev)
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self._empty = torch.zeros((0, 3), dtype=torch.int32, device=dev)
        self.dist = dist_group
        self.xchg = ops.step_exchange_view(self.ws, ent.rows, rel.rows, ent.ld) if dist_group is not None else None
        # entity-id partitioning (owner = id mod G: include/openea_hip.h, oea_part_*): the sharded main step of the
        # translational models.  OEA_DP_EXCHANGE=allreduce keeps the replicated update with one dense all-reduce.
        self.part = None
        import os
        if (dist_group is not None and not self.replicated and optimizer in ('Adagrad', 'SGD')
                and cfg.score_kind in (ops.SCORE_TRANSE, ops.SCORE_TRANSH, ops.SCORE_TRANSD) and self.exchange in ("step", "halo")):
            self._init_partition(optimizer)
        self.halo = None                  # workspace + buffers of the boundary-row exchange (halo_buffers, made on first use)
        self.halo_stats = [0, 0, 0, 0]    # bytes pushed, bytes pulled, largest rows sent in a step, steps (this rank)
        if self.exchange == "halo" and (self.part is None or getattr(self, "comm", None) is None):
            self.exchange = "step"        # no communicator for the one-call epoch: the dense per-step protocol (same result)
        if self.exchange == "halo" and self.comm.callbacks and self.comm.device_collectives:
            # the library's own RCCL communicator could not be made on an nccl group: the callback back end runs the dense
            # protocol's collectives on the registered device buffers, but has no device all-to-all -- dense protocol
            self.exchange = "step"

    # ---- dp_exchange = 'epoch': local steps, one exchange per epoch ----------------------------------------------
    @property
    def local_epochs(self):
        return self.dist is not None and not self.replicated and self.exchange == "epoch"

    def _state_tensors(self):
        return [t for t in (self.ent.var, self.rel.var, self.ent_acc, self.rel_acc) if t is not None]

    def epoch_begin(self):
        """remember tables + optimiser state as every rank holds them now (identical on all ranks; taken afresh every
        epoch: other trainers -- BootEA's alignment step, MTransE's mapping step -- move the same tables in between)"""
        cur = self._state_tensors()
        if self._snap is None:
            self._snap = [torch.empty_like(t) for t in cur]
        for t, snap in zip(cur, self._snap):
            snap.copy_(t)

    def epoch_sync(self):
        """reconcile the ranks' local copies: state = snapshot + sum over ranks of (local state - snapshot).  The sum is
        one all-reduce per tensor (RCCL: reduce-scatter to the owners + all-gather of the owned rows); every rank ends
        with the same bits and takes them as the next snapshot."""
        import torch.distributed as dist
        if self._snap is None:
            return
        from . import dist as mdist
        for cur, snap in zip(self._state_tensors(), self._snap):
            cur.sub_(snap)
            if mdist._staged(cur, self.dist):          # gloo (one-GPU wiring tests): stage on the host
                h = cur.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.dist)
                cur.copy_(h)
            else:
                dist.all_reduce(cur, op=dist.ReduceOp.SUM, group=self.dist)
            cur.add_(snap)

    def epoch_exchange_bytes(self):
        """bytes this rank sends (= receives) per epoch-end exchange (ring all-reduce of every state tensor)"""
        if not self.local_epochs:
            return 0
        import torch.distributed as dist
        g = dist.get_world_size(self.dist)
        return int(sum(t.numel() for t in self._state_tensors()) * 4 * 2 * (g - 1) / g)

    def _init_partition(self, optimizer):
        import torch.distributed as dist
        g, rk = dist.get_world_size(self.dist), dist.get_rank(self.dist)
        ent, rel, dev = self.ent, self.rel, self.dev
        rpr = ops.part_rows_per_rank(ent.rows, g)
        chunk = rpr * (ent.ld + 1)
        gdt = ops.scratch_dtype()                  # fp32, or int64 fixed point in the deterministic build: exact sums on the wire
        p = dict(world=g, rank=rk, rpr=rpr, chunk=chunk,
                 send=torch.empty(g * chunk, dtype=gdt, device=dev),
                 own=torch.empty(chunk, dtype=gdt, device=dev),
                 rel_x=torch.empty(rel.rows * (rel.ld + 1), dtype=gdt, device=dev),
                 upd=torch.empty((rpr, ent.ld), dtype=torch.float32, device=dev),
                 all=torch.empty((g, rpr, ent.ld), dtype=torch.float32, device=dev))
        if optimizer == 'Adagrad':                 # the state of the owned rows only: row j <-> entity id j * G + rank
            p['acc_own'] = torch.full((rpr, ent.ld), 0.1, dtype=torch.float32, device=dev)
            self.ent_acc = None                    # 1/G of the optimiser state per rank
        else:
            p['acc_own'] = None
        self.part = p
        # The epoch's partitioned steps come from ONE C call over the C ABI's communicator (oea_triple_epoch_range_comm:
        # RCCL on a multi-GPU node, host callbacks over the torch.distributed group otherwise -- models/dist.py:CAbiComm)
        # instead of six library calls + three torch.distributed calls per step.  OEA_DP_C_EPOCH=0 keeps the per-step
        # Python loop (the same protocol, _step_partitioned).
        self.comm = None
        import os as _os
        if _os.environ.get("OEA_DP_C_EPOCH", "1") != "0" and cfg_allows_c_epoch(self.cfg):
            from . import dist as mdist
            bufs = [p[k] for k in ('send', 'own', 'rel_x', 'upd', 'all')]
            if self.cfg.score_kind == ops.SCORE_TRANSH:
                p['nrm'] = ops.step_normal_views(self.ws, ent.rows, rel.rows, ent.ld)
                bufs += list(p['nrm'])
            try:
                self.comm = mdist.c_abi_comm(self.dist, bufs)
            except Exception as e:            # noqa: BLE001 -- no communicator: the per-step Python loop runs the same protocol
                import sys
                print("[openea_amd] one-call partitioned epoch unavailable (%s): per-step exchange from Python" % str(e)[:200],
                      file=sys.stderr)
                self.comm = None

    def halo_buffers(self, steps, max_batch, k):
        """workspace + exchange buffers of the boundary-row exchange, sized for `steps` steps of batches of <= max_batch rows"""
        key = (int(steps), int(max_batch), int(k))
        if self.halo is None or self.halo['key'] != key:
            import torch.distributed as dist
            self.halo = ops.halo_buffers(self.ent.rows, self.ent.ld, dist.get_world_size(self.dist), steps, max_batch, k, self.dev)
        return self.halo

    def close(self):
        """release the C ABI communicator of the partitioned step (RCCL communicator + staging buffers); the garbage collector
        does the same through CAbiComm.__del__"""
        comm, self.comm = getattr(self, "comm", None), None
        if comm is not None:
            comm.close()

    def _step_partitioned(self, pos, neg):
        """GRAD | pack | reduce-scatter + relation all-reduce | apply owned rows | all-gather | unpack"""
        from . import dist as mdist
        p = self.part
        ops.triple_step(self.ent.var, None, self.rel.var, self.rel_acc, self.ent.dim, pos, neg, self.cfg, self.ws, self.loss,
                        phase=ops.PHASE_GRAD)
        ops.part_pack(self.ws, self.ent.rows, self.rel.rows, self.ent.ld, p['world'], p['send'], p['rel_x'])
        mdist.reduce_scatter_(p['own'], p['send'], self.dist)
        mdist.allreduce_sum_(p['rel_x'], self.dist)
        transh = self.cfg.score_kind == ops.SCORE_TRANSH
        if transh:                     # the normal-vector table is relation-sized and replicated: its scratch is summed too
            if 'nrm' not in p:
                p['nrm'] = ops.step_normal_views(self.ws, self.ent.rows, self.rel.rows, self.ent.ld)
            mdist.allreduce_sum_(p['nrm'][0], self.dist)
            mdist.allreduce_sum_(p['nrm'][1], self.dist)
        import ctypes as _C
        n_items = int(ops.lib().oea_step_items(_C.byref(self.cfg), int(pos.shape[0]), 0 if neg is None else int(neg.shape[0])))
        ops.part_apply(self.ent.var, p['acc_own'], self.rel.var, self.rel_acc, p['world'], p['rank'], p['own'], p['rel_x'],
                       p['upd'], self.cfg, self.ws, n_items, self.loss)
        if transh:
            ops.step_apply_normals(self.ent.rows, self.rel.rows, self.ent.ld, self.cfg, self.ws)
        mdist.all_gather_into_(p['all'], p['upd'], self.dist)
        ops.part_unpack(self.ent.var, p['world'], p['rank'], p['all'])

    def _average_xchg(self):
        """replicated steps: every rank computed the same gradients from the same inputs; the average gives every replica
        the same bits (fp32 atomics reorder per process).  The fixed-point build's sums are identical already: G x / G = x"""
        g = torch.distributed.get_world_size(self.dist)
        if self.xchg.dtype == torch.int64:
            self.xchg.div_(g, rounding_mode="floor")
        else:
            self.xchg /= g

    def count_steps(self, n=1):
        """n more optimiser steps are about to run: cfg.opt_t = 1-based count of the first of them"""
        self.cfg.opt_t = self.t + 1
        self.t += n

    def step(self, pos, neg):
        """pos / neg: device int32 [n,3] (neg may be None)."""
        self.count_steps()
        if self.part is not None:
            self._step_partitioned(pos, neg)
        elif self.dist is None:
            ops.triple_step(self.ent.var, self.ent_acc, self.rel.var, self.rel_acc, self.ent.dim, pos, neg, self.cfg,
                            self.ws, self.loss)
        else:
            import torch.distributed as dist
            ops.triple_step(self.ent.var, self.ent_acc, self.rel.var, self.rel_acc, self.ent.dim, pos, neg, self.cfg,
                            self.ws, self.loss, phase=ops.PHASE_GRAD)
            dist.all_reduce(self.xchg, op=dist.ReduceOp.SUM, group=self.dist)
            if self.replicated:
                self._average_xchg()
            ops.triple_step(self.ent.var, self.ent_acc, self.rel.var, self.rel_acc, self.ent.dim, pos, neg, self.cfg,
                            self.ws, self.loss, phase=ops.PHASE_APPLY)

    def apply_entity_row_grads(self, ids, grads):
        """optimiser step for gradients w.r.t. the normalised entity rows `ids` computed outside the
        fused kernel (device [n, ld] fp32): scatter into the scratch, then the apply phase."""
        self._no_partition("apply_entity_row_grads")
        ops.step_scatter_ent_rows(self.ws, self.ent.rows, self.rel.rows, self.ent.ld, ids, grads)
        self.count_steps()
        if self.dist is not None:
            import torch.distributed as dist
            dist.all_reduce(self.xchg, op=dist.ReduceOp.SUM, group=self.dist)
            if self.replicated:
                self._average_xchg()
        ops.triple_step(self.ent.var, self.ent_acc, self.rel.var, self.rel_acc, self.ent.dim, self._empty, None,
                        self.cfg, self.ws, self.loss, phase=ops.PHASE_APPLY)

    def apply_scratch(self):
        """optimiser step for whatever a kernel outside the fused step has added to the gradient scratch
        (oea_mapping_step): the exchange (if any), then the apply phase."""
        self._no_partition("apply_scratch")
        self.count_steps()
        if self.dist is not None:
            import torch.distributed as dist
            dist.all_reduce(self.xchg, op=dist.ReduceOp.SUM, group=self.dist)
            if self.replicated:
                self._average_xchg()
        ops.triple_step(self.ent.var, self.ent_acc, self.rel.var, self.rel_acc, self.ent.dim, self._empty, None,
                        self.cfg, self.ws, self.loss, phase=ops.PHASE_APPLY)

    def _no_partition(self, what):
        """the replicated side entries (gradients computed outside the fused step) go through the dense exchange and the
        full optimiser state; a trainer partitioned by entity id keeps 1 / G of the state (ent_acc is None) -- the small
        steps that use them (MTransE's mapping step, BootEA's alignment step) build their own replicated trainer"""
        if self.part is not None:
            raise RuntimeError("TripleTrainer.%s on a trainer partitioned by entity id: build the trainer with "
                               "replicated=True (its optimiser state is then whole on every rank)" % what)

    def exchange_description(self):
        if self.dist is None:
            return None
        if self.local_epochs:
            return ("dp_exchange = 'epoch': local steps on this rank's share of every batch (fused epoch call, no collective); "
                    "per epoch one all-reduce of the changes of tables + optimiser state (= reduce-scatter to the row owners + "
                    "all-gather of the owned rows)")
        if self.part is not None and self.exchange == "halo":
            return ("owner = id mod G, boundary rows only: all-to-all of the gradient rows (+ flag) a rank's share of the batch "
                    "refers to (lists derived on every rank from the epoch's batches: no index travels), all-reduce of the relation "
                    "rows, all-to-all of the current values of the rows the next step's readers refer to; one dense all-gather "
                    "of the owned rows ends each call")
        if self.part is not None:
            return ("owner = id mod G: reduce-scatter of the packed gradient rows + touched flags ([G][rows/G][ld+1] fp32), "
                    "all-reduce of the relation rows, all-gather of the updated owned rows ([G][rows/G][ld])")
        return "dense all-reduce of the gradient scratch + touched flags, every rank applies every row"

    def exchange_bytes_per_step(self):
        """bytes this rank sends (= receives) per optimiser step in the data-parallel exchange: a ring all-reduce of the
        gradient scratch + touched flags moves 2 (G-1)/G of the buffer"""
        if self.dist is None:
            return 0
        import torch.distributed as dist
        g = dist.get_world_size(self.dist)
        if self.part is not None and self.exchange == "halo" and self.halo_stats[3] > 0:
            # measured: gradient rows sent + rows received per step (the closing all-gather of a call is amortised over its steps by
            # the caller: epoch_exchange_bytes has no halo term) + the relation all-reduce
            p = self.part
            nb = lambda t: t.numel() * t.element_size()
            return int((self.halo_stats[0] + self.halo_stats[1]) / self.halo_stats[3] + nb(p['rel_x']) * 2 * (g - 1) / g)
        if self.part is not None:       # reduce-scatter of the packed gradients + all-gather of the updated rows + relation all-reduce
            p = self.part            # the gradients travel as fp32 or (deterministic build) int64, the updated rows as fp32
            nb = lambda t: t.numel() * t.element_size()
            return int((nb(p['send']) + nb(p['all'])) * (g - 1) / g + nb(p['rel_x']) * 2 * (g - 1) / g)
        return int(self.xchg.numel() * self.xchg.element_size() * 2 * (g - 1) / g)

    def pop_loss(self):
        """epoch loss (sum of batch losses) -> host float; resets the accumulator.  Under data
        parallelism every rank holds the loss of its own slices; they are summed here."""
        if self.dist is not None and not self.replicated:
            import torch.distributed as dist
            dist.all_reduce(self.loss, op=dist.ReduceOp.SUM, group=self.dist)
        v = float(self.loss.item())
        self.loss.zero_()
        return v


class RelationTripleEpochs:
    """Positives + samplers of both KGs for the (pos, neg) relation-triple batches of
    generate_relation_triple_batch (batch.py:36-45)."""

    def __init__(self, kgs, batch_size, neg_triple_num, seed=0, dev=None, rank=0, world=1):
        self.kgs = kgs
        self.dev = dev or ops.device()
        self.batch_size, self.k = batch_size, neg_triple_num
        self.rank, self.world = rank, world
        self.seed = seed
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(int(seed))
        self.batches = EpochBatches(kgs.kg1.relation_triples_list, kgs.kg2.relation_triples_list, batch_size, self.dev)
        n_total = kgs.entities_num
        self.s1 = TripleSampler(kgs.kg1.relation_triples_set, kgs.kg1.entities_list, n_total, self.dev)
        self.s2 = TripleSampler(kgs.kg2.relation_triples_set, kgs.kg2.entities_list, n_total, self.dev)
        triples_num = len(self.batches.t1) + len(self.batches.t2)
        self.triple_steps = int(np.ceil(triples_num / batch_size))      # basic_model.py:255
        self.global_step = 0
        self._epoch_base = 0                  # global_step at the start of the current epoch (Philox step of its step 0)
        self._epoch_negs_ready = False        # _neg_all holds the current epoch's negatives
        self._plan = self._plan_next = None   # gathered-sum plans of the current / the prepared epoch (ops.step_plan_build)
        self._plan_ready = False              # _plan was built from the current epoch's positives and negatives
        self._plan_key = None                 # (n_ent, ld) of the tables the plans were sized for
        b = self.batches
        self.neg_buf = torch.empty(((b.b1 + b.b2) * max(self.k, 1), 3), dtype=torch.int32, device=self.dev)
        self.err = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._sides = None

    def set_neighbours(self, nbr1, nbr2):
        self.s1.set_neighbours(nbr1)
        self.s2.set_neighbours(nbr2)
        self._sides = None
        self._epoch_negs_ready = False        # negatives drawn ahead with the old lists are not used any further
        self._plan_ready = False              # ... nor the plan sorted from them

    def batch(self, step):
        """-> (pos [n,3], neg [n*k,3]) device tensors for step `step` of the current epoch.
        Under data parallelism this rank takes a contiguous share of the batch rows."""
        if getattr(self, "_prefetch", None) is not None and self.in_epoch == 0:   # a prepared epoch becomes current
            self._take_prefetch(torch.cuda.current_stream())
        pos, n_split = self.batches.pos(step)
        off = 0
        if self.world > 1:
            from .dist import shard_batch
            lo, hi, n_split = shard_batch(pos.shape[0], n_split, self.rank, self.world)
            pos, off = pos[lo:hi], lo                                               # same Philox streams as 1 GPU
        neg = None
        if self.k > 0 and pos.shape[0] > 0:
            if self._sides is None:
                self._sides = (self.s1.side(), self.s2.side())
            neg = self.neg_buf[: pos.shape[0] * self.k]
            ops.sample_negatives_pair(pos, n_split, self.k, self._sides[0], self._sides[1], self.seed,
                                      self.global_step, off, neg, self.err)
        self.global_step += 1
        return pos, neg

    def run_epoch(self, trainer):
        """All `triple_steps` steps of one epoch.  Single GPU: one C call enqueues the whole epoch
        (oea_triple_epoch); data parallel (or a trainer without the fused epoch call, fused_epoch = False): per-step loop
        with the exchange between the phases."""
        assert self.in_epoch == 0, "run_epoch in the middle of an epoch: use run_steps"
        return self.run_steps(trainer, len(self.batches.splits))

    @property
    def in_epoch(self):
        """steps of the current epoch already run"""
        return self.global_step - self._epoch_base

    def run_steps(self, trainer, n_steps):
        """The next n_steps optimiser steps, continuing where the previous call stopped (epochs end -- shuffle, next
        epoch's negatives -- wherever they fall).  Single GPU: ONE C call per epoch touched (oea_triple_epoch_range).
        -> positives consumed on this rank."""
        S = len(self.batches.splits)
        n = done = 0
        while done < n_steps:
            lo = self.in_epoch
            hi = min(S, lo + n_steps - done)
            n += self._run_range(trainer, lo, hi)
            done += hi - lo
        return n

    def _run_range(self, trainer, lo, hi):
        S = len(self.batches.splits)
        b = self.batches
        local = self.world > 1 and getattr(trainer, "local_epochs", False)
        c_part = getattr(trainer, "comm", None) is not None and getattr(trainer, "part", None) is not None   # one C call, RCCL inside
        if local and lo == 0:
            trainer.epoch_begin()
        if (self.world == 1 or local or c_part) and getattr(trainer, "fused_epoch", True):
            if self._sides is None:
                self._sides = (self.s1.side(), self.s2.side())
            main = torch.cuda.current_stream()
            if lo == 0:
                self._epoch_negs_ready = self._take_prefetch(main)      # the prepared epoch's positives (and negatives) become current
            if self.k:
                self._epoch_neg_buf()
            ev_start = None
            if lo == 0:
                ev_start = torch.cuda.Event()
                ev_start.record(main)                     # everything that still reads the spare buffers is before this
            have = self.k and self._epoch_negs_ready
            if hasattr(trainer, "count_steps"):
                trainer.count_steps(int((np.diff(b.offsets[lo:hi + 1]) > 0).sum()))
            # the boundary-row exchange plans a range from negatives that EXIST (drawn ahead for the epoch); a range that starts inside
            # an epoch whose negatives were dropped (set_neighbours between two run_steps calls) takes the dense exchange, which
            # samples step by step
            halo_ok = (not self.k) or have or lo == 0
            if c_part and getattr(trainer, "exchange", None) == "halo" and halo_ok:
                halo = trainer.halo_buffers(S, int(np.diff(b.offsets).max()), self.k)
                if getattr(self, "_off_dev", None) is None:          # (k = 0: no negatives buffer made them)
                    self._off_dev = torch.from_numpy(b.offsets).to(self.dev)
                    self._spl_dev = torch.from_numpy(b.splits).to(self.dev)
                st = ops.triple_epoch_halo(trainer.comm.handle, trainer.ent.var, trainer.part['acc_own'], trainer.rel.var, trainer.rel_acc,
                                           trainer.ent.dim, b.dall, b.offsets, b.splits, self.k,
                                           None if (have or not self.k) else self._sides[0],
                                           None if (have or not self.k) else self._sides[1], self.seed, self._epoch_base,
                                           self._neg_all if self.k else None, self.err if self.k else None, trainer.cfg,
                                           trainer.ws, trainer.loss, self._off_dev, self._spl_dev, trainer.part, halo, step_range=(lo, hi))
                trainer.halo_stats = [trainer.halo_stats[0] + st[0], trainer.halo_stats[1] + st[1], max(trainer.halo_stats[2], st[2]),
                                      trainer.halo_stats[3] + st[3]]
            elif c_part:
                ops.triple_epoch_comm(trainer.comm.handle, trainer.ent.var, trainer.part['acc_own'], trainer.rel.var, trainer.rel_acc,
                                      trainer.ent.dim, b.dall, b.offsets, b.splits, self.k,
                                      None if (have or not self.k) else self._sides[0],
                                      None if (have or not self.k) else self._sides[1], self.seed, self._epoch_base,
                                      self._neg_all if self.k else None, self.err if self.k else None, trainer.cfg,
                                      trainer.ws, trainer.loss, self._off_dev if self.k else None,
                                      self._spl_dev if self.k else None, trainer.part, step_range=(lo, hi))
            else:
                plan = None
                if self.world == 1 and not local and self.k and self._plan_wanted(trainer):
                    # the epoch's gathered-sum plan (include/openea_hip.h): sorted on the side stream with the negatives it is
                    # built from (_prefetch_next), or by the call itself when it draws them (first epoch, after a refresh)
                    plan = (self._plan, self._plan_ready and bool(have))
                ops.triple_epoch(trainer.ent.var, trainer.ent_acc, trainer.rel.var, trainer.rel_acc, trainer.ent.dim,
                                 b.dall, b.offsets, b.splits, self.k,
                                 None if (have or not self.k) else self._sides[0],
                                 None if (have or not self.k) else self._sides[1], self.seed, self._epoch_base,
                                 self._neg_all if self.k else None, self.err if self.k else None, trainer.cfg,
                                 trainer.ws, trainer.loss, self._off_dev if self.k else None,
                                 self._spl_dev if self.k else None, step_range=(lo, hi),
                                 shard=(self.rank, self.world) if local else (0, 1), plan=plan)
                if plan is not None and (have or lo == 0):
                    self._plan_ready = True               # built by the call (or already there) for the rest of this epoch
            if lo == 0 and self.k:
                self._epoch_negs_ready = True             # a range that starts the epoch draws all its negatives
            self.global_step += hi - lo
            n = int(b.offsets[hi] - b.offsets[lo])
            if local or (c_part and self.world > 1):      # this rank's share of those batches
                nb = np.diff(b.offsets[lo:hi + 1])
                n = int((nb * (self.rank + 1) // self.world - nb * self.rank // self.world).sum())
            if ev_start is not None:
                self._prefetch_next(ev_start, trainer)    # next epoch's shuffle + negatives (+ plan) on the side stream
            if hi == S:
                self._epoch_base = self.global_step
                if local:
                    trainer.epoch_sync()                  # the one exchange of the epoch
            return n
        n = 0
        for step in range(lo, hi):
            pos, neg = self.batch(step)
            if pos.shape[0] or trainer.dist is not None:     # DP: every rank joins every exchange, rows or not
                trainer.step(pos, neg)
            n += pos.shape[0]
        if hi == S:
            self.end_epoch()
        return n

    def _epoch_neg_buf(self):
        """negatives of a whole epoch (sampled ahead by one launch) + device copies of the batch layout."""
        b = self.batches
        if getattr(self, "_neg_all", None) is None:
            self._neg_all = torch.empty((int(b.offsets[-1]) * self.k, 3), dtype=torch.int32, device=self.dev)
            self._off_dev = torch.from_numpy(b.offsets).to(self.dev)
            self._spl_dev = torch.from_numpy(b.splits).to(self.dev)
        return self._neg_all

    # ---- next epoch prepared on a side stream while this one runs ------------------------------------------------
    # The shuffle (basic_model.py:234-235) and the negatives of epoch e+1 depend on nothing epoch e computes (the
    # sampler reads the triple set and the neighbour lists, not the tables), so they are enqueued on a second HIP
    # stream into spare buffers right after epoch e's kernels; the next run_epoch swaps the buffers in.  A change of
    # the neighbour lists in between (truncated-sampling refresh) drops the prepared negatives.
    def _plan_wanted(self, trainer):
        """does this trainer's epoch run on the gathered-sum plan?  Allocates the two plan workspaces on first use."""
        ent, rel = trainer.ent, trainer.rel
        if not ops.step_plan_supported(trainer.cfg, ent.rows, rel.rows, ent.ld, self.k):
            return False
        key = (ent.rows, ent.ld)
        if self._plan is None or self._plan_key != key:
            b = self.batches
            mb = int(np.diff(b.offsets).max()) if len(b.splits) else 0
            self._plan_dims = (int(b.offsets[-1]), len(b.splits), mb, ent.rows, ent.ld)
            self._plan = ops.step_plan_buffer(*self._plan_dims, dev=self.dev)
            self._plan_next = ops.step_plan_buffer(*self._plan_dims, dev=self.dev)
            self._plan_key, self._plan_ready = key, False
        return True

    def _prefetch_next(self, ev_start, trainer=None):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.dev)
            self._neg_next = torch.empty_like(self._neg_all) if self.k else None
        side = self._side
        next_base = self._epoch_base + len(self.batches.splits)     # Philox step of the NEXT epoch's step 0
        side.wait_event(ev_start)
        with torch.cuda.stream(side):
            self.batches.shuffle(self.gen, into_next=True)
            if self.k:
                ops.sample_negatives_epoch(self.batches.dall_next, self._off_dev, self._spl_dev, len(self.batches.splits), self.k,
                                           self._sides[0], self._sides[1], self.seed, next_base, self._neg_next, self.err)
            planned = False
            if (self.k and trainer is not None and self.world == 1 and not getattr(trainer, "local_epochs", False)
                    and getattr(trainer, "part", None) is None and self._plan_wanted(trainer)):
                n_total, steps, mb, n_ent, ld = self._plan_dims
                ops.step_plan_build(self.batches.dall_next, self._neg_next, self.k, self._off_dev, n_total, steps, mb, n_ent, ld,
                                    self._plan_next)
                planned = True
            self._prefetch_ev = torch.cuda.Event()
            self._prefetch_ev.record(side)
        self._prefetch = (self._sides, next_base, planned)

    def _take_prefetch(self, main):
        """make the prepared epoch current; -> True when its negatives are usable as they are."""
        pf = getattr(self, "_prefetch", None)
        if pf is None:
            return False
        self._prefetch = None
        main.wait_event(self._prefetch_ev)
        self.batches.swap()
        ok = self.k > 0 and pf[0] is self._sides and pf[1] == self._epoch_base
        if ok:
            self._neg_all, self._neg_next = self._neg_next, self._neg_all
        self._plan_ready = bool(ok and len(pf) > 2 and pf[2])
        if self._plan_ready:
            self._plan, self._plan_next = self._plan_next, self._plan
        return ok

    def end_epoch(self):
        """per-step path: the epoch's last batch() has been taken"""
        self._epoch_base = self.global_step
        self._epoch_negs_ready = False
        self._plan_ready = False
        if getattr(self, "_prefetch", None) is not None:      # already shuffled into the spare buffer: make it current
            self._take_prefetch(torch.cuda.current_stream())
            return
        self.batches.shuffle(self.gen)      # basic_model.py:234-235

    def check(self):
        """raise random.sample's error if a candidate list was smaller than the sample (one sync)."""
        torch.cuda.synchronize(self.dev)          # the sampler may have run on the side stream
        if int(self.err.item()) != 0:
            raise ValueError("Sample larger than population or is negative")


_IDS_CACHE = {}


def _entity_ids_on_device(entity_list, device):
    """a KG's entity id list as an int32 device tensor, converted once (100,000 Python ints -> numpy -> device cost 1.5 ms of a
    15.7 ms refresh).  The cache holds the list object itself (so its id cannot be reused by another list) and re-checks its
    length and three of its elements; the loaders never edit these lists in place."""
    n = len(entity_list)
    key = (id(entity_list), str(device))
    ends = (n, int(entity_list[0]), int(entity_list[n // 2]), int(entity_list[-1])) if n else (0,)
    hit = _IDS_CACHE.get(key)
    if hit is not None and hit[0] is entity_list and hit[1] == ends:
        return hit[2]
    ids = ops.to_ids(np.asarray(entity_list, np.int32), device)
    if len(_IDS_CACHE) > 16:
        _IDS_CACHE.clear()
    _IDS_CACHE[key] = (entity_list, ends, ids)
    return ids


def refresh_is_sharded(n, k, world_size):
    """Does a refresh of n entities' k nearest neighbours on world_size ranks shard its query rows?  A rank's row block against the
    whole table is a queries != candidates search (the general list path on the bf16 split: 8.0 ms per 50,000 x 100,000 rows,
    tools/r06/u.sh), followed by the all-gather of the [n, k] int32 table (priced at 300 GB/s of all-link xGMI).  The symmetric
    search of the whole table (upper triangle only, 10.3 ms at 100,000^2) run by EVERY rank needs no exchange and returns the same
    sets on all of them.  Sharding pays from two ranks on at the 100K shape (8.0 + 1.3 against 10.3 ms), not below the symmetric
    stream path's range.  OEA_REFRESH_MODE = shard | replicate overrides."""
    if world_size <= 1:
        return False
    mode = os.environ.get("OEA_REFRESH_MODE", "")
    if mode in ("shard", "replicate"):
        return mode == "shard"
    if n < 12288:                 # below the symmetric stream path's range (csrc/topk.hip plan_stream): nothing to replicate cheaply
        return True
    scale = (n / 1.0e5) ** 2
    sharded_ms = 16.0 * scale / world_size + 4.0 * n * k * (world_size - 1) / world_size / 300.0e6
    return sharded_ms < 10.3 * scale


def refresh_neighbours(ent, entity_list, k):
    """Truncated-sampling refresh (basic_model.py:267-289): embeddings of the KG's entities
    (normalised lookup) -> k nearest entity ids per entity, all on the device."""
    ids = _entity_ids_on_device(entity_list, ent.var.device)
    emb = ent.lookup(ids)
    from . import dist as mdist
    if refresh_is_sharded(emb.shape[0], k, mdist.world()[1]):      # query rows sharded over the ranks, table all-gathered
        return mdist.sharded_neighbours(emb, ent.dim, ids, k,
                                        lambda q, cand, d, kk, id_map: ops.topk_inner(q.contiguous(), cand, d, kk, id_map=id_map))
    return neighbours_device(emb, ent.dim, ids, k)
