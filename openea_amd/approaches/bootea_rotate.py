"""BootEA_RotatE (mirror of openea/approaches/bootea_rotate.py): BootEA's bootstrapping loop around RotatE scoring --
entities are complex vectors (two fp64 tables, real and imaginary parts), a relation is a vector of phases, a triple is
scored by sum_d |h_d e^{i theta_d} - t_d| against the margin gamma under -log sigmoid (bootea_rotate.py:59-81), the
optimiser is args.optimizer (Adam in run/args/bootea_rotate_args_15K.json).

Device side: oea_rotate_step / oea_rotate_lookup (csrc/rotate_step.hip), all in fp64 like the reference's variables.
Evaluation, bootstrapping and the neighbour search read `re + im` (bootea_rotate.py:111-146,160-167); those embeddings
are handed to the fp32 evaluation kernels (the reference evaluates the fp64 arrays with numpy: a rank can differ
where two similarities agree to fp32 precision)."""
import math
import time

import numpy as np
import torch

from .. import ops
from ..models.basic_model import BasicModel
from ..models.trainer import refresh_neighbours
from ..modules.base.initializers import init_embeddings
from ..modules.bootstrapping.alignment_finder import PairSim
from ..modules.finding.evaluation import early_stop
from ..modules.load import read as rd
from ..modules.load.kg import KG
from ..modules.utils.util import task_divide
from .bootea import bootstrapping, generate_pos_batch, generate_supervised_triples


class ComplexEntityTable:
    """re_ent_embeds and im_ent_embeds (bootea_rotate.py:50-55) stacked in one fp64 device array [2E, ld]: rows [0, E)
    real parts, [E, 2E) imaginary parts.  `lookup` is what every consumer outside the training step reads: the sum
    of the two (row-normalised) parts as an fp32 block."""

    def __init__(self, re_host, im_host, is_l2_norm, dev=None):
        self.rows, self.dim = re_host.shape                      # E, d
        self.is_l2_norm = bool(is_l2_norm)
        self.var = ops.to_table64(np.concatenate([re_host, im_host]).astype(np.float64), dev)
        self.ld = self.var.shape[1]

    def _ids(self, ids):
        if ids is None or hasattr(ids, "is_cuda"):
            return ids
        return ops.to_ids(np.asarray(ids, np.int32), self.var.device)

    def lookup(self, ids, sum_norm=None):
        """l2n?(re)[ids] + l2n?(im)[ids], normalised again when sum_norm (default: the l2_norm flag, as
        eval_kg*_useful_ent_embeddings does, bootea_rotate.py:128-140) -> device fp32 [n, pad4(dim)]."""
        sum_norm = self.is_l2_norm if sum_norm is None else sum_norm
        return ops.rotate_lookup(self.var, self.dim, self._ids(ids), self.is_l2_norm, sum_norm)

    def parts(self):
        """host fp64 (re, im), each [E, dim], normalised when the flag is set: `re_ent_embeds.eval()`."""
        v = self.var[:, :self.dim].cpu().numpy()
        if self.is_l2_norm:
            v = v / np.sqrt(np.maximum((v * v).sum(1, keepdims=True), 1e-12))
        return v[:self.rows], v[self.rows:]


class PhaseTable:
    """rel_embeds (bootea_rotate.py:56-57): fp64 [R, ld] phases (before the pi / embedding_range scaling)."""

    def __init__(self, host, is_l2_norm, dev=None):
        self.rows, self.dim = host.shape
        self.is_l2_norm = bool(is_l2_norm)
        self.var = ops.to_table64(host.astype(np.float64), dev)

    def eval(self, session=None):
        v = self.var[:, :self.dim].cpu().numpy()
        if self.is_l2_norm:
            v = v / np.sqrt(np.maximum((v * v).sum(1, keepdims=True), 1e-12))
        return v


class RotateTrainer:
    """One optimiser instance over the three variables (generate_optimizer, bootea_rotate.py:107-109 / 156-158): its own
    Adam moments and step count.  Same interface as TripleTrainer (step / pop_loss / dist), without the fused epoch
    call: RelationTripleEpochs drives it step by step."""
    fused_epoch = False

    def __init__(self, ent, rel, args, neg_group_k, dist_group=None, replicated=False):
        self.ent, self.rel, self.k = ent, rel, int(neg_group_k)
        self.optimizer = args.optimizer
        self.cfg = ops.make_rotate_cfg(args.gamma, args.dim, ent.is_l2_norm, rel.is_l2_norm, args.optimizer, args.learning_rate)
        self.ent_state = ops.rotate_state(ent.var, args.optimizer)
        self.rel_state = ops.rotate_state(rel.var, args.optimizer)
        dev = ent.var.device
        self.ws = ops.rotate_workspace(ent.rows, rel.rows, ent.ld, dev)
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self.t = 0
        self.dist, self.replicated = dist_group, bool(replicated)
        self.xchg = ops.rotate_exchange_view(self.ws, ent.rows, rel.rows, ent.ld) if dist_group is not None else None

    def _run(self, pos, neg, phase):
        ops.rotate_step(self.ent.var, self.ent_state, self.rel.var, self.rel_state, self.ent.dim, pos, neg,
                        self.k if neg is not None else 0, self.cfg, self.ws, self.loss, phase=phase)

    def step(self, pos, neg):
        self.t += 1
        self.cfg.t = self.t
        if self.dist is None:
            return self._run(pos, neg, ops.PHASE_BOTH)
        import torch.distributed as dist
        self._run(pos, neg, ops.PHASE_GRAD)
        dist.all_reduce(self.xchg, op=dist.ReduceOp.SUM, group=self.dist)
        if self.replicated:
            self.xchg /= dist.get_world_size(self.dist)
        self._run(pos, neg, ops.PHASE_APPLY)

    def pop_loss(self):
        if self.dist is not None and not self.replicated:
            import torch.distributed as dist
            dist.all_reduce(self.loss, op=dist.ReduceOp.SUM, group=self.dist)
        v = float(self.loss.item())
        self.loss.zero_()
        return v


class BootEA_RotatE(BasicModel):

    def __init__(self):
        super().__init__()
        self.ref_ent1 = None
        self.ref_ent2 = None
        self.pi = 3.14159265358979323846
        self.epsilon = 2.0
        self.embedding_range = None

    def init(self):
        self.embedding_range = (self.args.gamma + self.epsilon) / self.args.dim
        self._define_variables()
        self._define_embed_graph()
        self._define_alignment_graph()
        self.ref_ent1 = self.kgs.valid_entities1 + self.kgs.test_entities1
        self.ref_ent2 = self.kgs.valid_entities2 + self.kgs.test_entities2
        # customize parameters (bootea_rotate.py:43-47)
        assert self.args.alignment_module == 'swapping'
        assert self.args.neg_triple_num > 0.0
        assert self.args.truncated_epsilon > 0.0

    def _define_variables(self):
        """bootea_rotate.py:49-57: three float64 variables, drawn in the reference's order."""
        a, n_ent, n_rel = self.args, self.kgs.entities_num, self.kgs.relations_num
        re = init_embeddings([n_ent, a.dim], 're_ent_embeds', a.init, a.ent_l2_norm)
        im = init_embeddings([n_ent, a.dim], 'im_ent_embeds', a.init, a.ent_l2_norm)
        rel = init_embeddings([n_rel, a.dim], 'rel_embeds', a.init, a.rel_l2_norm)
        dev = re.var.device
        self.ent_embeds = ComplexEntityTable(re.raw(), im.raw(), a.ent_l2_norm, dev)
        self.rel_embeds = PhaseTable(rel.raw(), a.rel_l2_norm, dev)

    @property
    def re_ent_embeds(self):
        return self.ent_embeds.parts()[0]

    @property
    def im_ent_embeds(self):
        return self.ent_embeds.parts()[1]

    def _define_embed_graph(self):
        """bootea_rotate.py:96-109: -sum log sigmoid(gamma - dist+) - sum log sigmoid(dist- - gamma), one optimiser."""
        self.triple_loss = dict(loss='rotate-logsigmoid', gamma=self.args.gamma)
        self._trainer = RotateTrainer(self.ent_embeds, self.rel_embeds, self.args, self.args.neg_triple_num,
                                      dist_group=self._dist_group())
        self.triple_optimizer = self._trainer.cfg

    def _define_alignment_graph(self):
        """bootea_rotate.py:148-158: the positive half alone, with its own optimiser instance."""
        self.alignment_loss = dict(loss='rotate-logsigmoid-positive', gamma=self.args.gamma)
        self._align_trainer = RotateTrainer(self.ent_embeds, self.rel_embeds, self.args, 0, dist_group=self._dist_group(),
                                            replicated=True)
        self.alignment_optimizer = self._align_trainer.cfg

    # ---- what evaluation reads (bootea_rotate.py:111-146) -----------------------------------------------------------
    def _lookup(self, ids):
        return self.ent_embeds.lookup(ids, sum_norm=False)

    def eval_kg1_useful_ent_embeddings(self):
        return self.ent_embeds.lookup(self.kgs.useful_entities_list1)[:, :self.args.dim].cpu().numpy()

    def eval_kg2_useful_ent_embeddings(self):
        return self.ent_embeds.lookup(self.kgs.useful_entities_list2)[:, :self.args.dim].cpu().numpy()

    def save(self):
        """bootea_rotate.py:142-146: sklearn-normalised re + im, the raw phases, no mapping matrix."""
        re, im = self.ent_embeds.parts()
        ent = re + im
        norms = np.sqrt((ent * ent).sum(1, keepdims=True))
        ent = ent / np.where(norms == 0, 1.0, norms)
        rd.save_embeddings(self.out_folder, self.kgs, ent, self.rel_embeds.eval(), None, mapping_mat=None)

    def eval_ref_sim_mat(self):
        """bootea_rotate.py:160-167: l2_normalize(lookup(re + im, ref)) on both sides, similarities on demand."""
        r1 = self.ent_embeds.lookup(self.ref_ent1, sum_norm=True)
        r2 = self.ent_embeds.lookup(self.ref_ent2, sum_norm=True)
        return PairSim(r1, r2, self.args.dim)

    # ---- training (bootea_rotate.py:169-203) -------------------------------------------------------------------------
    def launch_training_k_epo(self, iter, iter_nums, triple_steps, steps_tasks, training_batch_queue, neighbors1,
                              neighbors2):
        for i in range(1, iter_nums + 1):
            epoch = (iter - 1) * iter_nums + i
            self.launch_triple_training_1epo(epoch, triple_steps, steps_tasks, training_batch_queue, neighbors1,
                                             neighbors2)

    def train_alignment(self, kg1: KG, kg2: KG, entities1, entities2, training_epochs):
        if entities1 is None or len(entities1) == 0:
            return
        newly_tris1, newly_tris2 = generate_supervised_triples(kg1.rt_dict, kg1.hr_dict, kg2.rt_dict, kg2.hr_dict,
                                                               entities1, entities2)
        total = len(newly_tris1) + len(newly_tris2)
        if total == 0:
            return
        steps = max(math.ceil(total / self.args.batch_size), 1)
        dev = self.ent_embeds.var.device
        for _ in range(training_epochs):
            t1 = time.time()
            for step in range(steps):
                b1, b2 = generate_pos_batch(newly_tris1, newly_tris2, step, self.args.batch_size)
                batch = list(b1) + list(b2)
                if not batch:
                    continue
                self._align_trainer.step(ops.to_ids(np.asarray(batch, np.int32), dev), None)
            alignment_loss_v = self._align_trainer.pop_loss() / total
            print("alignment_loss = {:.3f}, time = {:.3f} s".format(alignment_loss_v, time.time() - t1))

    def run(self):
        """bootea_rotate.py:205-248."""
        t = time.time()
        triples_num = self.kgs.kg1.relation_triples_num + self.kgs.kg2.relation_triples_num
        triple_steps = int(math.ceil(triples_num / self.args.batch_size))
        steps_tasks = task_divide(list(range(triple_steps)), self.args.batch_threads_num)
        neighbors1, neighbors2 = None, None
        labeled_align = set()
        sub_num = self.args.sub_epoch
        iter_nums = self.args.max_epoch // sub_num
        for i in range(1, iter_nums + 1):
            print("\niteration", i)
            self.launch_training_k_epo(i, sub_num, triple_steps, steps_tasks, None, neighbors1, neighbors2)
            if i * sub_num >= self.args.start_valid:
                flag = self.valid(self.args.stop_metric)
                self.flag1, self.flag2, self.early_stop = early_stop(self.flag1, self.flag2, flag)
                if (self.early_stop and i >= self.args.min_iter) or i == iter_nums:
                    break
            if i * sub_num >= self.args.start_bp:
                print("bootstrapping")
                labeled_align, entities1, entities2 = bootstrapping(self.eval_ref_sim_mat(), self.ref_ent1, self.ref_ent2,
                                                                    labeled_align, self.args.sim_th, self.args.k)
                self.train_alignment(self.kgs.kg1, self.kgs.kg2, entities1, entities2, self.args.align_times)
                if i * sub_num >= self.args.start_valid:
                    self.valid(self.args.stop_metric)
            if self.args.neg_sampling == "truncated":
                t1 = time.time()
                assert 0.0 < self.args.truncated_epsilon < 1.0
                neighbors_num1 = int((1 - self.args.truncated_epsilon) * self.kgs.kg1.entities_num)
                neighbors_num2 = int((1 - self.args.truncated_epsilon) * self.kgs.kg2.entities_num)
                neighbors1 = refresh_neighbours(self.ent_embeds, self.kgs.useful_entities_list1, neighbors_num1)
                neighbors2 = refresh_neighbours(self.ent_embeds, self.kgs.useful_entities_list2, neighbors_num2)
                torch.cuda.synchronize()
                ent_num = len(self.kgs.kg1.entities_list) + len(self.kgs.kg2.entities_list)
                print("generating neighbors of {} entities costs {:.3f} s.".format(ent_num, time.time() - t1))
        if self._epochs is not None:
            self._epochs.check()
        print("Training ends. Total time = {:.3f} s.".format(time.time() - t))
