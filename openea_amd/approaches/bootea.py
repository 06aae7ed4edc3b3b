"""BootEA (mirror of openea/approaches/bootea.py): AlignE + bootstrapping of likely alignment
+ the alignment loss -sum log sigmoid(-||h + r - t||^2) on the triples swapped through the newly
labelled pairs; BASELINE.json config 2."""
import math
import time

import numpy as np

from .. import ops
from ..models.trainer import TripleTrainer
from ..modules.base.losses import alignment_loss
from ..modules.bootstrapping.alignment_finder import PairSim, check_new_alignment, find_potential_alignment_mwgm
from ..modules.finding.evaluation import early_stop
from ..modules.load.kg import KG
from ..modules.utils.util import task_divide
from .aligne import AlignE


def bootstrapping(sim_mat, unaligned_entities1, unaligned_entities2, labeled_alignment, sim_th, k):
    """bootea.py:19-32."""
    curr_labeled_alignment = find_potential_alignment_mwgm(sim_mat, sim_th, k)
    if curr_labeled_alignment is not None:
        labeled_alignment = update_labeled_alignment_x(labeled_alignment, curr_labeled_alignment, sim_mat)
        labeled_alignment = update_labeled_alignment_y(labeled_alignment, sim_mat)
    if labeled_alignment is not None:
        newly_aligned_entities1 = [unaligned_entities1[pair[0]] for pair in labeled_alignment]
        newly_aligned_entities2 = [unaligned_entities2[pair[1]] for pair in labeled_alignment]
    else:
        newly_aligned_entities1, newly_aligned_entities2 = None, None
    return labeled_alignment, newly_aligned_entities1, newly_aligned_entities2


def update_labeled_alignment_x(pre_labeled_alignment, curr_labeled_alignment, sim_mat):
    """bootea.py:35-54: keep, per left entity, the partner with the larger similarity."""
    labeled = dict(pre_labeled_alignment)
    n1 = n2 = 0
    for i, j in curr_labeled_alignment:
        if labeled.get(i, -1) == i and j != i:
            n2 += 1
        if i in labeled:
            pre_j = labeled[i]
            if sim_mat[i, j] >= sim_mat[i, pre_j]:
                if pre_j == i and j != i:
                    n1 += 1
                labeled[i] = j
        else:
            labeled[i] = j
    print("update wrongly: ", n1, "greedy update wrongly: ", n2)
    out = set(labeled.items())
    check_new_alignment(out, context="after editing (<-)")
    return out


def update_labeled_alignment_y(labeled_alignment, sim_mat):
    """bootea.py:57-77: keep, per right entity, the left entity with the largest similarity."""
    by_j = {}
    for i, j in labeled_alignment:
        by_j.setdefault(j, set()).add(i)
    updated = set()
    for j, i_set in by_j.items():
        if len(i_set) == 1:
            updated.add((next(iter(i_set)), j))
        else:
            max_i, max_sim = -1, -10
            for i in i_set:
                if sim_mat[i, j] > max_sim:
                    max_sim, max_i = sim_mat[i, j], i
            updated.add((max_i, j))
    check_new_alignment(updated, context="after editing (->)")
    return updated


def generate_newly_triples(ent1, ent2, rt_dict1, hr_dict1):
    """bootea.py:117-123."""
    out = [(ent2, r, t) for r, t in rt_dict1.get(ent1, set())]
    out += [(h, r, ent2) for h, r in hr_dict1.get(ent1, set())]
    return out


def generate_supervised_triples(rt_dict1, hr_dict1, rt_dict2, hr_dict2, ents1, ents2):
    """bootea.py:107-114."""
    assert len(ents1) == len(ents2)
    t1, t2 = [], []
    for e1, e2 in zip(ents1, ents2):
        t1.extend(generate_newly_triples(e1, e2, rt_dict1, hr_dict1))
        t2.extend(generate_newly_triples(e2, e1, rt_dict2, hr_dict2))
    print("newly triples: {}, {}".format(len(t1), len(t2)))
    return t1, t2


def generate_pos_batch(triples1, triples2, step, batch_size):
    """bootea.py:126-138."""
    num1 = int(len(triples1) / (len(triples1) + len(triples2)) * batch_size)
    num2 = batch_size - num1
    return triples1[step * num1: min(step * num1 + num1, len(triples1))], \
        triples2[step * num2: min(step * num2 + num2, len(triples2))]


class BootEA(AlignE):

    def __init__(self):
        super().__init__()
        self.ref_ent1 = None
        self.ref_ent2 = None

    def init(self):
        self._define_variables()
        self._define_embed_graph()
        self._define_alignment_graph()
        self.ref_ent1 = self.kgs.valid_entities1 + self.kgs.test_entities1
        self.ref_ent2 = self.kgs.valid_entities2 + self.kgs.test_entities2
        self._check_args()

    def _define_alignment_graph(self):
        """bootea.py:190-199: its own AdagradOptimizer instance -> its own accumulators."""
        self.alignment_loss = alignment_loss()
        cfg, opt = self._step_cfg(self.alignment_loss, 0)
        self.alignment_optimizer = cfg
        self._align_trainer = TripleTrainer(self.ent_embeds, self.rel_embeds, cfg, opt, dist_group=self._dist_group(),
                                            replicated=True)

    def eval_ref_sim_mat(self):
        """bootea.py:214-219: l2_normalize(lookup(ref1)) . l2_normalize(lookup(ref2))^T, on demand."""
        d = self.args.dim
        r1 = self.ent_embeds.lookup(self.ref_ent1)
        r2 = self.ent_embeds.lookup(self.ref_ent2)
        ops.normalize_rows_(r1, d, sklearn=False)
        ops.normalize_rows_(r2, d, sklearn=False)
        return PairSim(r1, r2, d)

    def launch_training_k_epo(self, iter, iter_nums, triple_steps, steps_tasks, training_batch_queue, neighbors1,
                              neighbors2):
        for i in range(1, iter_nums + 1):
            epoch = (iter - 1) * iter_nums + i
            self.launch_triple_training_1epo(epoch, triple_steps, steps_tasks, training_batch_queue, neighbors1,
                                             neighbors2)

    def train_alignment(self, kg1: KG, kg2: KG, entities1, entities2, training_epochs):
        """bootea.py:228-249."""
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
        """bootea.py:269-317."""
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
                if self.early_stop or i == iter_nums:
                    break
            labeled_align, entities1, entities2 = bootstrapping(self.eval_ref_sim_mat(), self.ref_ent1, self.ref_ent2,
                                                                labeled_align, self.args.sim_th, self.args.k)
            self.train_alignment(self.kgs.kg1, self.kgs.kg2, entities1, entities2, 1)
            if i * sub_num >= self.args.start_valid:
                self.valid(self.args.stop_metric)
            neighbors1, neighbors2 = self._refresh_truncated_neighbours()
        if self._epochs is not None:
            self._epochs.check()
        print("Training ends. Total time = {:.3f} s.".format(time.time() - t))
