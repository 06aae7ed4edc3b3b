"""BasicModel: the framework template and API boundary of the hot path.

Mirror of openea/models/basic_model.py (class BasicModel:26): same method names, same hook
points (``_define_*``, ``launch_*_1epo``, ``_eval_*_embeddings``), same ``args`` keys, same log
lines -- on top of the device-resident engine in ``models/trainer.py`` instead of a TF session.
What changes underneath:

* tables, optimiser state, positives, triple set and neighbour lists live in HBM; an epoch is
  enqueued with one C call (no producer processes, no queue, no feed_dict);
* validation / test never copy embeddings to the host and never build the N1 x N2 matrix;
* the truncated-neighbour refresh (basic_model.py:267-289) is an MFMA similarity strip + radix
  select on the device; its result stays there for the sampler.
"""
import math
import os
import time

import numpy as np
import torch

from .. import ops
from ..modules.base.initializers import init_embeddings, orthogonal_host
from ..modules.base.losses import get_loss_func
from ..modules.base.optimizers import generate_optimizer
from ..modules.finding.evaluation import early_stop, test, valid
from ..modules.finding.similarity import sim
from ..modules.load import read as rd
from ..modules.utils.util import generate_out_folder, task_divide
from .trainer import RelationTripleEpochs, TripleTrainer, refresh_neighbours


class _Lookup:
    """device [n, ld] rows + their logical dim (what `tf.nn.embedding_lookup(...).eval()` returned
    as a host array in the reference)."""

    def __init__(self, tensor, dim):
        self.tensor, self.dim = tensor, dim


class BasicModel:

    def set_kgs(self, kgs):
        self.kgs = kgs

    def set_args(self, args):
        self.args = args
        self.out_folder = generate_out_folder(self.args.output, self.args.training_data, self.args.dataset_division,
                                              self.__class__.__name__)

    def init(self):
        # need to be overwrite
        pass

    def __init__(self):
        self.out_folder = None
        self.args = None
        self.kgs = None
        self.session = None            # kept for API compatibility; there is no TF session
        self.rel_embeds = None
        self.ent_embeds = None
        self.mapping_mat = None        # device [d, d] fp32 (MTransE) or None
        self.eye_mat = None
        self.triple_optimizer = None
        self.triple_loss = None
        self.mapping_optimizer = None
        self.mapping_loss = None
        self.flag1 = -1
        self.flag2 = -1
        self.early_stop = False
        self._trainer = None           # TripleTrainer
        self._epochs = None            # RelationTripleEpochs
        self._seed = 0

    # ------------------------------------------------------------------------------------------
    # graph definition hooks (basic_model.py:73-104)
    # ------------------------------------------------------------------------------------------
    def _define_variables(self):
        self.ent_embeds = init_embeddings([self.kgs.entities_num, self.args.dim], 'ent_embeds',
                                          self.args.init, self.args.ent_l2_norm)
        self.rel_embeds = init_embeddings([self.kgs.relations_num, self.args.dim], 'rel_embeds',
                                          self.args.init, self.args.rel_l2_norm)

    @staticmethod
    def _dist_group():
        """data parallelism is on whenever the process was launched under torch.distributed with more
        than one rank (one process per GPU; models/dist.py): tables replicated, each rank scores its
        slice of every batch, evaluation / neighbour search row-sharded."""
        from . import dist as mdist
        import torch.distributed as tdist
        return tdist.group.WORLD if mdist.world()[1] > 1 else None

    def _dist_kw(self):
        """keyword arguments of the MAIN step's trainer: the process group + how the ranks exchange (args.dp_exchange:
        'step' -- the G-rank job equals the single-GPU job --, 'allreduce', or 'epoch' -- BASELINE.json north_star: local
        steps, one exchange per epoch; models/trainer.py:TripleTrainer)"""
        return dict(dist_group=self._dist_group(), exchange=getattr(self.args, 'dp_exchange', None))

    def _step_cfg(self, loss_cfg, neg_group_k, normal=None):
        """normal: EmbeddingTable of TransH normal vectors -> the step scores projected rows and trains it
        (its own Adagrad accumulator, created here: one per generate_optimizer call)."""
        cfg = generate_optimizer(loss_cfg, self.args.learning_rate, opt=self.args.optimizer)
        nv = na = None
        if normal is not None:
            if cfg['optimizer'] not in ('Adagrad', 'SGD'):
                # oea_triple_step's TransH score trains the normal vectors with Adagrad or SGD (apply_normal_rows)
                raise NotImplementedError("TransH scoring with optimizer=%s: the normal-vector table is trained with Adagrad "
                                          "(the shipped args files) or SGD" % cfg['optimizer'])
            nv = normal.var
            na = torch.full_like(nv, 0.1) if cfg['optimizer'] == 'Adagrad' else None
        return ops.make_step_cfg(ent_l2_norm=self.ent_embeds.is_l2_norm, rel_l2_norm=self.rel_embeds.is_l2_norm,
                                 neg_group_k=neg_group_k, normal=nv, normal_acc=na, **cfg), cfg['optimizer']

    def _define_embed_graph(self):
        """basic_model.py:80-98: lookups + get_loss_func + generate_optimizer."""
        self.triple_loss = get_loss_func(self.args)
        k = self.args.neg_triple_num if self.args.loss != 'margin-based' else 0
        cfg, opt = self._step_cfg(self.triple_loss, k)
        self.triple_optimizer = cfg
        self._trainer = TripleTrainer(self.ent_embeds, self.rel_embeds, cfg, opt, **self._dist_kw())

    def _define_mapping_variables(self):
        """mapping.py:22-25: orthogonal d x d matrix + identity."""
        d = self.args.dim
        dev = self.ent_embeds.var.device
        self.mapping_mat = torch.from_numpy(orthogonal_host(np.random.RandomState(self._seed + 17), (d, d))).to(dev)
        self.eye_mat = torch.eye(d, dtype=torch.float32, device=dev)
        self._mapping_acc = torch.full_like(self.mapping_mat, 0.1)       # Adagrad accumulator of M

    def _define_mapping_graph(self):
        self.mapping_loss = "alpha * (sum||e2 - e1 M||^2 + sum (M M^T - I)^2)"     # mapping.py:17, losses.py:76-80
        from ..modules.base.optimizers import get_optimizer
        self.mapping_optimizer = get_optimizer(self.args.optimizer, self.args.learning_rate)
        if self.mapping_optimizer['optimizer'] not in ('Adagrad', 'SGD'):
            # oea_mapping_step updates M with Adagrad or SGD; reject here instead of failing in the first mapping epoch
            raise NotImplementedError("mapping matrix with optimizer=%s: the fused mapping step (csrc/mapping.hip) implements "
                                      "Adagrad (the shipped mtranse_args_*.json) and SGD" % self.args.optimizer)
        # a second optimizer instance in the reference (mapping.py:18) = its own Adagrad accumulators
        cfg, opt = self._step_cfg(dict(loss='positive', loss_norm='L2'), 0)
        self._mapping_trainer = TripleTrainer(self.ent_embeds, self.rel_embeds, cfg, opt, dist_group=self._dist_group(),
                                              replicated=True)

    # ------------------------------------------------------------------------------------------
    # evaluation (basic_model.py:106-138)
    # ------------------------------------------------------------------------------------------
    def _lookup(self, ids):
        return self.ent_embeds.lookup(ids)

    def _eval_valid_embeddings(self):
        if len(self.kgs.valid_links) > 0:
            embeds1 = self._lookup(self.kgs.valid_entities1)
            embeds2 = self._lookup(self.kgs.valid_entities2 + self.kgs.test_entities2)
        else:
            embeds1 = self._lookup(self.kgs.test_entities1)
            embeds2 = self._lookup(self.kgs.test_entities2)
        return embeds1, embeds2, self.mapping_mat

    def _eval_test_embeddings(self):
        embeds1 = self._lookup(self.kgs.test_entities1)
        embeds2 = self._lookup(self.kgs.test_entities2)
        return embeds1, embeds2, self.mapping_mat

    def _apply_mapping(self, embeds1, mapping):
        """np.matmul(embeds1, mapping) of evaluation.py:11 on the device (plain library GEMM)."""
        if mapping is None:
            return embeds1
        d = self.args.dim
        out = torch.zeros_like(embeds1)
        out[:, :d] = embeds1[:, :d] @ mapping
        return out

    def _with_dim(self, t):
        t.oea_dim = self.args.dim
        return t

    def valid(self, stop_metric):
        embeds1, embeds2, mapping = self._eval_valid_embeddings()
        embeds1 = self._apply_mapping(embeds1, mapping)
        hits1_12, mrr_12 = valid(self._with_dim(embeds1), self._with_dim(embeds2), None, self.args.top_k,
                                 self.args.test_threads_num, metric=self.args.eval_metric,
                                 normalize=self.args.eval_norm, csls_k=0, accurate=False)
        return hits1_12 if stop_metric == 'hits1' else mrr_12

    def test(self, save=True):
        embeds1, embeds2, mapping = self._eval_test_embeddings()
        embeds1 = self._apply_mapping(embeds1, mapping)
        e1, e2 = self._with_dim(embeds1), self._with_dim(embeds2)
        rest_12, _, _ = test(e1, e2, None, self.args.top_k, self.args.test_threads_num,
                             metric=self.args.eval_metric, normalize=self.args.eval_norm, csls_k=0, accurate=True)
        test(e1, e2, None, self.args.top_k, self.args.test_threads_num,
             metric=self.args.eval_metric, normalize=self.args.eval_norm, csls_k=self.args.csls, accurate=True)
        if save:
            ent_ids_rest_12 = [(self.kgs.test_entities1[i], self.kgs.test_entities2[j]) for i, j in rest_12]
            rd.save_results(self.out_folder, ent_ids_rest_12)

    def retest(self):
        """basic_model.py:140-182: reload the saved embeddings of this run's parent folder and evaluate them both
        ways + the stable matching."""
        import os
        from ..modules.finding.alignment import stable_alignment
        parts = self.out_folder.split("/")
        new_dir = "".join(p + "/" for p in parts[:len(parts) - 2])
        new_dir = new_dir + sorted(os.listdir(new_dir))[0] + "/"
        embeds = np.load(new_dir + "ent_embeds.npy")
        embeds1, embeds2 = embeds[self.kgs.test_entities1], embeds[self.kgs.test_entities2]
        mapping = None
        print(self.__class__.__name__, type(self.__class__.__name__))
        if self.__class__.__name__ == "GCN_Align":
            print(self.__class__.__name__, "loads attr embeds")
            attr_embeds = np.load(new_dir + "attr_embeds.npy")
            embeds1 = np.concatenate([embeds1 * self.args.beta, attr_embeds[self.kgs.test_entities1] * (1.0 - self.args.beta)], axis=1)
            embeds2 = np.concatenate([embeds2 * self.args.beta, attr_embeds[self.kgs.test_entities2] * (1.0 - self.args.beta)], axis=1)
        if os.path.exists(new_dir + "mapping_mat.npy"):
            print(self.__class__.__name__, "loads mapping mat")
            mapping = np.load(new_dir + "mapping_mat.npy")
        kw = dict(metric=self.args.eval_metric, normalize=self.args.eval_norm, csls_k=0, accurate=True)
        print("conventional test:")
        test(embeds1, embeds2, mapping, self.args.top_k, self.args.test_threads_num, **kw)
        print("conventional reversed test:")
        if mapping is not None:
            embeds1 = np.matmul(embeds1, mapping)
        test(embeds2, embeds1, None, self.args.top_k, self.args.test_threads_num, **kw)
        print("stable test:")
        stable_alignment(embeds1, embeds2, self.args.eval_metric, self.args.eval_norm, csls_k=0,
                         nums_threads=self.args.test_threads_num)
        print("stable test with csls:")
        stable_alignment(embeds1, embeds2, self.args.eval_metric, self.args.eval_norm, csls_k=self.args.csls,
                         nums_threads=self.args.test_threads_num)

    def save(self):
        """basic_model.py:184-188: same files, same .npy payloads (the NORMALISED tensors, as
        `self.ent_embeds.eval()` returns them in the reference)."""
        ent_embeds = self.ent_embeds.eval()
        rel_embeds = self.rel_embeds.eval()
        mapping_mat = self.mapping_mat.cpu().numpy() if self.mapping_mat is not None else None
        rd.save_embeddings(self.out_folder, self.kgs, ent_embeds, rel_embeds, None, mapping_mat=mapping_mat)

    def eval_kg1_ent_embeddings(self):
        return self._lookup(self.kgs.kg1.entities_list)[:, :self.args.dim].cpu().numpy()

    def eval_kg2_ent_embeddings(self):
        return self._lookup(self.kgs.kg2.entities_list)[:, :self.args.dim].cpu().numpy()

    def eval_kg1_useful_ent_embeddings(self):
        return self._lookup(self.kgs.useful_entities_list1)[:, :self.args.dim].cpu().numpy()

    def eval_kg2_useful_ent_embeddings(self):
        return self._lookup(self.kgs.useful_entities_list2)[:, :self.args.dim].cpu().numpy()

    # ------------------------------------------------------------------------------------------
    # training (basic_model.py:206-290)
    # ------------------------------------------------------------------------------------------
    def _ensure_epochs(self, with_negatives=True):
        if self._epochs is None:
            k = self.args.neg_triple_num if with_negatives else 0
            from . import dist as mdist
            rank, world = mdist.world()
            self._epochs = RelationTripleEpochs(self.kgs, self.args.batch_size, k, seed=self._seed,
                                                dev=self.ent_embeds.var.device, rank=rank, world=world)
        return self._epochs

    def launch_training_1epo(self, epoch, triple_steps, steps_tasks, training_batch_queue, neighbors1, neighbors2):
        self.launch_triple_training_1epo(epoch, triple_steps, steps_tasks, training_batch_queue, neighbors1, neighbors2)
        if self.args.alignment_module == 'mapping':
            self.launch_mapping_training_1epo(epoch, triple_steps)

    def launch_triple_training_1epo(self, epoch, triple_steps, steps_tasks, batch_queue, neighbors1, neighbors2):
        """basic_model.py:211-236.  `steps_tasks` / `batch_queue` belonged to the host producers and
        are ignored; `neighbors1/2` are device neighbour tables (or None = uniform sampling)."""
        start = time.time()
        epochs = self._ensure_epochs(True)
        if (neighbors1 is not None) != (epochs.s1.nbr is not None) or (neighbors1 is not None and
                                                                       epochs.s1.nbr is not neighbors1):
            epochs.set_neighbours(neighbors1, neighbors2)
        trained_samples_num = epochs.run_epoch(self._trainer)
        if epochs.world > 1:                      # pop_loss sums the ranks' losses: divide by the whole job's positives
            trained_samples_num = int(epochs.batches.offsets[-1])
        epoch_loss = self._trainer.pop_loss() / max(trained_samples_num, 1)
        print('epoch {}, avg. triple loss: {:.4f}, cost time: {:.4f}s'.format(epoch, epoch_loss, time.time() - start))

    def launch_mapping_training_1epo(self, epoch, triple_steps):
        """basic_model.py:238-250 / mtranse.py:84-96: triple_steps steps on |train| // steps random seed links each
        (random.sample); Adagrad on the entity rows (through the normalisation) and on M.  One step = the fused
        oea_mapping_step (3 kernels) + the apply phase of the step engine; the epoch's batches go up in one copy."""
        start = time.time()
        dev = self.mapping_mat.device
        if getattr(self, "_train_links_dev", None) is None:
            self._train_links_dev = ops.to_ids(np.asarray(self.kgs.train_links, np.int32), dev)           # [L, 2]
        links = self._train_links_dev
        n_batch = links.shape[0] // triple_steps
        # random.sample per step = the first n_batch of a random permutation of the links; drawn for all steps at once on the
        # device (keys + row-wise argsort; same seed -> same draws on every rank).  On the host this was the epoch: numpy
        # permutes all L links per rng.choice(..., replace=False) call -- 2 ms of a 4.2 ms epoch at 15K, 13 ms at 100K
        gen = torch.Generator(device=dev)
        gen.manual_seed(self._seed + 1000 + epoch)
        picks = torch.rand((triple_steps, links.shape[0]), device=dev, generator=gen).argsort(dim=1)[:, :n_batch]
        batches = links[picks.reshape(-1)].reshape(triple_steps, n_batch, 2).permute(0, 2, 1).contiguous()   # [steps, 2, n_batch]
        t = self._mapping_trainer
        loss_dev = torch.zeros(1, dtype=torch.float64, device=dev)
        opt = self.mapping_optimizer['optimizer']
        if t.dist is None:            # single process: ONE C call enqueues the epoch's steps (oea_mapping_epoch)
            t.count_steps(triple_steps)
            self._mapping_work = ops.mapping_epoch(self.ent_embeds.var, t.ent_acc, self.rel_embeds.var, t.rel_acc, self.args.dim,
                                                   self.ent_embeds.is_l2_norm, batches.contiguous(), self.mapping_mat,
                                                   self._mapping_acc if opt == 'Adagrad' else None, float(self.args.alpha),
                                                   float(self.args.learning_rate), opt, t.cfg, t.ws, loss_dev, t.loss,
                                                   getattr(self, "_mapping_work", None))
        for s in range(triple_steps if t.dist is not None else 0):
            self._mapping_work = ops.mapping_step(self.ent_embeds.var, self.args.dim, self.ent_embeds.is_l2_norm,
                                                  batches[s, 0], batches[s, 1], self.mapping_mat,
                                                  self._mapping_acc if opt == 'Adagrad' else None, float(self.args.alpha),
                                                  float(self.args.learning_rate), opt, t.ws, self.ent_embeds.rows,
                                                  self.rel_embeds.rows, loss_dev, getattr(self, "_mapping_work", None))
            t.apply_scratch()
        trained_samples_num = n_batch * triple_steps
        epoch_loss = float(loss_dev.item()) / max(trained_samples_num, 1)
        print('epoch {}, avg. mapping loss: {:.4f}, cost time: {:.4f}s'.format(epoch, epoch_loss, time.time() - start))

    def _refresh_truncated_neighbours(self):
        """basic_model.py:267-289 (every truncated_freq epochs) / bootea.py:296-316."""
        t1 = time.time()
        assert 0.0 < self.args.truncated_epsilon < 1.0
        neighbors_num1 = int((1 - self.args.truncated_epsilon) * self.kgs.kg1.entities_num)
        neighbors_num2 = int((1 - self.args.truncated_epsilon) * self.kgs.kg2.entities_num)
        neighbors1 = refresh_neighbours(self.ent_embeds, self.kgs.useful_entities_list1, neighbors_num1)
        neighbors2 = refresh_neighbours(self.ent_embeds, self.kgs.useful_entities_list2, neighbors_num2)
        torch.cuda.synchronize()
        ent_num = len(self.kgs.kg1.entities_list) + len(self.kgs.kg2.entities_list)
        print("\ngenerating neighbors of {} entities costs {:.3f} s.".format(ent_num, time.time() - t1))
        return neighbors1, neighbors2

    def run(self):
        t = time.time()
        triples_num = self.kgs.kg1.relation_triples_num + self.kgs.kg2.relation_triples_num
        triple_steps = int(math.ceil(triples_num / self.args.batch_size))
        steps_tasks = task_divide(list(range(triple_steps)), self.args.batch_threads_num)
        training_batch_queue = None
        neighbors1, neighbors2 = None, None
        for i in range(1, self.args.max_epoch + 1):
            self.launch_training_1epo(i, triple_steps, steps_tasks, training_batch_queue, neighbors1, neighbors2)
            if i >= self.args.start_valid and i % self.args.eval_freq == 0:
                flag = self.valid(self.args.stop_metric)
                self.flag1, self.flag2, self.early_stop = early_stop(self.flag1, self.flag2, flag)
                if self.early_stop or i == self.args.max_epoch:
                    break
            if self.args.neg_sampling == 'truncated' and i % self.args.truncated_freq == 0:
                if neighbors1 is not None:
                    del neighbors1, neighbors2
                # (the reference calls gc.collect() here, basic_model.py:277: it frees the host neighbour dicts; the
                #  device tables are released by reference count, and a full collection over the KG dictionaries
                #  costs ~0.1 s -- a hundred epochs' worth of training at 15K)
                neighbors1, neighbors2 = self._refresh_truncated_neighbours()
        if self._epochs is not None:
            self._epochs.check()
        print("Training ends. Total time = {:.3f} s.".format(time.time() - t))

    # ------------------------------------------------------------------------------------------
    # prediction (basic_model.py:292-413)
    # ------------------------------------------------------------------------------------------
    def predict(self, top_k=1, min_sim_value=None, output_file_name=None):
        """basic_model.py:292-355: top-k matches in both directions and/or a similarity floor."""
        d = self.args.dim
        embeds1 = self._apply_mapping(self._lookup(self.kgs.kg1.entities_list), self.mapping_mat)
        embeds2 = self._lookup(self.kgs.kg2.entities_list)
        sim_mat = sim(embeds1[:, :d].cpu().numpy(), embeds2[:, :d].cpu().numpy(), metric=self.args.eval_metric,
                      normalize=self.args.eval_norm, csls_k=0)
        matched = set()
        if top_k:
            assert top_k > 0
            for i in range(sim_mat.shape[0]):
                for j in np.argpartition(-sim_mat[i, :], top_k)[:top_k]:
                    matched.add((i, int(j)))
            for j in range(sim_mat.shape[1]):
                for i in np.argpartition(-sim_mat[:, j], top_k)[:top_k]:
                    matched.add((int(i), j))
        elif min_sim_value:
            matched = set(map(tuple, np.argwhere(sim_mat > min_sim_value)))
        else:
            raise ValueError("Either top_k or min_sim_value should have a value")
        kg1_id_to_uri = {v: k for k, v in self.kgs.kg1.entities_id_dict.items()}
        kg2_id_to_uri = {v: k for k, v in self.kgs.kg2.entities_id_dict.items()}
        res = [(kg1_id_to_uri[self.kgs.kg1.entities_list[i]], kg2_id_to_uri[self.kgs.kg2.entities_list[j]],
                sim_mat[i, j]) for i, j in matched]
        if output_file_name is not None:
            os.makedirs(self.out_folder, exist_ok=True)
            with open(self.out_folder + output_file_name, 'w', encoding='utf8') as file:
                for entity1, entity2, confidence in res:
                    file.write(str(entity1) + "\t" + str(entity2) + "\t" + str(confidence) + "\n")
            print(self.out_folder + output_file_name, "saved")
        return res

    def predict_entities(self, entities_file_path, output_file_name=None):
        """basic_model.py:354-413: the similarity of the (entity1 \\t entity2) URI pairs listed in a tsv file, as
        [(uri1, uri2, confidence)] (and written to <out_folder><output_file_name> if given)."""
        ids1, ids2 = [], []
        with open(entities_file_path, 'r', encoding='utf-8') as fh:
            for line in fh:
                uri1, uri2 = line.strip('\n').split('\t')[:2]
                ids1.append(self.kgs.kg1.entities_id_dict[uri1])
                ids2.append(self.kgs.kg2.entities_id_dict[uri2])
        distinct1, distinct2 = sorted(set(ids1)), sorted(set(ids2))
        row = {e: i for i, e in enumerate(distinct1)}
        col = {e: i for i, e in enumerate(distinct2)}
        d = self.args.dim
        embeds1 = self._apply_mapping(self._lookup(distinct1), self.mapping_mat)
        embeds2 = self._lookup(distinct2)
        sim_mat = sim(embeds1[:, :d].cpu().numpy(), embeds2[:, :d].cpu().numpy(), metric=self.args.eval_metric,
                      normalize=self.args.eval_norm, csls_k=0)
        uri1 = {v: k for k, v in self.kgs.kg1.entities_id_dict.items()}
        uri2 = {v: k for k, v in self.kgs.kg2.entities_id_dict.items()}
        res = [(uri1[a], uri2[b], sim_mat[row[a], col[b]]) for a, b in zip(ids1, ids2)]
        if output_file_name is not None:
            os.makedirs(self.out_folder, exist_ok=True)
            with open(self.out_folder + output_file_name, 'w', encoding='utf8') as fh:
                for entity1, entity2, confidence in res:
                    fh.write(str(entity1) + "\t" + str(entity2) + "\t" + str(confidence) + "\n")
            print(self.out_folder + output_file_name, "saved")
        return res
