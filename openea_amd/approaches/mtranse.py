"""MTransE (mirror of openea/approaches/mtranse.py:17-113): positive-only squared-L2 translational
loss + a d x d linear mapping trained on the seed links; BASELINE.json config 1."""
import math
import time

from ..models.basic_model import BasicModel
from ..models.trainer import TripleTrainer
from ..modules.base.losses import positive_loss
from ..modules.finding.evaluation import early_stop
from ..modules.utils.util import task_divide


class MTransE(BasicModel):

    def init(self):
        self._define_variables()
        self._define_mapping_variables()
        self._define_embed_graph()
        self._define_mapping_graph()
        # customize parameters (mtranse.py:30-37)
        assert self.args.init == 'unit'
        assert self.args.alignment_module == 'mapping'
        assert self.args.optimizer == 'Adagrad'
        assert self.args.eval_metric == 'inner'
        assert self.args.ent_l2_norm is True
        assert self.args.alpha > 1

    def _define_embed_graph(self):
        """mtranse.py:46-57: positive_loss(phs, prs, pts, 'L2') + Adagrad."""
        self.triple_loss = positive_loss('L2')
        cfg, opt = self._step_cfg(self.triple_loss, 0)
        self.triple_optimizer = cfg
        self._trainer = TripleTrainer(self.ent_embeds, self.rel_embeds, cfg, opt, **self._dist_kw())

    def launch_training_1epo(self, epoch, triple_steps, steps_tasks, training_batch_queue, neighbors1, neighbors2):
        self.launch_triple_training_1epo(epoch, triple_steps, steps_tasks, training_batch_queue, neighbors1, neighbors2)
        self.launch_mapping_training_1epo(epoch, triple_steps)

    def launch_triple_training_1epo(self, epoch, triple_steps, steps_tasks, batch_queue, neighbors1, neighbors2):
        """mtranse.py:63-82: positive batches only (generate_pos_batch_queue)."""
        start = time.time()
        epochs = self._ensure_epochs(with_negatives=False)
        trained_samples_num = epochs.run_epoch(self._trainer)
        epoch_loss = self._trainer.pop_loss() / max(trained_samples_num, 1)
        print('epoch {}, avg. triple loss: {:.4f}, cost time: {:.4f}s'.format(epoch, epoch_loss, time.time() - start))

    def run(self):
        """mtranse.py:98-112."""
        t = time.time()
        triples_num = self.kgs.kg1.relation_triples_num + self.kgs.kg2.relation_triples_num
        triple_steps = int(math.ceil(triples_num / self.args.batch_size))
        steps_tasks = task_divide(list(range(triple_steps)), self.args.batch_threads_num)
        for i in range(1, self.args.max_epoch + 1):
            self.launch_training_1epo(i, triple_steps, steps_tasks, None, None, None)
            if i >= self.args.start_valid and i % self.args.eval_freq == 0:
                flag = self.valid(self.args.stop_metric)
                self.flag1, self.flag2, self.early_stop = early_stop(self.flag1, self.flag2, flag)
                if self.early_stop or i == self.args.max_epoch:
                    break
        print("Training ends. Total time = {:.3f} s.".format(time.time() - t))
