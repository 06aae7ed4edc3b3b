"""AlignE (mirror of openea/approaches/aligne.py:10-66): limited loss + truncated negative sampling
+ parameter swapping (the seed-swapped triples come from KGs(mode='swapping'))."""
from ..models.basic_model import BasicModel
from ..models.trainer import TripleTrainer
from ..modules.base.losses import limited_loss


class AlignE(BasicModel):

    def init(self):
        self._define_variables()
        self._define_embed_graph()
        self._check_args()

    def _check_args(self):
        # customize parameters (aligne.py:21-38)
        assert self.args.init == 'normal'
        assert self.args.alignment_module == 'swapping'
        assert self.args.loss == 'limited'
        assert self.args.neg_sampling == 'truncated'
        assert self.args.optimizer == 'Adagrad'
        assert self.args.eval_metric == 'inner'
        assert self.args.loss_norm == 'L2'
        assert self.args.ent_l2_norm is True
        assert self.args.rel_l2_norm is True
        assert self.args.pos_margin >= 0.0
        assert self.args.neg_margin > self.args.pos_margin
        assert self.args.neg_triple_num > 1
        assert self.args.truncated_epsilon > 0.0
        assert self.args.learning_rate >= 0.01

    def _define_embed_graph(self):
        """aligne.py:47-66: limited_loss(..., balance=args.neg_margin_balance) + Adagrad."""
        self.triple_loss = limited_loss(self.args.pos_margin, self.args.neg_margin, self.args.loss_norm,
                                        balance=self.args.neg_margin_balance)
        cfg, opt = self._step_cfg(self.triple_loss, self.args.neg_triple_num)
        self.triple_optimizer = cfg
        self._trainer = TripleTrainer(self.ent_embeds, self.rel_embeds, cfg, opt, **self._dist_kw())
