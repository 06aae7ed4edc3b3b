"""BootEA_TransH (mirror of openea/approaches/bootea_transh.py): BootEA whose relation triples are scored
TransH-style -- h and t projected onto the hyperplane of the relation, h' = h - (h.n) n with
n = l2_normalize(l2_normalize(normal_vector)[r]) (bootea_transh.py:58-60,62-96).  The alignment loss keeps the
plain translation (bootea_transh.py:98-105), the bootstrapping / run loop is BootEA's (bootea_transh.py:186-229).
Device side: OEA_SCORE_TRANSH of the fused step (csrc/triple_step.hip: triple_transh_grouped, apply_normal_rows).
"""
from ..models.trainer import TripleTrainer
from ..modules.base.initializers import init_embeddings
from ..modules.base.losses import limited_loss
from .bootea import BootEA


class BootEA_TransH(BootEA):

    def _check_args(self):
        super()._check_args()
        assert self.args.loss_norm == 'L2'                       # bootea_transh.py:47

    def _define_variables(self):
        """bootea_transh.py:62-69."""
        super()._define_variables()
        self.normal_vector = init_embeddings([self.kgs.relations_num, self.args.dim], 'normal_vector', self.args.init, True)

    def _define_embed_graph(self):
        """bootea_transh.py:71-96: limited loss on the projected rows + Adagrad over all three tables."""
        self.triple_loss = limited_loss(self.args.pos_margin, self.args.neg_margin, self.args.loss_norm,
                                        balance=self.args.neg_margin_balance)
        cfg, opt = self._step_cfg(self.triple_loss, self.args.neg_triple_num, normal=self.normal_vector)
        self.triple_optimizer = cfg
        self._trainer = TripleTrainer(self.ent_embeds, self.rel_embeds, cfg, opt, **self._dist_kw())
