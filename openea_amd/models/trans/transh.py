"""TransH (openea/models/trans/transh.py:9-51): TransE whose entity rows are projected onto the hyperplane of the
triple's relation, e' = e - (e.n) n with n = l2_normalize(normal_vector)[r] normalised again at the lookup; margin loss
on the projected rows; the normal vectors are a third trained table.  Device side: OEA_SCORE_TRANSH of the fused step
with margin pairs (csrc/triple_step.hip: triple_projected)."""
from ...modules.base.initializers import init_embeddings
from ...modules.base.losses import margin_loss
from ..trainer import TripleTrainer
from .transe import TransE


class TransH(TransE):

    def _define_variables(self):
        """transh.py:14-21."""
        super()._define_variables()
        self.normal_vector = init_embeddings([self.kgs.relations_num, self.args.dim], 'normal_vector', self.args.init, True)

    def _define_embed_graph(self):
        """transh.py:23-46: margin_loss(projected rows) + one optimiser over the three tables."""
        self.triple_loss = margin_loss(self.args.margin, self.args.loss_norm)
        cfg, opt = self._step_cfg(self.triple_loss, 0, normal=self.normal_vector)
        self.triple_optimizer = cfg
        self._trainer = TripleTrainer(self.ent_embeds, self.rel_embeds, cfg, opt, **self._dist_kw())
