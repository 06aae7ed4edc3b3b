"""TransD (openea/models/trans/transd.py:9-57): every entity and relation has a second, "transfer" vector; a row is
projected with its own transfer vector and the relation's before the translation,
    e' = l2_normalize(e + (e . e_transfer) r_transfer)                                       (transd.py:56-57)
and the loss of get_loss_func is taken on the projected rows (transd.py:49-53).

Layout: the four variables of transd.py:16-24 live in TWO device tables -- rows [0, E) of `ent_embeds` are the entity
embeddings and rows [E, 2E) the entity transfer vectors, likewise [0, R) / [R, 2R) of `rel_embeds` -- because each pair
shares its l2_norm flag and its optimiser: the gradient scratch, the data-parallel exchange and the apply kernel of the
fused step then need no second code path (OEA_SCORE_TRANSD, include/openea_hip.h).  Lookups, evaluation and save() see
the first E / R rows only (EmbeddingTable.visible_rows)."""
import numpy as np

from ...modules.base.initializers import init_embeddings
from ...modules.base.losses import get_loss_func
from ..trainer import EmbeddingTable, TripleTrainer
from .transe import TransE


def _stacked(first, second, name):
    table = EmbeddingTable(np.concatenate([first.raw(), second.raw()]), first.is_l2_norm, name, dev=first.var.device,
                           visible_rows=first.rows)
    return table


class TransD(TransE):

    def _define_variables(self):
        """transd.py:14-24: the same four init_embeddings calls, in the same order, then stacked pairwise."""
        a, n_ent, n_rel = self.args, self.kgs.entities_num, self.kgs.relations_num
        ent = init_embeddings([n_ent, a.dim], 'ent_embeds', a.init, a.ent_l2_norm)
        rel = init_embeddings([n_rel, a.dim], 'rel_embeds', a.init, a.rel_l2_norm)
        ent_transfer = init_embeddings([n_ent, a.dim], 'ent_transfer', a.init, a.ent_l2_norm)
        rel_transfer = init_embeddings([n_rel, a.dim], 'rel_transfer', a.init, a.rel_l2_norm)
        self.ent_embeds = _stacked(ent, ent_transfer, 'ent_embeds')
        self.rel_embeds = _stacked(rel, rel_transfer, 'rel_embeds')

    @property
    def ent_transfer(self):
        """host [E, dim]: the (normalised) entity transfer vectors, as `self.ent_transfer.eval()` gave them."""
        e = self.ent_embeds
        return e.lookup(np.arange(e.visible_rows, e.rows, dtype=np.int32))[:, :e.dim].cpu().numpy()

    @property
    def rel_transfer(self):
        r = self.rel_embeds
        return r.lookup(np.arange(r.visible_rows, r.rows, dtype=np.int32))[:, :r.dim].cpu().numpy()

    def _define_embed_graph(self):
        """transd.py:26-53."""
        from ... import ops
        from ...modules.base.optimizers import generate_optimizer
        self.triple_loss = get_loss_func(self.args)
        merged = generate_optimizer(self.triple_loss, self.args.learning_rate, opt=self.args.optimizer)
        cfg = ops.make_step_cfg(ent_l2_norm=self.ent_embeds.is_l2_norm, rel_l2_norm=self.rel_embeds.is_l2_norm,
                                neg_group_k=0, transfer_bases=(self.ent_embeds.visible_rows, self.rel_embeds.visible_rows),
                                **merged)
        self.triple_optimizer = cfg
        self._trainer = TripleTrainer(self.ent_embeds, self.rel_embeds, cfg, merged['optimizer'], **self._dist_kw())
