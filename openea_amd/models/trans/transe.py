"""TransE (openea/models/trans/transe.py:9-49): shared ids for the seed pairs, margin loss on (pos i, neg i) pairs
with one uniform negative per positive, Adagrad -- BasicModel's graph as it stands, plus the argument contract."""
from ..basic_model import BasicModel


class TransE(BasicModel):

    def init(self):
        self._define_variables()
        self._define_embed_graph()
        self._check_args()

    def _check_args(self):
        """transe.py:20-29."""
        a = self.args
        required = dict(init='normal', alignment_module='sharing', loss='margin-based', neg_sampling='uniform',
                        optimizer='Adagrad', eval_metric='inner', loss_norm='L2', ent_l2_norm=True, rel_l2_norm=True,
                        neg_triple_num=1)
        for key, value in required.items():
            assert getattr(a, key) == value, "%s: %s must be %r" % (type(self).__name__, key, value)
