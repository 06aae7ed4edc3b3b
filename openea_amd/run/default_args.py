"""Hyper-parameter sets of the approaches on the hot path, as python dicts.

The values are the ones the reference ships in run/args/{mtranse,bootea,aligne,gcnalign,transh,transd,...}_args_{15K,100K}.json
(the `args_*` API: one attribute per key).  ``get_args(name, scale)`` returns an ``ARGs`` object that any
model accepts through ``set_args``; a reference JSON file loaded with ``load_args`` works the same way.
"""
from ..modules.args.args_hander import ARGs

_COMMON = dict(training_data="../../datasets/", output="../../output/results/", dataset_division="721_5fold",
               search_module="greedy", batch_threads_num=2, test_threads_num=4, ordered=True, start_valid=100,
               eval_freq=10, stop_metric="hits1", csls=10, top_k=[1, 5, 10, 50], is_save=True, max_epoch=2000)

_ARGS = {
    "MTransE": dict(embedding_module="MTransE", alignment_module="mapping", dim=100, init="unit", ent_l2_norm=True,
                    rel_l2_norm=True, loss_norm="L2", learning_rate=0.01, optimizer="Adagrad", batch_size=5000,
                    alpha=5, eval_metric="inner", eval_norm=True),
    "AlignE": dict(embedding_module="AlignE", alignment_module="swapping", dim=75, init="normal", ent_l2_norm=True,
                   rel_l2_norm=True, loss="limited", loss_norm="L2", learning_rate=0.01, optimizer="Adagrad",
                   batch_size=5000, pos_margin=0.01, neg_margin=2.0, neg_margin_balance=0.2, neg_sampling="truncated",
                   neg_triple_num=10, truncated_epsilon=0.9, truncated_freq=10, eval_metric="inner", eval_norm=False),
    "BootEA": dict(embedding_module="BootEA", alignment_module="swapping", dim=100, init="normal", ent_l2_norm=True,
                   rel_l2_norm=True, loss="limited", loss_norm="L2", learning_rate=0.01, optimizer="Adagrad",
                   batch_size=5000, pos_margin=0.01, neg_margin=2.0, neg_margin_balance=0.2, neg_sampling="truncated",
                   neg_triple_num=10, truncated_epsilon=0.9, truncated_freq=10, eval_metric="inner", eval_norm=False,
                   sim_th=0.7, k=10, likelihood_slice=10, sub_epoch=10),
    "BootEA_TransH": dict(embedding_module="BootEA_TransH", alignment_module="swapping", dim=100, init="normal",
                          ent_l2_norm=True, rel_l2_norm=True, loss="limited", loss_norm="L2", learning_rate=0.01,
                          optimizer="Adagrad", batch_size=5000, pos_margin=0.01, neg_margin=2.0, neg_margin_balance=0.2,
                          neg_sampling="truncated", neg_triple_num=10, truncated_epsilon=0.9, truncated_freq=10,
                          eval_metric="inner", eval_norm=False, sim_th=0.7, k=10, likelihood_slice=10, sub_epoch=10),
    "BootEA_RotatE": dict(embedding_module="BootEA_RotatE", alignment_module="swapping", dim=100, init="normal",
                          ent_l2_norm=True, rel_l2_norm=False, gamma=12.0, learning_rate=0.01, optimizer="Adam",
                          batch_size=5000, min_iter=40, neg_sampling="uniform", neg_triple_num=10, truncated_epsilon=0.9,
                          truncated_freq=10, batch_threads_num=4, start_valid=10, start_bp=5000, eval_metric="inner",
                          eval_norm=True, sim_th=0.75, k=10, sub_epoch=10, align_times=1),
    # run/args/trans{h,d}_args_*.json (TransE takes the same set: models/trans/transe.py:20-29 asserts it)
    **{name: dict(embedding_module=name, alignment_module="sharing", dim=100, init="normal", ent_l2_norm=True,
                  rel_l2_norm=True, loss="margin-based", loss_norm="L2", margin=1.5, neg_sampling="uniform",
                  neg_triple_num=1, learning_rate=0.01, optimizer="Adagrad", batch_size=5000, eval_metric="inner",
                  eval_norm=False) for name in ("TransE", "TransH", "TransD")},
    "GCN_Align": dict(embedding_module="GCN_Align", alignment_module="mapping", dim=100, neg_sampling="uniform",
                      neg_triple_num=5, learning_rate=8, batch_size=5000, test_threads_num=3, eval_metric="manhattan",
                      eval_norm=False, support_number=1, se_dim=100, ae_dim=100, hidden1=100, gamma=3,
                      early_stop=False, dropout=0, test_method="sa", beta=0.9),
    "RDGCN": dict(embedding_module="RDGCN", alignment_module="mapping", dim=300, neg_sampling="uniform",
                  neg_triple_num=125, learning_rate=0.002, batch_size=5000, test_threads_num=3, start_valid=30,
                  eval_metric="manhattan", eval_norm=False, gamma=1.0, dropout=0, beta=0.3, alpha=0.1),
    "AliNet": dict(embedding_module="AliNet", alignment_module="mapping", layer_dims=[500, 400, 300], init="xavier",
                   ent_l2_norm=True, rel_l2_norm=True, learning_rate=0.001, optimizer="Adam", batch_size=3000,
                   neg_margin=1.5, neg_margin_balance=0.1, dropout=0.0, neg_sampling="truncated", neg_triple_num=10,
                   truncated_epsilon=0.98, truncated_freq=10, start_valid=10, eval_metric="inner", is_save=False,
                   eval_norm=False, min_rel_win=50, start_augment=2, rel_param=0.01, num_features_nonzero=0,
                   sim_th=0.0, k=20),
}

# what changes at the 100K scale (run/args/*_100K.json)
_SCALE_100K = {
    "MTransE": dict(batch_size=20000),
    "AlignE": dict(batch_size=20000, truncated_epsilon=0.98),
    "BootEA": dict(batch_size=20000, truncated_epsilon=0.98),
    "BootEA_TransH": dict(batch_size=20000, truncated_epsilon=0.98),
    "TransE": dict(batch_size=20000),
    "TransH": dict(batch_size=20000),
    "TransD": dict(batch_size=20000),
    "GCN_Align": dict(batch_size=20000, learning_rate=25),
    "AliNet": dict(batch_size=20000, truncated_epsilon=0.995, min_rel_win=15),
    "RDGCN": dict(batch_size=20000, learning_rate=0.001, start_valid=50),
}


def get_args(name, scale="15K", **overrides):
    d = dict(_COMMON)
    d.update(_ARGS[name])
    if scale == "100K":
        d.update(_SCALE_100K.get(name, {}))
    d.update(overrides)
    return ARGs(d)
