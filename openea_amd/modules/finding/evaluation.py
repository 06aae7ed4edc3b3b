"""valid / test / early_stop (mirror of openea/modules/finding/evaluation.py:6-33)."""
import numpy as np

from .alignment import greedy_alignment


def _map(embeds1, mapping):
    if mapping is None:
        return embeds1
    if hasattr(embeds1, "is_cuda"):
        raise TypeError("pass host arrays when a mapping matrix is given")
    return np.matmul(embeds1, mapping)      # evaluation.py:11,22 (n x d times d x d, host)


def valid(embeds1, embeds2, mapping, top_k, threads_num, metric='inner', normalize=False, csls_k=0, accurate=False):
    _, hits1_12, mr_12, mrr_12 = greedy_alignment(_map(embeds1, mapping), embeds2, top_k, threads_num, metric,
                                                  normalize, csls_k, accurate)
    return hits1_12, mrr_12


def test(embeds1, embeds2, mapping, top_k, threads_num, metric='inner', normalize=False, csls_k=0, accurate=True):
    alignment_rest_12, hits1_12, mr_12, mrr_12 = greedy_alignment(_map(embeds1, mapping), embeds2, top_k, threads_num,
                                                                  metric, normalize, csls_k, accurate)
    return alignment_rest_12, hits1_12, mrr_12


def early_stop(flag1, flag2, flag):
    if flag <= flag2 <= flag1:
        print("\n == should early stop == \n")
        return flag2, flag, True
    return flag2, flag, False
