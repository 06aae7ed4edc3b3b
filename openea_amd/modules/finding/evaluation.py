"""valid / test / early_stop with the signatures of openea/modules/finding/evaluation.py:6-33; both wrappers are one
call of greedy_alignment on (embeds1 [x mapping], embeds2)."""
import numpy as np

from .alignment import greedy_alignment


def _aligned(embeds1, embeds2, mapping, top_k, threads_num, metric, normalize, csls_k, accurate):
    """-> (alignment_rest_12, hits1_12, mr_12, mrr_12); a mapping matrix is applied on the host (evaluation.py:11,22:
    an n x d by d x d product), which needs host arrays."""
    if mapping is not None:
        if hasattr(embeds1, "is_cuda"):
            raise TypeError("pass host arrays when a mapping matrix is given")
        embeds1 = np.matmul(embeds1, mapping)
    return greedy_alignment(embeds1, embeds2, top_k, threads_num, metric, normalize, csls_k, accurate)


def valid(embeds1, embeds2, mapping, top_k, threads_num, metric='inner', normalize=False, csls_k=0, accurate=False):
    """evaluation.py:6-14 -> (hits@1, mrr)."""
    result = _aligned(embeds1, embeds2, mapping, top_k, threads_num, metric, normalize, csls_k, accurate)
    return result[1], result[3]


def test(embeds1, embeds2, mapping, top_k, threads_num, metric='inner', normalize=False, csls_k=0, accurate=True):
    """evaluation.py:17-25 -> (alignment pairs, hits@1, mrr)."""
    result = _aligned(embeds1, embeds2, mapping, top_k, threads_num, metric, normalize, csls_k, accurate)
    return result[0], result[1], result[3]


def early_stop(flag1, flag2, flag):
    """evaluation.py:28-33: stop when the metric fell twice in a row; -> (flag2, flag, stop)."""
    stop = flag <= flag2 <= flag1
    if stop:
        print("\n == should early stop == \n")
    return flag2, flag, stop
