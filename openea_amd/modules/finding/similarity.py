"""Pairwise similarity + CSLS (mirror of openea/modules/finding/similarity.py).

``sim`` keeps the reference signature (numpy in, numpy N1 x N2 float32 out) for callers that
want the matrix; evaluation does NOT go through it (``alignment.greedy_alignment`` never
materialises the matrix).  The device-resident entry points take / return torch tensors.
"""
import numpy as np

from ... import ops


def device_metric(metric, normalize):
    """Map the reference's (metric, normalize) switch (similarity.py:34-51) onto a kernel metric
    and an extra row normalisation.  'cosine' without normalize is 1 - cdist(cosine) = the
    inner product of the L2-normalised rows."""
    if metric == 'inner':
        return 'inner', normalize
    if metric == 'cosine':
        return 'inner', True
    if metric == 'euclidean':
        return 'euclidean', normalize
    if metric == 'manhattan':
        return 'manhattan', normalize
    raise ValueError("unsupported metric %r (inner / cosine / euclidean / manhattan)" % (metric,))


def sim_device(t1, t2, dim, metric='inner', normalize=False, csls_k=0, inplace_ok=False):
    """device [n1, ld], [n2, ld] -> device [n1, n2] similarity (similarity.py:11-54)."""
    kmetric, norm = device_metric(metric, normalize)
    if norm:
        if not inplace_ok:
            t1, t2 = t1.clone(), t2.clone()
        ops.normalize_rows_(t1, dim, sklearn=True)
        ops.normalize_rows_(t2, dim, sklearn=True)
    s = ops.sim_matrix(t1, t2, dim, kmetric)
    if csls_k > 0:
        st = ops.sim_matrix(t2, t1, dim, kmetric)
        r = ops.row_topk_mean(s, csls_k)
        c = ops.row_topk_mean(st, csls_k)
        del st
        ops.csls_apply_(s, r, c)
    return s


def sim(embed1, embed2, metric='inner', normalize=False, csls_k=0):
    """similarity.py:11-54: returns the n1 x n2 float32 matrix on the host."""
    embed1 = np.asarray(embed1, dtype=np.float32)
    embed2 = np.asarray(embed2, dtype=np.float32)
    d = embed1.shape[1]
    s = sim_device(ops.to_table(embed1), ops.to_table(embed2), d, metric, normalize, csls_k, inplace_ok=True)
    return s.cpu().numpy()


def calculate_nearest_k(sim_mat, k):
    """similarity.py:80-83: mean of the k largest entries of each row."""
    s = ops.to_table(np.asarray(sim_mat, np.float32), ld=np.asarray(sim_mat).shape[1])
    return ops.row_topk_mean(s, k).cpu().numpy()


def csls_sim(sim_mat, k):
    """similarity.py:57-77."""
    sim_mat = np.ascontiguousarray(sim_mat, dtype=np.float32)
    n2 = sim_mat.shape[1]
    import torch
    s = torch.from_numpy(sim_mat).to(ops.device())
    st = s.t().contiguous()
    r = ops.row_topk_mean(s, k)
    c = ops.row_topk_mean(st, k)
    ops.csls_apply_(s, r, c)
    return s.cpu().numpy()


def csls_means_device(t1, t2, dim, kmetric, k, max_bytes=4 << 30, cols=True, return_grid=False):
    """Per-row and per-column top-k means WITHOUT holding the whole matrix: strips of rows of S
    and of S^T are produced and reduced one after the other (each strip <= max_bytes).
    cols=False: only the row means (second result None).  return_grid=True: -> (r, c, grid) where grid is the L1Grid of a
    manhattan evaluation (its kept strips are what the rank pass reads next) or None -- handed over explicitly: it used to
    travel in a function attribute, which pinned up to a third of the free memory after a direct call and could hand a stale
    quantisation to a later evaluation."""
    import os
    import torch
    if kmetric == 'inner':                 # one sweep, no strips of S / S^T (n1, n2 >= 4096)
        rc = ops.csls_means(t1, t2, dim, k)
        if rc is not None:                 # (cols=False -- a rank's block of rows in the sharded evaluation -- drops the column means:
            rc = rc if cols else (rc[0], None)     #  the sweep that finds both is still several times faster than the strips below)
            return (rc[0], rc[1], None) if return_grid else rc
    if (kmetric == 'manhattan' and min(t1.shape[0], t2.shape[0]) >= 2048 and k + 32 < min(t1.shape[0], t2.shape[0])
            and os.environ.get('OEA_L1_EVAL', 'grid') != 'f64'):
        # 16-bit grid distances + exact similarities of the k + margin nearest (certified): the same means without the fp64
        # distance of every pair, twice (ops.l1_grid_topk_means); the grid and the query strips go on to the rank pass
        r, c, grid = ops.csls_means_l1_grid(t1, t2, dim, k, cols=cols)
        return (r, c, grid) if return_grid else (r, c)

    def strip_means(a, b):
        n, m = a.shape[0], b.shape[0]
        rows_per = max(128, int(max_bytes // (4 * m)) // 128 * 128)
        out = torch.empty(n, dtype=torch.float32, device=a.device)
        for r0 in range(0, n, rows_per):
            s = ops.sim_matrix(a[r0:r0 + rows_per], b, dim, kmetric)
            out[r0:r0 + rows_per] = ops.row_topk_mean(s, k)
            del s
        return out
    r, c = strip_means(t1, t2), (strip_means(t2, t1) if cols else None)
    return (r, c, None) if return_grid else (r, c)


# ---- the reference's thread- / block-parallel spellings of the same products (similarity.py:86-127) ------------------
def sim_multi_threads(embeds1, embeds2, threads_num=16):
    """np.dot of row blocks in a process pool in the reference (similarity.py:105-116): one device call here."""
    return sim(embeds1, embeds2, metric='inner', normalize=False, csls_k=0)


def sim_multi_blocks(embeds1, embeds2, blocks_num=16):
    """similarity.py:119-127."""
    return sim(embeds1, embeds2, metric='inner', normalize=False, csls_k=0)


def csls_sim_multi_threads(sim_mat, k, nums_threads):
    """similarity.py:86-102: calculate_nearest_k over row blocks = calculate_nearest_k."""
    return calculate_nearest_k(sim_mat, k)
