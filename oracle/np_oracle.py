"""numpy restatement of the reference's host-side hot-path functions.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function cites the reference
lines (relative to /root/reference/src/openea/) it follows.  Heavy loops are delegated to
oracle/c/oracle.c through oracle.cport.
"""
import math

import numpy as np

from . import cport

# ----------------------------------------------------------------------------------------
# modules/utils/util.py
# ----------------------------------------------------------------------------------------


def task_divide(idx, n):
    """util.py:16-30."""
    total = len(idx)
    if n <= 0 or 0 == total:
        return [idx]
    if n > total:
        return [idx]
    elif n == total:
        return [[i] for i in idx]
    j = total // n
    tasks = [idx[i:i + j] for i in range(0, (n - 1) * j, j)]
    tasks.append(idx[(n - 1) * j:])
    return tasks


# ----------------------------------------------------------------------------------------
# modules/train/batch.py -- positive batching
# ----------------------------------------------------------------------------------------


def generate_pos_triples(triples, batch_size, step, is_fixed_size=False):
    """batch.py:48-57."""
    start = step * batch_size
    end = min(start + batch_size, len(triples))
    pos_batch = triples[start:end]
    if is_fixed_size and len(pos_batch) < batch_size:
        pos_batch = pos_batch + triples[:batch_size - len(pos_batch)]
    return pos_batch


def batch_sizes(n1, n2, batch_size):
    """batch.py:18-19 / 39-40: b1 = int(n1/(n1+n2)*B), b2 = B - b1 (python float arithmetic)."""
    b1 = int(n1 / (n1 + n2) * batch_size)
    return b1, batch_size - b1


def generate_pos_batch(triple_list1, triple_list2, batch_size, step):
    """batch.py:17-22."""
    b1, b2 = batch_sizes(len(triple_list1), len(triple_list2), batch_size)
    return generate_pos_triples(triple_list1, b1, step) + generate_pos_triples(triple_list2, b2, step)


# ----------------------------------------------------------------------------------------
# modules/train/batch.py -- neighbour search
# ----------------------------------------------------------------------------------------


def find_neighbours(frags, entity_list, sub_embed, embed, k):
    """batch.py:157-165.  Returns {entity -> sorted list of k neighbour entity ids}.

    The reference returns an UNORDERED k-set per entity (argpartition); ties at the k-th
    value are arbitrary there.  Restatement: (value desc, column asc) selection, listed in
    ascending column order."""
    entity_list = np.asarray(entity_list)
    idx = cport.topk_inner(sub_embed, embed, k)
    return {frags[i]: entity_list[idx[i]].tolist() for i in range(len(frags))}


def generate_neighbours(entity_embeds, entity_list, neighbors_num, threads_num):
    """batch.py:122-154 (both variants return the same dict)."""
    ent_frags = task_divide(np.array(entity_list), threads_num)
    ent_frag_indexes = task_divide(np.array(range(len(entity_list))), threads_num)
    dic = dict()
    for i in range(len(ent_frags)):
        dic.update(find_neighbours(ent_frags[i], np.array(entity_list),
                                   entity_embeds[ent_frag_indexes[i], :], entity_embeds,
                                   neighbors_num))
    return dic


# ----------------------------------------------------------------------------------------
# modules/finding/similarity.py
# ----------------------------------------------------------------------------------------


def normalize_rows(x):
    """sklearn.preprocessing.normalize (similarity.py:32-33)."""
    return cport.l2_normalize_rows(x)


def sim(embed1, embed2, metric='inner', normalize=False, csls_k=0):
    """similarity.py:11-54."""
    if normalize:
        embed1, embed2 = normalize_rows(embed1), normalize_rows(embed2)
    if metric == 'inner' or (metric == 'cosine' and normalize):
        sim_mat = cport.sim_matrix(embed1, embed2, 'inner')
    elif metric == 'euclidean':
        sim_mat = cport.sim_matrix(embed1, embed2, 'euclidean')
    elif metric == 'cosine':
        # 1 - cdist(cosine) == cosine similarity of the rows (similarity.py:42-44)
        sim_mat = cport.sim_matrix(normalize_rows(embed1), normalize_rows(embed2), 'inner')
    elif metric == 'manhattan':
        sim_mat = cport.sim_matrix(embed1, embed2, 'manhattan')
    else:
        raise ValueError(metric)
    if csls_k > 0:
        sim_mat = csls_sim(sim_mat, csls_k)
    return sim_mat


def calculate_nearest_k(sim_mat, k):
    """similarity.py:80-83 (mean of the k largest per row)."""
    return cport.topk_mean(sim_mat, k, axis=1)


def csls_sim(sim_mat, k):
    """similarity.py:57-77."""
    r = cport.topk_mean(sim_mat, k, axis=1)
    c = cport.topk_mean(sim_mat, k, axis=0)
    return cport.csls_apply(sim_mat, r, c)


# ----------------------------------------------------------------------------------------
# modules/finding/alignment.py + evaluation.py
# ----------------------------------------------------------------------------------------


def metrics_from_ranks(rank, top_k, total_num):
    """alignment.py:163-168 + 65-67: hits counts, hits %, MR, MRR from 0-based ranks."""
    rank = np.asarray(rank, np.int64)
    hits_cnt = [int((rank < k).sum()) for k in top_k]
    mr = float((rank + 1).sum()) / total_num
    mrr = float((1.0 / (rank + 1)).sum()) / total_num
    hits = np.array(hits_cnt) / total_num * 100
    hits = np.array([round(h, 3) for h in hits])
    return hits_cnt, hits, mr, mrr


def calculate_rank(idx, sim_mat, top_k, accurate, total_num):
    """alignment.py:146-168 (accurate semantics; stable tie rule, see oracle.c)."""
    assert 1 in top_k
    rank, argmax = _rank_rows(idx, sim_mat)
    hits_cnt, _, _, _ = metrics_from_ranks(rank, top_k, total_num)
    mr = float((rank.astype(np.int64) + 1).sum()) / total_num
    mrr = float((1.0 / (rank.astype(np.int64) + 1)).sum()) / total_num
    hits1_rest = {(int(idx[i]), int(argmax[i])) for i in range(len(idx))}
    return mr, mrr, hits_cnt, hits1_rest


def _rank_rows(idx, sim_rows):
    sim_rows = np.ascontiguousarray(sim_rows, np.float32)
    idx = np.asarray(idx)
    n, n2 = sim_rows.shape
    rank = np.empty(n, np.int32)
    argmax = np.empty(n, np.int32)
    for i in range(n):
        row = sim_rows[i]
        g = row[idx[i]]
        rank[i] = int((row > g).sum() + (row[:idx[i]] == g).sum())
        argmax[i] = int(np.argmax(row))  # first maximum
    return rank, argmax


def greedy_alignment(embed1, embed2, top_k, nums_threads, metric, normalize, csls_k, accurate):
    """alignment.py:13-84.  Returns (alignment_rest, hits1, mr, mrr) like the reference and
    keeps the raw integer results in greedy_alignment.last for bit-exact comparisons."""
    sim_mat = sim(embed1, embed2, metric=metric, normalize=normalize, csls_k=csls_k)
    num = sim_mat.shape[0]
    rank, argmax = cport.rank_from_matrix(sim_mat)
    hits_cnt, hits, mr, mrr = metrics_from_ranks(rank, top_k, num)
    alignment_rest = {(i, int(argmax[i])) for i in range(num)}
    assert len(alignment_rest) == num
    greedy_alignment.last = dict(rank=rank, argmax=argmax, hits_cnt=hits_cnt)
    return alignment_rest, hits[0], mr, mrr


def valid(embeds1, embeds2, mapping, top_k, threads_num, metric='inner', normalize=False,
          csls_k=0, accurate=False):
    """evaluation.py:6-14."""
    if mapping is not None:
        embeds1 = np.matmul(embeds1, mapping)
    _, hits1_12, mr_12, mrr_12 = greedy_alignment(embeds1, embeds2, top_k, threads_num, metric,
                                                  normalize, csls_k, accurate)
    return hits1_12, mrr_12


def test(embeds1, embeds2, mapping, top_k, threads_num, metric='inner', normalize=False,
         csls_k=0, accurate=True):
    """evaluation.py:17-25."""
    if mapping is not None:
        embeds1 = np.matmul(embeds1, mapping)
    rest, hits1_12, mr_12, mrr_12 = greedy_alignment(embeds1, embeds2, top_k, threads_num, metric,
                                                     normalize, csls_k, accurate)
    return rest, hits1_12, mrr_12


def early_stop(flag1, flag2, flag):
    """evaluation.py:28-33."""
    if flag <= flag2 <= flag1:
        return flag2, flag, True
    return flag2, flag, False


# ----------------------------------------------------------------------------------------
# TF1 graph pieces (restatements of TF1 semantics; the graphs' composition is pinned by tests/golden/tf_graphs.npz)
# ----------------------------------------------------------------------------------------


def l2_normalize(x, eps=1e-12):
    """tf.nn.l2_normalize(x, 1): x * rsqrt(max(sum(x^2, 1), eps)) (initializers.py:26)."""
    x = np.asarray(x)
    ss = np.maximum((x.astype(np.float64) ** 2).sum(1, keepdims=True), eps)
    return (x / np.sqrt(ss)).astype(x.dtype)


def truncated_normal(rng, shape, stddev):
    """tf.truncated_normal: N(0, stddev) re-drawn outside 2 sigma."""
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


def triple_step(ent, ent_acc, rel, rel_acc, pos, neg, **cfg):
    """One optimiser step of the translational graph (see oracle.c:oracle_triple_step)."""
    return cport.triple_step(ent, ent_acc, rel, rel_acc, pos, neg, **cfg)


def mapping_step(ent, mapping, acc_ent, acc_map, links, alpha, lr, ent_l2_norm=True):
    """MTransE mapping step (mapping.py:9-19, losses.py:76-80) with Adagrad on BOTH the
    entity table and M (generate_optimizer without var_list, optimizers.py:4-7).
    loss = alpha * (sum ||e2 - e1 M||^2 + sum (M M^T - I)^2).  fp64 internals."""
    links = np.asarray(links, np.int64).reshape(-1, 2)
    d = ent.shape[1]
    v = ent.astype(np.float64)
    M = mapping.astype(np.float64)
    ss = np.maximum((v ** 2).sum(1, keepdims=True), 1e-12)
    inv = 1.0 / np.sqrt(ss) if ent_l2_norm else np.ones_like(ss)
    y = v * inv
    e1, e2 = y[links[:, 0]], y[links[:, 1]]
    diff = e2 - e1 @ M
    orth = M @ M.T - np.eye(d)
    loss = alpha * ((diff ** 2).sum() + (orth ** 2).sum())
    g_e2 = alpha * 2.0 * diff
    g_e1 = -alpha * 2.0 * diff @ M.T
    g_M = alpha * (-2.0 * e1.T @ diff + 4.0 * orth @ M)
    gy = np.zeros_like(v)
    np.add.at(gy, links[:, 0], g_e1)
    np.add.at(gy, links[:, 1], g_e2)
    touched = np.zeros(len(v), bool)
    touched[links.ravel()] = True
    if ent_l2_norm:
        gv = (gy - y * (y * gy).sum(1, keepdims=True)) * inv
    else:
        gv = gy
    a = acc_ent.astype(np.float64)
    a[touched] += gv[touched] ** 2
    v[touched] -= lr * gv[touched] / np.sqrt(a[touched])
    am = acc_map.astype(np.float64) + g_M ** 2
    M -= lr * g_M / np.sqrt(am)
    ent[...] = v.astype(np.float32)
    acc_ent[...] = a.astype(np.float32)
    mapping[...] = M.astype(np.float32)
    acc_map[...] = am.astype(np.float32)
    return float(loss)


# ----------------------------------------------------------------------------------------
# GCN-Align (approaches/gcn_align.py)
# ----------------------------------------------------------------------------------------


def gcn_func(triples):
    """gcn_align.py:610-624: r2f[r] = #distinct heads / #triples."""
    head, cnt = {}, {}
    for h, r, t in triples:
        cnt[r] = cnt.get(r, 0) + 1
        head.setdefault(r, set()).add(h)
    return {r: len(head[r]) / cnt[r] for r in cnt}


def gcn_ifunc(triples):
    """gcn_align.py:626-640: r2if[r] = #distinct tails / #triples."""
    tail, cnt = {}, {}
    for h, r, t in triples:
        cnt[r] = cnt.get(r, 0) + 1
        tail.setdefault(r, set()).add(t)
    return {r: len(tail[r]) / cnt[r] for r in cnt}


def gcn_weighted_adj(e, triples):
    """gcn_align.py:642-664 -> scipy COO with row = second key, col = first key."""
    import scipy.sparse as sp
    r2f, r2if = gcn_func(triples), gcn_ifunc(triples)
    M = {}
    for h, r, t in triples:
        if h == t:
            continue
        M[(h, t)] = M.get((h, t), 0.0) + max(r2if[r], 0.3)
        M[(t, h)] = M.get((t, h), 0.0) + max(r2f[r], 0.3)
    row = [key[1] for key in M]
    col = [key[0] for key in M]
    data = [M[key] for key in M]
    return sp.coo_matrix((data, (row, col)), shape=(e, e))


def gcn_normalize_adj(adj):
    """gcn_align.py:566-573 (note: adj.dot(D).transpose().dot(D))."""
    import scipy.sparse as sp
    adj = sp.coo_matrix(adj)
    rowsum = np.array(adj.sum(1))
    with np.errstate(divide='ignore'):
        d_inv_sqrt = np.power(rowsum, -0.5).flatten()
    d_inv_sqrt[np.isinf(d_inv_sqrt)] = 0.
    d_mat_inv_sqrt = sp.diags(d_inv_sqrt)
    return adj.dot(d_mat_inv_sqrt).transpose().dot(d_mat_inv_sqrt).tocoo()


def gcn_preprocess_adj(adj):
    """gcn_align.py:575-578 -> (coords[nnz,2], values, shape)."""
    import scipy.sparse as sp
    a = gcn_normalize_adj(adj + sp.eye(adj.shape[0]))
    coords = np.vstack((a.row, a.col)).transpose()
    return coords, a.data, a.shape


def spmm(coords, values, x, n_rows):
    """tf.sparse_tensor_dense_matmul(A, X), fp32 (gcn_align.py:83)."""
    return cport.spmm_coo(coords[:, 0], coords[:, 1], values, x, n_rows)


def align_loss_and_grad(out, ILL, gamma, k, neg_left, neg_right, neg2_left, neg2_right):
    """gcn_align.py:298-320: L1 hinge over t seed links x k negatives, both sides,
    / (2 k t).  Returns (loss, d loss / d out) in fp64."""
    out = out.astype(np.float64)
    ILL = np.asarray(ILL, np.int64)
    t = len(ILL)
    left, right = ILL[:, 0], ILL[:, 1]
    dpos = out[left] - out[right]
    A = np.abs(dpos).sum(1)
    D = A + gamma
    g = np.zeros_like(out)
    loss = 0.0
    for nl, nr in ((neg_left, neg_right), (neg2_left, neg2_right)):
        nl = np.asarray(nl, np.int64)
        nr = np.asarray(nr, np.int64)
        dneg = out[nl] - out[nr]
        B = np.abs(dneg).sum(1)
        L = D[:, None] - B.reshape(t, k)
        mask = L > 0
        loss += L[mask].sum()
        # d/dA: +mask summed over k ; d/dB: -mask
        ca = mask.sum(1).astype(np.float64)
        sp_ = np.sign(dpos) * ca[:, None]
        np.add.at(g, left, sp_)
        np.add.at(g, right, -sp_)
        sn = -np.sign(dneg) * mask.reshape(-1)[:, None]
        np.add.at(g, nl, sn)
        np.add.at(g, nr, -sn)
    scale = 1.0 / (2.0 * k * t)
    return loss * scale, g * scale


def gcn_se_epoch(W, coords, values, ILL, gamma, k, negs, lr, features=None):
    """One full-batch SGD epoch of a GCN-Align unit (gcn_align.py:498-539, 204-267, 737-785): T = l2_normalize(W)
    (trunc_normal returns the normalised tensor, gcn_align.py:52-56); structure unit (featureless): H1 = relu(A T);
    attribute unit (features = scipy sparse X [n, f], W [f, d]): H1 = relu(A (X T)); out = A H1; loss = align_loss;
    W -= lr * dW.  fp64 internals; returns (loss, out_before_update fp32).  The forward pass and the gradient are pinned
    by the reference's own GCN_Align_Unit code run under tests/golden/tf_shim.py (tests/golden/tf_graphs.npz)."""
    Wd = W.astype(np.float64)
    ss = np.maximum((Wd ** 2).sum(1, keepdims=True), 1e-12)
    inv = 1.0 / np.sqrt(ss)
    T = Wd * inv
    import scipy.sparse as sp
    n = int(max(coords[:, 0].max(), coords[:, 1].max())) + 1 if features is None else features.shape[0]
    n = W.shape[0] if features is None else n
    A = sp.csr_matrix((np.asarray(values, np.float32).astype(np.float64),
                       (coords[:, 0], coords[:, 1])), shape=(n, n))
    X = None if features is None else sp.csr_matrix(features, dtype=np.float64)
    x = T if X is None else X @ T
    pre1 = A @ x
    H1 = np.maximum(pre1, 0.0)
    out = A @ H1
    loss, g_out = align_loss_and_grad(out, ILL, gamma, k, *negs)
    g_H1 = A.T @ g_out
    g_pre1 = g_H1 * (pre1 > 0)
    g_x = A.T @ g_pre1
    g_T = g_x if X is None else X.T @ g_x
    g_W = (g_T - T * (T * g_T).sum(1, keepdims=True)) * inv
    W[...] = (Wd - lr * g_W).astype(np.float32)
    return float(loss), out.astype(np.float32)


# ----------------------------------------------------------------------------------------
# Sparse attention pieces (alinet.py:656-677, rdgcn.py:202-215) -- which grouping tf.sparse_softmax applies is unpinned (H3)
# ----------------------------------------------------------------------------------------
# H3, the argument for the DEFAULT grouping 'runs' (every maximal run of CONSECUTIVE entries with the same row index is one softmax
# group) -- written down so that it can be reviewed against the TensorFlow 1.x sources (not vendored, not installable here;
# README.md:110 pins 1.8 / 1.12; tests/golden/make_tf1_golden.py settles it in five minutes wherever TensorFlow exists):
#   1. ORDER OF THE FED TENSOR.  AliNet's adjacency is scipy's `(D^-1/2 A D^-1/2).tocoo()` of a CSC product (alinet.py:27-41,51):
#      tocoo() of a CSC matrix enumerates column by column, so indices arrive COLUMN-major (checked against scipy in the survey
#      and in tests/test_graph_golden.py).  tf.SparseTensor keeps the order it is given; nothing in alinet.py calls
#      tf.sparse_reorder.
#   2. `adjs[0] * con_sa` (alinet.py:667-668) is SparseDenseCwiseMul (sparse_dense_binary_op_shared.cc): it maps over the
#      entries in place -- indices and their order unchanged.
#   3. `tf.sparse_add(con_sa_1, con_sa_2)` (alinet.py:669; sparse_add_op.cc, SparseAddOp::Compute) is a two-pointer MERGE of the
#      two index lists: while both have entries it compares a_indices[i] with b_indices[j] (sparse::DimComparator::cmp over the
#      dimensions): 0 -> emit the sum and advance both, -1 -> emit a, +1 -> emit b.  The two operands carry the SAME index list
#      (both are `adjs[0] * dense`), so every comparison is 0: the output has the operands' entries, summed, in the operands'
#      order -- column-major again.  (The op documents that it ASSUMES lexicographic order and does not check it.)
#   4. `tf.sparse_softmax` (alinet.py:673; sparse_softmax_op.cc, SparseSoftmaxOp::Compute) wraps indices / values in a
#      sparse::SparseTensor with the default order {0, 1} WITHOUT sorting or validating, then iterates
#      `st.group({0, ..., rank - 2})`.  sparse::GroupIterable (util/sparse/group_iterator.h) advances
#      `while (next_loc < N && GroupMatches(ix, loc, next_loc)) ++next_loc`: a group is a maximal run of CONSECUTIVE entries
#      whose group dimensions agree.  That the tensor really is sorted in that order is a DCHECK in SparseTensor::group
#      ("Group dimension is not in the same order as the sort order") -- compiled out of release wheels.  Each group's values
#      are replaced by exp(v - max) / sum in place, output index i = input index i.
#   5. Hence on the column-major tensor of 1. a group = consecutive entries of one ROW inside one COLUMN's listing = a single
#      entry wherever the pattern has no duplicate coordinates: every alpha is exp(0) / exp(0) = 1 and the layer aggregates the
#      unweighted neighbour sum ('runs').  If TensorFlow instead canonicalised the order somewhere on this path ('reorder'), or
#      if the reference is read as what its authors intended ('row': softmax over a node's whole neighbourhood), the other two
#      groupings apply; kernels, oracle and bench carry all three (approaches/alinet.py: softmax_grouping).
#   RDGCN (rdgcn.py:202-215) feeds `r_mat` in python-set order with duplicate (h, t) pairs: the same rule applies there --
#   groups = runs of equal h in THAT order -- and is what its restatement takes as data (seg_ptr).


def segment_softmax(logits, seg_offsets):
    """softmax within each segment [seg_offsets[s], seg_offsets[s+1]) (tf.sparse_softmax
    groups by the leading index; which entries form a group is DATA here, see H3)."""
    out = np.empty_like(logits, dtype=np.float64)
    lg = logits.astype(np.float64)
    for s in range(len(seg_offsets) - 1):
        a, b = seg_offsets[s], seg_offsets[s + 1]
        if b > a:
            m = lg[a:b].max()
            e = np.exp(lg[a:b] - m)
            out[a:b] = e / e.sum()
    return out


def leaky_relu(x, alpha=0.2):
    """tf.nn.leaky_relu default alpha = 0.2."""
    return np.where(x > 0, x, alpha * x)


def sparse_attn_forward(z, v, seg_ptr, seg_row, colidx, n_rows, slope=0.2):
    """alinet.py:670-676 / rdgcn.py:207-211: alpha = softmax over each segment of leaky_relu(z);
    out[seg_row[s]] += sum_e alpha_e v[colidx[e]].  fp64.  Returns (out, alpha)."""
    z = np.asarray(z, np.float64)
    v = np.asarray(v, np.float64)
    e = np.where(z > 0, z, slope * z)
    alpha = segment_softmax(e, seg_ptr)
    out = np.zeros((n_rows, v.shape[1]))
    for s in range(len(seg_ptr) - 1):
        a, b = seg_ptr[s], seg_ptr[s + 1]
        if b > a:
            out[seg_row[s]] += alpha[a:b] @ v[colidx[a:b]]
    return out, alpha


def sparse_attn_backward(z, v, alpha, dout, seg_ptr, seg_row, colidx, slope=0.2):
    """analytic gradient of sparse_attn_forward w.r.t. z and v (checked against finite differences in
    tests/test_oracle_golden.py)."""
    z = np.asarray(z, np.float64)
    v = np.asarray(v, np.float64)
    dout = np.asarray(dout, np.float64)
    dz = np.zeros_like(z)
    dv = np.zeros_like(v)
    for s in range(len(seg_ptr) - 1):
        a, b = seg_ptr[s], seg_ptr[s + 1]
        if b <= a:
            continue
        cols = colidx[a:b]
        d_alpha = v[cols] @ dout[seg_row[s]]
        c = (alpha[a:b] * d_alpha).sum()
        dz[a:b] = alpha[a:b] * (d_alpha - c) * np.where(z[a:b] > 0, 1.0, slope)
        np.add.at(dv, cols, alpha[a:b, None] * dout[seg_row[s]][None, :])
    return dz, dv


def adam_tf(p, g, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer step t (1-based), fp64 internals, in place on float arrays."""
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    m[...] = beta1 * m + (1 - beta1) * g
    v[...] = beta2 * v + (1 - beta2) * g * g
    p[...] = p - lr_t * m / (np.sqrt(v) + eps)


def galeshapley(suitor_pref_dict, reviewer_pref_dict, max_iteration):
    """alignment.py:170-221 restated (full preference lists, sequential rounds, a displaced suitor keeps the
    reviewer at the head of its list until it loses the comparison in a later round)."""
    suitor_pref_dict = {s: list(v) for s, v in suitor_pref_dict.items()}
    suitors = list(suitor_pref_dict.keys())
    matching, rev_matching = {}, {}
    for _ in range(max_iteration):
        if len(suitors) <= 0:
            break
        for s in suitors:
            r = suitor_pref_dict[s][0]
            if r not in matching.values():
                matching[s] = r
                rev_matching[r] = s
            else:
                r_partner = rev_matching.get(r)
                if reviewer_pref_dict[r].index(s) < reviewer_pref_dict[r].index(r_partner):
                    del matching[r_partner]
                    matching[s] = r
                    rev_matching[r] = s
                else:
                    suitor_pref_dict[s].remove(r)
        suitors = sorted(set(suitor_pref_dict.keys()) - set(matching.keys()))
    return matching


def stable_alignment(sim_mat, cut=100):
    """alignment.py:87-134 on a given similarity matrix: argsort both ways (stable, ties by index) + galeshapley."""
    sim_mat = np.asarray(sim_mat)
    kg1 = {i: np.argsort(-sim_mat[i], kind="stable").tolist() for i in range(sim_mat.shape[0])}
    kg2 = {j: np.argsort(-sim_mat[:, j], kind="stable").tolist() for j in range(sim_mat.shape[1])}
    return galeshapley(kg1, kg2, cut)


# ----------------------------------------------------------------------------------------
# AliNet negative links (alinet.py:988-1006) -- the device formulation restated
# ----------------------------------------------------------------------------------------
def _feistel_f(r, key, rnd):
    m = 0xFFFFFFFF
    v = (r * 0x9E3779B1 + key + rnd * 0x85EBCA6B) & m
    v ^= v >> 15
    v = (v * 0x2C1B3C6D) & m
    v ^= v >> 12
    v = (v * 0x297A2D39) & m
    v ^= v >> 15
    return v


def perm_index(i, n, key):
    """image of i under the keyed pseudo-random permutation of [0, n): 4-round Feistel network on the index bits +
    cycle walking (random.sample = the first `count` images)."""
    bits = 2
    while (1 << bits) < n:
        bits += 2
    half = bits >> 1
    mask = (1 << half) - 1
    x = i
    while True:
        l, r = x >> half, x & mask
        for rnd in range(4):
            l, r = r, l ^ (_feistel_f(r, key, rnd) & mask)
        x = (l << half) | r
        if x < n:
            return x


def _mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d); x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b); x ^= x >> np.uint32(16)
    return x


def epoch_layout_perm(n, seed, epoch, second):
    """The epoch's permutation of one KG's triple list (basic_model.py:234-235 random.shuffle(kgs.kg1/kg2.relation_triples_list)
    draws it from Python's Mersenne Twister; here it is a keyed bijection so that it can be evaluated per element): 6 alternating
    Feistel rounds on the ceil(log2 n)-bit index keyed by Philox4x32-10 of (epoch, list, 0x5eed, {0, 1}) under the seed,
    cycle-walked into [0, n).  -> int64 [n], perm[i] = the list position read for shuffled position i."""
    from . import cport
    key2 = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32)
    wa = cport.philox(np.array([epoch, 1 if second else 0, 0x5eed, 0], np.uint32), key2)
    wb = cport.philox(np.array([epoch, 1 if second else 0, 0x5eed, 1], np.uint32), key2)
    key = [wa[0], wa[1], wa[2], wa[3], wb[0], wb[1]]
    bits = 1
    while bits < 32 and (1 << bits) < n:
        bits += 1
    bits = max(bits, 2)
    rbits = bits // 2
    lbits = bits - rbits
    lmask, rmask = np.uint32((1 << lbits) - 1), np.uint32((1 << rbits) - 1)
    x = np.arange(n, dtype=np.uint32)
    todo = np.ones(n, bool)
    with np.errstate(over="ignore"):
        while todo.any():
            v = x[todo]
            l, r = v >> np.uint32(rbits), v & rmask
            for q in range(0, 6, 2):
                l = (l ^ _mix32(r ^ key[q])) & lmask
                r = (r ^ _mix32(l ^ key[q + 1])) & rmask
            v = (l << np.uint32(rbits)) | r
            x[todo] = v
            todo[todo] = v >= n
    return x.astype(np.int64)


def epoch_layout(t1, t2, slot, seed, epoch):
    """batch.py:17-22 over the shuffled lists: layout row j = (KG1 list ++ KG2 list)[perm(slot[j])], each list permuted on its own."""
    n1 = len(t1)
    full = np.concatenate([t1, t2])
    p = np.concatenate([epoch_layout_perm(n1, seed, epoch, False), n1 + epoch_layout_perm(len(t2), seed, epoch, True)]) if len(t2) else \
        epoch_layout_perm(n1, seed, epoch, False)
    return full[p[np.asarray(slot)]]


def link_negatives(n_pos, k, seed, step, pos_links=None, ents1=None, ents2=None, nbr1=None, row1=None, nbr2=None,
                   row2=None, exclude=()):
    """alinet.py:988-1006: uniform -> k rounds of zip(sample(ents1, n_pos), sample(ents2, n_pos)); truncated -> per link
    (e1, c) for c in sample(neighbors1[e1], k) and (c, e2) for c in sample(neighbors2[e2], k); then
    set(pairs) - exclude.  -> (pairs int32 [m, 2] in draw order, valid bool [m]: first of equal pairs, not excluded)."""
    from . import cport
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32)
    pairs = []
    if nbr1 is None:
        for rnd in range(k):
            w = cport.philox(np.array([rnd, step, 1, 0], np.uint32), key)
            for i in range(n_pos):
                pairs.append((int(ents1[perm_index(i, len(ents1), int(w[0]))]), int(ents2[perm_index(i, len(ents2), int(w[1]))])))
    else:
        nbr_k = nbr1.shape[1]
        for link in range(n_pos):
            e1, e2 = int(pos_links[link][0]), int(pos_links[link][1])
            w = cport.philox(np.array([link, step, 2, 0], np.uint32), key)
            for s in range(k):
                pairs.append((e1, int(nbr1[row1[e1], perm_index(s, nbr_k, int(w[0]))])))
            for s in range(k):
                pairs.append((int(nbr2[row2[e2], perm_index(s, nbr_k, int(w[1]))]), e2))
    seen, valid = set(), []
    exclude = set(exclude)
    for p in pairs:
        valid.append(p not in seen and p not in exclude)
        seen.add(p)
    return np.asarray(pairs, np.int32).reshape(-1, 2), np.asarray(valid, bool)


# ----------------------------------------------------------------------------------------------------
# RotatE step of BootEA_RotatE (approaches/bootea_rotate.py:50-109,148-158) -- loss and gradients reproduce the reference's
# own graph code run under tests/golden/tf_shim.py; optimiser arithmetic unpinned vs TF (hand-restated
# autodiff + optimisers); the gradients are pinned by finite differences in tests/test_oracle_golden.py.
# ----------------------------------------------------------------------------------------------------
def _l2n_rows(x):
    """tf.nn.l2_normalize(x, 1): x * rsqrt(max(sum x^2, 1e-12))."""
    return x / np.sqrt(np.maximum((x * x).sum(1, keepdims=True), 1e-12))


def rotate_loss(ent, rel, pos, neg, gamma, phase_scale, ent_l2_norm=True, rel_l2_norm=False):
    """The loss alone (forward only), written as the TF graph reads: lookup_all (bootea_rotate.py:83-94),
    _generate_scores (:59-70), _generate_loss (:72-81).  ent = [re ; im] stacked [2E, d], fp64."""
    E = ent.shape[0] // 2
    re, im = ent[:E], ent[E:]
    if ent_l2_norm:
        re, im = _l2n_rows(re), _l2n_rows(im)
    rl = _l2n_rows(rel) if rel_l2_norm else rel

    def scores(tr, is_pos):
        if tr is None or len(tr) == 0:
            return np.zeros(0)
        h, r, t = tr[:, 0], tr[:, 1], tr[:, 2]
        theta = rl[r] * phase_scale
        rr, ir = np.cos(theta), np.sin(theta)
        re_s = re[h] * rr - im[h] * ir - re[t]
        im_s = re[h] * ir + im[h] * rr - im[t]
        sc = gamma - np.sqrt(re_s ** 2 + im_s ** 2).sum(-1)
        return sc if is_pos else -sc

    def log_sigmoid(x):
        return -np.logaddexp(0.0, -x)
    return -log_sigmoid(scores(pos, True)).sum() - log_sigmoid(scores(neg, False)).sum()


def rotate_step(ent, rel, pos, neg, state, *, gamma, phase_scale, ent_l2_norm=True, rel_l2_norm=False, optimizer="Adam",
                lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8):
    """One optimiser step in place on fp64 ent [2E, d] / rel [R, d]; returns the batch loss.
    state: dict created by the caller ({} at first); holds Adagrad accumulators (0.1) or Adam m, v, t.
    Gradient by hand (duplicates summed with np.add.at), back through the row normalisations, then
    tf.train.{GradientDescent,Adagrad,Adam}Optimizer on the WHOLE variables (Adam moves every row)."""
    E, d = ent.shape[0] // 2, ent.shape[1]
    inv_e = 1.0 / np.sqrt(np.maximum((ent * ent).sum(1, keepdims=True), 1e-12)) if ent_l2_norm else np.ones((2 * E, 1))
    inv_r = 1.0 / np.sqrt(np.maximum((rel * rel).sum(1, keepdims=True), 1e-12)) if rel_l2_norm else np.ones((len(rel), 1))
    ye, yr = ent * inv_e, rel * inv_r
    ge, gr = np.zeros_like(ent), np.zeros_like(rel)
    loss = 0.0
    for tr, is_pos in ((pos, True), (neg, False)):
        if tr is None or len(tr) == 0:
            continue
        tr = np.asarray(tr)
        h, r, t = tr[:, 0], tr[:, 1], tr[:, 2]
        theta = yr[r] * phase_scale
        c, s = np.cos(theta), np.sin(theta)
        rh, ih, rt, it = ye[h], ye[E + h], ye[t], ye[E + t]
        a = rh * c - ih * s - rt
        b = rh * s + ih * c - it
        n = np.sqrt(a * a + b * b)
        dist = n.sum(1)
        x = dist - gamma if is_pos else gamma - dist
        loss += np.logaddexp(0.0, x).sum()
        coef = 1.0 / (1.0 + np.exp(-x))
        coef = coef if is_pos else -coef
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = np.where(n > 0, 1.0 / n, 0.0)
        da, db = coef[:, None] * a * inv, coef[:, None] * b * inv
        np.add.at(ge, h, da * c + db * s)
        np.add.at(ge, E + h, db * c - da * s)
        np.add.at(ge, t, -da)
        np.add.at(ge, E + t, -db)
        np.add.at(gr, r, (da * (-rh * s - ih * c) + db * (rh * c - ih * s)) * phase_scale)

    def back(y, g, inv, on, raw):
        if not on:
            return g
        ss = (raw * raw).sum(1, keepdims=True)
        ydg = np.where(ss > 1e-12, (y * g).sum(1, keepdims=True), 0.0)
        return (g - y * ydg) * inv
    gv_e = back(ye, ge, inv_e, ent_l2_norm, ent)
    gv_r = back(yr, gr, inv_r, rel_l2_norm, rel)
    if optimizer == "Adam":
        state["t"] = t_ = state.get("t", 0) + 1
        lr_t = lr * np.sqrt(1.0 - beta2 ** t_) / (1.0 - beta1 ** t_)
        for name, var, g in (("ent", ent, gv_e), ("rel", rel, gv_r)):
            m = state.setdefault("m_" + name, np.zeros_like(var))
            v = state.setdefault("v_" + name, np.zeros_like(var))
            m *= beta1
            m += (1.0 - beta1) * g
            v *= beta2
            v += (1.0 - beta2) * g * g
            var -= lr_t * m / (np.sqrt(v) + eps)
    elif optimizer == "Adagrad":
        for name, var, g in (("ent", ent, gv_e), ("rel", rel, gv_r)):
            acc = state.setdefault("acc_" + name, np.full_like(var, 0.1))
            acc += g * g
            var -= lr * g / np.sqrt(acc)
    else:
        ent -= lr * gv_e
        rel -= lr * gv_r
    return float(loss)


def rotate_lookup(ent, ids, part_norm=True, sum_norm=False):
    """re + im of the looked-up rows as the evaluation reads them (bootea_rotate.py:111-146,160-167) -> fp32."""
    E = ent.shape[0] // 2
    re, im = ent[:E][ids], ent[E:][ids]
    if part_norm:
        re, im = _l2n_rows(re), _l2n_rows(im)
    out = re + im
    if sum_norm:
        out = _l2n_rows(out)
    return out.astype(np.float32)


def halo_plan(pos_all, neg_all, k, offsets, n_ent, world):
    """Boundary-row plan of the partitioned step (no reference counterpart: the reference is single-device; restated from the
    protocol in DESIGN.md section 6 / include/openea_hip.h:oea_halo_plan, test infrastructure only).  Rank r's share of step s =
    rows [nb r / world, nb (r + 1) / world) of the batch; the rows it refers to = heads and tails of its positives and of their k
    negatives; owner of entity id = id mod world, local index = id div world.
    -> (counts int64 [steps, world, world], lists: dict (s, r, o) -> ascending local indices)"""
    pos_all = np.asarray(pos_all)
    neg_all = None if neg_all is None else np.asarray(neg_all)
    steps = len(offsets) - 1
    counts = np.zeros((steps, world, world), np.int64)
    lists = {}
    for s in range(steps):
        b0, nb = int(offsets[s]), int(offsets[s + 1] - offsets[s])
        for r in range(world):
            lo, hi = b0 + nb * r // world, b0 + nb * (r + 1) // world
            ids = [pos_all[lo:hi, 0], pos_all[lo:hi, 2]]
            if k > 0 and hi > lo:
                ids += [neg_all[lo * k:hi * k, 0], neg_all[lo * k:hi * k, 2]]
            ids = np.unique(np.concatenate(ids)) if hi > lo else np.zeros(0, np.int64)
            for o in range(world):
                mine = np.sort(ids[ids % world == o] // world)
                lists[(s, r, o)] = mine
                counts[s, r, o] = len(mine)
    return counts, lists


def step_plan(pos_all, neg_all, k, offsets, n_ent, hub_entries=8):
    """Gathered-sum plan of a translational epoch (no reference counterpart: it is the bookkeeping of the scatter-add TF performs on
    the gradient of tf.nn.embedding_lookup, models/basic_model.py:89-98, done once per epoch; restated from
    include/openea_hip.h:oea_step_plan_build, test infrastructure only).  A positive whose k negatives are all corruptions of it on
    ONE side (batch.py:101-107: one Bernoulli per sampling round) refers to its head and tail row:
        tail side (every negative keeps the head):  head += A_p,  tail -= B_p
        head side (every negative keeps the tail):  head += B_p,  tail -= A_p
    with slots A_p = 2 * (index inside the batch), B_p = A_p + 1; other positives are left to the atomic path.
    -> dict(ukeys uint64 [(step << row_bits) | row, ascending], uoff, vals uint32 [sign << 31 | slot, batch order inside a key],
            step_first int64 [steps + 1], row_bits, pflags uint32 [n])"""
    pos_all = np.asarray(pos_all, np.int64)
    neg = np.asarray(neg_all, np.int64).reshape(len(pos_all), k, 3) if k > 0 else np.zeros((len(pos_all), 0, 3), np.int64)
    steps = len(offsets) - 1
    row_bits = max(int(max(n_ent - 1, 1)).bit_length(), 1)
    keys, vals = [], []
    for s in range(steps):
        for p in range(int(offsets[s]), int(offsets[s + 1])):
            h, r, t = pos_all[p]
            same_h, same_t, same_r = neg[p, :, 0] == h, neg[p, :, 2] == t, neg[p, :, 1] == r
            corruption = same_r & (same_h | same_t)
            if not corruption.all() or not (same_h.all() or same_t.all()):
                continue
            tails = bool(same_h.all())                       # (a positive whose negatives all equal it counts as tail side)
            slot = 2 * (p - int(offsets[s]))
            keys += [(s << row_bits) | int(h), (s << row_bits) | int(t)]
            vals += [slot + (0 if tails else 1), (1 << 31) | (slot + (1 if tails else 0))]
    keys, vals = np.asarray(keys, np.uint64), np.asarray(vals, np.uint32)
    order = np.argsort(keys, kind="stable")
    keys, vals = keys[order], vals[order]
    ukeys, first = np.unique(keys, return_index=True)
    uoff = np.concatenate([first, [len(keys)]]).astype(np.uint32)
    step_first = np.searchsorted(ukeys, (np.arange(steps + 1, dtype=np.uint64) << np.uint64(row_bits)))
    # hubs: a row with more than hub_entries references in one step takes its positives' gradient through the atomic scratch;
    # pflags[p] bit 0 / 1 = positive p's head / tail reference is to such a row
    pflags = np.zeros(len(pos_all), np.uint32)
    for i in np.nonzero(np.diff(uoff.astype(np.int64)) > hub_entries)[0]:
        s = int(ukeys[i]) >> row_bits
        for v in vals[uoff[i]:uoff[i + 1]]:
            pflags[int(offsets[s]) + ((int(v) & 0x7fffffff) >> 1)] |= 2 if (int(v) >> 31) else 1
    return dict(ukeys=ukeys, uoff=uoff, vals=vals, step_first=step_first.astype(np.int64), row_bits=row_bits, pflags=pflags)


def sample_negatives_replay(pos, k, triples, cand_head, cand_tail, replay, max_try=10):
    """generate_neg_triples_fast (modules/train/batch.py:89-119) with its random numbers REPLAYED from a record: per positive up to
    max_try rounds; round i corrupts the head when replay[p, i, 0] == 1 (np.random.binomial(1, 0.5), batch.py:99), else the tail;
    the entities drawn are the candidate list's entries at positions replay[p, i, 1 : 1 + need] (random.sample(candidates, need),
    batch.py:101,104: distinct), need = negatives still missing; drawn triples that are true triples are dropped (set difference,
    batch.py:110) except in the last round (batch.py:106-108); accepted negatives are appended in draw order (the reference appends
    them in python-set order: a positive's negatives are compared as a multiset).
    cand_head[p] / cand_tail[p]: the candidate list of positive p's head / tail (neighbor.get(entity, entities_list)).
    -> int32 [n_pos * k, 3]"""
    tri = set(map(tuple, np.asarray(triples).tolist()))
    out = np.zeros((len(pos) * k, 3), np.int32)
    for p, (h, r, t) in enumerate(np.asarray(pos).tolist()):
        got = []
        for i in range(max_try):
            need = k - len(got)
            head = int(replay[p, i, 0]) == 1
            cands = cand_head[p] if head else cand_tail[p]
            drawn = [int(cands[int(j)]) for j in replay[p, i, 1:1 + need]]
            neg = [(e, r, t) if head else (h, r, e) for e in drawn]
            got += neg if i == max_try - 1 else [x for x in neg if x not in tri]
            if len(got) == k:
                break
        assert len(got) == k
        out[p * k:(p + 1) * k] = got
    return out
