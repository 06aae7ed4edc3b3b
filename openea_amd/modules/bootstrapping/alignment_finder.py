"""Bootstrapping candidate selection (mirror of openea/modules/bootstrapping/alignment_finder.py).

SURVEY 8(f) rank 1 -- downstream of the kNN / similarity kernels.  Candidate selection
(threshold filter ∩ per-row top-k, alignment_finder.py:28-76) runs on the device through the
same top-k kernel as the neighbour search.  The matching step of the reference calls graph_tool
(`max_cardinality_matching(heuristic=True, weight=..., minimize=False)`: a linear-time HEURISTIC
maximal matching, alignment_finder.py:83-112) or igraph (exact maximum-weight bipartite matching,
alignment_finder.py:115-140); neither library is available here:
  * heuristic=True  -> deterministic greedy weight-descending maximal matching (the classic
    1/2-approximation; graph_tool's heuristic is itself unspecified / randomised),
  * heuristic=False -> EXACT maximum-weight bipartite matching through scipy's sparse assignment
    solver (same optimum as igraph's, ties aside).
"""
import time

import numpy as np

from ... import ops


class PairSim:
    """sim_mat[i, j] of eval_ref_sim_mat (bootea.py:214-219) evaluated on demand from the two
    L2-normalised reference embedding blocks instead of a materialised n x n matrix."""

    def __init__(self, embeds1, embeds2):
        self.e1 = np.asarray(embeds1, np.float32)
        self.e2 = np.asarray(embeds2, np.float32)
        self.shape = (len(self.e1), len(self.e2))

    def __getitem__(self, ij):
        i, j = ij
        return float(np.dot(self.e1[i], self.e2[j]))

    def pairs(self, ii, jj):
        return np.einsum('nd,nd->n', self.e1[ii], self.e2[jj])


def search_nearest_k_device(sim, k):
    """alignment_finder.py:66-76: the k nearest columns of every row -> int32 [n, k] (host)."""
    d = sim.e1.shape[1]
    return ops.topk_inner(ops.to_table(sim.e1), ops.to_table(sim.e2), d, k).cpu().numpy()


def find_alignment(sim, sim_th, k):
    """alignment_finder.py:28-51: pairs with sim > sim_th that are among the row's k nearest."""
    assert k > 0
    idx = search_nearest_k_device(sim, k)
    ii = np.repeat(np.arange(idx.shape[0]), k)
    jj = idx.reshape(-1)
    w = sim.pairs(ii, jj)
    keep = w > sim_th
    if not keep.any():
        return None, None
    return list(zip(ii[keep].tolist(), jj[keep].tolist())), w[keep]


def check_new_alignment(aligned_pairs, context="check alignment"):
    """alignment_finder.py:143-151."""
    if aligned_pairs is None or len(aligned_pairs) == 0:
        print("{}, empty aligned pairs".format(context))
        return
    num = sum(1 for x, y in aligned_pairs if x == y)
    print("{}, right alignment: {}/{}={:.3f}".format(context, num, len(aligned_pairs), num / len(aligned_pairs)))


def greedy_weight_matching(pairs, weights):
    """one-to-one selection: edges by weight descending (ties: smaller (i, j) first)."""
    order = sorted(range(len(pairs)), key=lambda e: (-weights[e], pairs[e]))
    used_i, used_j, out = set(), set(), set()
    for e in order:
        i, j = pairs[e]
        if i not in used_i and j not in used_j:
            used_i.add(i)
            used_j.add(j)
            out.add((i, j))
    return out


def max_weight_matching(pairs, weights):
    """exact maximum-weight bipartite matching of the candidate edges (mwgm_igraph, alignment_finder.py:115-140):
    every left node gets a private dummy partner so that a FULL matching of the left side always exists, and
    with cost C - w on real edges / C on dummy edges the minimum-cost full matching maximises the real weight."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import min_weight_full_bipartite_matching
    pairs = list(pairs)
    w = np.asarray(weights, np.float64)
    lefts = sorted({p[0] for p in pairs})
    rights = sorted({p[1] for p in pairs})
    li = {x: i for i, x in enumerate(lefts)}
    ri = {y: j for j, y in enumerate(rights)}
    nl, nr = len(lefts), len(rights)
    big = float(max(w.max(), 0.0)) + 1.0
    keep = w > 0                                            # a non-positive edge never improves the total
    rows = np.concatenate([[li[p[0]] for p, kp in zip(pairs, keep) if kp], np.arange(nl)]).astype(np.int64)
    cols = np.concatenate([[ri[p[1]] for p, kp in zip(pairs, keep) if kp], nr + np.arange(nl)]).astype(np.int64)
    cost = np.concatenate([big - w[keep], np.full(nl, big)])
    r, c = min_weight_full_bipartite_matching(sp.csr_matrix((cost, (rows, cols)), shape=(nl, nr + nl)))
    return {(lefts[i], rights[j]) for i, j in zip(r, c) if j < nr}


def find_potential_alignment_mwgm(sim, sim_th, k, heuristic=True):
    """alignment_finder.py:12-25."""
    t = time.time()
    pairs, w = find_alignment(sim, sim_th, k)
    if pairs is None:
        return None
    check_new_alignment(pairs, context="after filtering by sim and nearest k")
    t1 = time.time()
    selected = greedy_weight_matching(pairs, w) if heuristic else max_weight_matching(pairs, w)
    check_new_alignment(selected, context="after mwgm")
    print("mwgm costs time: {:.3f} s".format(time.time() - t1))
    print("selecting potential alignment costs time: {:.3f} s".format(time.time() - t))
    return selected


def find_potential_alignment_greedily(sim, sim_th):
    """alignment_finder.py:8-9."""
    pairs, _ = find_alignment(sim, sim_th, 1)
    return None if pairs is None else set(pairs)
