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
    L2-normalised reference embedding blocks instead of a materialised n x n matrix.  The blocks stay on the
    device for the candidate search; single lookups (update_labeled_alignment_*) use a host copy."""

    def __init__(self, embeds1, embeds2, dim=None):
        if hasattr(embeds1, "is_cuda"):
            self.t1, self.t2 = embeds1, embeds2
            self.dim = int(dim if dim is not None else embeds1.shape[1])
            self._e1 = self._e2 = None
        else:
            self._e1 = np.asarray(embeds1, np.float32)
            self._e2 = np.asarray(embeds2, np.float32)
            self.dim = self._e1.shape[1]
            self.t1 = self.t2 = None
        self.shape = (len(embeds1), len(embeds2))

    @property
    def e1(self):
        if self._e1 is None:
            self._e1 = self.t1[:, :self.dim].cpu().numpy()
        return self._e1

    @property
    def e2(self):
        if self._e2 is None:
            self._e2 = self.t2[:, :self.dim].cpu().numpy()
        return self._e2

    def tables(self):
        if self.t1 is None:
            self.t1, self.t2 = ops.to_table(self._e1), ops.to_table(self._e2)
        return self.t1, self.t2

    def __getitem__(self, ij):
        i, j = ij
        return float(np.dot(self.e1[i], self.e2[j]))

    def pairs(self, ii, jj):
        return np.einsum('nd,nd->n', self.e1[ii], self.e2[jj])


def search_nearest_k_device(sim, k):
    """alignment_finder.py:66-76: the k nearest columns of every row -> int32 [n, k] (host)."""
    t1, t2 = sim.tables()
    return ops.topk_inner(t1, t2, sim.dim, k).cpu().numpy()


def find_alignment_arrays(sim, sim_th, k):
    """alignment_finder.py:28-51 without python tuples: (rows, cols, weights) host arrays of the pairs with
    sim > sim_th that are among the row's k nearest; candidate search AND the pair similarities on the device."""
    import torch
    assert k > 0
    t1, t2 = sim.tables()
    idx = ops.topk_inner(t1, t2, sim.dim, k)                              # device [n, k]
    n = idx.shape[0]
    ii = torch.arange(n, dtype=torch.int32, device=idx.device).repeat_interleave(k)
    jj = idx.reshape(-1).contiguous()
    w = ops.pair_dots(t1, t2, sim.dim, ii, jj)
    keep = (w > sim_th).cpu().numpy()
    if not keep.any():
        return None
    return ii.cpu().numpy()[keep], jj.cpu().numpy()[keep], w.cpu().numpy()[keep]


def find_alignment(sim, sim_th, k):
    """alignment_finder.py:28-51: pairs with sim > sim_th that are among the row's k nearest."""
    r = find_alignment_arrays(sim, sim_th, k)
    if r is None:
        return None, None
    return list(zip(r[0].tolist(), r[1].tolist())), r[2]


def check_new_alignment(aligned_pairs, context="check alignment"):
    """alignment_finder.py:143-151 (also takes a (rows, cols) pair of arrays)."""
    if aligned_pairs is None or len(aligned_pairs) == 0:
        print("{}, empty aligned pairs".format(context))
        return
    if isinstance(aligned_pairs, tuple) and len(aligned_pairs) == 2 and hasattr(aligned_pairs[0], "dtype"):
        num, total = int((aligned_pairs[0] == aligned_pairs[1]).sum()), len(aligned_pairs[0])
        if total == 0:
            print("{}, empty aligned pairs".format(context))
            return
    else:
        num, total = sum(1 for x, y in aligned_pairs if x == y), len(aligned_pairs)
    print("{}, right alignment: {}/{}={:.3f}".format(context, num, total, num / total))


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
    cand = find_alignment_arrays(sim, sim_th, k)
    if cand is None:
        return None
    ii, jj, w = cand
    check_new_alignment((ii, jj), context="after filtering by sim and nearest k")
    t1 = time.time()
    if heuristic:                       # native weight-descending selection (csrc/match.hip)
        m = ops.greedy_matching(ii, jj, w)
        selected = set(zip(ii[m].tolist(), jj[m].tolist()))
    else:
        selected = max_weight_matching(list(zip(ii.tolist(), jj.tolist())), w)
    check_new_alignment(selected, context="after mwgm")
    print("mwgm costs time: {:.3f} s".format(time.time() - t1))
    print("selecting potential alignment costs time: {:.3f} s".format(time.time() - t))
    return selected


def find_potential_alignment_greedily(sim, sim_th):
    """alignment_finder.py:8-9."""
    pairs, _ = find_alignment(sim, sim_th, 1)
    return None if pairs is None else set(pairs)


# ---- the reference's host-matrix helpers under their own names (alignment_finder.py:54-140) --------------------------
def filter_sim_mat(mat, threshold, greater=True, equal=False):
    """alignment_finder.py:54-63: the (row, column) pairs on the chosen side of the threshold, as a set."""
    mat = np.asarray(mat)
    keep = (mat >= threshold if equal else mat > threshold) if greater else (mat <= threshold if equal else mat < threshold)
    rows, cols = np.nonzero(keep)
    return set(zip(rows, cols))


def search_nearest_k(sim_mat, k):
    """alignment_finder.py:66-76 on a host matrix: {(i, j)} for the k nearest columns j of every row i (np.argpartition,
    like the reference; PairSim inputs go through the device search)."""
    assert k > 0
    if isinstance(sim_mat, PairSim):
        near = search_nearest_k_device(sim_mat, k)
    else:
        near = np.argpartition(-np.asarray(sim_mat), k, axis=1)[:, :k]
    pairs = {(i, j) for i in range(near.shape[0]) for j in near[i]}
    assert len(pairs) == near.shape[0] * k
    return pairs


def _pair_weights(pairs, sim_mat):
    pairs = list(pairs)
    return pairs, np.array([sim_mat[i, j] for i, j in pairs], np.float64)


def mwgm_igraph(pairs, sim_mat):
    """alignment_finder.py:124-140 (igraph maximum_bipartite_matching = the exact maximum-weight matching)."""
    pairs, weights = _pair_weights(pairs, sim_mat)
    return max_weight_matching(pairs, weights)


def mwgm_graph_tool(pairs, sim_mat):
    """alignment_finder.py:83-121 (graph_tool's heuristic matching) -> the greedy weight-descending matching."""
    pairs, weights = _pair_weights(pairs, sim_mat)
    return greedy_weight_matching(pairs, weights)


def mwgm(pairs, sim_mat, func):
    """alignment_finder.py:79-80."""
    return func(pairs, sim_mat)
