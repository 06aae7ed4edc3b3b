"""Greedy alignment search + Hits@k / MR / MRR (mirror of openea/modules/finding/alignment.py).

The N1 x N2 similarity matrix is never written: the rank of the gold column and the argmax are
counted inside the similarity tiles (csrc/sim_rank.hip).  Ranks are exact in BOTH modes -- the
reference's quick mode (accurate=False) only trusts Hits@k and leaves MR/MRR to
argpartition's internal order (alignment.py:157-162); Hits@k agree in both.
"""
import collections.abc
import time

import numpy as np

from ... import ops
from .similarity import csls_means_device, device_metric


class AlignmentPairs(collections.abc.Set):
    """{(i, argmax[i])} -- what greedy_alignment returns as `alignment_rest`.  The reference builds a Python set of tuples
    (alignment.py:47-63); for the 70,000 test pairs of a 100K dataset that alone is 17 ms of host time, more than the whole
    evaluation on the device.  This is that set without the tuples: len, iteration, membership, equality with a real set and
    the set algebra of collections.abc.Set (whose results are real sets) behave the same; the callers of the reference
    only iterate it (`for i, j in rest_12`, basic_model.py:146-148)."""
    __slots__ = ("_am",)

    def __init__(self, argmax):
        self._am = np.asarray(argmax)

    def __len__(self):
        return len(self._am)

    def __iter__(self):
        return zip(range(len(self._am)), self._am.tolist())

    def __contains__(self, pair):
        try:
            i, j = pair
        except (TypeError, ValueError):
            return False
        return isinstance(i, (int, np.integer)) and 0 <= i < len(self._am) and int(self._am[i]) == j

    @classmethod
    def _from_iterable(cls, it):
        return set(it)

    def __repr__(self):
        return "AlignmentPairs(%d pairs)" % len(self._am)


def greedy_alignment_device(t1, t2, dim, top_k, metric, normalize, csls_k):
    """device [n1, ld], [n2, ld] (gold of row i = row i of t2) ->
    (rank int32[n1] device, argmax int32[n1] device, hits counts, rank_sum, rr_sum)."""
    kmetric, norm = device_metric(metric, normalize)
    if norm:
        t1, t2 = t1.clone(), t2.clone()
        ops.normalize_rows_(t1, dim, sklearn=True)
        ops.normalize_rows_(t2, dim, sklearn=True)
    from ...models import dist as mdist
    rk, ws = mdist.world()
    if ws > 1:
        return _greedy_alignment_sharded(t1, t2, dim, top_k, kmetric, csls_k, rk, ws)
    r = c = grid = None
    if csls_k > 0:
        r, c, grid = csls_means_device(t1, t2, dim, kmetric, csls_k, return_grid=True)
    if kmetric == 'inner' and 1 <= len(top_k) <= 8 and ops.tile_glds():
        if ops.eval_bf16_enabled(t1.shape[0], t2.shape[0]):
            # certified bf16 prefilter (with or without the CSLS means): the same ranks / nearest candidates at 3/16 of the fp32
            # matrix time; None = its record buffer overflowed (tables of near-duplicates): the fp32 sweep below
            res = ops.rank_eval_metrics_bf16(t1, t2, dim, top_k, csls_r=r, csls_c=c)
            if res is not None:
                return res
        return ops.rank_eval_metrics(t1, t2, dim, top_k, r, c)             # prologue + sweep: two launches, one copy back
    if grid is not None:       # manhattan + CSLS: the rank pass reads the strips the means pass left behind
        rank, argmax = ops.rank_eval_l1_grid(t1, t2, dim, csls_r=r, csls_c=c, grid=grid)
    else:
        rank, argmax = ops.rank_eval(t1, t2, dim, kmetric, r, c)
    hits, rank_sum, rr_sum = ops.rank_metrics(rank, top_k)
    return rank, argmax, hits, rank_sum, rr_sum


def _greedy_alignment_sharded(t1, t2, dim, top_k, kmetric, csls_k, rk, ws):
    """torch.distributed is initialised: this rank ranks its block of query rows against the full
    (replicated) candidate block -- no data-path collective; the integer sums are all-reduced and
    the per-row outputs all-gathered (models/dist.py).  CSLS: row means of the own block, column
    means of the own block of CANDIDATE rows (all-gathered)."""
    from ...models import dist as mdist
    n1, n2 = t1.shape[0], t2.shape[0]
    r = c = None
    if csls_k > 0:
        lo1, hi1 = mdist.shard_range(n1, rk, ws)
        lo2, hi2 = mdist.shard_range(n2, rk, ws)
        r_loc, _ = csls_means_device(t1[lo1:hi1], t2, dim, kmetric, csls_k, cols=False)
        c_loc, _ = csls_means_device(t2[lo2:hi2], t1, dim, kmetric, csls_k, cols=False)
        r, c = mdist.allgather_rows(r_loc, n1), mdist.allgather_rows(c_loc, n2)

    def rank_fn(block, cand, d, off):
        return ops.rank_eval(block, cand, d, kmetric, None if r is None else r[off: off + block.shape[0]].contiguous(),
                             c, gold_offset=off)
    hits, rank_sum, rr_sum, argmax = mdist.sharded_rank_metrics(t1, t2, dim, top_k, rank_fn)
    return None, argmax, hits, rank_sum, rr_sum


def greedy_alignment(embed1, embed2, top_k, nums_threads, metric, normalize, csls_k, accurate):
    """alignment.py:13-84: same arguments, same return (alignment_rest, hits1, mr, mrr), same
    log lines.  `nums_threads` is accepted and ignored (it only forked host workers)."""
    t = time.time()
    assert 1 in top_k
    e1 = embed1 if hasattr(embed1, "is_cuda") else ops.to_table(np.asarray(embed1, np.float32))
    e2 = embed2 if hasattr(embed2, "is_cuda") else ops.to_table(np.asarray(embed2, np.float32))
    dim = embed1.shape[1] if not hasattr(embed1, "is_cuda") else getattr(embed1, "oea_dim", embed1.shape[1])
    num = e1.shape[0]
    rank, argmax, hits_cnt, rank_sum, rr_sum = greedy_alignment_device(e1, e2, dim, top_k, metric, normalize, csls_k)
    am = argmax.cpu().numpy()
    alignment_rest = AlignmentPairs(am)               # (one pair per row: the reference's len(alignment_rest) == num holds)
    assert len(alignment_rest) == num
    hits = np.array(hits_cnt) / num * 100
    for i in range(len(hits)):
        hits[i] = round(hits[i], 3)
    mr = rank_sum / num
    mrr = rr_sum / num
    cost = time.time() - t
    if accurate:
        if csls_k > 0:
            print("accurate results with csls: csls={}, hits@{} = {}%, mr = {:.3f}, mrr = {:.6f}, time = {:.3f} s ".
                  format(csls_k, top_k, hits, mr, mrr, cost))
        else:
            print("accurate results: hits@{} = {}%, mr = {:.3f}, mrr = {:.6f}, time = {:.3f} s ".
                  format(top_k, hits, mr, mrr, cost))
    else:
        if csls_k > 0:
            print("quick results with csls: csls={}, hits@{} = {}%, time = {:.3f} s ".format(csls_k, top_k, hits, cost))
        else:
            print("quick results: hits@{} = {}%, time = {:.3f} s ".format(top_k, hits, cost))
    greedy_alignment.last = dict(rank=rank, argmax=argmax, hits_cnt=hits_cnt, rank_sum=rank_sum, rr_sum=rr_sum)
    return alignment_rest, hits[0], mr, mrr


def calculate_rank(idx, sim_mat, top_k, accurate, total_num):
    """alignment.py:146-168 on an explicit row block of a similarity matrix (host arrays):
    gold of row i is column idx[i]."""
    import torch
    assert 1 in top_k
    s = torch.from_numpy(np.ascontiguousarray(sim_mat, np.float32)).to(ops.device())
    rank, argmax = ops.rank_rows(s, ops.to_ids(np.asarray(idx, np.int32), s.device))
    rank_h = rank.cpu().numpy().astype(np.int64)
    mr = float((rank_h + 1).sum()) / total_num
    mrr = float((1.0 / (rank_h + 1)).sum()) / total_num
    hits = [int((rank_h < k).sum()) for k in top_k]
    hits1_rest = {(int(idx[i]), int(a)) for i, a in enumerate(argmax.cpu().numpy())}
    return mr, mrr, hits, hits1_rest


def galeshapley_topk(cand, cand_sim, sim_lookup, max_iteration):
    """The reference's Gale-Shapley loop (alignment.py:170-221) on truncated preference lists.

    cand[s]      suitor s's reviewers in preference order (the reference holds the full argsort; a suitor
                 drops at most one reviewer per round, so `max_iteration` entries are all it can ever use),
    cand_sim     the matching similarities (unused by the loop, kept for callers),
    sim_lookup   (s, r) -> similarity: "reviewer r prefers s to its partner" is `index(s) < index(partner)` in
                 r's argsort of its column in the reference, i.e. a larger similarity (ties: smaller suitor id).
    Round structure as in the reference: suitors are visited in list order, matches change immediately, a
    displaced suitor proposes again next round and only then drops the reviewer that left it.
    -> dict suitor -> reviewer."""
    n = len(cand)
    ptr = [0] * n
    matching, rev_matching = {}, {}
    suitors = list(range(n))
    for _ in range(max_iteration):
        if not suitors:
            break
        for s in suitors:
            if ptr[s] >= len(cand[s]):
                continue
            r = int(cand[s][ptr[s]])
            partner = rev_matching.get(r)
            if partner is None:
                matching[s] = r
                rev_matching[r] = s
            else:
                a, b = sim_lookup(s, r), sim_lookup(partner, r)
                if a > b or (a == b and s < partner):
                    del matching[partner]
                    matching[s] = r
                    rev_matching[r] = s
                else:
                    ptr[s] += 1
        suitors = sorted(set(range(n)) - set(matching.keys()))
    return matching


def arg_sort(idx, sim_mat, prefix1, prefix2):
    """alignment.py:136-143: {prefix1 + row id: [prefix2 + column, ...] by descending similarity} (host matrix)."""
    order = np.argsort(-np.asarray(sim_mat), axis=1)
    return {prefix1 + str(idx[i]): [prefix2 + str(j) for j in order[i]] for i in range(len(idx))}


def galeshapley(suitor_pref_dict, reviewer_pref_dict, max_iteration):
    """alignment.py:170-224 with the reference's round structure (every unmatched suitor proposes once per round to the
    head of its list; a rejected suitor drops that reviewer; at most max_iteration rounds), on rank tables instead of
    list.index / `in dict.values()` scans.  Mutates the suitors' lists like the reference does."""
    rank_of = {r: {s: i for i, s in enumerate(prefs)} for r, prefs in reviewer_pref_dict.items()}
    matching, holder = {}, {}
    waiting = list(suitor_pref_dict.keys())
    for _ in range(max_iteration):
        if not waiting:
            break
        for s in waiting:
            r = suitor_pref_dict[s][0]
            if r not in holder:
                matching[s], holder[r] = r, s
            elif rank_of[r][s] < rank_of[r][holder[r]]:
                del matching[holder[r]]
                matching[s], holder[r] = r, s
            else:
                suitor_pref_dict[s].remove(r)
        waiting = list(set(suitor_pref_dict.keys()) - set(matching.keys()))
    return matching


def stable_alignment(embed1, embed2, metric, normalize, csls_k, nums_threads, cut=100, sim_mat=None):
    """alignment.py:87-134: stable (Gale-Shapley, at most `cut` rounds) matching of the two embedding blocks and
    its precision.  The similarity block stays on the device; each suitor's `cut` best reviewers come from the
    top-k select kernel (sorted by value on the host) instead of a full n x n argsort in both directions."""
    import torch
    from .similarity import sim_device
    t = time.time()
    if sim_mat is None:
        e1 = embed1 if hasattr(embed1, "is_cuda") else ops.to_table(np.asarray(embed1, np.float32))
        e2 = embed2 if hasattr(embed2, "is_cuda") else ops.to_table(np.asarray(embed2, np.float32))
        dim = embed1.shape[1] if not hasattr(embed1, "is_cuda") else getattr(embed1, "oea_dim", embed1.shape[1])
        s = sim_device(e1, e2, dim, metric, normalize, csls_k)
    else:
        s = torch.from_numpy(np.ascontiguousarray(sim_mat, np.float32)).to(ops.device())
    n1, n2 = s.shape
    k = min(cut, n2)
    ld = (n2 + 31) // 32 * 32
    sp = torch.empty((n1, ld), dtype=torch.float32, device=s.device)       # the select kernel reads 16-byte aligned rows
    sp[:, :n2] = s
    idx = ops.topk_rows(sp, k, nc=n2).long()
    vals = torch.gather(s, 1, idx).cpu().numpy()
    idx = idx.cpu().numpy()
    order = np.lexsort((idx, -vals.astype(np.float64)), axis=1)            # value desc, column asc
    cand = np.take_along_axis(idx, order, axis=1)
    cand_sim = np.take_along_axis(vals, order, axis=1)
    print("generating candidate lists costs time {:.3f} s ".format(time.time() - t))
    t = time.time()
    cache = {}

    def lookup(i, j):
        v = cache.get((i, j))
        if v is None:
            v = cache[(i, j)] = float(s[i, j].item())
        return v
    for i in range(n1):                                                    # the candidates' own values are known already
        for j, v in zip(cand[i].tolist(), cand_sim[i].tolist()):
            cache[(i, j)] = v
    matching = galeshapley_topk(cand, cand_sim, lookup, cut)
    n = sum(1 for i, j in matching.items() if i == j)
    cost = time.time() - t
    print("stable alignment precision = {:.3f}%, time = {:.3f} s ".format(n / max(len(matching), 1) * 100, cost))
    return matching
