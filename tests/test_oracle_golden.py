"""Pin the oracle against the reference's own numpy functions (fixtures made by
tests/golden/make_golden.py, which imports the reference in place)."""
import json
import os

import numpy as np
import pytest

from oracle import cport, np_oracle as orc

TOP_K = [1, 5, 10, 50]
CASES = [('inner', False), ('inner', True), ('cosine', True), ('cosine', False),
         ('euclidean', False), ('manhattan', False)]


@pytest.fixture(scope="module")
def ali(golden_dir):
    return np.load(os.path.join(golden_dir, "alignment.npz"))


@pytest.mark.parametrize("metric,normalize", CASES)
@pytest.mark.parametrize("csls", [0, 10])
def test_greedy_alignment_matches_reference(ali, metric, normalize, csls):
    key = '%s_%d_%d' % (metric, int(normalize), csls)
    e1, e2 = ali['e1'], ali['e2']
    s = orc.sim(e1, e2, metric=metric, normalize=normalize, csls_k=csls)
    ref_rows = ali['sim_' + key]
    if metric == 'manhattan' and csls == 0:
        # fp64 sequential restatement of scipy cdist: bit-exact
        assert np.array_equal(s[:len(ref_rows)], ref_rows)
    else:
        np.testing.assert_allclose(s[:len(ref_rows)], ref_rows, rtol=0, atol=3e-6)
    rest, hits1, mr, mrr = orc.greedy_alignment(e1, e2, TOP_K, 1, metric, normalize, csls, True)
    last = orc.greedy_alignment.last
    ref_rank = ali['rank_' + key]
    # integer results: identical except where the reference's own fp32 summation order
    # produces a near-tie; the fixture was chosen so there are none.
    assert np.array_equal(last['rank'], ref_rank)
    assert np.array_equal(last['argmax'], ali['argmax_' + key])
    ref_hits1, ref_mr, ref_mrr = ali['stats_' + key]
    assert hits1 == ref_hits1
    assert abs(mr - ref_mr) < 1e-9 and abs(mrr - ref_mrr) < 1e-9
    assert ali['hits1_quick_' + key][0] == ref_hits1  # quick mode agrees on hits (reference)


def test_valid_with_mapping(ali):
    hits1, mrr = orc.valid(ali['e1'], ali['e2'], ali['mapping'], TOP_K, 1, metric='inner', normalize=True)
    assert hits1 == ali['valid_mapping'][0]
    # quick-mode MRR of the reference is only meaningful for rows ranked < 50


def test_csls_blocks(golden_dir):
    g = np.load(os.path.join(golden_dir, "csls.npz"))
    s = g['s']
    np.testing.assert_allclose(orc.calculate_nearest_k(s, 10), g['nearest_rows'], rtol=2e-6)
    np.testing.assert_allclose(orc.calculate_nearest_k(s.T, 10), g['nearest_cols'], rtol=2e-6)
    np.testing.assert_allclose(orc.csls_sim(s, 10), g['csls'], rtol=0, atol=2e-6)


def test_neighbours(golden_dir):
    g = np.load(os.path.join(golden_dir, "neighbours.npz"))
    k = int(g['k'])
    dic = orc.generate_neighbours(g['emb'], g['entity_list'].tolist(), k, 4)
    assert sorted(dic.keys()) == g['keys'].tolist()
    for key, ref in zip(g['keys'], g['nbrs']):
        got = dic[int(key)]
        assert len(got) == k and len(set(got)) == k
        assert int(key) in got                      # lists contain the entity itself
        assert got == ref.tolist()                  # no boundary ties in the fixture


def test_pos_batch_task_divide_early_stop(golden_dir):
    g = np.load(os.path.join(golden_dir, "pos_batch.npz"))
    t1 = [tuple(x) for x in g['t1'].tolist()]
    t2 = [tuple(x) for x in g['t2'].tolist()]
    for step in (0, 1, 3, 9):
        got = np.array(orc.generate_pos_batch(t1, t2, 200, step), np.int32).reshape(-1, 3)
        assert np.array_equal(got, g['step%d' % step])
    misc = json.load(open(os.path.join(golden_dir, "misc.json")))
    for key, ref in misc['task_divide'].items():
        total, n = map(int, key.split('_'))
        got = [list(map(int, x)) for x in orc.task_divide(list(range(total)), n)]
        assert got == ref
    for f1, f2, f, r0, r1, r2 in misc['early_stop']:
        assert orc.early_stop(f1, f2, f) == (r0, r1, r2)


def test_reference_sampler_invariants(golden_dir):
    """Properties of the REFERENCE sampler's output that the Philox restatement
    must share; the restatement itself is checked in test_sampler.py."""
    g = np.load(os.path.join(golden_dir, "neg_sampling.npz"))
    tri = set(map(tuple, g['triples'].tolist()))
    pos = g['pos']
    for name in ('neg_uniform', 'neg_truncated'):
        neg = g[name].reshape(len(pos), 10, 3)
        for p in range(len(pos)):
            h, r, t = pos[p]
            for nh, nr, nt in neg[p]:
                assert nr == r and ((nh == h) != (nt == t) or (nh == h and nt == t))
        frac_true = np.mean([tuple(x) in tri for x in g[name].tolist()])
        assert frac_true < 0.01


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    out = cport.philox([0, 0, 0, 0], [0, 0])
    assert [hex(x) for x in out] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    out = cport.philox([0xffffffff] * 4, [0xffffffff] * 2)
    assert [hex(x) for x in out] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    out = cport.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])
    assert [hex(x) for x in out] == ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_sparse_attention_oracle_gradient_by_finite_differences():
    rng = np.random.RandomState(0)
    n_rows, n_cols, d = 7, 9, 5
    seg_ptr = np.array([0, 3, 3, 4, 9, 11])          # an empty segment and two segments on one row
    seg_row = np.array([0, 1, 2, 4, 4])
    colidx = rng.randint(0, n_cols, 11)
    z = rng.standard_normal(11)
    v = rng.standard_normal((n_cols, d))
    w = rng.standard_normal((n_rows, d))             # loss = sum(out * w)
    out, alpha = orc.sparse_attn_forward(z, v, seg_ptr, seg_row, colidx, n_rows)
    dz, dv = orc.sparse_attn_backward(z, v, alpha, w, seg_ptr, seg_row, colidx)
    eps = 1e-6
    for e in range(11):
        zp, zm = z.copy(), z.copy()
        zp[e] += eps
        zm[e] -= eps
        num = ((orc.sparse_attn_forward(zp, v, seg_ptr, seg_row, colidx, n_rows)[0] -
                orc.sparse_attn_forward(zm, v, seg_ptr, seg_row, colidx, n_rows)[0]) * w).sum() / (2 * eps)
        assert abs(num - dz[e]) < 1e-6
    for (j, c) in [(0, 0), (3, 2), (8, 4), (int(colidx[5]), 1)]:
        vp, vm = v.copy(), v.copy()
        vp[j, c] += eps
        vm[j, c] -= eps
        num = ((orc.sparse_attn_forward(z, vp, seg_ptr, seg_row, colidx, n_rows)[0] -
                orc.sparse_attn_forward(z, vm, seg_ptr, seg_row, colidx, n_rows)[0]) * w).sum() / (2 * eps)
        assert abs(num - dv[j, c]) < 1e-6


def test_transh_oracle_gradients_by_finite_differences():
    """oracle_triple_step_transh restates TF1 autodiff of bootea_transh.py:58-96 by hand (PARITY UNPINNED vs TF):
    pin the hand-derived gradients against central differences of the loss written independently in numpy."""
    from oracle import cport
    rng = np.random.RandomState(0)
    n_ent, n_rel, d = 12, 3, 7
    ent = rng.standard_normal((n_ent, d)).astype(np.float32)
    rel = rng.standard_normal((n_rel, d)).astype(np.float32)
    nrm = rng.standard_normal((n_rel, d)).astype(np.float32)
    pos = np.array([[0, 1, 2], [3, 1, 4], [5, 0, 6]], np.int32)
    neg = np.array([[0, 1, 7], [8, 1, 2], [3, 1, 9], [5, 0, 1], [10, 0, 6], [5, 0, 11]], np.int32)

    def l2n(x):
        return x / np.sqrt(np.maximum((x * x).sum(1, keepdims=True), 1e-12))

    def loss_of(e, r, n):
        e, r, n = l2n(e), l2n(r), l2n(l2n(n))

        def sc(tr):
            h, t, rr, nn = e[tr[:, 0]], e[tr[:, 2]], r[tr[:, 1]], n[tr[:, 1]]
            hp = h - (h * nn).sum(1, keepdims=True) * nn
            tp = t - (t * nn).sum(1, keepdims=True) * nn
            return ((hp + rr - tp) ** 2).sum(1)
        return np.maximum(sc(pos) - 0.01, 0).sum() + 0.2 * np.maximum(2.0 - sc(neg), 0).sum()

    lr = 1e-3
    tabs = [ent.copy(), rel.copy(), nrm.copy()]
    loss = cport.triple_step_transh(tabs[0], None, tabs[1], None, tabs[2], None, pos, neg, loss="limited", pos_margin=0.01,
                                    neg_margin=2.0, balance=0.2, optimizer="SGD", lr=lr)
    base = [ent.astype(np.float64), rel.astype(np.float64), nrm.astype(np.float64)]
    assert abs(loss - loss_of(*base)) < 1e-6
    eps = 1e-4
    for idx in range(3):
        analytic = (base[idx] - tabs[idx]) / lr                    # SGD: v' = v - lr * grad
        num = np.zeros_like(analytic)
        for i in range(analytic.shape[0]):
            for k in range(d):
                a = [x.copy() for x in base]
                b = [x.copy() for x in base]
                a[idx][i, k] += eps
                b[idx][i, k] -= eps
                num[i, k] = (loss_of(*a) - loss_of(*b)) / (2 * eps)
        assert np.abs(analytic - num).max() < 5e-4 * max(np.abs(num).max(), 1.0)   # fp32 table rounding of the update


@pytest.mark.parametrize("loss,l1", [("limited", False), ("margin-based", False), ("logistic", False), ("limited", True)])
def test_triple_step_oracle_gradients_by_finite_differences(loss, l1):
    """oracle_triple_step (losses.py:15-73 + l2_normalize + SGD) against central differences of the loss written
    independently in numpy -- pins the hand-derived TF1 gradient (through the normalisation, duplicates summed)."""
    from oracle import cport
    rng = np.random.RandomState(1)
    n_ent, n_rel, d = 10, 3, 6
    ent = rng.standard_normal((n_ent, d)).astype(np.float32)
    rel = rng.standard_normal((n_rel, d)).astype(np.float32)
    pos = np.array([[0, 1, 2], [3, 1, 4], [0, 0, 6], [7, 2, 0]], np.int32)           # entity 0 occurs three times
    neg = np.array([[0, 1, 7], [8, 1, 4], [9, 0, 6], [7, 2, 5]], np.int32)

    def l2n(x):
        return x / np.sqrt(np.maximum((x * x).sum(1, keepdims=True), 1e-12))

    def score(e, r, tr):
        dd = e[tr[:, 0]] + r[tr[:, 1]] - e[tr[:, 2]]
        return np.abs(dd).sum(1) if l1 else (dd * dd).sum(1)

    def loss_of(e, r):
        e, r = l2n(e), l2n(r)
        sp_, sn = score(e, r, pos), score(e, r, neg)
        if loss == "margin-based":
            return np.maximum(1.5 + sp_ - sn, 0).sum()
        if loss == "limited":
            return np.maximum(sp_ - 0.01, 0).sum() + 0.2 * np.maximum(2.0 - sn, 0).sum()
        return np.log1p(np.exp(sp_)).sum() + np.log1p(np.exp(-sn)).sum()

    lr = 1e-3
    e1, r1 = ent.copy(), rel.copy()
    got = cport.triple_step(e1, None, r1, None, pos, neg, loss=loss, loss_norm="L1" if l1 else "L2", margin=1.5,
                            pos_margin=0.01, neg_margin=2.0, balance=0.2, optimizer="SGD", lr=lr)
    base = [ent.astype(np.float64), rel.astype(np.float64)]
    assert abs(got - loss_of(*base)) < 1e-6 * max(1.0, abs(got))
    eps = 1e-5 if l1 else 1e-4
    for idx, new in enumerate((e1, r1)):
        analytic = (base[idx] - new) / lr
        num = np.zeros_like(analytic)
        for i in range(analytic.shape[0]):
            for k in range(d):
                a = [x.copy() for x in base]
                b = [x.copy() for x in base]
                a[idx][i, k] += eps
                b[idx][i, k] -= eps
                num[i, k] = (loss_of(*a) - loss_of(*b)) / (2 * eps)
        assert np.abs(analytic - num).max() < 5e-4 * max(np.abs(num).max(), 1.0)


def test_gcn_epoch_oracle_gradient_by_finite_differences():
    """gcn_se_epoch (gcn_align.py:204-320,498-539 restated): dW against central differences of the hinge loss."""
    from oracle import np_oracle as orc
    rng = np.random.RandomState(2)
    n, d, t, k = 14, 5, 4, 3
    W = rng.standard_normal((n, d)).astype(np.float32)
    coords = np.array([(i, j) for i in range(n) for j in range(n) if rng.rand() < 0.3], np.int64)
    values = rng.rand(len(coords)).astype(np.float32)
    ILL = np.stack([rng.permutation(n)[:t], rng.permutation(n)[:t]], 1)
    negs = (np.repeat(ILL[:, 0], k), rng.randint(0, n, t * k), rng.randint(0, n, t * k), np.repeat(ILL[:, 1], k))
    import scipy.sparse as sp
    A = sp.csr_matrix((values.astype(np.float64), (coords[:, 0], coords[:, 1])), shape=(n, n))

    def loss_of(Wd):
        T = Wd / np.sqrt(np.maximum((Wd ** 2).sum(1, keepdims=True), 1e-12))
        out = A @ np.maximum(A @ T, 0.0)
        a = np.abs(out[ILL[:, 0]] - out[ILL[:, 1]]).sum(1)
        l = 0.0
        for nl, nr in ((negs[0], negs[1]), (negs[2], negs[3])):
            b = np.abs(out[nl] - out[nr]).sum(1).reshape(t, k)
            l += np.maximum(a[:, None] + 3.0 - b, 0).sum()
        return l / (2.0 * k * t)

    lr = 1e-3
    W1 = W.copy()
    loss, _ = orc.gcn_se_epoch(W1, coords, values, ILL, 3.0, k, negs, lr)
    Wd = W.astype(np.float64)
    assert abs(loss - loss_of(Wd)) < 1e-9
    analytic = (Wd - W1) / lr
    num = np.zeros_like(analytic)
    eps = 1e-6
    for i in range(n):
        for c in range(d):
            a, b = Wd.copy(), Wd.copy()
            a[i, c] += eps
            b[i, c] -= eps
            num[i, c] = (loss_of(a) - loss_of(b)) / (2 * eps)
    assert np.abs(analytic - num).max() < 5e-4 * max(np.abs(num).max(), 1.0)


@pytest.mark.parametrize("loss,l1", [("margin-based", False), ("limited", False), ("margin-based", True)])
def test_transd_oracle_gradients_by_finite_differences(loss, l1):
    """oracle_triple_step_transd restates TF1 autodiff of models/trans/transd.py:16-57 by hand (PARITY UNPINNED vs
    TF): pin the gradients of all four variables against central differences of the loss written in numpy."""
    from oracle import cport
    rng = np.random.RandomState(3)
    E, R, d = 9, 3, 6
    ent = rng.standard_normal((2 * E, d)).astype(np.float32)       # [ent_embeds ; ent_transfer]
    rel = rng.standard_normal((2 * R, d)).astype(np.float32)       # [rel_embeds ; rel_transfer]
    pos = np.array([[0, 1, 2], [3, 1, 4], [0, 0, 6], [7, 2, 0]], np.int32)
    neg = np.array([[0, 1, 7], [8, 1, 4], [5, 0, 6], [7, 2, 5]], np.int32)

    def l2n(x):
        return x / np.sqrt(np.maximum((x * x).sum(1, keepdims=True), 1e-12))

    def loss_of(e, r):
        e, r = l2n(e), l2n(r)

        def sc(tr):
            h, t, hp, tp = e[tr[:, 0]], e[tr[:, 2]], e[E + tr[:, 0]], e[E + tr[:, 2]]
            rr, rp = r[tr[:, 1]], r[R + tr[:, 1]]
            hh = l2n(h + (h * hp).sum(1, keepdims=True) * rp)
            tt = l2n(t + (t * tp).sum(1, keepdims=True) * rp)
            dlt = hh + rr - tt
            return np.abs(dlt).sum(1) if l1 else (dlt ** 2).sum(1)
        if loss == "margin-based":
            return np.maximum(0.7 + sc(pos) - sc(neg), 0).sum()
        return np.maximum(sc(pos) - 0.01, 0).sum() + 0.2 * np.maximum(2.5 - sc(neg), 0).sum()

    lr = 1e-3
    tabs = [ent.copy(), rel.copy()]
    got = cport.triple_step_transd(tabs[0], None, tabs[1], None, pos, neg, loss=loss, loss_norm="L1" if l1 else "L2",
                                   margin=0.7, pos_margin=0.01, neg_margin=2.5, balance=0.2, optimizer="SGD", lr=lr)
    base = [ent.astype(np.float64), rel.astype(np.float64)]
    assert abs(got - loss_of(*base)) < 1e-6
    assert got > 0
    eps = 1e-5
    for idx in range(2):
        analytic = (base[idx] - tabs[idx]) / lr
        num = np.zeros_like(analytic)
        for i in range(analytic.shape[0]):
            for k in range(d):
                a = [x.copy() for x in base]
                b = [x.copy() for x in base]
                a[idx][i, k] += eps
                b[idx][i, k] -= eps
                num[i, k] = (loss_of(*a) - loss_of(*b)) / (2 * eps)
        assert np.abs(num).max() > 1e-2
        assert np.abs(analytic - num).max() < 5e-4 * max(np.abs(num).max(), 1.0)


def test_transh_margin_oracle_gradients_by_finite_differences():
    """the margin-loss form of the TransH step (models/trans/transh.py:16-51: margin_loss on projected rows)."""
    from oracle import cport
    rng = np.random.RandomState(4)
    n_ent, n_rel, d = 10, 3, 5
    ent = rng.standard_normal((n_ent, d)).astype(np.float32)
    rel = rng.standard_normal((n_rel, d)).astype(np.float32)
    nrm = rng.standard_normal((n_rel, d)).astype(np.float32)
    pos = np.array([[0, 1, 2], [3, 1, 4], [5, 0, 6]], np.int32)
    neg = np.array([[0, 1, 7], [8, 1, 4], [5, 0, 9]], np.int32)

    def l2n(x):
        return x / np.sqrt(np.maximum((x * x).sum(1, keepdims=True), 1e-12))

    def loss_of(e, r, n):
        e, r, n = l2n(e), l2n(r), l2n(l2n(n))

        def sc(tr):
            h, t, rr, nn = e[tr[:, 0]], e[tr[:, 2]], r[tr[:, 1]], n[tr[:, 1]]
            return ((h - (h * nn).sum(1, keepdims=True) * nn + rr - t + (t * nn).sum(1, keepdims=True) * nn) ** 2).sum(1)
        return np.maximum(1.5 + sc(pos) - sc(neg), 0).sum()

    lr = 1e-3
    tabs = [ent.copy(), rel.copy(), nrm.copy()]
    got = cport.triple_step_transh(tabs[0], None, tabs[1], None, tabs[2], None, pos, neg, loss="margin-based", margin=1.5,
                                   optimizer="SGD", lr=lr)
    base = [ent.astype(np.float64), rel.astype(np.float64), nrm.astype(np.float64)]
    assert abs(got - loss_of(*base)) < 1e-6 and got > 0
    eps = 1e-5
    for idx in range(3):
        analytic = (base[idx] - tabs[idx]) / lr
        num = np.zeros_like(analytic)
        for i in range(analytic.shape[0]):
            for k in range(d):
                a = [x.copy() for x in base]
                b = [x.copy() for x in base]
                a[idx][i, k] += eps
                b[idx][i, k] -= eps
                num[i, k] = (loss_of(*a) - loss_of(*b)) / (2 * eps)
        assert np.abs(analytic - num).max() < 5e-4 * max(np.abs(num).max(), 1.0)


@pytest.mark.parametrize("ent_norm,rel_norm", [(True, False), (True, True), (False, False)])
def test_rotate_oracle_gradients_by_finite_differences(ent_norm, rel_norm):
    """np_oracle.rotate_step restates TF1 autodiff of bootea_rotate.py:59-109 by hand (PARITY UNPINNED vs TF): pin the
    gradients to the real / imaginary / phase tables against central differences of rotate_loss, which is written the
    way the TF graph reads."""
    rng = np.random.RandomState(7)
    E, R, d = 8, 3, 5
    ent = rng.standard_normal((2 * E, d)) * 0.5
    rel = rng.standard_normal((R, d)) * 0.3
    pos = np.array([[0, 1, 2], [3, 1, 4], [0, 0, 6]])
    neg = np.array([[0, 1, 7], [5, 1, 2], [3, 1, 0], [0, 0, 1], [2, 0, 6], [7, 2, 6]])
    kw = dict(gamma=1.2, phase_scale=2.6, ent_l2_norm=ent_norm, rel_l2_norm=rel_norm)
    lr = 1e-6
    e1, r1 = ent.copy(), rel.copy()
    got = orc.rotate_step(e1, r1, pos, neg, {}, optimizer="SGD", lr=lr, **kw)
    assert abs(got - orc.rotate_loss(ent, rel, pos, neg, **kw)) < 1e-10
    eps = 1e-6
    for idx, (base, after) in enumerate(((ent, e1), (rel, r1))):
        analytic = (base - after) / lr
        num = np.zeros_like(base)
        for i in range(base.shape[0]):
            for k in range(d):
                a, b = [ent.copy(), rel.copy()], [ent.copy(), rel.copy()]
                a[idx][i, k] += eps
                b[idx][i, k] -= eps
                num[i, k] = (orc.rotate_loss(a[0], a[1], pos, neg, **kw) - orc.rotate_loss(b[0], b[1], pos, neg, **kw)) / (2 * eps)
        assert np.abs(num).max() > 1e-2
        assert np.abs(analytic - num).max() < 1e-6 * max(np.abs(num).max(), 1.0)


def test_rotate_oracle_adam_moves_every_row():
    """tf.train.AdamOptimizer semantics: rows without gradient keep moving while their first moment decays."""
    rng = np.random.RandomState(8)
    ent, rel = rng.standard_normal((8, 4)), rng.standard_normal((2, 4))
    st = {}
    kw = dict(gamma=1.0, phase_scale=2.0, ent_l2_norm=False, rel_l2_norm=False, optimizer="Adam", lr=0.01)
    orc.rotate_step(ent, rel, np.array([[0, 0, 1]]), None, st, **kw)
    before = ent.copy()
    orc.rotate_step(ent, rel, np.array([[2, 1, 3]]), None, st, **kw)      # rows 0, 1 (and 4, 5) get no gradient now
    assert st["t"] == 2 and np.abs(ent[0] - before[0]).max() > 1e-4 and np.abs(ent[4] - before[4]).max() > 1e-4


# ---------------------------------------------------------------------------------------------------------------------
# The reference's OWN graph code (losses.py, basic_model.py, the approach classes' _define_*_graph methods), executed
# under tests/golden/tf_shim.py by tests/golden/make_tf_graph_golden.py: loss value + finite-difference gradients
# w.r.t. every variable.  The oracle's step must reproduce both (its gradient read back from one SGD step).
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tfg(golden_dir):
    return np.load(os.path.join(golden_dir, "tf_graphs.npz"))


def _sgd_grads(before, after, lr):
    return [(b.astype(np.float64) - a.astype(np.float64)) / lr for b, a in zip(before, after)]


def _check(tfg, tag, names, loss, grads, tol=5e-4):
    ref_loss = float(tfg[tag + "_loss"][0])
    assert abs(loss - ref_loss) <= 1e-6 * max(abs(ref_loss), 1.0), (tag, loss, ref_loss)
    for name, g in zip(names, grads):
        ref = tfg["%s_grad_%s" % (tag, name)]
        assert np.abs(ref).max() > 1e-3
        assert np.abs(g - ref).max() <= tol * max(np.abs(ref).max(), 1.0), (tag, name, np.abs(g - ref).max())


@pytest.mark.parametrize("tag,kw,neg_key", [
    ("aligne_triple", dict(loss="limited", pos_margin=0.01, neg_margin=2.0, balance=0.2), "neg2"),
    ("bootea_triple", dict(loss="limited", pos_margin=0.01, neg_margin=2.0, balance=0.2), "neg2"),
    ("bootea_align", dict(loss="align"), None),
    ("mtranse_triple", dict(loss="positive"), None),
    ("transe_triple", dict(loss="margin-based", margin=1.5), "neg1"),
])
def test_triple_step_oracle_equals_reference_graph(tfg, tag, kw, neg_key):
    """basic_model.py:80-98 / aligne.py:47-66 / bootea.py:190-199 / mtranse.py:46-57 / models/trans/transe.py:31-49."""
    ent = tfg[tag + "_var_ent_embeds"].astype(np.float32)
    rel = tfg[tag + "_var_rel_embeds"].astype(np.float32)
    e1, r1 = ent.copy(), rel.copy()
    lr = 1e-3
    loss = cport.triple_step(e1, None, r1, None, tfg["pos"], tfg[neg_key] if neg_key else None, loss_norm="L2",
                             ent_l2_norm=True, rel_l2_norm=True, optimizer="SGD", lr=lr, **kw)
    _check(tfg, tag, ["ent_embeds", "rel_embeds"], loss, _sgd_grads([ent, rel], [e1, r1], lr))


@pytest.mark.parametrize("tag,kw,neg_key", [
    ("bootea_transh_triple", dict(loss="limited", pos_margin=0.01, neg_margin=2.0, balance=0.2), "neg2"),
    ("transh_triple", dict(loss="margin-based", margin=1.5), "neg1"),
])
def test_transh_step_oracle_equals_reference_graph(tfg, tag, kw, neg_key):
    """bootea_transh.py:58-96 and models/trans/transh.py:16-51 (normal vectors normalised at creation and in _calc)."""
    tabs = [tfg["%s_var_%s" % (tag, n)].astype(np.float32) for n in ("ent_embeds", "rel_embeds", "normal_vector")]
    new = [t.copy() for t in tabs]
    lr = 1e-3
    loss = cport.triple_step_transh(new[0], None, new[1], None, new[2], None, tfg["pos"], tfg[neg_key], loss_norm="L2",
                                    ent_l2_norm=True, rel_l2_norm=True, optimizer="SGD", lr=lr, **kw)
    _check(tfg, tag, ["ent_embeds", "rel_embeds", "normal_vector"], loss, _sgd_grads(tabs, new, lr))


def test_transd_step_oracle_equals_reference_graph(tfg):
    """models/trans/transd.py:16-57: four variables; the oracle (like the device step) takes them stacked."""
    tag = "transd_triple"
    v = {n: tfg["%s_var_%s" % (tag, n)].astype(np.float32) for n in ("ent_embeds", "rel_embeds", "ent_transfer", "rel_transfer")}
    ent = np.concatenate([v["ent_embeds"], v["ent_transfer"]])
    rel = np.concatenate([v["rel_embeds"], v["rel_transfer"]])
    e1, r1 = ent.copy(), rel.copy()
    lr = 1e-3
    loss = cport.triple_step_transd(e1, None, r1, None, tfg["pos"], tfg["neg1"], loss="margin-based", margin=1.5, loss_norm="L2",
                                    ent_l2_norm=True, rel_l2_norm=True, optimizer="SGD", lr=lr)
    ge, gr = _sgd_grads([ent, rel], [e1, r1], lr)
    E, R = len(v["ent_embeds"]), len(v["rel_embeds"])
    _check(tfg, tag, ["ent_embeds", "ent_transfer", "rel_embeds", "rel_transfer"], loss, [ge[:E], ge[E:], gr[:R], gr[R:]])


def test_mapping_step_oracle_equals_reference_graph(tfg):
    """modules/base/mapping.py:9-19 + losses.py:76-80 (alpha * (map loss + orthogonality loss))."""
    tag = "mtranse_mapping"
    ent = tfg[tag + "_var_ent_embeds"].astype(np.float32)
    M = tfg[tag + "_var_mapping_matrix"].astype(np.float32)
    e1, m1 = ent.copy(), M.copy()
    big = 1e12                                   # Adagrad with a huge accumulator = a plain gradient step of lr / 1e6
    loss = orc.mapping_step(e1, m1, np.full_like(ent, big), np.full_like(M, big), np.array([[0, 2], [3, 4], [8, 9]]),
                            alpha=float(tfg["mtranse_alpha"][0]), lr=1e3, ent_l2_norm=True)
    _check(tfg, tag, ["ent_embeds", "mapping_matrix"], loss, _sgd_grads([ent, M], [e1, m1], 1e-3), tol=2e-3)


@pytest.mark.parametrize("tag,neg_key", [("rotate_triple", "neg2"), ("rotate_align", None)])
def test_rotate_oracle_equals_reference_graph(tfg, tag, neg_key):
    """bootea_rotate.py:59-109 (lookup_all, _generate_scores, _generate_loss) and :148-158, in float64."""
    ent = np.concatenate([tfg[tag + "_var_re_ent_embeds"], tfg[tag + "_var_im_ent_embeds"]])
    rel = tfg[tag + "_var_rel_embeds"].copy()
    kw = dict(gamma=float(tfg["rotate_gamma"][0]), phase_scale=float(tfg["rotate_phase_scale"][0]), ent_l2_norm=True, rel_l2_norm=False)
    neg = tfg[neg_key] if neg_key else None
    assert abs(orc.rotate_loss(ent, rel, tfg["pos"], neg, **kw) - float(tfg[tag + "_loss"][0])) < 1e-10
    e1, r1 = ent.copy(), rel.copy()
    lr = 1e-6
    loss = orc.rotate_step(e1, r1, tfg["pos"], neg, {}, optimizer="SGD", lr=lr, **kw)
    ge, gr = _sgd_grads([ent, rel], [e1, r1], lr)
    E = len(ent) // 2
    _check(tfg, tag, ["re_ent_embeds", "im_ent_embeds", "rel_embeds"], loss, [ge[:E], ge[E:], gr], tol=1e-5)


@pytest.mark.parametrize("tag", ["gcn_se", "gcn_ae"])
def test_gcn_unit_oracle_equals_reference_graph(tfg, tag):
    """gcn_align.py:498-539 (GCN_Align_Unit: two GraphConvolution layers + align_loss), structure and attribute units,
    built by the reference's own code under the numpy TensorFlow stand-in."""
    import scipy.sparse as sp
    W = tfg[tag + "_var_weights"].astype(np.float32)
    coords, values = tfg["gcn_support_coords"], tfg["gcn_support_values"]
    n = int(coords.max()) + 1
    feats = None
    if tag == "gcn_ae":
        fc = tfg["gcn_feat_coords"]
        feats = sp.csr_matrix((np.ones(len(fc)), (fc[:, 0], fc[:, 1])), shape=(n, W.shape[0]))
    negs = [tfg["gcn_" + k] for k in ("neg_left", "neg_right", "neg2_left", "neg2_right")]
    w1 = W.copy()
    lr = 1e-3
    loss, out = orc.gcn_se_epoch(w1, coords, values, tfg["gcn_ill"], 3.0, 3, negs, lr, features=feats)
    np.testing.assert_allclose(out, tfg[tag + "_outputs"], rtol=1e-5, atol=1e-6)
    _check(tfg, tag, ["weights"], loss, _sgd_grads([W], [w1], lr))


@pytest.mark.parametrize("mode", ["truncated", "uniform"])
def test_sampler_statistics_match_reference(golden_dir, mode):
    """The Philox restatement of generate_neg_triples_fast (batch.py:89-119) cannot agree with python's `random` draw by
    draw; over 300 batches it must agree in distribution with 300 runs of the REFERENCE sampler
    (tests/golden/neg_stats.npz): how often a positive's k negatives all corrupt the head, how often a family mixes both
    sides (a re-draw after a true triple flips a new coin), and the same histogram over the candidate-list positions."""
    g = np.load(os.path.join(golden_dir, "neg_sampling.npz"))
    st = np.load(os.path.join(golden_dir, "neg_stats.npz"))
    runs, k = int(st["runs"][0]), 10
    pos, ents = g["pos"], g["entity_list"]
    table = cport.tripleset_build(g["triples"])
    ent_pos = np.full(int(ents.max()) + 1, -1, np.int32)
    ent_pos[ents] = np.arange(len(ents), dtype=np.int32)
    nbr = g["nbr"] if mode == "truncated" else None
    n_cand = nbr.shape[1] if nbr is not None else len(ents)
    pure = mixed = 0
    hist = np.zeros(n_cand, np.int64)
    where = {int(e): i for i, e in enumerate(ents)}
    for step in range(runs):
        neg = cport.sample_negatives(pos, k, table, ents, ent_pos if nbr is not None else None, nbr, seed=77, step=step)
        neg = neg.reshape(len(pos), k, 3)
        head_side = neg[:, :, 0] != pos[:, None, 0]
        tail_side = neg[:, :, 2] != pos[:, None, 2]
        pure += int(head_side.all(1).sum())
        mixed += int((head_side.any(1) & tail_side.any(1)).sum())
        drawn = np.where(head_side, neg[:, :, 0], neg[:, :, 2])
        owner = np.where(head_side, pos[:, None, 0], pos[:, None, 2])
        valid = head_side | tail_side
        if nbr is None:
            idx = ent_pos[drawn[valid]]
        else:                                   # position of the drawn entity in the corrupted entity's neighbour list
            rows = nbr[ent_pos[owner[valid]]]
            idx = (rows == drawn[valid][:, None]).argmax(1)
        hist += np.bincount(idx, minlength=n_cand)
    families = runs * len(pos)
    for ours, key in ((pure, mode + "_pure_head"), (mixed, mode + "_mixed")):
        ref = int(st[key][0])
        assert abs(ours - ref) / families < 0.012, (key, ours, ref)          # sigma of the difference ~ 0.0025
    ref_hist = st[mode + "_hist"].astype(np.float64)
    ours_n, ref_n = hist / hist.sum(), ref_hist / ref_hist.sum()
    # neither histogram is flat (positions that would give a true triple are rejected and re-drawn): the two samplers
    # must show the same profile; sigma of the difference of two multinomial frequencies ~ sqrt(2 p / N)
    assert np.abs(ours_n - ref_n).max() < 6 * np.sqrt(2 * ref_n.max() / ref_hist.sum())


def _families(neg, k):
    """a positive's k negatives as a sorted list (the reference appends a round's negatives in python-set order)"""
    neg = np.asarray(neg).reshape(-1, k, 3)
    return [sorted(map(tuple, fam.tolist())) for fam in neg]


@pytest.mark.parametrize("case", ["truncated", "uniform", "dense"])
def test_sampler_replay_equals_the_reference_run(golden_dir, case):
    """generate_neg_triples_fast itself (modules/train/batch.py:89-119), its python random numbers recorded
    (tests/golden/make_neg_replay_golden.py): the oracle's restatement of the algorithm, fed the SAME draws, yields the reference's
    negatives -- every positive's family of k, as a multiset.  Half of the positives of these fixtures need more than one round
    (true triples drawn and removed), the dense case runs into the last round, which keeps them."""
    from oracle import np_oracle
    g = np.load(os.path.join(golden_dir, "neg_replay.npz"))
    if case == "dense":
        tri, ents, pos, k, max_try = g["dense_triples"], g["dense_entities"], g["dense_pos"], 5, 3
        ch = ct = [ents] * len(pos)
    else:
        s = np.load(os.path.join(golden_dir, "neg_sampling.npz"))
        tri, ents, pos, k, max_try = s["triples"], s["entity_list"], s["pos"], 10, 10
        row = {int(e): i for i, e in enumerate(ents)}
        ch = [s["nbr"][row[int(h)]] if case == "truncated" else ents for h in pos[:, 0]]
        ct = [s["nbr"][row[int(t)]] if case == "truncated" else ents for t in pos[:, 2]]
    replay, ref = g["replay_" + case], g["neg_" + case]
    got = np_oracle.sample_negatives_replay(pos, k, tri, ch, ct, replay, max_try)
    assert _families(got, k) == _families(ref, k)
    rounds = (replay[:, :, 0] >= 0).sum(1)
    assert (rounds > 1).sum() >= len(pos) // 3                      # the record exercises the retry rounds
    if case == "dense":
        assert rounds.max() == max_try                              # ... and the last round, which keeps true triples
        tset = set(map(tuple, tri.tolist()))
        assert any(tuple(x) in tset for x in ref.tolist())


@pytest.mark.parametrize("n", [1, 2, 3, 17, 4096, 4097, 100003])
def test_epoch_layout_permutation_is_a_bijection(n):
    """oracle.epoch_layout_perm (the restatement the device layout is held to; basic_model.py:234-235 shuffles with Python's MT,
    any uniform permutation is the same algorithm): every index exactly once for any n, another epoch / list / seed another
    permutation, and no trace of the identity (a position's image is uncorrelated with it)."""
    from oracle import np_oracle as orc
    p = orc.epoch_layout_perm(n, 17, 3, False)
    assert np.array_equal(np.sort(p), np.arange(n))
    if n >= 4096:
        q, r, t = orc.epoch_layout_perm(n, 17, 4, False), orc.epoch_layout_perm(n, 17, 3, True), orc.epoch_layout_perm(n, 18, 3, False)
        for other in (q, r, t):
            assert (p == other).mean() < 0.01
            assert abs(np.corrcoef(p, other)[0, 1]) < 0.05
        assert abs(np.corrcoef(np.arange(n), p)[0, 1]) < 0.05
        assert abs(np.abs(np.diff(p)).mean() / n - 1 / 3) < 0.02          # neighbours land a third of the list apart, as for uniform draws
