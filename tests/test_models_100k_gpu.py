"""Model-level parity at the 100K shapes (BASELINE.json configs 3 / 4: AliNet on EN-DE-100K, RDGCN on EN-FR-100K; GCN-Align
beside them): every layer of the device models is held to a float64 numpy restatement of the cited reference lines,
fed with the DEVICE's input of that layer, on hub rows + sampled rows (a full float64 forward of a 200,000 x 500
model is minutes of host time; per-layer checks on checked inputs leave no layer unverified).  GCN-Align's
full-batch epoch runs against the oracle's epoch (scipy, whole graph).  One optimiser step per model shows the backward
runs at this size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

BN_SCALE = 1.0 / np.sqrt(1.0 + 1e-3)          # keras BatchNormalization, inference mode, initial statistics


def _args(name, tmp_path, shape, **kw):
    from openea_amd.run.default_args import get_args
    return get_args(name, output=str(tmp_path) + "/out/", training_data="synthetic/%s/" % shape, dataset_division="fold1/", **kw)


def _rows(rng, n, hubs=8, rand=56):
    return np.unique(np.concatenate([np.arange(hubs), rng.choice(n, rand, replace=False)]))


def _h(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _edges(g):
    return g.e_rows.cpu().numpy(), g.e_cols.cpu().numpy(), g.e_vals.cpu().numpy().astype(np.float64)


def _csr_rows(op, rows):
    """rows of a CsrOperand as (cols, vals) lists"""
    rp, ci, va = op.rowptr.cpu().numpy(), op.colidx.cpu().numpy(), op.vals.cpu().numpy().astype(np.float64)
    return [(ci[rp[r]:rp[r + 1]], va[rp[r]:rp[r + 1]]) for r in rows]


def _attention_rows(er, ec, z, v_of, rows, grouping, slope=0.2):
    """tf.sparse_softmax + sparse_tensor_dense_matmul for the output rows `rows`: softmax groups = whole rows ('row') or
    maximal runs of consecutive equal rows in the fed edge order ('runs'); out[r] = sum over its groups."""
    out = []
    for r in rows:
        pos = np.flatnonzero(er == r)
        acc = 0.0
        if len(pos):
            cuts = np.flatnonzero(np.diff(pos) != 1) + 1 if grouping != 'row' else np.zeros(0, np.int64)
            for run in np.split(pos, cuts):
                lg = np.where(z[run] > 0, z[run], slope * z[run])
                a = np.exp(lg - lg.max())
                a /= a.sum()
                acc = acc + a @ v_of(ec[run])
        out.append(acc if len(pos) else None)
    return out


def _close(dev_rows, ref_rows, what, atol=2e-4):
    worst = 0.0
    for d, r in zip(dev_rows, ref_rows):
        if r is None:
            assert not d.any(), what
            continue
        worst = max(worst, float(np.abs(d - r).max()))
    assert worst <= atol, "%s: max abs deviation %.3e" % (what, worst)
    return worst


def test_alinet_layers_100k_shape(tmp_path, capsys):
    """alinet.py:539-677, 784-850 at the EN-DE-100K shape, layer_dims of the shipped args (500-400-300)."""
    from openea_amd.approaches import AliNet
    from openea_amd.modules.load.synth import make_kgs
    kgs = make_kgs("EN-DE-100K-V1", mode="mapping", seed=0)
    m = AliNet()
    m.set_args(_args("AliNet", tmp_path, "EN-DE-100K-V1", max_epoch=1, start_valid=1000))
    m.set_kgs(kgs)
    m.init()
    rng = np.random.RandomState(11)
    rows = _rows(rng, kgs.entities_num)
    report = []
    with torch.no_grad():
        x = m.init_embedding
        layer_num = len(m.args.layer_dims) - 1
        for i in range(layer_num):
            gc = m.one_hop_layers[i]
            one = gc.call(x)
            xh = None

            def bn_rows(ids, bn, src=x):
                return _h(src[torch.from_numpy(np.asarray(ids, np.int64)).to(src.device)]) * (_h(bn.gamma) * BN_SCALE) + _h(bn.beta)
            # GraphConvolution: tanh(A (BN(x) W) + b)
            K, b = _h(gc.kernel), _h(gc.bias)
            ref = [np.tanh(vals @ (bn_rows(cols, gc.bn) @ K) + b) for cols, vals in _csr_rows(gc.graph.fwd, rows)]
            report.append(("gcn layer %d" % i, _close(_h(one)[rows], ref, "AliNet GraphConvolution %d" % i)))
            if i < layer_num - 1:
                at = m.two_hop_layers[i]
                two = at.call(x)
                er, ec, ev = _edges(at.graph)
                K0, K1, K2 = _h(at.kernel), _h(at.kernel1), _h(at.kernel2)

                def s_of(ids, kern):
                    xb = bn_rows(ids, at.bn)
                    return np.tanh(((xb @ kern) * xb).sum(1))
                touched = np.flatnonzero(np.isin(er, rows))
                s1 = dict(zip(rows.tolist(), s_of(rows, K1)))
                cols_needed = np.unique(ec[touched])
                s2 = dict(zip(cols_needed.tolist(), s_of(cols_needed, K2)))
                z = np.zeros(len(er))
                z[touched] = ev[touched] * np.array([s1[r] for r in er[touched]]) + ev[touched] * np.array([s2[c] for c in ec[touched]])
                mapped = {}

                def v_of(ids):
                    miss = [c for c in ids.tolist() if c not in mapped]
                    if miss:
                        mapped.update(zip(miss, bn_rows(miss, at.bn) @ K0))
                    return np.stack([mapped[c] for c in ids.tolist()])
                ref = [None if a is None else np.tanh(a) for a in _attention_rows(er, ec, z, v_of, rows, at.graph.grouping)]
                report.append(("attention layer %d (%s)" % (i, at.graph.grouping),
                               _close(_h(two)[rows], ref, "AliNet attention %d" % i)))
                hw = m.highways[i]
                out = hw.call(two, one)
                g_ = _h(hw.bn.gamma) * BN_SCALE
                i1, i2 = _h(two)[rows] * g_ + _h(hw.bn.beta), _h(one)[rows] * g_ + _h(hw.bn.beta)
                gate = np.maximum(np.tanh(i1 @ _h(hw.weight)), 0.0)
                report.append(("highway %d" % i, _close(_h(out)[rows], list(np.tanh(i2 * (1 - gate) + i1 * gate)), "AliNet highway %d" % i)))
                x = out
            else:
                x = one
            del xh
    with capsys.disabled():
        print("\nAliNet at the EN-DE-100K shape (%d entities, 1-hop nnz %d, 2-hop nnz %d): max abs deviation per layer on %d rows: %s"
              % (kgs.entities_num, m.adj[0].nnz, m.adj[1].nnz, len(rows), ", ".join("%s %.2e" % r for r in report)))
    # one optimiser step: the backward runs at this size and moves every variable it should
    before = [p.detach().clone() for p in m._params]
    pos, neg, valid = m.device_input_batch(m.args.batch_size)
    loss = m.train_step(pos, neg, neg_valid=valid)
    assert np.isfinite(float(loss.detach()))
    assert all(bool((a != b.detach()).any()) for a, b in zip(before[:3], m._params[:3]))


def test_rdgcn_layers_100k_shape(tmp_path, capsys):
    """rdgcn.py:184-338 at the EN-FR-100K shape (d = 300): relation features, dense dual attention, primal sparse attention
    with relation-derived logits, diagonal GCN layers, highway gates."""
    from openea_amd.approaches import RDGCN
    from openea_amd.modules.load.synth import make_kgs
    kgs = make_kgs("EN-FR-100K-V1", mode="mapping", seed=0)
    m = RDGCN()
    m.set_args(_args("RDGCN", tmp_path, "EN-FR-100K-V1", max_epoch=1, start_valid=1000, random_name_init=True))
    m.set_kgs(kgs)
    m.init()
    L = m.gcn_model
    p = {k: _h(v) for k, v in L.p.items()}
    rng = np.random.RandomState(12)
    rows = _rows(rng, kgs.entities_num)
    report = []

    def dense_att(in_fts, f1, b1, f2, b2, values, A, bias):
        logits = (in_fts @ f1 + b1) + (in_fts @ f2 + b2).T
        logits = A * logits
        lg = np.where(logits > 0, logits, 0.2 * logits) + bias
        e = np.exp(lg - lg.max(1, keepdims=True))
        return np.maximum((e / e.sum(1, keepdims=True)) @ values, 0.0)

    with torch.no_grad():
        x0 = L.primal_X_0
        A, bias = _h(L.dual_A), _h(L.dual_bias)
        # compute_r (rdgcn.py:258-266): the whole [R, 2d] matrix (R = a few hundred)
        r0 = L.compute_r(x0)
        hm = [c for c in _csr_rows(L.head_mean.fwd, range(L.count_r))]
        tm = [c for c in _csr_rows(L.tail_mean.fwd, range(L.count_r))]
        x0h = _h(x0)
        ref_r0 = np.stack([np.concatenate([v1 @ x0h[c1], v2 @ x0h[c2]]) for (c1, v1), (c2, v2) in zip(hm, tm)])
        report.append(("compute_r", _close(list(_h(r0)), list(ref_r0), "RDGCN compute_r")))
        d1 = L.add_self_att_layer(r0)
        ref_d1 = dense_att(_h(r0) @ p['sa_w'], p['sa_f1'], p['sa_b1'], p['sa_f2'], p['sa_b2'], _h(r0), A, bias)
        report.append(("dual self attention", _close(list(_h(d1)), list(ref_d1), "RDGCN self attention", atol=5e-4)))
        er, ec, _ = _edges(L.r_graph)
        edge_rel = L.edge_rel.cpu().numpy()

        def sparse_att(inlayer, dual_layer, w, b):
            dt = (_h(dual_layer) @ w + b).reshape(-1)
            z = dt[edge_rel]
            xin = _h(inlayer)
            return [None if a is None else np.maximum(a, 0.0)
                    for a in _attention_rows(er, ec, z, lambda ids: xin[ids], rows, L.r_graph.grouping)]
        a1 = L.add_sparse_att_layer(x0, d1, L.p['pa1_w'], L.p['pa1_b'])
        report.append(("primal attention 1 (%s)" % L.r_graph.grouping,
                       _close(_h(a1)[rows], sparse_att(x0, d1, p['pa1_w'], p['pa1_b']), "RDGCN primal attention 1")))
        x1 = x0 + L.alpha * a1
        r1 = L.compute_r(x1)
        d2 = L.add_dual_att_layer(d1, r1)
        ref_d2 = dense_att(_h(r1) @ p['da_w'] + p['da_b'], p['da_f1'], p['da_b1'], p['da_f2'], p['da_b2'], _h(d1), A, bias)
        report.append(("dual attention", _close(list(_h(d2)), list(ref_d2), "RDGCN dual attention", atol=5e-4)))
        a2 = L.add_sparse_att_layer(x1, d2, L.p['pa2_w'], L.p['pa2_b'])
        report.append(("primal attention 2", _close(_h(a2)[rows], sparse_att(x1, d2, p['pa2_w'], p['pa2_b']), "RDGCN primal attention 2")))
        x2 = x0 + L.beta * a2
        cur = x2
        for tag in ("1", "2"):
            diag = L.add_diag_layer(cur, L.p['diag' + tag])
            curh = _h(cur)
            ref = [np.maximum(vals @ (curh[cols] * p['diag' + tag]), 0.0) for cols, vals in _csr_rows(L.M.fwd, rows)]
            report.append(("diag GCN " + tag, _close(_h(diag)[rows], ref, "RDGCN diag layer " + tag)))
            nxt = L.highway(cur, diag, L.p['hw%s_w' % tag], L.p['hw%s_b' % tag])
            gate = 1.0 / (1.0 + np.exp(-(curh[rows] @ p['hw%s_w' % tag] + p['hw%s_b' % tag])))
            report.append(("highway " + tag, _close(_h(nxt)[rows], list(gate * _h(diag)[rows] + (1.0 - gate) * curh[rows]), "RDGCN highway " + tag)))
            cur = nxt
        out = L.forward()
        assert torch.allclose(out, cur, rtol=0, atol=1e-5)   # the pieces above ARE the forward (fp32 atomics reorder between runs)
    with capsys.disabled():
        print("\nRDGCN at the EN-FR-100K shape (%d entities, %d relations, %d attention edges): max abs deviation per layer: %s"
              % (kgs.entities_num, L.count_r, len(er), ", ".join("%s %.2e" % r for r in report)))


def _segment_attention_f64(er, ec, z, v, n_rows, grouping, slope=0.2):
    """tf.sparse_softmax + sparse_tensor_dense_matmul over the WHOLE graph in float64, vectorised: softmax groups = rows
    ('row': edges sorted by row first) or maximal runs of consecutive equal rows in the fed order ('runs')."""
    import scipy.sparse as sp
    if grouping == 'row':
        order = np.lexsort((ec, er))
        er, ec, z = er[order], ec[order], z[order]
    starts = np.concatenate([[0], np.flatnonzero(np.diff(er)) + 1])
    lg = np.where(z > 0, z, slope * z)
    seg_of = np.repeat(np.arange(len(starts)), np.diff(np.concatenate([starts, [len(er)]])))
    e = np.exp(lg - np.maximum.reduceat(lg, starts)[seg_of])
    alpha = e / np.add.reduceat(e, starts)[seg_of]
    return sp.csr_matrix((alpha, (er, ec)), shape=(n_rows, v.shape[0])) @ v      # duplicates are summed: several groups per row


def test_alinet_end_to_end_forward_100k_shape(tmp_path, capsys):
    """the WHOLE AliNet forward at the EN-DE-100K shape (500-400-300), not layer by layer on the device's
    own intermediate inputs -- a float64 numpy / scipy restatement of alinet.py:574-677,784-826 from the initial embedding
    to both layer outputs, every row."""
    import scipy.sparse as sp
    from openea_amd.approaches import AliNet
    from openea_amd.modules.load.synth import make_kgs
    kgs = make_kgs("EN-DE-100K-V1", mode="mapping", seed=0)
    m = AliNet()
    m.set_args(_args("AliNet", tmp_path, "EN-DE-100K-V1", max_epoch=1, start_valid=1000))
    m.set_kgs(kgs)
    m.init()
    rng = np.random.RandomState(3)
    with torch.no_grad():                      # move the BatchNorm affines and biases off their initial (1, 0) so that they matter
        for p in m._params[1:]:
            if p.dim() == 1:
                p.add_(torch.from_numpy(rng.standard_normal(p.shape[0]).astype(np.float32) * 0.05).to(p.device))
        outs = [_h(o) for o in m._forward()]
    n = kgs.entities_num

    def csr(op):
        return sp.csr_matrix((op.vals.cpu().numpy().astype(np.float64), op.colidx.cpu().numpy(), op.rowptr.cpu().numpy()), shape=op.shape)

    def bn(x, b):
        return x * (_h(b.gamma) * BN_SCALE) + _h(b.beta)
    x = _h(m.init_embedding)
    g0, g1, at, hw = m.one_hop_layers[0], m.one_hop_layers[1], m.two_hop_layers[0], m.highways[0]
    a1 = csr(g0.graph.fwd)
    one = np.tanh(a1 @ (bn(x, g0.bn) @ _h(g0.kernel)) + _h(g0.bias))
    xb = bn(x, at.bn)
    s1 = np.tanh(((xb @ _h(at.kernel1)) * xb).sum(1))
    s2 = np.tanh(((xb @ _h(at.kernel2)) * xb).sum(1))
    er, ec, ev = _edges(at.graph)
    two = np.tanh(_segment_attention_f64(er, ec, ev * s1[er] + ev * s2[ec], xb @ _h(at.kernel), n, at.graph.grouping))
    i1, i2 = bn(two, hw.bn), bn(one, hw.bn)
    gate = np.maximum(np.tanh(i1 @ _h(hw.weight)), 0.0)
    x1 = np.tanh(i2 * (1 - gate) + i1 * gate)
    x2 = np.tanh(csr(g1.graph.fwd) @ (bn(x1, g1.bn) @ _h(g1.kernel)) + _h(g1.bias))
    d1, d2 = float(np.abs(outs[0] - x1).max()), float(np.abs(outs[1] - x2).max())
    with capsys.disabled():
        print("\nAliNet end-to-end forward at the EN-DE-100K shape (%d entities, grouping %r, %d of %d softmax groups single-edge): "
              "max abs deviation from the float64 forward over ALL rows: layer-1 output %.2e, layer-2 output %.2e"
              % (n, at.graph.grouping, int((np.diff(at.graph.seg_ptr_host) == 1).sum()), len(at.graph.seg_row_host), d1, d2))
    assert d1 <= 1e-4 and d2 <= 1e-4


def test_rdgcn_end_to_end_forward_100k_shape(tmp_path, capsys):
    """the WHOLE RDGCN forward (rdgcn.py:184-338) at the EN-FR-100K shape, d = 300, in float64 numpy / scipy from the input
    layer to the output embedding, every row -- against Layer.forward() on the device."""
    import scipy.sparse as sp
    from openea_amd.approaches import RDGCN
    from openea_amd.modules.load.synth import make_kgs
    kgs = make_kgs("EN-FR-100K-V1", mode="mapping", seed=0)
    m = RDGCN()
    m.set_args(_args("RDGCN", tmp_path, "EN-FR-100K-V1", max_epoch=1, start_valid=1000, random_name_init=True))
    m.set_kgs(kgs)
    m.init()
    L = m.gcn_model
    with torch.no_grad():
        out = _h(L.forward())
    p = {k: _h(v) for k, v in L.p.items()}
    n = kgs.entities_num

    def csr(op):
        return sp.csr_matrix((op.vals.cpu().numpy().astype(np.float64), op.colidx.cpu().numpy(), op.rowptr.cpu().numpy()), shape=op.shape)
    hm, tm, M = csr(L.head_mean.fwd), csr(L.tail_mean.fwd), csr(L.M.fwd)
    A, bias = _h(L.dual_A), _h(L.dual_bias)
    er, ec, _ = _edges(L.r_graph)
    edge_rel = L.edge_rel.cpu().numpy()

    def compute_r(x):
        return np.concatenate([hm @ x, tm @ x], axis=1)

    def dense_att(in_fts, f1, b1, f2, b2, values):
        logits = A * ((in_fts @ f1 + b1) + (in_fts @ f2 + b2).T)
        lg = np.where(logits > 0, logits, 0.2 * logits) + bias
        e = np.exp(lg - lg.max(1, keepdims=True))
        return np.maximum((e / e.sum(1, keepdims=True)) @ values, 0.0)

    def sparse_att(x, dual, w, b):
        z = (dual @ w + b).reshape(-1)[edge_rel]
        return np.maximum(_segment_attention_f64(er, ec, z, x, n, L.r_graph.grouping), 0.0)
    x0 = _h(L.primal_X_0)
    r0 = compute_r(x0)
    d1 = dense_att(r0 @ p['sa_w'], p['sa_f1'], p['sa_b1'], p['sa_f2'], p['sa_b2'], r0)
    x1 = x0 + L.alpha * sparse_att(x0, d1, p['pa1_w'], p['pa1_b'])
    d2 = dense_att(compute_r(x1) @ p['da_w'] + p['da_b'], p['da_f1'], p['da_b1'], p['da_f2'], p['da_b2'], d1)
    cur = x0 + L.beta * sparse_att(x1, d2, p['pa2_w'], p['pa2_b'])
    for tag in ("1", "2"):
        diag = np.maximum(M @ (cur * p['diag' + tag]), 0.0)
        gate = 1.0 / (1.0 + np.exp(-(cur @ p['hw%s_w' % tag] + p['hw%s_b' % tag])))
        cur = gate * diag + (1.0 - gate) * cur
    dev = float(np.abs(out - cur).max())
    with capsys.disabled():
        print("\nRDGCN end-to-end forward at the EN-FR-100K shape (%d entities, %d relations, %d attention edges, grouping %r): "
              "max abs deviation of the output embedding from the float64 forward over ALL rows %.2e (|out| max %.2f)"
              % (n, L.count_r, len(er), L.r_graph.grouping, dev, float(np.abs(cur).max())))
    assert dev <= 1e-4 * max(1.0, float(np.abs(cur).max()))


def test_gcn_align_epoch_100k_shape(tmp_path, capsys):
    """gcn_align.py:498-539, 204-267, 737-785: two full-batch epochs of the structure unit at the EN-FR-100K shape against
    the oracle's epoch (fp64, scipy, whole graph), adjacency from the oracle's own restatement of :610-664."""
    import scipy.sparse as sp
    from openea_amd import ops
    from openea_amd.approaches import GCN_Align
    from openea_amd.modules.load.synth import make_kgs
    from oracle import np_oracle as orc
    kgs = make_kgs("EN-FR-100K-V1", mode="mapping", seed=0)
    m = GCN_Align()
    d = 100
    m.set_args(_args("GCN_Align", tmp_path, "EN-FR-100K-V1", se_dim=d, ae_dim=d, max_epoch=2))
    m.set_kgs(kgs)
    m.init()
    se = m.model_se
    W0 = se.W[:, :d].cpu().numpy().copy()
    triples = kgs.kg1.relation_triples_list + kgs.kg2.relation_triples_list
    coords, values, shape = orc.gcn_preprocess_adj(orc.gcn_weighted_adj(kgs.entities_num, triples))
    a_ref = sp.coo_matrix((values, (coords[:, 0], coords[:, 1])), shape=shape).tocsr()
    a_dev = sp.csr_matrix((se.adj.fwd.vals.cpu().numpy(), se.adj.fwd.colidx.cpu().numpy(), se.adj.fwd.rowptr.cpu().numpy()), shape=shape)
    assert abs(a_ref - a_dev).max() < 1e-6                   # device-built support == the oracle's (python-dict) restatement
    train = np.asarray(kgs.train_links, np.int32)
    k, t = m.args.neg_triple_num, len(train)
    rng = np.random.RandomState(5)
    negs_h = (np.repeat(train[:, 0], k), rng.choice(kgs.entities_num, t * k), rng.choice(kgs.entities_num, t * k),
              np.repeat(train[:, 1], k))
    negs_d = tuple(ops.to_ids(x.astype(np.int32)) for x in negs_h)
    W_ref = W0.copy()
    for _ in range(2):
        se.train_step(negs_d)
        loss_ref, out_ref = orc.gcn_se_epoch(W_ref, coords, values.astype(np.float32), train, m.args.gamma, k, negs_h,
                                             m.args.learning_rate)
    W = se.W[:, :d].cpu().numpy()
    dev_w = np.linalg.norm(W - W_ref) / np.linalg.norm(W_ref)
    dev_o = float(np.abs(se.outputs[:, :d].cpu().numpy() - out_ref).max())
    with capsys.disabled():
        print("\nGCN-Align SE unit at the EN-FR-100K shape (%d entities, nnz %d): table relative L2 deviation %.2e, output max abs %.2e"
              % (kgs.entities_num, a_dev.nnz, dev_w, dev_o))
    assert dev_w <= 1e-4
    assert dev_o <= 5e-5
