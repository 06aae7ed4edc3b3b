"""BASELINE.json full sizes (EN-FR-100K: 70,000 test pairs, 100,000 entities per KG, k = 2,000
neighbours, batch 20,000 x 10 negatives) through size-independent properties + oracle spot checks
on sampled rows (the oracle cannot finish the full problems in seconds)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    from openea_amd import ops as _ops
    _ops.lib()
    return _ops


def _unit_rows(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def test_rank_eval_70k(ops):
    from oracle import cport
    rng = np.random.RandomState(0)
    n, d = 70000, 100
    e1 = _unit_rows(rng, n, d)
    e2 = e1 + 0.35 * _unit_rows(rng, n, d)
    t1, t2 = ops.to_table(e1), ops.to_table(e2)
    rank, argmax = ops.rank_eval(t1, t2, d, "inner")
    rank_h, am_h = rank.cpu().numpy(), argmax.cpu().numpy()
    # oracle spot check on 96 sampled query rows (each needs all 70,000 candidates)
    rows = rng.choice(n, 96, replace=False)
    s = cport.sim_matrix(e1[rows], e2, "inner")
    g = s[np.arange(len(rows)), rows]
    cols = np.arange(n)[None, :]
    ref = ((s > g[:, None]) | ((s == g[:, None]) & (cols < rows[:, None]))).sum(1)
    assert np.array_equal(rank_h[rows], ref)
    assert np.array_equal(am_h[rows], s.argmax(1))
    # properties: ranks in range; metrics kernel == host reductions; Hits monotone in k
    assert rank_h.min() >= 0 and rank_h.max() < n
    hits, rs, rr = ops.rank_metrics(rank, [1, 5, 10, 50])
    assert hits == [int((rank_h < k).sum()) for k in (1, 5, 10, 50)] and hits == sorted(hits)
    assert rs == int((rank_h.astype(np.int64) + 1).sum())
    # idempotence: a set evaluated against itself ranks every gold first
    r2, a2 = ops.rank_eval(t1, t1, d, "inner")
    assert int(r2.max().item()) == 0 and bool((a2 == torch.arange(n, device=a2.device, dtype=torch.int32)).all().item())
    # sharded evaluation (two query blocks with gold offsets) == unsharded
    lo = 33333
    ra, _ = ops.rank_eval(t1[:lo], t2, d, "inner", gold_offset=0)
    rb, _ = ops.rank_eval(t1[lo:], t2, d, "inner", gold_offset=lo)
    assert np.array_equal(np.concatenate([ra.cpu().numpy(), rb.cpu().numpy()]), rank_h)


def test_rank_eval_70k_alinet_width(ops, monkeypatch):
    """AliNet's evaluation shape (alinet.py:948-966: [init, out0, out1] = 500 + 400 + 300 columns, each block L2-normalised;
    alinet_args_100K.json: eval_metric inner, csls 10; 70,000 test pairs).  The product path (certified bf16 prefilter, K-blocked
    accumulation, 256 x 128 tiles on the three-stage ring) against the ORACLE on sampled query rows (each needs all 70,000
    candidates at 1,200 columns), against the exact fp32 sweep on every row, with and without CSLS means; the prefilter must not
    have fallen back and its record count stays small on a table whose Hits@1 is about one half."""
    from oracle import cport
    from openea_amd.modules.finding.similarity import csls_means_device
    rng = np.random.RandomState(12)
    n, dims = 70000, (500, 400, 300)
    d = sum(dims)
    b1s, b2s = [], []
    for db in dims:
        b1 = rng.standard_normal((n, db)).astype(np.float32)
        b2 = (b1 + 8.0 * rng.standard_normal((n, db)).astype(np.float32)).astype(np.float32)
        b1s.append(b1 / np.linalg.norm(b1, axis=1, keepdims=True))
        b2s.append(b2 / np.linalg.norm(b2, axis=1, keepdims=True))
    e1, e2 = np.concatenate(b1s, 1), np.concatenate(b2s, 1)
    del b1s, b2s
    t1, t2 = ops.to_table(e1), ops.to_table(e2)
    assert ops.eval_bf16_enabled(n, n)
    st = {}
    rank, argmax = ops.rank_eval_bf16(t1, t2, d, stats=st)
    assert not st["fallback"] and st["records"] < 64 * n, st
    rank_h, am_h = rank.cpu().numpy(), argmax.cpu().numpy()
    assert 0.3 < float((rank_h == 0).mean()) < 0.7                    # Hits@1 between 0.3 and 0.7
    rows = rng.choice(n, 32, replace=False)
    s = cport.sim_matrix(e1[rows], e2, "inner")
    g = s[np.arange(len(rows)), rows]
    ref = ((s > g[:, None]) | ((s == g[:, None]) & (np.arange(n)[None, :] < rows[:, None]))).sum(1)
    assert np.array_equal(rank_h[rows], ref) and np.array_equal(am_h[rows], s.argmax(1))
    r32, a32 = ops.rank_eval(t1, t2, d, "inner", allow_bf16=False)   # the exact fp32 sweep, every row
    assert torch.equal(rank, r32) and torch.equal(argmax, a32)
    rr, cc = csls_means_device(t1, t2, d, "inner", 10)
    monkeypatch.setenv("OEA_CSLS_BF16", "0")
    rr32, cc32 = csls_means_device(t1, t2, d, "inner", 10)
    monkeypatch.delenv("OEA_CSLS_BF16")
    assert torch.equal(rr, rr32) and torch.equal(cc, cc32)           # the means of the bf16 sweep ARE the fp32 sweep's
    rc, ac = ops.rank_eval_bf16(t1, t2, d, csls_r=rr, csls_c=cc)
    rc32, ac32 = ops.rank_eval(t1, t2, d, "inner", rr, cc, allow_bf16=False)
    assert torch.equal(rc, rc32) and torch.equal(ac, ac32)


def _l1_rank_ref(s, rows, csls_r=None, csls_c=None):
    """ranks / nearest candidates of the sampled query rows from their full similarity strips (alignment.py:146-168: the number of
    candidates ranked before the gold, ties by column; with CSLS (2 s - r) - c in fp32, similarity.py:57-77)"""
    if csls_r is not None:
        from oracle import cport
        s = cport.csls_apply(s, csls_r[rows], csls_c)
    g = s[np.arange(len(rows)), rows]
    cols = np.arange(s.shape[1])[None, :]
    return ((s > g[:, None]) | ((s == g[:, None]) & (cols < rows[:, None]))).sum(1), s.argmax(1)


@pytest.mark.parametrize("table", ["half", "clustered"])
def test_rank_eval_70k_manhattan_rdgcn_width(ops, table, monkeypatch):
    """RDGCN-100K / GCN-Align test(): eval_metric manhattan (run/args/rdgcn_args_100K.json:26, similarity.py:46-48), 70,000 test
    pairs at 300 columns, plain and with csls = 10 (basic_model.py:132-135).  The product path -- 16-bit grid distances of every
    pair, exact fp64 chains only where the grid's error bound leaves a comparison open, CSLS means from the k + 32 nearest on the
    grid, the rank pass over the kept strips (19.6 GB) -- against the C ORACLE on 32 sampled query rows (all 70,000 candidates
    each, the sequential fp64 chain of scipy's cdist) and against the all-pairs fp64 device path on EVERY row; the CSLS means are
    compared bit for bit.  'half': Hits@1 about one half; 'clustered': 20,000 candidates in 40 tight clusters and every gold
    inside one -- hundreds of candidates within the grid's error of the gold distance (the band / exact-pair paths at size)."""
    from oracle import cport
    from openea_amd.modules.finding.similarity import csls_means_device
    rng = np.random.RandomState(21)
    n, d, k = 70000, 300, 10
    e1 = _unit_rows(rng, n, d)
    if table == "half":
        e2 = (e1 + 2.0 * rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d)).astype(np.float32)
    else:
        e2 = (e1 + 0.05 * rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d)).astype(np.float32)
        for c0 in range(10000, 30000, 500):                           # 40 clusters of 500 near-duplicates
            e2[c0:c0 + 500] = e2[c0] + 2e-4 * rng.standard_normal((500, d)).astype(np.float32)
        e1[10000:30000] = e2[10000:30000] + 1e-4 * rng.standard_normal((20000, d)).astype(np.float32)
    t1, t2 = ops.to_table(e1), ops.to_table(e2)
    rows = np.sort(np.concatenate([rng.choice(np.arange(10000, 30000), 16, replace=False),
                                   rng.choice(np.setdiff1d(np.arange(n), np.arange(10000, 30000)), 16, replace=False)]))
    s_rows = cport.sim_matrix(e1[rows], e2, "manhattan")              # [32, 70,000]: the oracle's strips of the sampled rows
    # ---- plain ---------------------------------------------------------------------------------------------------------------
    monkeypatch.setenv("OEA_L1_EVAL", "grid")
    rank, argmax = ops.rank_eval(t1, t2, d, "manhattan")
    monkeypatch.setenv("OEA_L1_EVAL", "f64")
    rank64, argmax64 = ops.rank_eval(t1, t2, d, "manhattan")          # every pair in fp64 on the device
    assert torch.equal(rank, rank64) and torch.equal(argmax, argmax64)
    rk_ref, am_ref = _l1_rank_ref(s_rows, rows)
    assert np.array_equal(rank.cpu().numpy()[rows], rk_ref) and np.array_equal(argmax.cpu().numpy()[rows], am_ref)
    h1 = float((rank == 0).float().mean().item())
    if table == "half":
        assert 0.3 < h1 < 0.7, h1                                      # Hits@1 between 0.3 and 0.7
    # ---- csls = 10: means, then ranks of (2 s - r) - c -------------------------------------------------------------------------
    r64, c64 = csls_means_device(t1, t2, d, "manhattan", k)           # strips of every pair in fp64 + row_topk_mean
    rk64, am64 = ops.rank_eval(t1, t2, d, "manhattan", r64, c64)
    monkeypatch.setenv("OEA_L1_EVAL", "grid")
    r, c, grid = csls_means_device(t1, t2, d, "manhattan", k, return_grid=True)
    assert grid is not None and len(grid.strips) > 0                   # the query strips stay for the rank pass (they fit)
    assert torch.equal(r, r64) and torch.equal(c, c64)                 # the means, bit for bit
    assert np.array_equal(r.cpu().numpy()[rows], cport.topk_mean(s_rows, k))
    rk, am = ops.rank_eval_l1_grid(t1, t2, d, csls_r=r, csls_c=c, grid=grid)
    assert not grid.strips
    assert torch.equal(rk, rk64) and torch.equal(am, am64)
    rk_ref, am_ref = _l1_rank_ref(s_rows, rows, r.cpu().numpy(), c.cpu().numpy())
    assert np.array_equal(rk.cpu().numpy()[rows], rk_ref) and np.array_equal(am.cpu().numpy()[rows], am_ref)
    # the call test() makes (greedy_alignment_device): the same numbers
    from openea_amd.modules.finding.alignment import greedy_alignment_device
    rk2, am2, hits, rs, rr = greedy_alignment_device(t1, t2, d, [1, 5, 10, 50], "manhattan", False, k)
    assert torch.equal(rk2, rk) and torch.equal(am2, am) and hits[0] == int((rk == 0).sum().item())


def test_rank_eval_70k_bf16_csls_row_blocks(ops):
    """the row-sharded inner-product evaluation with CSLS at BootEA-100K's size (70,000^2 x 100): two blocks of
    query rows through oea_rank_eval_bf16_csls with their gold offsets and their slices of the row means == the unsharded call
    == the exact fp32 sweep."""
    from openea_amd.modules.finding.similarity import csls_means_device
    rng = np.random.RandomState(5)
    n, d = 70000, 100
    e1 = _unit_rows(rng, n, d)
    e2 = e1 + 2.0 * _unit_rows(rng, n, d)
    t1, t2 = ops.to_table(e1), ops.to_table(e2)
    assert ops.eval_bf16_enabled(n, n)
    rr, cc = csls_means_device(t1, t2, d, "inner", 10)
    st = {}
    rank, argmax = ops.rank_eval_bf16(t1, t2, d, csls_r=rr, csls_c=cc, stats=st)
    assert not st["fallback"]
    r32, a32 = ops.rank_eval(t1, t2, d, "inner", rr, cc, allow_bf16=False)
    assert torch.equal(rank, r32) and torch.equal(argmax, a32)
    assert 0.02 < float((rank == 0).float().mean().item()) < 0.98     # neither all golds first nor none: both epilogue paths run
    lo = 33333
    sa, sb = {}, {}
    ra, aa = ops.rank_eval_bf16(t1[:lo], t2, d, gold_offset=0, csls_r=rr[:lo].contiguous(), csls_c=cc, stats=sa)
    rb, ab = ops.rank_eval_bf16(t1[lo:], t2, d, gold_offset=lo, csls_r=rr[lo:].contiguous(), csls_c=cc, stats=sb)
    assert not sa["fallback"] and not sb["fallback"]
    assert torch.equal(torch.cat([ra, rb]), rank) and torch.equal(torch.cat([aa, ab]), argmax)


def test_rank_eval_manhattan_10k5(ops):
    from oracle import cport
    rng = np.random.RandomState(1)
    n, d = 10500, 200             # GCN-Align test size: concat(se, ae) = 200 dims, manhattan
    e1 = _unit_rows(rng, n, d)
    e2 = e1 + 0.5 * _unit_rows(rng, n, d)
    rank, argmax = ops.rank_eval(ops.to_table(e1), ops.to_table(e2), d, "manhattan")
    rows = rng.choice(n, 64, replace=False)
    s = cport.sim_matrix(e1[rows], e2, "manhattan")
    g = s[np.arange(len(rows)), rows]
    ref = ((s > g[:, None]) | ((s == g[:, None]) & (np.arange(n)[None, :] < rows[:, None]))).sum(1)
    assert np.array_equal(rank.cpu().numpy()[rows], ref)


def test_neighbours_100k(ops):
    from oracle import cport
    rng = np.random.RandomState(2)
    n, d, k = 100000, 100, 2000          # int((1 - 0.98) * 100000) = 2000
    assert int((1 - 0.98) * 100000) == k
    emb = _unit_rows(rng, n, d)
    t = ops.to_table(emb)
    ids = np.arange(n, dtype=np.int32) * 2
    out = ops.topk_inner(t, t, d, k, id_map=ops.to_ids(ids))
    out_h = out.cpu().numpy()
    assert out_h.shape == (n, k)
    assert np.all(np.diff(out_h, axis=1) > 0)                       # ascending, unique per row
    assert np.all((out_h == ids[:, None]).any(1))                   # contains the entity itself
    rows = rng.choice(n, 16, replace=False)
    ref = cport.topk_inner(emb[rows], emb, k)
    assert np.array_equal(out_h[rows], ids[ref])


def test_step_100k_shape(ops, capsys):
    """EN-FR-100K batch shape (20,000 positives x 10 negatives, 200,000 entities, dim 100): one fused step against the
    C oracle's step on the same batch, at the north-star tolerance (every embedding within 1e-4 L2)."""
    from _tol import assert_rows_close
    from openea_amd.models.trainer import EmbeddingTable, TripleTrainer
    from oracle import cport
    rng = np.random.RandomState(3)
    n_ent, n_rel, d, B, k = 200000, 700, 100, 20000, 10
    ent_h = (rng.standard_normal((n_ent, d)) / np.sqrt(d)).astype(np.float32)
    rel_h = (rng.standard_normal((n_rel, d)) / np.sqrt(d)).astype(np.float32)
    ent = EmbeddingTable(ent_h, True, "e")
    rel = EmbeddingTable(rel_h, True, "r")
    w = 1.0 / np.arange(1, n_ent + 1) ** 0.9                                  # Zipf heads: hub rows collect many gradients
    pos = np.stack([rng.choice(n_ent, B, p=w / w.sum()), rng.randint(0, n_rel, B), rng.randint(0, n_ent, B)], 1).astype(np.int32)
    neg = np.repeat(pos, k, 0)
    flip = rng.rand(len(neg)) < 0.5
    neg[flip, 0] = rng.randint(0, n_ent, int(flip.sum()))
    neg[~flip, 2] = rng.randint(0, n_ent, int((~flip).sum()))
    kw = dict(loss="limited", loss_norm="L2", pos_margin=0.01, neg_margin=2.0, balance=0.2, optimizer="Adagrad", lr=0.01)
    tg = TripleTrainer(ent, rel, ops.make_step_cfg(neg_group_k=k, **kw))
    e0 = ent.var.clone()
    losses = []
    e_ref, r_ref = ent_h.copy(), rel_h.copy()
    ea, ra = np.full_like(e_ref, 0.1), np.full_like(r_ref, 0.1)
    for _ in range(2):                                                         # two steps: the accumulators matter in the second
        tg.step(ops.to_ids(pos), ops.to_ids(neg))
        losses.append(cport.triple_step(e_ref, ea, r_ref, ra, pos, neg, **kw))
    loss_g = tg.pop_loss()
    touched = (ent.var != e0).any(1)
    n_touched = int(touched.sum().item())
    uniq = len(np.unique(np.concatenate([pos[:, 0], pos[:, 2], neg[:, 0], neg[:, 2]])))
    assert 0 < n_touched <= uniq                                     # only referenced rows move
    with capsys.disabled():
        assert_rows_close(ent.raw(), e_ref, "100K-shape step, entity table")
        assert_rows_close(rel.raw(), r_ref, "100K-shape step, relation table")
    assert abs(loss_g - sum(losses)) <= 1e-5 * abs(sum(losses))
    np.testing.assert_allclose(tg.ent_acc[:, :d].cpu().numpy(), ea, rtol=2e-3, atol=1e-6)   # acc = 0.1 + sum g^2, g sums of ~100 fp32 atomics
    assert not bool((tg.ws[: tg.ws.numel() - 8 * 4096] != 0).any().item())


def test_graph_operators_100k_shape(ops):
    """EN-DE-100K-like graph (200,000 nodes, ~1.6 M edges with hubs): the aggregate is linear and matches the oracle on
    sampled rows; the sparse attention's weights sum to one per row, its output is linear in v, and a constant v
    passes through unchanged."""
    import scipy.sparse as sp
    from openea_amd.models.graph_ops import EdgeGraph, sparse_attention, spmm
    rng = np.random.RandomState(3)
    n, nnz, d = 200000, 1600000, 64
    w = 1.0 / np.arange(1, n + 1) ** 0.9
    rows = rng.choice(n, nnz, p=w / w.sum())
    cols = rng.randint(0, n, nnz)
    vals = rng.rand(nnz).astype(np.float32)
    g = EdgeGraph(rows, cols, vals, (n, n), ops.device())
    x1 = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda()
    x2 = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).cuda()
    y1, y2, y12 = spmm(g, x1), spmm(g, x2), spmm(g, x1 + x2)
    assert torch.allclose(y12, y1 + y2, rtol=1e-4, atol=1e-3)
    a = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    a.sum_duplicates()
    pick = np.concatenate([np.arange(8), rng.choice(n, 64, replace=False)])            # hubs + random rows
    ref = a[pick].astype(np.float64) @ x1.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(y1[torch.from_numpy(pick).cuda()].cpu().numpy(), ref, rtol=1e-4, atol=2e-3)
    # attention
    z = torch.from_numpy(rng.standard_normal(g.nnz).astype(np.float32)).cuda()
    ones = torch.ones((n, d), device="cuda")
    out_c = sparse_attention(g, z, ones)
    has = torch.zeros(n, dtype=torch.bool, device="cuda")
    has[g.e_rows] = True
    assert torch.allclose(out_c[has], torch.ones_like(out_c[has]), rtol=0, atol=1e-4)   # softmax weights sum to 1 per row
    assert float(out_c[~has].abs().max()) == 0.0                                        # rows without edges stay zero
    o1, o2, o12 = sparse_attention(g, z, x1), sparse_attention(g, z, x2), sparse_attention(g, z, x1 + x2)
    assert torch.allclose(o12, o1 + o2, rtol=1e-4, atol=1e-4)
    assert float(o1.abs().max()) <= float(x1.abs().max()) + 1e-4                       # convex combination of v rows
    # oracle (float64 softmax + aggregate over the row's edges in the graph's canonical order) on hub + random rows
    er, ec = g.e_rows.cpu().numpy(), g.e_cols.cpu().numpy()
    zh, xh = z.cpu().numpy().astype(np.float64), x1.cpu().numpy().astype(np.float64)
    o1h = o1.cpu().numpy()
    order = np.argsort(er, kind="stable")
    starts = np.searchsorted(er[order], np.arange(n + 1))
    for r in np.concatenate([np.arange(6), rng.choice(n, 48, replace=False)]):
        e = order[starts[r]:starts[r + 1]]
        if len(e) == 0:
            assert not o1h[r].any()
            continue
        lg = np.where(zh[e] > 0, zh[e], 0.2 * zh[e])
        a = np.exp(lg - lg.max())
        a /= a.sum()
        np.testing.assert_allclose(o1h[r], a @ xh[ec[e]], rtol=1e-4, atol=2e-5)


def test_symmetric_neighbour_search_equals_general_path(ops):
    """queries == candidates (q is c): only the tiles on and above the diagonal are computed and feed rows and columns
    (topk_append_sym_kernel); the result equals the general list path's (a copy of the table as candidates) bit for bit,
    at a size with several work items per query tile and a ragged last tile, and equals the oracle on sampled rows."""
    from oracle import cport
    rng = np.random.RandomState(7)
    n, d, k = 40100, 64, 800
    assert ops.lib().oea_topk_sym_workspace_bytes(n, k) > 0
    emb = _unit_rows(rng, n, d)
    emb[5] = emb[4]                                                  # duplicate rows: ties across the diagonal
    emb[n - 1] = emb[0]
    t = ops.to_table(emb)
    sym = ops.topk_inner(t, t, d, k).cpu().numpy()
    gen = ops.topk_inner(t, t.clone(), d, k).cpu().numpy()
    assert np.array_equal(sym, gen)
    rows = np.concatenate([[0, 4, 5, n - 1], rng.choice(n, 12, replace=False)])
    assert np.array_equal(sym[rows], cport.topk_inner(emb[rows], emb, k))


@pytest.mark.parametrize("case", ["tiny streams", "tiny streams, dry pool"])
def test_stream_neighbour_search_overflow_paths_stay_exact(ops, case, monkeypatch):
    """the stream form with its per-wave streams shrunk to 256 records: nearly every wave overflows, its tiles are recomputed
    (topk_stream_redo_kernel) and the records go through the shared overflow pool (topk_overflow_kernel); with a pool of 64
    chunks most of them are lost and the rows they belonged to must arrive through the strip fallback.  Same neighbour sets."""
    from oracle import cport
    monkeypatch.setenv("OEA_TOPK_STREAM_CAP", "256")
    if case.endswith("dry pool"):
        monkeypatch.setenv("OEA_TOPK_OVF_CHUNKS", "64")
    rng = np.random.RandomState(13)
    n, d, k = 33100, 48, 500
    emb = _unit_rows(rng, n, d)
    t = ops.to_table(emb)
    out = ops.topk_inner(t, t, d, k).cpu().numpy()
    monkeypatch.delenv("OEA_TOPK_STREAM_CAP")
    monkeypatch.delenv("OEA_TOPK_OVF_CHUNKS", raising=False)
    ref_dev = ops.topk_inner(t, t, d, k).cpu().numpy()
    assert np.array_equal(out, ref_dev)
    rows = np.concatenate([[0, 127, 128, n - 1], rng.choice(n, 12, replace=False)])
    assert np.array_equal(out[rows], cport.topk_inner(emb[rows], emb, k))


def test_neighbour_search_on_clustered_rows_spills_and_stays_exact(ops):
    """clustered embeddings (blocks of ~600 consecutive near-duplicate rows, as trained tables have them): a row's neighbours
    crowd into a few candidate tiles, its list segments overflow into the spill list (and some rows into the strip
    fallback); both list paths must still equal the oracle."""
    from oracle import cport
    rng = np.random.RandomState(11)
    n, d, k = 40100, 64, 800
    centres = rng.standard_normal((n // 600 + 1, d)).astype(np.float32)
    emb = centres[np.arange(n) // 600] + 0.15 * rng.standard_normal((n, d)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    t = ops.to_table(emb)
    sym = ops.topk_inner(t, t, d, k).cpu().numpy()
    gen = ops.topk_inner(t, t.clone(), d, k).cpu().numpy()
    assert np.array_equal(sym, gen)
    rows = np.concatenate([[0, 599, 600, n - 1], rng.choice(n, 20, replace=False)])
    assert np.array_equal(sym[rows], cport.topk_inner(emb[rows], emb, k))
    own = (sym // 600 == (np.arange(n) // 600)[:, None]).sum(1)           # the own cluster's ~600 rows lead every list
    assert own.min() >= 500


def test_epoch_layout_100k_shape_is_a_permutation_of_each_list(ops):
    """oea_epoch_layout at the EN-FR-100K shape (basic_model.py:234-235 shuffles both KGs' triple lists, batch.py:17-22 lays the
    batches out: 40 batches of 20,000 over ~800,000 triples): every slot of the layout holds a distinct triple of the right list,
    KG1's slice in front of KG2's in every batch, the layout equals the numpy restatement of the keyed permutation, and two epochs
    share no more positions than chance allows."""
    from openea_amd.modules.train.batch import EpochBatches
    from oracle import np_oracle as orc
    rng = np.random.RandomState(5)
    n1, n2 = 410000, 390000
    t1 = np.stack([rng.randint(0, 100000, n1), rng.randint(0, 300, n1), np.arange(n1)], 1).astype(np.int32)          # unique by column 2
    t2 = np.stack([rng.randint(100000, 200000, n2), rng.randint(0, 300, n2), np.arange(n1, n1 + n2)], 1).astype(np.int32)
    b = EpochBatches(t1, t2, 20000)
    gen = torch.Generator(device=ops.device())
    gen.manual_seed(23)
    slot = b.slot.cpu().numpy()
    layouts = []
    for e in range(2):
        b.shuffle(gen)
        d = b.dall.cpu().numpy()
        layouts.append(d.copy())
        ids = d[:, 2].astype(np.int64)
        assert len(np.unique(ids)) == len(ids)                                               # nothing twice
        assert np.array_equal((ids >= n1), (slot >= n1))                                     # every slot draws from its own list
        assert np.array_equal(d, np.concatenate([t1, t2])[ids])                              # rows travel whole
        for s in (0, len(b.splits) // 2, len(b.splits) - 1):
            o0, o1, sp = int(b.offsets[s]), int(b.offsets[s + 1]), int(b.splits[s])
            assert (ids[o0:o0 + sp] < n1).all() and (ids[o0 + sp:o1] >= n1).all()
        assert np.array_equal(d, orc.epoch_layout(t1, t2, slot, 23, e + 1))
    assert (layouts[0][:, 2] == layouts[1][:, 2]).mean() < 1e-3
