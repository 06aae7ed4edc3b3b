"""Host-side graph builders and bootstrapping helpers against outputs of the REFERENCE's own functions
(tests/golden/graphs.npz, written by tests/golden/make_graph_golden.py from approaches/gcn_align.py, alinet.py, rdgcn.py,
bootea.py, modules/bootstrapping/alignment_finder.py, modules/finding/alignment.py on the synthetic "tiny" KG pair).
The device halves (candidate search, stable matching on the device similarity) are at the end, marked gpu."""
import contextlib
import io
import os
import types

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(HERE, "golden", "graphs.npz"))


@pytest.fixture(scope="module")
def kgs():
    from openea_amd.modules.load.synth import make_kgs
    return make_kgs("tiny", mode="mapping", seed=0)


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def coo_sorted(rows, cols, values):
    rows, cols = np.asarray(rows, np.int64), np.asarray(cols, np.int64)
    values = np.asarray(values, np.float64)
    order = np.lexsort((cols, rows))
    return np.stack([rows[order].astype(np.float64), cols[order].astype(np.float64), values[order]], axis=1)


def triples_sorted(triples):
    return np.array(sorted(tuple(int(x) for x in t) for t in triples), np.int64).reshape(-1, 3)


def assert_coo(mine, ref, tol=1e-12):
    assert mine.shape == ref.shape
    assert np.array_equal(mine[:, :2], ref[:, :2])
    np.testing.assert_allclose(mine[:, 2], ref[:, 2], rtol=tol, atol=tol)


def test_gcn_align_adjacency_matches_reference(g, kgs):
    """gcn_align.py:610-664 (functionality weights, weighted adjacency) and :566-578 (D^-1/2 (A + I) D^-1/2)."""
    from openea_amd.approaches.gcn_align import GCN_Utils, load_attr
    triples = kgs.kg1.relation_triples_list + kgs.kg2.relation_triples_list
    u = GCN_Utils(types.SimpleNamespace(), kgs)
    r2f, r2if = u.func(triples), u.ifunc(triples)
    np.testing.assert_array_equal(np.array([r2f[r] for r in sorted(r2f)]), g["gcn_r2f"])
    np.testing.assert_array_equal(np.array([r2if[r] for r in sorted(r2if)]), g["gcn_r2if"])
    adj = u.get_weighted_adj(kgs.entities_num, triples)
    assert_coo(coo_sorted(adj.row, adj.col, adj.data), g["gcn_adj"])
    sup = u.preprocess_adj(adj).tocoo()
    assert_coo(coo_sorted(sup.row, sup.col, sup.data), g["gcn_support"])
    attr_kgs = types.SimpleNamespace(
        kg1=types.SimpleNamespace(entity_attributes_dict={e: {(e * 7 + j) % 23 for j in range(1 + e % 4)} for e in range(0, 60, 2)}),
        kg2=types.SimpleNamespace(entity_attributes_dict={e: {(e * 5 + j) % 23 for j in range(1 + e % 3)} for e in range(1, 60, 2)}))
    np.testing.assert_array_equal(np.asarray(load_attr(60, attr_kgs).todense(), np.float32), g["gcn_attr"])


def test_alinet_builders_match_reference(g, kgs):
    """alinet.py:155-181 (1-hop adjacency), :250-287 (2-hop triples incl. the pattern cut), :399-416 (seed-edge
    enhancement), :138-144."""
    from openea_amd.approaches import alinet
    sup1 = [a for a, _ in kgs.train_links]
    sup2 = [b for _, b in kgs.train_links]
    kg1, kg2 = alinet.AKG(kgs.kg1.relation_triples_set), alinet.AKG(kgs.kg2.relation_triples_set)
    en1, en2 = quiet(alinet.enhance_triples, kg1, kg2, sup1, sup2)
    assert np.array_equal(triples_sorted(en1), g["alinet_enhanced1"]) and np.array_equal(triples_sorted(en2), g["alinet_enhanced2"])
    half = len(kgs.test_entities1) // 2
    linked = set(sup1 + sup2 + kgs.valid_entities1 + kgs.valid_entities2 + kgs.test_entities1[:half] + kgs.test_entities2[:half])
    for name, kg in (("kg1", kg1), ("kg2", kg2)):
        assert np.array_equal(triples_sorted(quiet(alinet.generate_2hop_triples, kg, linked_ents=linked)), g["alinet_2hop_" + name])
        assert np.array_equal(triples_sorted(quiet(alinet.generate_2hop_triples, kg)), g["alinet_2hop_all_" + name])
    one = alinet.no_weighted_adj(kgs.entities_num, list(kg1.triples | kg2.triples | en1 | en2))
    one = one.tocoo() if hasattr(one, "tocoo") else one
    if isinstance(one, tuple):
        mine = coo_sorted(one[0][:, 0], one[0][:, 1], one[1])
    else:
        mine = coo_sorted(one.row, one.col, one.data)
    assert_coo(mine, g["alinet_one_adj"])
    rel_ht = alinet.generate_rel_ht(sorted(kgs.kg1.relation_triples_set))
    assert np.array_equal(np.array([len(rel_ht[r]) for r in sorted(rel_ht)]), g["alinet_rel_ht_sizes"])


def test_rdgcn_structures_match_reference(g, kgs):
    """rdgcn.py:17-72: per-relation head / tail sets, the relation-labelled edge list in triple order, the primal
    adjacency 1 / sqrt(deg deg) with the reference's degree rule."""
    from openea_amd.approaches import rdgcn
    triples = kgs.kg1.relation_triples_list + kgs.kg2.relation_triples_list
    n_ent, n_rel = kgs.entities_num, kgs.relations_num
    head, tail, ind, val = rdgcn.rfunc(triples, n_ent, n_rel)
    assert np.array_equal(np.array([len(head.get(r, ())) for r in range(n_rel)]), g["rdgcn_head_sizes"])
    assert np.array_equal(np.array([len(tail.get(r, ())) for r in range(n_rel)]), g["rdgcn_tail_sizes"])
    for sets, dense in ((head, g["rdgcn_head_r"]), (tail, g["rdgcn_tail_r"])):
        mine = np.zeros_like(dense)
        for r, ents in sets.items():
            mine[list(ents), r] = 1
        assert np.array_equal(mine, dense)
    assert np.array_equal(np.concatenate([ind, val[:, None]], 1), g["rdgcn_r_mat"])
    rows, cols, vals = rdgcn.get_sparse_tensor(triples, n_ent)
    assert_coo(coo_sorted(rows, cols, vals), g["rdgcn_primal"], tol=1e-7)        # ours hands the values over as fp32


def test_bootstrapping_helpers_match_reference(g, kgs):
    """bootea.py:35-77 (label editing), :107-138 (supervised triples, positive batches)."""
    from openea_amd.approaches import bootea
    sim_mat = np.matmul(g["boot_e1"], g["boot_e2"].T)
    pre = {(i, (i * 7) % 90) for i in range(0, 90, 3)}
    cur = {(i, (i * 11) % 90) for i in range(0, 90, 2)}
    lab_x = quiet(bootea.update_labeled_alignment_x, pre, cur, sim_mat)
    assert np.array_equal(np.array(sorted(lab_x)), g["boot_update_x"])
    assert np.array_equal(np.array(sorted(quiet(bootea.update_labeled_alignment_y, lab_x, sim_mat))), g["boot_update_y"])
    sup1 = [a for a, _ in kgs.train_links][:25]
    sup2 = [b for _, b in kgs.train_links][:25]
    t1, t2 = quiet(bootea.generate_supervised_triples, kgs.kg1.rt_dict, kgs.kg1.hr_dict, kgs.kg2.rt_dict, kgs.kg2.hr_dict, sup1, sup2)
    assert np.array_equal(triples_sorted(t1), g["boot_sup_triples1"]) and np.array_equal(triples_sorted(t2), g["boot_sup_triples2"])
    b1, b2 = bootea.generate_pos_batch(sorted(t1), sorted(t2), 2, 37)
    assert np.array_equal(np.array(list(b1) + list(b2)).reshape(-1, 3), g["boot_pos_batch"])


def test_stable_matching_oracle_matches_reference(g):
    """alignment.py:87-224 printed 'stable alignment precision = x%': the oracle's Gale-Shapley restatement on the
    same similarity matrices (plain and CSLS) reaches the same precision."""
    from oracle import np_oracle as orc
    for csls in (0, 5):
        s = orc.sim(g["boot_e1"], g["boot_e2"], metric="inner", normalize=False, csls_k=csls)
        match = orc.stable_alignment(s)
        pairs = match.items() if isinstance(match, dict) else match
        correct = sum(1 for i, j in pairs if i == j)
        assert abs(correct / len(s) * 100 - float(g["stable_precision_csls%d" % csls][0])) < 1e-3


def test_writers_are_byte_compatible_with_reference(kgs, tmp_path):
    """read.py:282-366: every file save_embeddings / save_results write (.npy payloads, id tsv files, the text dumps of the
    embeddings, the result pairs) has the bytes the reference's own writer produced for the same inputs."""
    import hashlib
    import json
    from openea_amd.modules.load import read as rd
    golden = json.load(open(os.path.join(HERE, "golden", "writers.json")))
    wr = np.random.RandomState(3)
    ent = (wr.standard_normal((kgs.entities_num, 5)) * np.array([1, 1e-3, 1e3, 1e-8, 1])).astype(np.float32)
    rel = wr.standard_normal((kgs.relations_num, 5)).astype(np.float32)
    folder = str(tmp_path) + "/out/"
    quiet(rd.save_embeddings, folder, kgs, ent, rel, None, mapping_mat=np.eye(5, dtype=np.float32))
    quiet(rd.save_results, folder, [(3, 4), (10, 7), (5, 5)])
    assert sorted(os.listdir(folder)) == sorted(golden)
    for name, digest in golden.items():
        assert hashlib.sha256(open(folder + name, "rb").read()).hexdigest() == digest, name


# ---- device halves ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_candidate_search_and_stable_matching_on_device(g, capsys):
    """alignment_finder.py:28-76 (threshold & top-k candidates, nearest-k lists) and alignment.py:87-134 through the
    device path: same candidate sets, same precision."""
    pytest.importorskip("torch")
    from openea_amd.modules.bootstrapping import alignment_finder as af
    from openea_amd.modules.finding.alignment import stable_alignment
    sim = af.PairSim(g["boot_e1"], g["boot_e2"])
    for th, k in ((0.5, 5), (0.7, 10), (0.2, 3)):
        pairs, _ = af.find_alignment(sim, th, k)
        assert np.array_equal(np.array(sorted(pairs), np.int64).reshape(-1, 2), g["boot_find_%g_%d" % (th, k)])
    near = af.search_nearest_k_device(sim, 7)
    mine = sorted((i, int(j)) for i in range(near.shape[0]) for j in near[i])
    assert np.array_equal(np.array(mine, np.int64), g["boot_nearest_7"])
    for csls in (0, 5):
        stable_alignment(g["boot_e1"], g["boot_e2"], "inner", False, csls, 1)
        out = capsys.readouterr().out
        line = [ln for ln in out.splitlines() if "stable alignment precision" in ln][-1]
        assert abs(float(line.split("=")[1].split("%")[0]) - float(g["stable_precision_csls%d" % csls][0])) < 1e-3


@pytest.mark.gpu
def test_rdgcn_hard_negatives_on_device(g):
    """rdgcn.py:75-87 (scipy cdist cityblock + argsort[:k]) -> the fp64 manhattan tiles + row select: the same k
    entities for every seed (the reference lists them by ascending distance; the loss sums over them)."""
    pytest.importorskip("torch")
    from openea_amd import ops
    from openea_amd.approaches.rdgcn import get_neg
    layer = ops.to_table(g["rdgcn_neg_layer"])
    got = get_neg(ops.to_ids(g["rdgcn_neg_ill"].astype(np.int32)), layer, 24, 9).cpu().numpy().reshape(40, 9)
    assert np.array_equal(np.sort(got, axis=1), np.sort(g["rdgcn_neg"], axis=1))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["gcn_se", "gcn_ae"])
def test_gcn_units_on_device_equal_reference_graph(tag):
    """The device GCN_Align_Unit (structure and attribute unit) on the inputs of tests/golden/tf_graphs.npz: outputs, loss
    and the weight after one SGD epoch equal what the reference's own GCN_Align_Unit graph gives (its loss and its
    finite-difference gradient, gcn_align.py:498-539 under tests/golden/tf_shim.py)."""
    pytest.importorskip("torch")
    import scipy.sparse as sp
    from openea_amd import ops
    from openea_amd.approaches.gcn_align import DeviceCSR, GCN_Align_Unit
    t = np.load(os.path.join(HERE, "golden", "tf_graphs.npz"))
    W = t[tag + "_var_weights"].astype(np.float32)
    coords, values = t["gcn_support_coords"], t["gcn_support_values"]
    n, d = int(coords.max()) + 1, W.shape[1]
    dev = ops.device()
    adj = DeviceCSR(sp.csr_matrix((values, (coords[:, 0], coords[:, 1])), shape=(n, n)), dev)
    feats = None
    if tag == "gcn_ae":
        fc = t["gcn_feat_coords"]
        feats = DeviceCSR(sp.csr_matrix((np.ones(len(fc)), (fc[:, 0], fc[:, 1])), shape=(n, W.shape[0])), dev)
    lr = 1e-2
    unit = GCN_Align_Unit(types.SimpleNamespace(neg_triple_num=3, gamma=3.0, learning_rate=lr), adj, W.shape[0], d, t["gcn_ill"],
                          features=feats)
    unit.W[:, :d] = ops.to_table(W)[:, :d]
    negs = tuple(ops.to_ids(t["gcn_" + k].astype(np.int32)) for k in ("neg_left", "neg_right", "neg2_left", "neg2_right"))
    unit.train_step(negs)
    np.testing.assert_allclose(unit.outputs[:, :d].cpu().numpy(), t[tag + "_outputs"], rtol=1e-5, atol=1e-5)
    ref_loss = float(t[tag + "_loss"][0])
    assert abs(unit.pop_loss() - ref_loss) <= 1e-5 * ref_loss
    grad = (W.astype(np.float64) - unit.W[:, :d].cpu().numpy().astype(np.float64)) / lr
    ref = t[tag + "_grad_weights"]
    assert np.abs(grad - ref).max() <= 1e-3 * max(np.abs(ref).max(), 1.0)


@pytest.mark.gpu
def test_rdgcn_layer_on_device_equals_reference_graph():
    """rdgcn.py:162-338 (Layer.build: relation features -> dual self / dual attention -> primal sparse attention, twice;
    two diagonal GCN layers with highway gates; L1 hinge) built by the reference's own code under tests/golden/tf_shim.py:
    with the 22 variables copied over in creation order, our Layer gives the same output layer, the same loss and
    (autograd) the reference graph's finite-difference gradients."""
    torch = pytest.importorskip("torch")
    from openea_amd import ops
    from openea_amd.approaches import rdgcn
    t = np.load(os.path.join(HERE, "golden", "tf_graphs.npz"))
    tri1 = [tuple(int(x) for x in r) for r in t["rdgcn_tri1"]]
    tri2 = [tuple(int(x) for x in r) for r in t["rdgcn_tri2"]]
    n, nr, d, k = 36, 5, 4, 3
    kgs = types.SimpleNamespace(train_links=[tuple(int(x) for x in p) for p in t["rdgcn_links"]], entities_num=n, relations_num=nr,
                                kg1=types.SimpleNamespace(relation_triples_list=tri1), kg2=types.SimpleNamespace(relation_triples_list=tri2))
    args = types.SimpleNamespace(dim=d, dropout=0.0, gamma=1.0, neg_triple_num=k, alpha=0.1, beta=0.3)
    dev = ops.device()
    layer = rdgcn.Layer(args, kgs, t["rdgcn_var_v00"].astype(np.float32), dev)
    params = layer.params()
    assert len(params) == int(t["rdgcn_n_vars"][0]) == 22
    with torch.no_grad():
        for i, p in enumerate(params):
            v = t["rdgcn_var_v%02d" % i]
            v = v[0] if v.ndim == 3 else v                       # conv1d kernels are [1, C, F]
            p.copy_(torch.from_numpy(v.astype(np.float32)).reshape(p.shape).to(dev))
    out = layer.forward()
    np.testing.assert_allclose(out.detach().cpu().numpy()[:, :d], t["rdgcn_outputs"], rtol=1e-4, atol=1e-5)
    negs = tuple(ops.to_ids(t["rdgcn_" + kk].astype(np.int32)) for kk in ("neg_left", "neg_right", "neg2_left", "neg2_right"))
    loss = layer.loss(out, negs)
    ref_loss = float(t["rdgcn_loss"][0])
    assert abs(float(loss.detach()) - ref_loss) <= 1e-5 * ref_loss
    loss.backward()
    checked = 0
    for i, p in enumerate(params):
        ref = t["rdgcn_grad_v%02d" % i]
        ref = ref[0] if ref.ndim == 3 else ref
        got = p.grad.detach().cpu().numpy().reshape(ref.shape)
        assert np.abs(got - ref).max() <= 2e-3 * max(np.abs(ref).max(), 0.05), i
        checked += 1
    assert checked == 22


@pytest.mark.gpu
@pytest.mark.parametrize("grouping", ["runs", "row", "reorder"])
def test_alinet_model_on_device_equals_reference_graph(grouping):
    """alinet.py:539-677 (GraphConvolution, AliNetGraphAttentionLayer, HighwayLayer with keras BatchNormalization in
    inference mode), :784-866 (_define_model, compute_loss, compute_rel_loss) built by the reference's own
    `_generate_rel_graph` under tests/golden/tf_shim.py: with the 17 variables copied over by name, our layers give the
    same layer outputs, the same loss and (autograd through the HIP aggregate / attention operators) the reference
    graph's finite-difference gradients."""
    torch = pytest.importorskip("torch")
    from openea_amd import ops
    from openea_amd.approaches import alinet
    from openea_amd.models.graph_ops import EdgeGraph
    t = np.load(os.path.join(HERE, "golden", "tf_graphs.npz"))
    n, dims, dev = 30, [8, 8, 4], ops.device()
    # preprocess_adj (alinet.py:44-57) hands the edges over in COLUMN-major order; the stand-in's sparse_softmax groups runs
    # of consecutive equal rows (TF-1's kernel on a non-canonical tensor) -> grouping="runs" on the same order
    assert not (np.diff(t["alinet_two_coords"][:, 0]) >= 0).all()
    m = alinet.AliNet()
    m.args = types.SimpleNamespace(layer_dims=dims, neg_margin=1.5, neg_margin_balance=0.1, rel_param=0.01, dropout=0.0,
                                   learning_rate=0.001)
    m.kgs = types.SimpleNamespace(entities_num=n)
    m.dev, m._rng, m.rel_win_size = dev, np.random.RandomState(0), 3
    m.adj = [EdgeGraph(t["alinet_one_coords"][:, 0], t["alinet_one_coords"][:, 1], t["alinet_one_values"], (n, n), dev),
             EdgeGraph(t["alinet_two_coords"][:, 0], t["alinet_two_coords"][:, 1], t["alinet_two_values"], (n, n), dev, grouping=grouping)]
    tag = "alinet" if grouping == "runs" else "alinet_" + grouping          # the reference graph under the matching tf.sparse_softmax stand-in
    m._get_variable()
    m._define_model()
    g0, g1, att, hw = m.one_hop_layers[0], m.one_hop_layers[1], m.two_hop_layers[0], m.highways[0]
    by_name = {"init_embedding": m.init_embedding, "gcn_0_kernel_0": g0.kernel, "gcn_0_bias": g0.bias, "bn1_gamma": g0.bn.gamma,
               "bn1_beta": g0.bn.beta, "alinet_0_kernel": att.kernel, "alinet_0_kernel_1": att.kernel1, "alinet_0_kernel_2": att.kernel2,
               "bn2_gamma": att.bn.gamma, "bn2_beta": att.bn.beta, "highwaykernel": hw.weight, "bn3_gamma": hw.bn.gamma,
               "bn3_beta": hw.bn.beta, "gcn_1_kernel_0": g1.kernel, "gcn_1_bias": g1.bias, "bn4_gamma": g1.bn.gamma, "bn4_beta": g1.bn.beta}
    assert sorted(by_name) == sorted(str(x) for x in t["alinet_var_names"]) and len(by_name) == len(m._params)
    with torch.no_grad():
        for name, p in by_name.items():
            p.copy_(torch.from_numpy(t["alinet_var_" + name].astype(np.float32)).reshape(p.shape).to(dev))
    outs = m._forward()
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().cpu().numpy(), t[tag + "_out%d" % i], rtol=1e-4, atol=1e-5)
    emb = m._concat_train(outs)
    pos = torch.from_numpy(t["alinet_pos"]).to(dev)
    neg = torch.from_numpy(t["alinet_neg"]).to(dev)
    loss = m.compute_loss(emb, pos, neg) + m.compute_rel_loss(emb, torch.from_numpy(t["alinet_hs"]).to(dev),
                                                              torch.from_numpy(t["alinet_ts"]).to(dev))
    ref_loss = float(t[tag + "_loss"][0])
    assert abs(float(loss.detach()) - ref_loss) <= 1e-5 * ref_loss
    loss.backward()
    for name, p in by_name.items():
        ref = t[tag + "_grad_" + name]
        # a parameter without a gradient: the attention logits' kernels when every softmax group is one edge (their
        # gradient is exactly zero -- the reference's finite differences say the same)
        got = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().numpy().reshape(ref.shape)
        assert np.abs(got - ref).max() <= 2e-3 * max(np.abs(ref).max(), 0.05), name


@pytest.mark.gpu
@pytest.mark.parametrize("tag,kw,neg_key,kind", [
    ("aligne_triple", dict(loss="limited", pos_margin=0.01, neg_margin=2.0, balance=0.2), "neg2", "transe"),
    ("bootea_triple", dict(loss="limited", pos_margin=0.01, neg_margin=2.0, balance=0.2), "neg2", "transe"),
    ("bootea_align", dict(loss="align"), None, "transe"),
    ("mtranse_triple", dict(loss="positive"), None, "transe"),
    ("transe_triple", dict(loss="margin-based", margin=1.5), "neg1", "transe"),
    ("bootea_transh_triple", dict(loss="limited", pos_margin=0.01, neg_margin=2.0, balance=0.2), "neg2", "transh"),
    ("transh_triple", dict(loss="margin-based", margin=1.5), "neg1", "transh"),
    ("transd_triple", dict(loss="margin-based", margin=1.5), "neg1", "transd"),
])
def test_device_step_equals_reference_graph(tag, kw, neg_key, kind):
    """The HIP translational step itself against the reference's own graph code (tests/golden/tf_graphs.npz): batch loss and
    the gradient of every variable, read back from one SGD step."""
    torch = pytest.importorskip("torch")
    from openea_amd import ops
    t = np.load(os.path.join(HERE, "golden", "tf_graphs.npz"))
    names = {"transe": ["ent_embeds", "rel_embeds"], "transh": ["ent_embeds", "rel_embeds", "normal_vector"],
             "transd": ["ent_embeds", "ent_transfer", "rel_embeds", "rel_transfer"]}[kind]
    host = {nm: t["%s_var_%s" % (tag, nm)].astype(np.float32) for nm in names}
    d = host["ent_embeds"].shape[1]
    if kind == "transd":
        ent_h = np.concatenate([host["ent_embeds"], host["ent_transfer"]])
        rel_h = np.concatenate([host["rel_embeds"], host["rel_transfer"]])
    else:
        ent_h, rel_h = host["ent_embeds"], host["rel_embeds"]
    ent, rel = ops.to_table(ent_h), ops.to_table(rel_h)
    nrm = ops.to_table(host["normal_vector"]) if kind == "transh" else None
    lr = 1e-3
    cfg = ops.make_step_cfg(loss_norm="L2", ent_l2_norm=True, rel_l2_norm=True, optimizer="SGD", lr=lr, neg_group_k=0, normal=nrm,
                            transfer_bases=(len(host["ent_embeds"]), len(host["rel_embeds"])) if kind == "transd" else None, **kw)
    ws = ops.step_workspace(ent.shape[0], rel.shape[0], ent.shape[1])
    acc = torch.zeros(1, dtype=torch.float64, device=ent.device)
    pos = ops.to_ids(t["pos"].astype(np.int32))
    neg = ops.to_ids(t[neg_key].astype(np.int32)) if neg_key else None
    ops.triple_step(ent, None, rel, None, d, pos, neg, cfg, ws, acc)
    ref_loss = float(t[tag + "_loss"][0])
    assert abs(float(acc.item()) - ref_loss) <= 2e-5 * ref_loss
    ge = (ent_h.astype(np.float64) - ent[:, :d].cpu().numpy()) / lr
    gr = (rel_h.astype(np.float64) - rel[:, :d].cpu().numpy()) / lr
    E, R = len(host["ent_embeds"]), len(host["rel_embeds"])
    got = {"ent_embeds": ge[:E], "rel_embeds": gr[:R]}
    if kind == "transd":
        got.update(ent_transfer=ge[E:], rel_transfer=gr[R:])
    if kind == "transh":
        got["normal_vector"] = (host["normal_vector"].astype(np.float64) - nrm[:, :d].cpu().numpy()) / lr
    for nm in names:
        ref = t["%s_grad_%s" % (tag, nm)]
        assert np.abs(got[nm] - ref).max() <= 1e-3 * max(np.abs(ref).max(), 1.0), nm


@pytest.mark.gpu
@pytest.mark.parametrize("tag,neg_key", [("rotate_triple", "neg2"), ("rotate_align", None)])
def test_device_rotate_step_equals_reference_graph(tag, neg_key):
    """oea_rotate_step (fp64) against bootea_rotate.py's own graph code: loss to 1e-10, gradients to the accuracy of the
    finite differences."""
    torch = pytest.importorskip("torch")
    from openea_amd import ops
    t = np.load(os.path.join(HERE, "golden", "tf_graphs.npz"))
    re_, im_, rel_h = t[tag + "_var_re_ent_embeds"], t[tag + "_var_im_ent_embeds"], t[tag + "_var_rel_embeds"]
    d, E = re_.shape[1], len(re_)
    ent_h = np.concatenate([re_, im_])
    ent, rel = ops.to_table64(ent_h), ops.to_table64(rel_h)
    cfg = ops.make_rotate_cfg(float(t["rotate_gamma"][0]), d, ent_l2_norm=True, rel_l2_norm=False, optimizer="SGD", lr=1e-6)
    assert abs(cfg.phase_scale - float(t["rotate_phase_scale"][0])) < 1e-9
    ws = ops.rotate_workspace(E, len(rel_h), ent.shape[1])
    acc = torch.zeros(1, dtype=torch.float64, device=ent.device)
    neg = ops.to_ids(t[neg_key].astype(np.int32)) if neg_key else None
    ops.rotate_step(ent, None, rel, None, d, ops.to_ids(t["pos"].astype(np.int32)), neg, 0, cfg, ws, acc)
    assert abs(float(acc.item()) - float(t[tag + "_loss"][0])) < 1e-10
    ge = (ent_h - ent[:, :d].cpu().numpy()) / 1e-6
    gr = (rel_h - rel[:, :d].cpu().numpy()) / 1e-6
    for got, nm in ((ge[:E], "re_ent_embeds"), (ge[E:], "im_ent_embeds"), (gr, "rel_embeds")):
        ref = t["%s_grad_%s" % (tag, nm)]
        assert np.abs(got - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1.0), nm


@pytest.mark.gpu
def test_device_mapping_step_equals_reference_graph():
    """oea_mapping_step + the apply phase (SGD) against the reference's own add_mapping_module graph
    (modules/base/mapping.py:9-19, losses.py:76-80): loss, gradient of the entity table and of the mapping matrix."""
    torch = pytest.importorskip("torch")
    from openea_amd import ops
    t = np.load(os.path.join(HERE, "golden", "tf_graphs.npz"))
    tag = "mtranse_mapping"
    ent_h = t[tag + "_var_ent_embeds"].astype(np.float32)
    m_h = t[tag + "_var_mapping_matrix"].astype(np.float32)
    n_ent, d = ent_h.shape
    te = ops.to_table(ent_h)
    tm = torch.from_numpy(m_h).to(te.device).contiguous()
    rel = ops.to_table(np.zeros((2, d), np.float32))
    lr, alpha = 1e-3, float(t["mtranse_alpha"][0])
    cfg = ops.make_step_cfg(loss="positive", optimizer="SGD", lr=lr, ent_l2_norm=True, rel_l2_norm=True)
    ws = ops.step_workspace(n_ent, 2, te.shape[1])
    loss = torch.zeros(1, dtype=torch.float64, device=te.device)
    dummy = torch.zeros(1, dtype=torch.float64, device=te.device)
    empty = torch.zeros((0, 3), dtype=torch.int32, device=te.device)
    ids1, ids2 = ops.to_ids(np.array([0, 3, 8], np.int32)), ops.to_ids(np.array([2, 4, 9], np.int32))
    ops.mapping_step(te, d, True, ids1, ids2, tm, None, alpha, lr, "SGD", ws, n_ent, 2, loss, None)
    ops.triple_step(te, None, rel, None, d, empty, None, cfg, ws, dummy, phase=ops.PHASE_APPLY)
    ref_loss = float(t[tag + "_loss"][0])
    assert abs(float(loss.item()) - ref_loss) <= 2e-5 * ref_loss
    for got, ref in (((ent_h.astype(np.float64) - te[:, :d].cpu().numpy()) / lr, t[tag + "_grad_ent_embeds"]),
                     ((m_h.astype(np.float64) - tm.cpu().numpy()) / lr, t[tag + "_grad_mapping_matrix"])):
        assert np.abs(got - ref).max() <= 2e-3 * max(np.abs(ref).max(), 1.0)


def test_alinet_2hop_array_form_equals_set_form(kgs):
    from openea_amd.approaches import alinet
    kg = alinet.AKG(kgs.kg1.relation_triples_set)
    as_set = quiet(alinet.generate_2hop_triples, kg)
    as_arr = quiet(alinet.generate_2hop_triples, kg, as_array=True)
    assert as_arr.dtype == np.int64 and np.array_equal(as_arr, triples_sorted(as_set))


def test_host_matrix_helpers_match_reference(g):
    """alignment_finder.py:54-76 (filter_sim_mat, search_nearest_k) and alignment.py:136-224 (arg_sort, galeshapley with
    its round limit) under the reference's names, on host matrices."""
    from openea_amd.modules.bootstrapping import alignment_finder as af
    from openea_amd.modules.finding import alignment as ali
    sim_mat = np.matmul(g["boot_e1"], g["boot_e2"].T)
    for name, (greater, equal) in (("gt", (True, False)), ("ge", (True, True)), ("lt", (False, False)), ("le", (False, True))):
        got = af.filter_sim_mat(sim_mat, float(g["boot_filter_th_" + name][0]), greater, equal)
        assert np.array_equal(np.array(sorted(got), np.int64).reshape(-1, 2), g["boot_filter_" + name])
    near = af.search_nearest_k(sim_mat, 7)
    assert np.array_equal(np.array(sorted((int(i), int(j)) for i, j in near), np.int64), g["boot_nearest_7"])
    idx = list(range(sim_mat.shape[0]))
    for rounds in (3, 100):
        suitors = ali.arg_sort(idx, sim_mat, "x", "y")
        reviewers = ali.arg_sort(idx, sim_mat.T, "y", "x")
        match = ali.galeshapley(suitors, reviewers, rounds)
        got = np.array(sorted((int(a[1:]), int(b[1:])) for a, b in match.items()), np.int64).reshape(-1, 2)
        assert np.array_equal(got, g["gs_match_%d" % rounds])
    # the matchers under the reference's names: one-to-one, and the exact one is at least as heavy as the greedy one
    pairs = {(int(i), int(j)) for i, j in g["boot_find_0.5_5"]}
    exact, greedy = af.mwgm(pairs, sim_mat, af.mwgm_igraph), af.mwgm(pairs, sim_mat, af.mwgm_graph_tool)
    for m in (exact, greedy):
        assert len({i for i, _ in m}) == len(m) == len({j for _, j in m}) and set(m) <= pairs
    assert sum(sim_mat[i, j] for i, j in exact) >= sum(sim_mat[i, j] for i, j in greedy) - 1e-9


# ---- the same builders on the device (csrc/graph_build.hip) against the reference's outputs ------------------------------
@pytest.mark.gpu
def test_device_gcn_align_adjacency_matches_reference(g, kgs):
    """oea_build_weighted_adj == gcn_align.py:610-664 + 566-578 (reference fixtures), entry ORDER of the support included."""
    from openea_amd import ops
    from openea_amd.approaches.gcn_align import GCN_Utils
    triples = kgs.kg1.relation_triples_list + kgs.kg2.relation_triples_list
    n_rel = 1 + max(r for _, r, _ in triples)
    b = ops.build_weighted_adj(triples, kgs.entities_num, n_rel, raw=True)
    rels = sorted({r for _, r, _ in triples})
    np.testing.assert_array_equal(b["r2f"][rels], g["gcn_r2f"])
    np.testing.assert_array_equal(b["r2if"][rels], g["gcn_r2if"])
    assert_coo(coo_sorted(*b["adj"]), g["gcn_adj"])
    assert_coo(coo_sorted(*b["support"]), g["gcn_support"])
    u = GCN_Utils(types.SimpleNamespace(), kgs)
    sup = u.preprocess_adj(u.get_weighted_adj(kgs.entities_num, triples)).tocoo()
    assert np.array_equal(sup.row, b["support"][0]) and np.array_equal(sup.col, b["support"][1])     # scipy's (col, row) order


@pytest.mark.gpu
def test_device_alinet_builders_match_reference(g, kgs):
    """oea_build_unweighted_adj / oea_build_2hop == alinet.py:155-181, 250-287 (reference fixtures)."""
    from openea_amd.approaches import alinet
    sup1 = [a for a, _ in kgs.train_links]
    sup2 = [b for _, b in kgs.train_links]
    kg1, kg2 = alinet.AKG(kgs.kg1.relation_triples_set), alinet.AKG(kgs.kg2.relation_triples_set)
    en1, en2 = quiet(alinet.enhance_triples, kg1, kg2, sup1, sup2)
    half = len(kgs.test_entities1) // 2
    linked = set(sup1 + sup2 + kgs.valid_entities1 + kgs.valid_entities2 + kgs.test_entities1[:half] + kgs.test_entities2[:half])
    for name, kg in (("kg1", kg1), ("kg2", kg2)):
        assert np.array_equal(triples_sorted(quiet(alinet.generate_2hop_triples_device, kg, linked)), g["alinet_2hop_" + name])
        assert np.array_equal(triples_sorted(quiet(alinet.generate_2hop_triples_device, kg)), g["alinet_2hop_all_" + name])
    triples = list(kg1.triples | kg2.triples | en1 | en2)
    one = alinet.no_weighted_adj_device(kgs.entities_num, triples)
    assert_coo(coo_sorted(one.row, one.col, one.data), g["alinet_one_adj"])
    host = alinet.no_weighted_adj(kgs.entities_num, triples)
    assert np.array_equal(host.row, one.row) and np.array_equal(host.col, one.col)      # the order the 'runs' grouping sees
    np.testing.assert_allclose(one.data, host.data, rtol=1e-14)


@pytest.mark.gpu
def test_device_rdgcn_structures_match_reference(g, kgs):
    """oea_build_primal_adj / oea_build_dual_adj == rdgcn.py:45-72, 268-277."""
    from openea_amd import ops
    from openea_amd.approaches import rdgcn
    triples = kgs.kg1.relation_triples_list + kgs.kg2.relation_triples_list
    n_ent, n_rel = kgs.entities_num, kgs.relations_num
    rows, cols, vals = ops.build_primal_adj(triples, n_ent)
    assert_coo(coo_sorted(rows, cols, vals), g["rdgcn_primal"], tol=1e-7)
    h_rows, h_cols, h_vals = rdgcn.get_sparse_tensor(triples, n_ent)
    assert np.array_equal(h_vals[np.lexsort((h_cols, h_rows))], vals[np.lexsort((cols, rows))])      # bit-identical fp32 values
    head, tail, _, _ = rdgcn.rfunc(triples, n_ent, n_rel)
    count_r = len(head)
    dual = ops.build_dual_adj(triples, count_r).cpu().numpy()
    assert np.array_equal(dual, rdgcn.dual_adjacency(head, tail, count_r))


@pytest.mark.gpu
def test_device_builders_equal_host_builders_15k():
    """the device builders on the EN-FR-15K-shaped synthetic pair: same entries as the numpy restatements (which equal the
    reference's functions on the fixtures above), values to 1e-13, 2-hop triples identical."""
    from openea_amd import ops
    from openea_amd.approaches import alinet, rdgcn
    from openea_amd.approaches.gcn_align import GCN_Utils
    from openea_amd.modules.load.synth import make_kgs
    kgs15 = make_kgs("EN-FR-15K-V1", mode="mapping", seed=0)
    triples = kgs15.kg1.relation_triples_list + kgs15.kg2.relation_triples_list
    n_ent, n_rel = kgs15.entities_num, kgs15.relations_num
    u = GCN_Utils(types.SimpleNamespace(), kgs15)
    sup = u.preprocess_adj(u.get_weighted_adj(n_ent, triples)).tocoo()
    b = ops.build_weighted_adj(triples, n_ent, n_rel)
    assert np.array_equal(sup.row, b["support"][0]) and np.array_equal(sup.col, b["support"][1])
    np.testing.assert_allclose(b["support"][2], sup.data, rtol=1e-13)
    kg1 = alinet.AKG(kgs15.kg1.relation_triples_set)
    linked = set(kgs15.train_entities1 + kgs15.valid_entities1 + kgs15.test_entities1)
    for le in (None, linked):
        a = quiet(alinet.generate_2hop_triples, kg1, le, as_array=True)
        d = quiet(alinet.generate_2hop_triples_device, kg1, le)
        assert a.shape == d.shape and np.array_equal(a, d)
    one_h, one_d = alinet.no_weighted_adj(n_ent, triples), alinet.no_weighted_adj_device(n_ent, triples)
    assert np.array_equal(one_h.row, one_d.row) and np.array_equal(one_h.col, one_d.col)
    np.testing.assert_allclose(one_d.data, one_h.data, rtol=1e-14)
    hr, hc, hv = rdgcn.get_sparse_tensor(triples, n_ent)
    dr, dc, dv = ops.build_primal_adj(triples, n_ent)
    assert np.array_equal(coo_sorted(hr, hc, hv), coo_sorted(dr, dc, dv))
    head, tail, _, _ = rdgcn.rfunc(triples, n_ent, n_rel)
    assert np.array_equal(ops.build_dual_adj(triples, len(head)).cpu().numpy(), rdgcn.dual_adjacency(head, tail, len(head)))
