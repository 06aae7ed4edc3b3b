"""Golden fixtures for the host-side graph builders and the bootstrapping helpers: outputs of the REFERENCE's own
functions (approaches/gcn_align.py, alinet.py, rdgcn.py, bootea.py, modules/bootstrapping/alignment_finder.py,
modules/finding/alignment.py) on the synthetic "tiny" KG pair.

Run in the build container only (needs /root/reference):  python tests/golden/make_graph_golden.py
The reference modules import TensorFlow, igraph, gensim, ... at module level; they are imported here under stub
modules (nothing of TF runs: only numpy / scipy / pandas / pure-python functions are called).  tf.SparseTensor is
replaced by a recorder so that rdgcn.rfunc / get_sparse_tensor hand back their indices and values.
"""
import contextlib
import importlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
ROOT = '/root/reference/src/openea'
sys.path.insert(0, REPO)


class Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        m = Stub(self.__name__ + '.' + k)
        setattr(self, k, m)
        return m

    def __call__(self, *a, **k):
        return None


def import_reference():
    for name in ('tensorflow', 'igraph', 'graph_tool', 'graph_tool.all', 'gensim', 'gensim.models', 'gensim.models.word2vec',
                 'Levenshtein', 'scipy.sparse.linalg.eigen', 'scipy.sparse.linalg.eigen.arpack'):
        sys.modules[name] = Stub(name)
    sys.modules['tensorflow'].SparseTensor = lambda indices, values, dense_shape: dict(indices=indices, values=values,
                                                                                     dense_shape=dense_shape)

    def _pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    _pkg('openea', ROOT)
    _pkg('openea.modules', ROOT + '/modules')
    for sub in ('utils', 'load', 'train', 'finding', 'args', 'base', 'bootstrapping'):
        _pkg('openea.modules.' + sub, ROOT + '/modules/' + sub)
    _pkg('openea.models', ROOT + '/models')
    _pkg('openea.approaches', ROOT + '/approaches')
    ref = types.SimpleNamespace()
    ref.gcn = importlib.import_module('openea.approaches.gcn_align')
    ref.alinet = importlib.import_module('openea.approaches.alinet')
    ref.rdgcn = importlib.import_module('openea.approaches.rdgcn')
    ref.bootea = importlib.import_module('openea.approaches.bootea')
    ref.finder = importlib.import_module('openea.modules.bootstrapping.alignment_finder')
    ref.ali = importlib.import_module('openea.modules.finding.alignment')
    return ref


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def coo_sorted(coords, values):
    """(coords [n,2], values [n]) -> rows sorted by (row, col) as one float64 [n,3] array."""
    coords = np.asarray(coords, np.int64).reshape(-1, 2)
    values = np.asarray(values, np.float64).reshape(-1)
    order = np.lexsort((coords[:, 1], coords[:, 0]))
    return np.concatenate([coords[order].astype(np.float64), values[order, None]], axis=1)


def triples_sorted(triples):
    return np.array(sorted(tuple(int(x) for x in t) for t in triples), np.int64).reshape(-1, 3)


def main():
    ref = import_reference()
    from openea_amd.modules.load.synth import make_kgs
    kgs = make_kgs("tiny", mode="mapping", seed=0)
    n_ent, n_rel = kgs.entities_num, kgs.relations_num
    triples = kgs.kg1.relation_triples_list + kgs.kg2.relation_triples_list
    out = {}

    # ---- GCN-Align (gcn_align.py:542-675): functionality weights, weighted adjacency, normalised support ---------
    utils = ref.gcn.GCN_Utils(types.SimpleNamespace(), kgs)
    r2f, r2if = utils.func(triples), utils.ifunc(triples)
    out['gcn_r2f'] = np.array([r2f[r] for r in sorted(r2f)], np.float64)
    out['gcn_r2if'] = np.array([r2if[r] for r in sorted(r2if)], np.float64)
    adj = utils.get_weighted_adj(n_ent, triples)
    out['gcn_adj'] = coo_sorted(np.stack([adj.row, adj.col], 1), adj.data)
    coords, values, _ = utils.preprocess_adj(adj)
    out['gcn_support'] = coo_sorted(coords, values)
    attr_kgs = types.SimpleNamespace(
        kg1=types.SimpleNamespace(entity_attributes_dict={e: {(e * 7 + j) % 23 for j in range(1 + e % 4)} for e in range(0, 60, 2)}),
        kg2=types.SimpleNamespace(entity_attributes_dict={e: {(e * 5 + j) % 23 for j in range(1 + e % 3)} for e in range(1, 60, 2)}))
    out['gcn_attr'] = quiet(ref.gcn.load_attr, 60, attr_kgs)

    # ---- AliNet (alinet.py:138-287, 399-416, 459-493): 1-hop adjacency, 2-hop triples, seed-edge enhancement -----
    sup1 = [a for a, _ in kgs.train_links]
    sup2 = [b for _, b in kgs.train_links]
    kg1 = quiet(ref.alinet.AKG, kgs.kg1.relation_triples_set)
    kg2 = quiet(ref.alinet.AKG, kgs.kg2.relation_triples_set)
    en1, en2 = quiet(ref.alinet.enhance_triples, kg1, kg2, sup1, sup2)
    out['alinet_enhanced1'], out['alinet_enhanced2'] = triples_sorted(en1), triples_sorted(en2)
    half = len(kgs.test_entities1) // 2          # a strict subset, so that remove_unlinked_triples removes something
    linked = set(sup1 + sup2 + kgs.valid_entities1 + kgs.valid_entities2 + kgs.test_entities1[:half] + kgs.test_entities2[:half])
    for name, kg in (('kg1', kg1), ('kg2', kg2)):
        out['alinet_2hop_' + name] = triples_sorted(quiet(ref.alinet.generate_2hop_triples, kg, linked_ents=linked))
        out['alinet_2hop_all_' + name] = triples_sorted(quiet(ref.alinet.generate_2hop_triples, kg))
    one_adj, _ = quiet(ref.alinet.no_weighted_adj, n_ent, list(kg1.triples | kg2.triples | en1 | en2), False)
    out['alinet_one_adj'] = coo_sorted(one_adj[0], one_adj[1])
    rel_ht = ref.alinet.generate_rel_ht(sorted(kgs.kg1.relation_triples_set))
    out['alinet_rel_ht_sizes'] = np.array([len(rel_ht[r]) for r in sorted(rel_ht)], np.int64)

    # ---- RDGCN (rdgcn.py:17-72): relation incidence, primal adjacency --------------------------------------------
    head, tail, head_r, tail_r, r_mat = ref.rdgcn.rfunc(triples, n_ent, n_rel)
    out['rdgcn_head_r'], out['rdgcn_tail_r'] = head_r.astype(np.float32), tail_r.astype(np.float32)
    out['rdgcn_head_sizes'] = np.array([len(head.get(r, ())) for r in range(n_rel)], np.int64)
    out['rdgcn_tail_sizes'] = np.array([len(tail.get(r, ())) for r in range(n_rel)], np.int64)
    out['rdgcn_r_mat'] = np.concatenate([np.asarray(r_mat['indices'], np.int64), np.asarray(r_mat['values'], np.int64)[:, None]], 1)
    pos = ref.rdgcn.get_sparse_tensor(triples, n_ent)
    out['rdgcn_primal'] = coo_sorted(pos['indices'], pos['values'])

    # hard negatives (rdgcn.py:75-87): the k L1-nearest entities of every seed entity, in ascending distance
    rng = np.random.RandomState(5)
    layer = rng.standard_normal((300, 24)).astype(np.float32)
    ill = rng.permutation(300)[:40]
    out['rdgcn_neg_layer'], out['rdgcn_neg_ill'] = layer, ill.astype(np.int64)
    out['rdgcn_neg'] = ref.rdgcn.get_neg(ill, layer, 9).reshape(40, 9).astype(np.int64)

    # ---- bootstrapping (alignment_finder.py:12-76, bootea.py:35-138) ----------------------------------------------
    rng = np.random.RandomState(99)
    e1 = rng.standard_normal((90, 16)).astype(np.float32)
    e2 = (e1 + 0.6 * rng.standard_normal((90, 16))).astype(np.float32)
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 /= np.linalg.norm(e2, axis=1, keepdims=True)
    sim_mat = np.matmul(e1, e2.T)
    out['boot_e1'], out['boot_e2'] = e1, e2
    for th, k in ((0.5, 5), (0.7, 10), (0.2, 3)):
        pairs = quiet(ref.finder.find_alignment, sim_mat, th, k)
        out['boot_find_%g_%d' % (th, k)] = np.array(sorted(pairs) if pairs else [], np.int64).reshape(-1, 2)
    out['boot_nearest_7'] = np.array(sorted(ref.finder.search_nearest_k(sim_mat, 7)), np.int64)
    pre = {(i, (i * 7) % 90) for i in range(0, 90, 3)}
    cur = {(i, (i * 11) % 90) for i in range(0, 90, 2)}
    lab_x = quiet(ref.bootea.update_labeled_alignment_x, pre, cur, sim_mat)
    out['boot_update_x'] = np.array(sorted(lab_x), np.int64)
    out['boot_update_y'] = np.array(sorted(quiet(ref.bootea.update_labeled_alignment_y, lab_x, sim_mat)), np.int64)
    ents1, ents2 = sup1[:25], sup2[:25]
    t1, t2 = quiet(ref.bootea.generate_supervised_triples, kgs.kg1.rt_dict, kgs.kg1.hr_dict, kgs.kg2.rt_dict, kgs.kg2.hr_dict,
                   ents1, ents2)
    out['boot_sup_triples1'], out['boot_sup_triples2'] = triples_sorted(t1), triples_sorted(t2)
    b1, b2 = ref.bootea.generate_pos_batch(sorted(t1), sorted(t2), 2, 37)
    out['boot_pos_batch'] = np.array(list(b1) + list(b2), np.int64).reshape(-1, 3)

    for name, (greater, equal) in (('gt', (True, False)), ('ge', (True, True)), ('lt', (False, False)), ('le', (False, True))):
        th = float(sim_mat[3, 4]) if equal else 0.55            # an exact hit for the inclusive variants
        out['boot_filter_' + name] = np.array(sorted(ref.finder.filter_sim_mat(sim_mat, th, greater, equal)), np.int64).reshape(-1, 2)
        out['boot_filter_th_' + name] = np.array([th])
    # Gale-Shapley on the reference's own preference dictionaries (alignment.py:136-143, 170-224)
    idx = list(range(sim_mat.shape[0]))
    for rounds in (3, 100):
        suitors = ref.ali.arg_sort(idx, sim_mat, 'x', 'y')
        reviewers = ref.ali.arg_sort(idx, sim_mat.T, 'y', 'x')
        match = ref.ali.galeshapley(suitors, reviewers, rounds)
        out['gs_match_%d' % rounds] = np.array(sorted((int(a[1:]), int(b[1:])) for a, b in match.items()), np.int64).reshape(-1, 2)

    # ---- stable matching (alignment.py:87-224) ----------------------------------------------------------------------
    for csls in (0, 5):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref.ali.stable_alignment(e1, e2, 'inner', False, csls, 1)
        line = [ln for ln in buf.getvalue().splitlines() if 'stable alignment precision' in ln][-1]
        out['stable_precision_csls%d' % csls] = np.array([float(line.split('=')[1].split('%')[0])])

    # ---- writers (read.py:282-366): byte content of every file save_embeddings / save_results produce ---------------
    import hashlib
    import json
    import tempfile
    ref_read = importlib.import_module('openea.modules.load.read')
    wr = np.random.RandomState(3)
    ent = (wr.standard_normal((n_ent, 5)) * np.array([1, 1e-3, 1e3, 1e-8, 1])).astype(np.float32)
    rel = wr.standard_normal((n_rel, 5)).astype(np.float32)
    digests = {}
    with tempfile.TemporaryDirectory() as tmp:
        folder = tmp + "/out/"
        quiet(ref_read.save_embeddings, folder, kgs, ent, rel, None, mapping_mat=np.eye(5, dtype=np.float32))
        quiet(ref_read.save_results, folder, [(3, 4), (10, 7), (5, 5)])
        for name in sorted(os.listdir(folder)):
            digests[name] = hashlib.sha256(open(folder + name, 'rb').read()).hexdigest()
    with open(os.path.join(HERE, 'writers.json'), 'w') as fh:
        json.dump(digests, fh, indent=1)

    np.savez_compressed(os.path.join(HERE, 'graphs.npz'), **out)
    print('wrote', os.path.join(HERE, 'graphs.npz'), {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
