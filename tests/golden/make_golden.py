"""Generate golden fixtures by running the REFERENCE's own numpy functions in place.

Run in the build container only (needs /root/reference, which does not exist on the GPU
box):  python tests/golden/make_golden.py
The fixtures (small .npz / .json files next to this script) are committed; tests only read
them.  Import recipe: SURVEY.md Appendix C (stub `tensorflow`, bare package objects so that
openea/__init__.py -- which imports TF/igraph/gensim -- never runs).
"""
import contextlib
import importlib
import io
import json
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = '/root/reference/src/openea'


def import_reference():
    sys.modules['tensorflow'] = types.ModuleType('tensorflow')

    def _pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    _pkg('openea', ROOT)
    _pkg('openea.modules', ROOT + '/modules')
    for sub in ('utils', 'load', 'train', 'finding', 'args'):
        _pkg('openea.modules.' + sub, ROOT + '/modules/' + sub)
    ref = types.SimpleNamespace()
    ref.bat = importlib.import_module('openea.modules.train.batch')
    ref.sim = importlib.import_module('openea.modules.finding.similarity')
    ref.ali = importlib.import_module('openea.modules.finding.alignment')
    ref.ev = importlib.import_module('openea.modules.finding.evaluation')
    ref.util = importlib.import_module('openea.modules.utils.util')
    return ref


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def make_embeds(rng, n1, n2, d, noise):
    """e2[i] = e1[i] + noise for i < n1 (gold of row i is column i), extra rows random."""
    e1 = rng.standard_normal((n1, d)).astype(np.float32) / np.sqrt(d)
    e2 = rng.standard_normal((n2, d)).astype(np.float32) / np.sqrt(d)
    e2[:n1] = e1 + noise * rng.standard_normal((n1, d)).astype(np.float32) / np.sqrt(d)
    return e1.astype(np.float32), e2.astype(np.float32)


def main():
    ref = import_reference()
    rng = np.random.RandomState(1234)
    top_k = [1, 5, 10, 50]

    # ---- 1. greedy_alignment over metrics / csls ---------------------------------------
    e1, e2 = make_embeds(rng, 300, 420, 40, 0.9)
    out = {'e1': e1, 'e2': e2}
    cases = [('inner', False), ('inner', True), ('cosine', True), ('cosine', False),
             ('euclidean', False), ('manhattan', False)]
    for metric, normalize in cases:
        for csls in (0, 10):
            key = '%s_%d_%d' % (metric, int(normalize), csls)
            s = quiet(ref.sim.sim, e1, e2, metric=metric, normalize=normalize, csls_k=csls)
            out['sim_' + key] = s[:24].astype(np.float32)   # first 24 rows only (fixture size)
            # reference's own rank bookkeeping on its own sim matrix, accurate mode
            rank = np.array([int(np.where((-s[i]).argsort() == i)[0][0]) for i in range(len(s))])
            out['rank_' + key] = rank.astype(np.int32)
            rest, hits1, mr, mrr = quiet(ref.ali.greedy_alignment, e1, e2, top_k, 1, metric,
                                         normalize, csls, True)
            out['argmax_' + key] = np.array(sorted(rest), np.int32)[:, 1]
            out['stats_' + key] = np.array([hits1, mr, mrr], np.float64)
            # quick mode: hits must agree with accurate mode
            rest_q, hits1_q, _, _ = quiet(ref.ali.greedy_alignment, e1, e2, top_k, 1, metric,
                                          normalize, csls, False)
            out['hits1_quick_' + key] = np.array([hits1_q])
    # valid()/test() wrappers with a mapping matrix
    M = rng.standard_normal((40, 40)).astype(np.float32) / np.sqrt(40)
    out['mapping'] = M
    hits1, mrr = quiet(ref.ev.valid, e1, e2, M, top_k, 1, metric='inner', normalize=True)
    out['valid_mapping'] = np.array([hits1, mrr])
    np.savez_compressed(os.path.join(HERE, 'alignment.npz'), **out)

    # ---- 2. csls building blocks --------------------------------------------------------
    s = rng.standard_normal((64, 90)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'csls.npz'), s=s,
                        nearest_rows=ref.sim.calculate_nearest_k(s, 10),
                        nearest_cols=ref.sim.calculate_nearest_k(s.T, 10),
                        csls=ref.sim.csls_sim(s, 10))

    # ---- 3. neighbour search ------------------------------------------------------------
    n, d, k = 600, 32, 59
    emb = rng.standard_normal((n, d)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    entity_list = (np.arange(n) * 2 + 1).tolist()          # KG2-style odd ids
    dic = ref.bat.generate_neighbours_single_thread(emb, entity_list, k, 4)
    keys = np.array(sorted(dic.keys()), np.int32)
    nb = np.array([sorted(dic[kk]) for kk in keys], np.int32)
    np.savez_compressed(os.path.join(HERE, 'neighbours.npz'), emb=emb,
                        entity_list=np.array(entity_list, np.int32), k=k, keys=keys, nbrs=nb)

    # ---- 4. positive batching + task_divide + early_stop --------------------------------
    t1 = [(int(a), int(b), int(c)) for a, b, c in rng.randint(0, 500, (1037, 3))]
    t2 = [(int(a), int(b), int(c)) for a, b, c in rng.randint(0, 500, (811, 3))]
    pb = {}
    for step in (0, 1, 3, 9):
        pb['step%d' % step] = np.array(ref.bat.generate_pos_batch(t1, t2, 200, step), np.int32).reshape(-1, 3)
    np.savez_compressed(os.path.join(HERE, 'pos_batch.npz'), t1=np.array(t1, np.int32),
                        t2=np.array(t2, np.int32), **pb)
    misc = {'task_divide': {}, 'early_stop': []}
    for total, nn in ((10, 3), (10, 10), (10, 11), (10, 0), (18, 2), (0, 3), (29, 4)):
        misc['task_divide']['%d_%d' % (total, nn)] = [list(map(int, x)) for x in
                                                       ref.util.task_divide(list(range(total)), nn)]
    for f1, f2, f in ((-1, -1, 10.0), (-1, 10.0, 9.0), (10.0, 9.0, 8.0), (5.0, 6.0, 5.5),
                      (7.0, 7.0, 7.0)):
        r = quiet(ref.ev.early_stop, f1, f2, f)
        misc['early_stop'].append([f1, f2, f, r[0], r[1], bool(r[2])])
    with open(os.path.join(HERE, 'misc.json'), 'w') as fh:
        json.dump(misc, fh, indent=1)

    # ---- 5. negative sampling: reference outputs for distribution / invariant tests -----
    random.seed(7)
    np.random.seed(7)
    n_ent, n_rel, n_tri = 400, 12, 3000
    ents = list(range(0, 2 * n_ent, 2))
    tri = set()
    while len(tri) < n_tri:
        h = ents[min(int(rng.zipf(1.6)) - 1, n_ent - 1)]
        t = ents[rng.randint(n_ent)]
        tri.add((h, int(rng.randint(n_rel)), t))
    tri_list = sorted(tri)
    pos = tri_list[:256]
    knb = 40
    nbr = {e: random.sample(ents, knb) for e in ents}
    neg_u = ref.bat.generate_neg_triples_fast(pos, tri, ents, 10, neighbor=None)
    neg_t = ref.bat.generate_neg_triples_fast(pos, tri, ents, 10, neighbor=nbr)
    np.savez_compressed(os.path.join(HERE, 'neg_sampling.npz'), triples=np.array(tri_list, np.int32),
                        entity_list=np.array(ents, np.int32), pos=np.array(pos, np.int32),
                        nbr=np.array([nbr[e] for e in ents], np.int32),
                        neg_uniform=np.array(neg_u, np.int32), neg_truncated=np.array(neg_t, np.int32))
    print('golden fixtures written to', HERE)


if __name__ == '__main__':
    main()


def make_load_golden():
    """id assignment / swapping triples of the reference's loader on a tiny synthetic dataset written
    by openea_amd.modules.load.synth.write_dataset (URIs are plain strings)."""
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from openea_amd.modules.load.synth import write_dataset
    ref_kgs = importlib.import_module('openea.modules.load.kgs')
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        folder = write_dataset(tmp + "/tiny/", "tiny", seed=4) 
        for mode in ("mapping", "sharing", "swapping"):
            kgs = quiet(ref_kgs.read_kgs_from_folder, folder, "721_5fold/1/", mode, True)
            out[mode] = {
                "ent_ids1": kgs.kg1.entities_id_dict, "ent_ids2": kgs.kg2.entities_id_dict,
                "rel_ids1": kgs.kg1.relations_id_dict, "rel_ids2": kgs.kg2.relations_id_dict,
                "train_links": [list(map(int, x)) for x in kgs.train_links],
                "valid_links": [list(map(int, x)) for x in kgs.valid_links],
                "test_links": [list(map(int, x)) for x in kgs.test_links],
                "entities_num": kgs.entities_num, "relations_num": kgs.relations_num,
                "kg1_triples": sorted(map(list, kgs.kg1.relation_triples_set)),
                "kg2_triples": sorted(map(list, kgs.kg2.relation_triples_set)),
                "kg1_local_triples": len(kgs.kg1.local_relation_triples_set),
            }
        # reversed reading (kgs.py:102-123) and the DBP15K / DWY100K layout (kgs.py:134-169, with and without
        # remove_unlinked) on the same tiny dataset
        def summary(kgs):
            return {"ent_ids1": kgs.kg1.entities_id_dict, "ent_ids2": kgs.kg2.entities_id_dict,
                    "rel_ids1": kgs.kg1.relations_id_dict, "rel_ids2": kgs.kg2.relations_id_dict,
                    "train_links": sorted(list(map(int, x)) for x in kgs.train_links),
                    "valid_links": sorted(list(map(int, x)) for x in kgs.valid_links),
                    "test_links": sorted(list(map(int, x)) for x in kgs.test_links),
                    "entities_num": kgs.entities_num, "relations_num": kgs.relations_num,
                    "kg1_triples": sorted(map(list, kgs.kg1.relation_triples_set)),
                    "kg2_triples": sorted(map(list, kgs.kg2.relation_triples_set))}
        out["reversed_mapping"] = summary(quiet(ref_kgs.read_reversed_kgs_from_folder, folder, "721_5fold/1/", "mapping", True))
        dbp = make_dbp_layout(folder, tmp + "/dbp15k_tiny/")
        for remove in (False, True):
            out["dbp_%d" % remove] = summary(quiet(ref_kgs.read_kgs_from_folder, dbp, "0_3/", "mapping", True, remove))
    with open(os.path.join(HERE, 'load.json'), 'w') as fh:
        json.dump(out, fh)
    print('load golden written')


def make_dbp_layout(src_folder, dst_folder, division="0_3/"):
    """the tiny dataset in the DBP15K / DWY100K file layout; a third of the links is dropped so that remove_unlinked has work"""
    import shutil
    os.makedirs(dst_folder + division, exist_ok=True)
    shutil.copy(src_folder + "rel_triples_1", dst_folder + division + "triples_1")
    shutil.copy(src_folder + "rel_triples_2", dst_folder + division + "triples_2")
    train = open(src_folder + "721_5fold/1/train_links").read().splitlines()
    test = open(src_folder + "721_5fold/1/test_links").read().splitlines()
    open(dst_folder + division + "sup_ent_ids", "w").write("\n".join(train[: len(train) * 2 // 3]) + "\n")
    open(dst_folder + division + "ref_pairs", "w").write("\n".join(test[: len(test) * 2 // 3]) + "\n")
    return dst_folder


if __name__ == '__main__':
    import_reference()
    make_load_golden()
