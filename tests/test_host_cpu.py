"""CPU tests of the host side: the C ABI library loads and exports every symbol the header declares,
the loader reproduces the reference's id layout (golden from the reference's own loader), the batch
index arithmetic, args objects, and the product path failing loudly without a GPU."""
import json
import os
import re
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from openea_amd import _lib
    header = open(os.path.join(ROOT, "include", "openea_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(oea_[a-z0-9_]+)\s*\(", header))
    declared -= {"oea_store", "oea_step_cfg", "oea_sampler_side"}
    assert len(declared) >= 30
    lib = _lib.load(require_device=False)          # binds every prototype; AttributeError on a missing symbol
    for name in sorted(declared):
        assert hasattr(lib, name), "header declares %s but the library does not export it" % name
        assert name in _lib.PROTOTYPES, "no ctypes prototype for %s" % name
    assert set(_lib.PROTOTYPES) <= declared
    assert lib.oea_version() >= 100


def test_struct_layouts_match_header():
    import ctypes as C
    from openea_amd import _lib
    assert C.sizeof(_lib.StepCfg) == 88            # 12 x 4-byte fields + 2 pointers (8-byte aligned) + 2 x int32 + 3 floats + int32 of oea_step_cfg
    assert C.sizeof(_lib.SamplerSide) == 64        # 5 pointers/u64 + 2 int32 + filter pointer + u64
    assert C.sizeof(_lib.RotateCfg) == 72          # 6 doubles + int64 + 4 x int32 of oea_rotate_cfg
    assert C.sizeof(_lib.CsrSplit) == 80           # 4 pointers + 3 int32 (+ pad) + 2 pointers + int64 + pointer of oea_csr_split
    assert C.sizeof(_lib.AttnGraph) == 30 * 8      # 13 pointers + 17 int64 of oea_attn_graph


def test_host_side_planners_of_the_library():
    """pure host entry points (no device): workspace planners answer without a GPU and follow their documented domains."""
    from openea_amd import _lib
    lib = _lib.load(require_device=False)
    # symmetric neighbour search: the stream form from 12,288 rows (round 4), the segment lists from 32,768, up to the select's
    # segment table (~140,000 rows)
    assert lib.oea_topk_sym_workspace_bytes(10000, 200) == 0
    assert 0 < lib.oea_topk_sym_workspace_bytes(15000, 1499) < 2 << 30
    assert 0 < lib.oea_topk_sym_workspace_bytes(20000, 400) < 2 << 30
    need = lib.oea_topk_sym_workspace_bytes(100000, 2000)
    assert 5 << 30 < need < 12 << 30                # ~8 GB: record streams + compact lists + overflow pool (segment lists: ~30 GB)
    assert lib.oea_topk_sym_workspace_bytes(400000, 8000) == 0
    assert lib.oea_topk_workspace_bytes(1000, 100000) == 1000 * 100000 * 4
    # one-sweep CSLS means: from 4,096 x 4,096 on, k <= 32
    assert lib.oea_csls_means_workspace_bytes(1000, 1000, 10) == 0
    assert lib.oea_csls_means_workspace_bytes(10500, 10500, 10) > 0
    assert lib.oea_csls_means_workspace_bytes(70000, 70000, 10) > 0
    assert lib.oea_csls_means_workspace_bytes(10500, 10500, 64) == 0
    # sparse attention: statistics of every sub-segment / segment + one value per CSR slot of the call's ranges
    g = _lib.AttnGraph()
    g.n_sub, g.n_seg, g.agg_slot1, g.t_slot1 = 1000, 600, 5000, 7000
    assert lib.oea_sparse_attn_workspace_floats(g) >= 2 * 1000 + 2 * 600 + 7000
    # entity-id partition arithmetic
    assert lib.oea_part_rows_per_rank(30000, 8) == 3750 and lib.oea_part_rows_per_rank(30001, 8) == 3751
    assert lib.oea_part_send_floats(30000, 76, 8) == 8 * 3750 * 77


def test_host_side_plans_of_round_3():
    """Planners that run without a device: the row chunks of the X^T dY kernel (workspace = chunks x k1 x k2 partial tiles, none
    when one chunk covers the rows), and the chunked per-row sums behind RDGCN's per-relation logit gradient (every gather
    position in exactly one chunk of <= 2,048, chunks grouped by source row)."""
    import torch
    from openea_amd import _lib
    from openea_amd.models.graph_ops import gather_few_plan
    lib = _lib.load(require_device=False)
    assert lib.oea_gemm_tn_workspace_floats(100, 8, 12) == 0                      # one chunk: written in place
    for m, k1, k2 in ((200000, 500, 400), (200000, 300, 300), (5000, 300, 300), (70001, 500, 400)):
        n = lib.oea_gemm_tn_workspace_floats(m, k1, k2)
        assert n % (k1 * k2) == 0
        chunks = n // (k1 * k2)
        tiles = -(-k1 // 128) * -(-k2 // 128)
        assert 2 <= chunks <= -(-m // 256) and chunks * tiles <= 2048 + tiles, (m, k1, k2, chunks)
    assert lib.oea_colsum_blocks(0) == 0 and lib.oea_colsum_blocks(500) == 1 and lib.oea_colsum_blocks(200000) == 3125
    assert lib.oea_colsum_blocks(10 ** 7) == 4096
    rng = np.random.RandomState(4)
    n_src, n_idx = 50, 30000
    idx = torch.tensor(np.minimum(rng.zipf(1.3, n_idx) - 1, n_src - 1))
    order, chunk_ptr, row_chunk_ptr = gather_few_plan(idx, n_src)
    cp, rcp = chunk_ptr.numpy(), row_chunk_ptr.numpy()
    assert cp[0] == 0 and cp[-1] == n_idx and (np.diff(cp) > 0).all() and np.diff(cp).max() <= 2048
    assert sorted(order.tolist()) == list(range(n_idx))
    for r in range(n_src):
        pos = order[cp[rcp[r]]: cp[rcp[r + 1]]].long()
        assert (idx[pos] == r).all() and len(pos) == int((idx == r).sum())
        assert (np.diff(pos.numpy()) > 0).all()                                    # position order kept inside a row


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from openea_amd import OpenEAHipError, ops
    with pytest.raises(OpenEAHipError, match="no HIP device|no CPU fallback"):
        ops.lib()
    from openea_amd.modules.finding.similarity import sim
    with pytest.raises(OpenEAHipError):
        sim(np.zeros((4, 8), np.float32), np.zeros((4, 8), np.float32))
    from openea_amd.modules.train import batch as bat
    with pytest.raises(OpenEAHipError):
        bat.generate_neg_triples_fast([(0, 0, 1)], {(0, 0, 1)}, [0, 1, 2], 1)


def test_no_oracle_import_in_product():
    """the product package never imports oracle/ (only tests, smoke() and bench's cpu_baseline may)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "openea_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)


@pytest.mark.parametrize("mode", ["mapping", "sharing", "swapping"])
def test_loader_matches_reference(golden_dir, mode):
    from openea_amd.modules.load.kgs import read_kgs_from_folder
    from openea_amd.modules.load.synth import write_dataset
    g = json.load(open(os.path.join(golden_dir, "load.json")))[mode]
    with tempfile.TemporaryDirectory() as tmp:
        folder = write_dataset(tmp + "/tiny/", "tiny", seed=4)
        kgs = read_kgs_from_folder(folder, "721_5fold/1/", mode, True)
    assert kgs.kg1.entities_id_dict == g["ent_ids1"] and kgs.kg2.entities_id_dict == g["ent_ids2"]
    assert kgs.kg1.relations_id_dict == g["rel_ids1"] and kgs.kg2.relations_id_dict == g["rel_ids2"]
    assert [list(x) for x in kgs.train_links] == g["train_links"]
    assert [list(x) for x in kgs.valid_links] == g["valid_links"]
    assert [list(x) for x in kgs.test_links] == g["test_links"]
    assert kgs.entities_num == g["entities_num"] and kgs.relations_num == g["relations_num"]
    assert sorted(map(list, kgs.kg1.relation_triples_set)) == g["kg1_triples"]
    assert sorted(map(list, kgs.kg2.relation_triples_set)) == g["kg2_triples"]
    assert len(kgs.kg1.local_relation_triples_set) == g["kg1_local_triples"]
    if mode == "mapping":        # KG1 even / KG2 odd ids in descending frequency (read.py:64-92)
        assert all(v % 2 == 0 for v in kgs.kg1.entities_id_dict.values())
        assert all(v % 2 == 1 for v in kgs.kg2.entities_id_dict.values())


@pytest.mark.parametrize("variant", ["reversed_mapping", "dbp_0", "dbp_1"])
def test_reversed_and_dbp_loaders_match_reference(golden_dir, variant):
    """kgs.py:102-123 (KGs and links turned round) and :134-189 (DBP15K / DWY100K layout: triples_{1,2}, sup_ent_ids,
    ref_pairs; remove_unlinked iterates the two filters to a fixed point) against the reference loader's output."""
    import shutil
    from openea_amd.modules.load.kgs import read_kgs_from_folder, read_reversed_kgs_from_folder
    from openea_amd.modules.load.synth import write_dataset
    g = json.load(open(os.path.join(golden_dir, "load.json")))[variant]
    with tempfile.TemporaryDirectory() as tmp:
        folder = write_dataset(tmp + "/tiny/", "tiny", seed=4)
        if variant == "reversed_mapping":
            kgs = read_reversed_kgs_from_folder(folder, "721_5fold/1/", "mapping", True)
        else:
            dst = tmp + "/dbp15k_tiny/"
            os.makedirs(dst + "0_3/")
            shutil.copy(folder + "rel_triples_1", dst + "0_3/triples_1")
            shutil.copy(folder + "rel_triples_2", dst + "0_3/triples_2")
            train = open(folder + "721_5fold/1/train_links").read().splitlines()
            test = open(folder + "721_5fold/1/test_links").read().splitlines()
            open(dst + "0_3/sup_ent_ids", "w").write("\n".join(train[: len(train) * 2 // 3]) + "\n")
            open(dst + "0_3/ref_pairs", "w").write("\n".join(test[: len(test) * 2 // 3]) + "\n")
            kgs = read_kgs_from_folder(dst, "0_3/", "mapping", True, variant == "dbp_1")      # dispatched on the folder name
    assert kgs.kg1.entities_id_dict == g["ent_ids1"] and kgs.kg2.entities_id_dict == g["ent_ids2"]
    assert kgs.kg1.relations_id_dict == g["rel_ids1"] and kgs.kg2.relations_id_dict == g["rel_ids2"]
    for part in ("train_links", "valid_links", "test_links"):
        assert sorted(list(x) for x in getattr(kgs, part)) == g[part]
    assert kgs.entities_num == g["entities_num"] and kgs.relations_num == g["relations_num"]
    assert sorted(map(list, kgs.kg1.relation_triples_set)) == g["kg1_triples"]
    assert sorted(map(list, kgs.kg2.relation_triples_set)) == g["kg2_triples"]


def test_pos_batching_matches_reference(golden_dir):
    from openea_amd.modules.train import batch as bat
    g = np.load(os.path.join(golden_dir, "pos_batch.npz"))
    t1 = [tuple(x) for x in g['t1'].tolist()]
    t2 = [tuple(x) for x in g['t2'].tolist()]
    for step in (0, 1, 3, 9):
        got = np.array(bat.generate_pos_batch(t1, t2, 200, step), np.int32).reshape(-1, 3)
        assert np.array_equal(got, g['step%d' % step])
    assert bat.batch_sizes(47334 + 18534, 40864 + 16028, 5000) == (int((47334 + 18534) / 122760 * 5000), 5000 - int((47334 + 18534) / 122760 * 5000))


def test_util_and_early_stop_match_reference(golden_dir):
    from openea_amd.modules.finding.evaluation import early_stop
    from openea_amd.modules.utils.util import merge_dic, task_divide
    misc = json.load(open(os.path.join(golden_dir, "misc.json")))
    for key, ref in misc['task_divide'].items():
        total, n = map(int, key.split('_'))
        assert [list(map(int, x)) for x in task_divide(list(range(total)), n)] == ref
    for f1, f2, f, r0, r1, r2 in misc['early_stop']:
        assert early_stop(f1, f2, f) == (r0, r1, r2)
    assert merge_dic({1: 2}, {1: 3, 4: 5}) == {1: 3, 4: 5}


def test_args_objects(tmp_path):
    from openea_amd.modules.args.args_hander import load_args
    from openea_amd.run.default_args import get_args
    a = get_args("BootEA")
    assert (a.dim, a.batch_size, a.neg_triple_num, a.truncated_epsilon, a.loss, a.optimizer) == (100, 5000, 10, 0.9, "limited", "Adagrad")
    assert int((1 - a.truncated_epsilon) * 15000) == 1499
    a100 = get_args("BootEA", "100K")
    assert a100.batch_size == 20000 and int((1 - a100.truncated_epsilon) * 100000) == 2000
    p = tmp_path / "args.json"
    p.write_text(json.dumps({"dim": 75, "embedding_module": "AlignE", "top_k": [1, 5]}))
    b = load_args(str(p))
    assert b.dim == 75 and b.top_k == [1, 5]


def test_save_formats(tmp_path):
    """ent_embeds.npy payload + id tsv files as the reference's save_embeddings writes them."""
    from openea_amd.modules.load import read as rd
    from openea_amd.modules.load.synth import make_kgs
    kgs = make_kgs("tiny", "mapping")
    ent = np.arange(kgs.entities_num * 4, dtype=np.float32).reshape(-1, 4)
    rd.save_embeddings(str(tmp_path) + "/", kgs, ent, ent[:kgs.relations_num], None, mapping_mat=np.eye(4, dtype=np.float32))
    back = np.load(str(tmp_path) + "/ent_embeds.npy")
    assert back.dtype == np.float32 and back.flags.c_contiguous and np.array_equal(back, ent)
    ids = rd.read_dict(str(tmp_path) + "/kg1_ent_ids")
    assert ids == kgs.kg1.entities_id_dict
    line = open(str(tmp_path) + "/kg1_ent_embeds_txt").readline().split(' ')
    assert line[0] in kgs.kg1.entities_id_dict and len(line) == 5
    rd.save_results(str(tmp_path) + "/", [(1, 2), (3, 4)])
    assert rd.read_pair_ids(str(tmp_path) + "/alignment_results_12") == [(1, 2), (3, 4)]


def test_bootstrapping_matchings():
    """alignment_finder.py:83-140 stand-ins: the exact matcher reaches the brute-force optimum, the greedy one is a
    maximal one-to-one matching with at least half of it."""
    import itertools
    from openea_amd.modules.bootstrapping.alignment_finder import greedy_weight_matching, max_weight_matching
    rng = np.random.RandomState(0)
    for trial in range(20):
        nl, nr = rng.randint(2, 7), rng.randint(2, 7)
        pairs = [(i, j) for i in range(nl) for j in range(nr) if rng.rand() < 0.6]
        if not pairs:
            continue
        w = rng.rand(len(pairs)) + 0.05
        wd = dict(zip(pairs, w))
        best = 0.0
        for m in range(1, min(nl, nr) + 1):
            for sub in itertools.combinations(pairs, m):
                if len({p[0] for p in sub}) == m and len({p[1] for p in sub}) == m:
                    best = max(best, sum(wd[p] for p in sub))
        exact = max_weight_matching(pairs, w)
        assert len({p[0] for p in exact}) == len(exact) == len({p[1] for p in exact})
        assert abs(sum(wd[p] for p in exact) - best) < 1e-9
        greedy = greedy_weight_matching(pairs, w)
        assert len({p[0] for p in greedy}) == len(greedy) == len({p[1] for p in greedy})
        assert sum(wd[p] for p in greedy) >= 0.5 * best - 1e-12
        free_l = set(range(nl)) - {p[0] for p in greedy}
        free_r = set(range(nr)) - {p[1] for p in greedy}
        assert not any(p[0] in free_l and p[1] in free_r for p in pairs)       # maximal


def test_rdgcn_name_vectors(tmp_path):
    """rdgcn.py:415-464 restated: names -> 4 word ids (unknown / padding -> zero vector) -> summed word vectors."""
    from openea_amd.approaches.rdgcn import name_vectors, read_word_vectors
    vec = tmp_path / "w.vec"
    rng = np.random.RandomState(0)
    vocab = ["alpha", "beta", "gamma", "delta", "e7"]
    mat = rng.standard_normal((len(vocab), 6))
    with open(vec, "w") as f:
        f.write("%d %d\n" % (len(vocab), 6))
        for w, v in zip(vocab, mat):
            f.write(w + " " + " ".join("%.6f" % x for x in v) + " \n")        # fastText lines end with a space
    words, word_em = read_word_vectors(str(vec))
    assert words == vocab and word_em.shape == (len(vocab) + 1, 6) and not word_em[-1].any()
    np.testing.assert_allclose(word_em[:-1], np.round(mat, 6), atol=1e-9)
    names = {0: "alpha beta", 2: "gamma, (delta) unknownword alpha beta", 3: "e7"}
    emb, ids = name_vectors(names, 5, words, word_em)
    u = len(vocab)
    assert ids.tolist() == [[0, 1, u, u], [u, u, u, u], [2, 3, u, 0], [4, u, u, u], [u, u, u, u]]
    np.testing.assert_allclose(emb[0], word_em[0] + word_em[1])
    np.testing.assert_allclose(emb[2], word_em[2] + word_em[3] + word_em[0])      # punctuation stripped, 4-word window
    assert not emb[1].any() and not emb[4].any()
    # padding quirk: with no unknown word anywhere the pad id is the LAST vocabulary word
    emb2, ids2 = name_vectors({0: "alpha"}, 1, words, word_em)
    assert ids2.tolist() == [[0, 4, 4, 4]]


def test_galeshapley_topk_equals_reference_loop():
    """truncated preference lists + similarity comparisons == the reference's loop on full argsorted lists."""
    from oracle import np_oracle as orc
    from openea_amd.modules.finding.alignment import galeshapley_topk
    rng = np.random.RandomState(0)
    for n, cut in ((12, 100), (40, 100), (40, 5), (25, 3)):
        s = rng.rand(n, n)
        s[np.arange(n), np.arange(n)] += 0.3
        if n == 25:
            s = np.round(s, 1)                                   # ties: index order decides, as in a stable argsort
        ref = orc.stable_alignment(s, cut)
        order = np.argsort(-s, axis=1, kind="stable")[:, :min(cut, n)]
        got = galeshapley_topk(order, np.take_along_axis(s, order, 1), lambda i, j: float(s[i, j]), cut)
        assert got == ref
        assert len(set(got.values())) == len(got)


def test_rdgcn_dual_adjacency_equals_set_loops():
    """rdgcn.py:268-277 (R^2 python set intersections) == the sparse incidence product used here, bit for bit."""
    from openea_amd.approaches.rdgcn import dual_adjacency
    rng = np.random.RandomState(0)
    R, E = 23, 200
    head = {r: set(rng.randint(0, E, rng.randint(1, 40)).tolist()) for r in range(R) if r != 5}
    tail = {r: set(rng.randint(0, E, rng.randint(1, 40)).tolist()) for r in range(R) if r != 7}
    ref = np.zeros((R, R), np.float32)
    for i in range(R):
        hi, ti = head.get(i, set()), tail.get(i, set())
        for j in range(R):
            hj, tj = head.get(j, set()), tail.get(j, set())
            a_h = len(hi & hj) / len(hi | hj) if (hi | hj) else 0.0
            a_t = len(ti & tj) / len(ti | tj) if (ti | tj) else 0.0
            ref[i, j] = a_h + a_t
    assert np.array_equal(dual_adjacency(head, tail, R), ref)


def test_native_greedy_matching_equals_python():
    """oea_greedy_matching (host C++) == the python reference loop, ties included."""
    from openea_amd import ops
    from openea_amd.modules.bootstrapping.alignment_finder import greedy_weight_matching
    rng = np.random.RandomState(3)
    for n_edges in (0, 1, 500, 5000):
        pairs = list(dict.fromkeys((int(a), int(b)) for a, b in zip(rng.randint(0, 80, n_edges), rng.randint(0, 90, n_edges))))
        w = np.round(rng.rand(len(pairs)), 1).astype(np.float32)
        m = ops.greedy_matching([p[0] for p in pairs], [p[1] for p in pairs], w)
        assert {p for p, keep in zip(pairs, m) if keep} == greedy_weight_matching(pairs, w)


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE: one torch.distributed.run with N ranks on 127.0.0.1 and the same flags"""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_entity_id_cache_of_the_neighbour_refresh(monkeypatch):
    """refresh_neighbours converts a KG's entity list to a device tensor once: the same list object hits the cache, a copy or an
    edited list does not (models/trainer.py:_entity_ids_on_device)."""
    from openea_amd.models import trainer
    calls = []
    monkeypatch.setattr(trainer.ops, "to_ids", lambda a, d=None: (calls.append(1), np.array(a))[1])
    trainer._IDS_CACHE.clear()
    ents = [3, 4, 5, 6]
    a = trainer._entity_ids_on_device(ents, "cpu")
    assert trainer._entity_ids_on_device(ents, "cpu") is a and len(calls) == 1
    trainer._entity_ids_on_device(list(ents), "cpu")
    assert len(calls) == 2
    ents.append(9)
    assert len(trainer._entity_ids_on_device(ents, "cpu")) == 5 and len(calls) == 3
    ents[2] = 77                                   # an edit the three probes see
    assert trainer._entity_ids_on_device(ents, "cpu")[2] == 77 and len(calls) == 4
    trainer._IDS_CACHE.clear()


def test_alignment_pairs_behave_as_the_reference_set():
    """greedy_alignment's `alignment_rest` without the Python tuples (modules/finding/alignment.py:AlignmentPairs): the same set"""
    from openea_amd.modules.finding.alignment import AlignmentPairs
    am = np.array([2, 0, 2, 1], np.int32)
    ref = set(zip(range(4), am.tolist()))
    got = AlignmentPairs(am)
    assert len(got) == 4 and got == ref and ref == got and set(got) == ref
    assert (0, 2) in got and (1, 2) not in got and (9, 0) not in got and "x" not in got
    assert sorted(got) == sorted(ref) and [(i, j) for i, j in got] == [(0, 2), (1, 0), (2, 2), (3, 1)]
    assert (got & {(0, 2), (5, 5)}) == {(0, 2)} and (got - {(0, 2)}) == ref - {(0, 2)} and (got | {(7, 7)}) == ref | {(7, 7)}
    assert got != ref - {(0, 2)} and not (got < ref) and got <= ref


def test_bench_line_is_compact_strict_json():
    """the LAST stdout line of bench.py must reach the driver: BENCH_r04's 21.8 KB line was not parsed (the driver keeps an 8 KB
    tail).  compact_line() of a full-size result (round 4's own detail, with kernel-name-keyed counter dictionaries) stays
    under 4 KB, is strict JSON, and carries the contract's keys + roofline + roofline_eval + cpu_baseline."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_driver_like.json")))
    assert len(json.dumps(full)) > 20000
    # the round-5 layout: the side shape block is shape_15k, the eval legs sit in extra
    full["extra"]["shape_15k"] = full["extra"].pop("shape_100k")
    full["extra"]["gnn"]["alinet_eval_70000x1200"] = {"inner_ms": 21.0, "inner_csls10_ms": 55.5, "frac": 0.67, "csls_frac": 0.5,
                                                      "peak": 833.3, "bf16_prefilter": True, "records_per_row": 12.5, "fallback": False,
                                                      "note": "x" * 3000}
    full["extra"]["exchange_phases"] = {"grad_us": 1.0, "pack_us": 2.0, "note": "y" * 2000, "steps_timed": 20}
    full["extra"]["junk"] = {"k" * 90: list(range(500))}
    line = bench.compact_line(full, "bench_detail.json")
    assert "\n" not in line and len(line.encode()) <= bench.COMPACT_LIMIT < 8000

    def no_constants(x):
        raise ValueError("non-strict JSON constant %s" % x)
    j = json.loads(line, parse_constant=no_constants)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_eval", "cpu_baseline", "extra", "detail"):
        assert k in j, k
    assert j["value"] == full["value"] and j["ms_per_step"] == full["ms_per_step"] and "workload" in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["frac"] == round(r["achieved"] / r["peak"], 4) and "traffic" in r and r["avg_kernel_us"] > 0
    assert j["roofline_eval"]["bound"] == "mfma" and 0 < j["roofline_eval"]["frac"] <= 1
    assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["cores"] >= 1 and j["cpu_baseline"]["kind"] == "port"
    x = j["extra"]
    assert x["shape_15k"]["value"] > 0 and x["gnn"]["alinet_eval_70000x1200"]["inner_ms"] == 21.0
    assert "note" not in x["gnn"]["alinet_eval_70000x1200"] and "junk" not in x and "note" not in x["exchange_phases"]
    # a block that alone would push the line past the limit is dropped, never the contract's keys
    full["extra"]["shape_15k"]["value"] = 1.0
    full["config"]["workload"] = "w" * 5000
    full["extra"]["gnn"] = {"error": "e" * 9000}
    line2 = bench.compact_line(full, None)
    assert len(line2.encode()) <= bench.COMPACT_LIMIT and json.loads(line2)["roofline"]["frac"] == r["frac"]


def test_refresh_shards_only_where_it_pays(monkeypatch):
    """refresh_neighbours on several ranks (basic_model.py:267-289 runs on one device): a rank's row block against the whole table
    cannot use the symmetric search, so sharding pays only where (general search / N + all-gather of the [n, k] table) undercuts
    every rank running the symmetric search itself; both modes return the same sets."""
    from openea_amd.models.trainer import refresh_is_sharded
    monkeypatch.delenv("OEA_REFRESH_MODE", raising=False)
    assert not refresh_is_sharded(100000, 2000, 1)
    assert refresh_is_sharded(100000, 2000, 2) and refresh_is_sharded(100000, 2000, 8)
    assert not refresh_is_sharded(100000, 20000, 2)               # a table whose all-gather costs more than the search it saves
    assert refresh_is_sharded(9000, 100, 2)                       # below the symmetric path's range
    monkeypatch.setenv("OEA_REFRESH_MODE", "shard")
    assert refresh_is_sharded(100000, 2000, 2)
    monkeypatch.setenv("OEA_REFRESH_MODE", "replicate")
    assert not refresh_is_sharded(100000, 2000, 8) and not refresh_is_sharded(100000, 2000, 1)
