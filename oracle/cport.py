"""ctypes binding of oracle/c/oracle.c (test infrastructure only -- see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

METRIC = {"inner": 0, "manhattan": 1, "euclidean": 2}
LOSS = {"margin-based": 0, "limited": 1, "logistic": 2, "positive": 3, "align": 4}
OPT = {"SGD": 0, "Adagrad": 1}


def build(force=False):
    src = os.path.join(_HERE, "c", "oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "c", "oracle.c")
        if not os.path.exists(_LIB_PATH) or (os.path.exists(src) and os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
            build()                                   # missing or older than its source
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_triple_step.restype = C.c_double
        _lib.oracle_sample_negatives.restype = C.c_int
        _lib.oracle_tripleset_contains.restype = C.c_int
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class StepCfg(C.Structure):
    _fields_ = [("loss_kind", C.c_int), ("l1", C.c_int), ("margin", C.c_float),
                ("pos_margin", C.c_float), ("neg_margin", C.c_float), ("balance", C.c_float),
                ("ent_l2_norm", C.c_int), ("rel_l2_norm", C.c_int), ("opt_kind", C.c_int),
                ("lr", C.c_float)]


def num_threads():
    return lib().oracle_num_threads()


def set_num_threads(n):
    """OpenMP threads of the parallel loops (bench.py's cpu_baseline: 1 thread and all host cores)"""
    lib().oracle_set_num_threads(C.c_int(int(n)))


def l2_normalize_rows(x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().oracle_l2_normalize_rows(_p(x), C.c_int(x.shape[0]), C.c_int(x.shape[1]), _p(out))
    return out


def sim_matrix(e1, e2, metric="inner"):
    e1, e2 = _f32(e1), _f32(e2)
    out = np.empty((e1.shape[0], e2.shape[0]), np.float32)
    lib().oracle_sim_matrix(_p(e1), C.c_int(e1.shape[0]), _p(e2), C.c_int(e2.shape[0]),
                            C.c_int(e1.shape[1]), C.c_int(METRIC[metric]), _p(out))
    return out


def topk_mean(s, k, axis=1):
    """mean of the k largest entries along `axis` of a 2-D fp32 matrix."""
    s = _f32(s)
    n1, n2 = s.shape
    if axis == 1:
        out = np.empty(n1, np.float32)
        lib().oracle_topk_mean(_p(s), C.c_int(n1), C.c_int(n2), C.c_size_t(n2), C.c_size_t(1),
                               C.c_int(k), _p(out))
    else:
        out = np.empty(n2, np.float32)
        lib().oracle_topk_mean(_p(s), C.c_int(n2), C.c_int(n1), C.c_size_t(1), C.c_size_t(n2),
                               C.c_int(k), _p(out))
    return out


def csls_apply(s, r, c):
    s = _f32(s).copy()
    r, c = _f32(r), _f32(c)
    lib().oracle_csls_apply(_p(s), C.c_int(s.shape[0]), C.c_int(s.shape[1]), _p(r), _p(c))
    return s


def rank_from_matrix(s):
    s = _f32(s)
    rank = np.empty(s.shape[0], np.int32)
    argmax = np.empty(s.shape[0], np.int32)
    lib().oracle_rank_from_matrix(_p(s), C.c_int(s.shape[0]), C.c_int(s.shape[1]), _p(rank), _p(argmax))
    return rank, argmax


def rank_eval(e1, e2, metric="inner", csls_r=None, csls_c=None):
    e1, e2 = _f32(e1), _f32(e2)
    rank = np.empty(e1.shape[0], np.int32)
    argmax = np.empty(e1.shape[0], np.int32)
    r = _f32(csls_r) if csls_r is not None else None
    c = _f32(csls_c) if csls_c is not None else None
    lib().oracle_rank_eval(_p(e1), C.c_int(e1.shape[0]), _p(e2), C.c_int(e2.shape[0]),
                           C.c_int(e1.shape[1]), C.c_int(METRIC[metric]), _p(r), _p(c),
                           _p(rank), _p(argmax))
    return rank, argmax


def topk_mean_rows(e1, e2, k, metric="inner"):
    e1, e2 = _f32(e1), _f32(e2)
    out = np.empty(e1.shape[0], np.float32)
    lib().oracle_topk_mean_rows(_p(e1), C.c_int(e1.shape[0]), _p(e2), C.c_int(e2.shape[0]),
                                C.c_int(e1.shape[1]), C.c_int(METRIC[metric]), C.c_int(k), _p(out))
    return out


def topk_inner(q, c, k):
    q, c = _f32(q), _f32(c)
    out = np.empty((q.shape[0], k), np.int32)
    lib().oracle_topk_inner(_p(q), C.c_int(q.shape[0]), _p(c), C.c_int(c.shape[0]),
                            C.c_int(q.shape[1]), C.c_int(k), _p(out))
    return out


def philox(ctr, key):
    ctr = np.ascontiguousarray(ctr, np.uint32)
    key = np.ascontiguousarray(key, np.uint32)
    out = np.empty(4, np.uint32)
    lib().oracle_philox4x32_10(_p(ctr), _p(key), _p(out))
    return out


def tripleset_capacity(n):
    cap = 16
    while cap < 2 * max(n, 1):
        cap *= 2
    return cap


def tripleset_build(triples):
    triples = _i32(triples).reshape(-1, 3)
    cap = tripleset_capacity(len(triples))
    table = np.empty(cap, np.uint64)
    lib().oracle_tripleset_build(_p(triples), C.c_int(len(triples)), _p(table), C.c_uint64(cap))
    return table


def tripleset_contains(table, h, r, t):
    return bool(lib().oracle_tripleset_contains(_p(table), C.c_uint64(len(table)), C.c_int(h),
                                                C.c_int(r), C.c_int(t)))


def sample_negatives(pos, k, table, entity_list, ent_pos=None, nbr=None, seed=0, step=0,
                     pos_offset=0, max_try=10):
    pos = _i32(pos).reshape(-1, 3)
    entity_list = _i32(entity_list)
    out = np.empty((len(pos) * k, 3), np.int32)
    nbr_k = 0
    if nbr is not None:
        nbr = _i32(nbr)
        nbr_k = nbr.shape[1]
        ent_pos = _i32(ent_pos)
    err = lib().oracle_sample_negatives(_p(pos), C.c_int(len(pos)), C.c_int(k), _p(table),
                                        C.c_uint64(len(table)), _p(entity_list),
                                        C.c_int(len(entity_list)), _p(ent_pos), _p(nbr),
                                        C.c_int(nbr_k), C.c_uint64(seed), C.c_uint32(step),
                                        C.c_uint32(pos_offset), C.c_int(max_try), _p(out))
    if err:
        raise ValueError("Sample larger than population or is negative")  # random.sample's error
    return out


def triple_step(ent, ent_acc, rel, rel_acc, pos, neg, *, loss="limited", loss_norm="L2",
                margin=0.0, pos_margin=0.01, neg_margin=2.0, balance=1.0, ent_l2_norm=True,
                rel_l2_norm=True, optimizer="Adagrad", lr=0.01):
    """In-place step on fp32 tables; returns the batch loss (float)."""
    assert ent.dtype == np.float32 and ent.flags.c_contiguous
    assert rel.dtype == np.float32 and rel.flags.c_contiguous
    pos = _i32(pos).reshape(-1, 3)
    neg = _i32(neg).reshape(-1, 3) if neg is not None and len(neg) else None
    cfg = StepCfg(LOSS[loss], 1 if loss_norm == "L1" else 0, margin, pos_margin, neg_margin,
                  balance, int(bool(ent_l2_norm)), int(bool(rel_l2_norm)), OPT[optimizer], lr)
    return lib().oracle_triple_step(_p(ent), _p(ent_acc), C.c_int(ent.shape[0]), _p(rel),
                                    _p(rel_acc), C.c_int(rel.shape[0]), C.c_int(ent.shape[1]),
                                    _p(pos), C.c_int(len(pos)), _p(neg),
                                    C.c_int(0 if neg is None else len(neg)), C.byref(cfg))


def triple_step_transh(ent, ent_acc, rel, rel_acc, nrm, nrm_acc, pos, neg, *, loss="limited", loss_norm="L2",
                       margin=0.0, pos_margin=0.01, neg_margin=2.0, balance=1.0, ent_l2_norm=True,
                       rel_l2_norm=True, optimizer="Adagrad", lr=0.01):
    """TransH step (bootea_transh.py:58-96) in place on fp32 tables; returns the batch loss."""
    for a in (ent, rel, nrm):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    pos = _i32(pos).reshape(-1, 3)
    neg = _i32(neg).reshape(-1, 3) if neg is not None and len(neg) else None
    cfg = StepCfg(LOSS[loss], 1 if loss_norm == "L1" else 0, margin, pos_margin, neg_margin,
                  balance, int(bool(ent_l2_norm)), int(bool(rel_l2_norm)), OPT[optimizer], lr)
    f = lib().oracle_triple_step_transh
    f.restype = C.c_double
    return f(_p(ent), _p(ent_acc), C.c_int(ent.shape[0]), _p(rel), _p(rel_acc), _p(nrm), _p(nrm_acc),
             C.c_int(rel.shape[0]), C.c_int(ent.shape[1]), _p(pos), C.c_int(len(pos)), _p(neg),
             C.c_int(0 if neg is None else len(neg)), C.byref(cfg))


def triple_step_transd(ent, ent_acc, rel, rel_acc, pos, neg, *, loss="margin-based", loss_norm="L2", margin=1.0,
                       pos_margin=0.01, neg_margin=2.0, balance=1.0, ent_l2_norm=True, rel_l2_norm=True,
                       optimizer="Adagrad", lr=0.01):
    """TransD step (models/trans/transd.py:16-57) in place on STACKED fp32 tables, ent = [ent_embeds ; ent_transfer]
    and rel = [rel_embeds ; rel_transfer]; returns the batch loss."""
    for a in (ent, rel):
        assert a.dtype == np.float32 and a.flags.c_contiguous and a.shape[0] % 2 == 0
    pos = _i32(pos).reshape(-1, 3)
    neg = _i32(neg).reshape(-1, 3) if neg is not None and len(neg) else None
    cfg = StepCfg(LOSS[loss], 1 if loss_norm == "L1" else 0, margin, pos_margin, neg_margin,
                  balance, int(bool(ent_l2_norm)), int(bool(rel_l2_norm)), OPT[optimizer], lr)
    f = lib().oracle_triple_step_transd
    f.restype = C.c_double
    return f(_p(ent), _p(ent_acc), C.c_int(ent.shape[0]), _p(rel), _p(rel_acc), C.c_int(rel.shape[0]),
             C.c_int(ent.shape[1]), _p(pos), C.c_int(len(pos)), _p(neg), C.c_int(0 if neg is None else len(neg)),
             C.byref(cfg))


def spmm_coo(rows, cols, vals, x, n_rows):
    rows, cols, vals, x = _i32(rows), _i32(cols), _f32(vals), _f32(x)
    y = np.empty((n_rows, x.shape[1]), np.float32)
    lib().oracle_spmm_coo(_p(rows), _p(cols), _p(vals), C.c_int64(len(rows)), _p(x),
                          C.c_int(n_rows), C.c_int(x.shape[1]), _p(y))
    return y
