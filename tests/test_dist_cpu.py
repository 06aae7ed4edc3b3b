"""world_size-2 gloo tests (CPU) of the multi-GPU layer: shard arithmetic, the row-sharded
evaluation / neighbour merges and the sum-exchange of the data-parallel step.  The per-shard
compute is the ORACLE here (test side only); on a GPU box the same code runs the HIP kernels."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openea_amd.models import dist as odist
from oracle import cport


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(fn, world, *args):
    port = _free_port()
    mp.spawn(_entry, args=(world, port, fn, args), nprocs=world, join=True)


def _entry(rank, world, port, fn, args):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world, *args)
    finally:
        dist.destroy_process_group()


def test_shard_arithmetic():
    for n in (0, 1, 7, 10500, 70000):
        for w in (1, 2, 3, 8):
            rs = [odist.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1 and sizes == odist.shard_sizes(n, w)
    # batch sharding: local splits tile the global split, offsets tile the batch
    for n, sp in ((5000, 2683), (11, 0), (11, 11), (9, 4)):
        for w in (1, 2, 4, 8):
            tot_split, cover = 0, 0
            for r in range(w):
                lo, hi, ls = odist.shard_batch(n, sp, r, w)
                assert 0 <= ls <= hi - lo and lo == cover
                cover = hi
                tot_split += ls
            assert cover == n and tot_split == sp


def _eval_worker(rank, world, seed):
    rng = np.random.RandomState(seed)
    n1, n2, d = 203, 301, 24
    e1 = rng.standard_normal((n1, d)).astype(np.float32)
    e2 = rng.standard_normal((n2, d)).astype(np.float32)
    e2[:n1] += e1

    def rank_fn(q, c, dim, gold_offset):
        # oracle on the query block: emulate the gold offset by evaluating against a rolled candidate view
        qn, cn = q.numpy(), c.numpy()
        s = cport.sim_matrix(qn, cn, "inner")
        idx = np.arange(len(qn)) + gold_offset
        g = s[np.arange(len(qn)), idx]
        cols = np.arange(cn.shape[0])[None, :]
        rk = ((s > g[:, None]) | ((s == g[:, None]) & (cols < idx[:, None]))).sum(1).astype(np.int32)
        return torch.from_numpy(rk), torch.from_numpy(s.argmax(1).astype(np.int32))

    hits, rsum, rr, am = odist.sharded_rank_metrics(torch.from_numpy(e1), torch.from_numpy(e2), d, [1, 5, 10, 50],
                                                    rank_fn)
    rk_ref, am_ref = cport.rank_eval(e1, e2, "inner")
    assert hits == [int((rk_ref < k).sum()) for k in (1, 5, 10, 50)]
    assert rsum == int((rk_ref.astype(np.int64) + 1).sum())
    assert abs(rr - float((1.0 / (rk_ref + 1.0)).sum())) < 1e-9
    assert np.array_equal(am.numpy(), am_ref)


def test_sharded_eval_matches_unsharded():
    _run(_eval_worker, 2, 7)


def _nbr_worker(rank, world, seed):
    rng = np.random.RandomState(seed)
    n, d, k = 157, 16, 9
    emb = rng.standard_normal((n, d)).astype(np.float32)
    ids = (np.arange(n) * 2 + 1).astype(np.int32)

    def topk_fn(q, c, dim, kk, id_map):
        return torch.from_numpy(id_map.numpy()[cport.topk_inner(q.numpy(), c.numpy(), kk)])

    out = odist.sharded_neighbours(torch.from_numpy(emb), d, torch.from_numpy(ids), k, topk_fn)
    assert np.array_equal(out.numpy(), ids[cport.topk_inner(emb, emb, k)])


def test_sharded_neighbours_match_unsharded():
    _run(_nbr_worker, 2, 3)


def _dp_step_worker(rank, world, seed):
    """data-parallel translational step = sum-exchange of per-rank gradient scratches, then the same
    update everywhere.  Emulated with the oracle: each rank's scratch is (table_after_SGD - table)
    for its batch slice at lr = -1 (SGD makes the update linear in the gradient), summed with
    all_reduce; must equal the single-process step on the whole batch."""
    rng = np.random.RandomState(seed)
    n_ent, n_rel, d, n_pos, k = 120, 7, 16, 64, 3
    ent = rng.standard_normal((n_ent, d)).astype(np.float32)
    rel = rng.standard_normal((n_rel, d)).astype(np.float32)
    pos = np.stack([rng.randint(0, n_ent, n_pos), rng.randint(0, n_rel, n_pos), rng.randint(0, n_ent, n_pos)], 1).astype(np.int32)
    neg = np.repeat(pos, k, 0)
    neg[:, 2] = rng.randint(0, n_ent, len(neg))
    kw = dict(loss="limited", loss_norm="L2", pos_margin=0.01, neg_margin=2.0, balance=0.2, ent_l2_norm=False,
              rel_l2_norm=False, optimizer="SGD", lr=-1.0)
    lo, hi, _ = odist.shard_batch(n_pos, 0, rank, world)
    e_loc, r_loc = ent.copy(), rel.copy()
    cport.triple_step(e_loc, None, r_loc, None, pos[lo:hi], neg[lo * k:hi * k], **kw)
    g = torch.from_numpy(np.concatenate([(e_loc - ent).ravel(), (r_loc - rel).ravel()]).astype(np.float64))
    odist.allreduce_sum_(g)
    e_ref, r_ref = ent.copy(), rel.copy()
    cport.triple_step(e_ref, None, r_ref, None, pos, neg, **kw)
    g_ref = np.concatenate([(e_ref - ent).ravel(), (r_ref - rel).ravel()])
    np.testing.assert_allclose(g.numpy(), g_ref, rtol=0, atol=5e-6)


def test_data_parallel_exchange_equals_big_batch():
    _run(_dp_step_worker, 2, 11)


def _gather_worker(rank, world):
    n = 11
    lo, hi = odist.shard_range(n, rank, world)
    local = torch.arange(lo, hi, dtype=torch.int32).view(-1, 1).repeat(1, 3)
    full = odist.allgather_rows(local, n)
    assert np.array_equal(full.numpy(), np.arange(n, dtype=np.int32)[:, None].repeat(3, 1))


def test_allgather_rows_uneven():
    _run(_gather_worker, 2)


def _spmm_worker(rank, world, seed):
    """row-sharded aggregate y = A . x (models/graph_ops.py:CsrOperand.apply): nnz-balanced row
    blocks, each rank computes its block (oracle SpMM here), one all-gather -> the unsharded product."""
    import scipy.sparse as sp
    from oracle import cport
    rng = np.random.RandomState(seed)
    n, d = 301, 12
    rows = np.minimum(rng.zipf(1.6, 4000) - 1, n - 1)                    # hubs at the low ids
    a = sp.coo_matrix((rng.rand(4000).astype(np.float32), (rows, rng.randint(0, n, 4000))), shape=(n, n)).tocsr()
    a.sum_duplicates(); a.sort_indices()
    x = rng.standard_normal((n, d)).astype(np.float32)
    bounds = odist.balanced_bounds(a.indptr, world)
    assert bounds[0] == 0 and bounds[-1] == n and all(b1 >= b0 for b0, b1 in zip(bounds, bounds[1:]))
    nnz_blocks = [int(a.indptr[bounds[r + 1]] - a.indptr[bounds[r]]) for r in range(world)]
    assert max(nnz_blocks) <= 0.75 * a.nnz                                # equal ROW counts would give rank 0 ~90 %
    lo, hi = bounds[rank], bounds[rank + 1]
    blk = a[lo:hi].tocoo()
    out = torch.zeros((n, d), dtype=torch.float32)
    out[lo:hi] = torch.from_numpy(cport.spmm_coo(blk.row, blk.col, blk.data, x, hi - lo))
    odist.allgather_blocks(out, bounds)
    coo = a.tocoo()
    assert np.array_equal(out.numpy(), cport.spmm_coo(coo.row, coo.col, coo.data, x, n))


def test_row_sharded_aggregate_equals_unsharded():
    _run(_spmm_worker, 2, 5)


class _FakeLib:
    """stand-in for libopenea_hip.so's communicator entry points (no GPU here): records the calls"""

    def __init__(self, rank, fail_unique_id=False, fail_init_on=()):
        self.rank, self.fail_unique_id, self.fail_init_on, self.calls = rank, fail_unique_id, fail_init_on, []

    def oea_comm_unique_id(self, uid):
        self.calls.append("unique_id")
        if self.fail_unique_id:
            return -1
        uid.raw = bytes(range(128))
        return 0

    def oea_comm_init(self, buf, rank, world, handle):
        self.calls.append("init")
        assert bytes(buf.raw) == bytes(range(128))
        if rank in self.fail_init_on:
            return -1
        handle._obj.value = 1000 + rank
        return 0

    def oea_comm_init_callbacks(self, rank, world, fn, user, handle):
        self.calls.append("init_callbacks")
        handle._obj.value = 2000 + rank
        return 0

    def oea_comm_set_alltoallv(self, h, fn):
        return 0

    def oea_comm_destroy(self, h):
        self.calls.append("destroy")
        return 0

    def oea_last_error(self):
        return b"fake failure"


def _comm_setup_worker(rank, world, mode):
    import openea_amd._lib as _lib
    from openea_amd import ops
    fake = _FakeLib(rank, fail_unique_id=(mode == "no_rccl"), fail_init_on=(1,) if mode == "init_fails_on_1" else ())
    ops.lib = lambda: fake
    _lib.load = lambda: fake
    odist.dist.get_backend = lambda group=None: "nccl"           # take the RCCL branch over the gloo group
    odist._collective_device = lambda backend: "cpu"
    c = odist.CAbiComm(None)
    if mode == "ok":
        assert not c.callbacks and c.handle.value == 1000 + rank and "init_callbacks" not in fake.calls
    else:
        # rank 0 could not make an id / rank 1 could not join: EVERY rank ends on the callbacks back end (no hang, no mismatch)
        assert c.callbacks and c.handle.value == 2000 + rank and fake.calls[-1] == "init_callbacks"
        if mode == "no_rccl":
            assert "init" not in fake.calls                    # oea_comm_init only ever runs with a valid id
        else:
            assert ("destroy" in fake.calls) == (rank == 0)      # the communicator made on rank 0 alone is dropped
    dist.barrier()
    c.close()
    c.close()
    assert fake.calls.count("destroy") == (2 if (mode == "init_fails_on_1" and rank == 0) else 1)


def test_cabi_comm_setup_agrees_on_the_back_end_when_rccl_fails_on_one_rank():
    """when rank 0's oea_comm_unique_id fails the other ranks used to wait in broadcast_object_list while
    rank 0 went on to the all-reduce -- mismatched collectives.  Now every rank runs broadcast (id or None) -> MIN all-reduce."""
    for mode in ("ok", "no_rccl", "init_fails_on_1"):
        _run(_comm_setup_worker, 2, mode)
