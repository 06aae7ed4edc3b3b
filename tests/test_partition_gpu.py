"""Entity-id partitioning of the translational step (owner = id mod G, include/openea_hip.h: oea_part_*): G processes on ONE
GPU (gloo rendezvous; the collectives are staged through the host, the kernels are the product's) run the same AlignE-style
steps on their batch shards; the tables must equal the single-process run on the whole batch within the north-star
tolerance, every replica must hold the same bits, and each rank keeps 1/G of the optimiser state."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ["OEA_ROOT"])
world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
group = None
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["OEA_PORT"], rank=rank, world_size=world)
    group = dist.group.WORLD
torch.cuda.set_device(0)
from openea_amd import ops
from openea_amd.models.trainer import EmbeddingTable, TripleTrainer
from openea_amd.models.dist import shard_range
rng = np.random.RandomState(5)
n_ent, n_rel, d, B, k = 1003, 19, 40, 1200, 3                  # n_ent not a multiple of 2 or 4: padding slots in the last stripe
ent_h = (rng.standard_normal((n_ent, d)) / np.sqrt(d)).astype(np.float32)
rel_h = (rng.standard_normal((n_rel, d)) / np.sqrt(d)).astype(np.float32)
opt = os.environ["OEA_OPT"]
cfg = ops.make_step_cfg(loss="limited", loss_norm="L2", pos_margin=0.01, neg_margin=2.0, balance=0.2, optimizer=opt, lr=0.02,
                        neg_group_k=k)
ent, rel = EmbeddingTable(ent_h, True, "e"), EmbeddingTable(rel_h, True, "r")
tr = TripleTrainer(ent, rel, cfg, opt, dist_group=group)
assert (tr.part is not None) == (world > 1)
for step in range(4):
    pos = np.stack([rng.randint(0, n_ent, B), rng.randint(0, n_rel, B), rng.randint(0, n_ent, B)], 1).astype(np.int32)
    neg = np.repeat(pos, k, 0)
    neg[:, 2] = rng.randint(0, n_ent, len(neg))
    lo, hi = shard_range(B, rank, world)
    if step == 3 and world > 1:                                  # the last rank gets an EMPTY shard and still joins every collective
        lo, hi = shard_range(B, rank, world - 1) if rank < world - 1 else (B, B)
    tr.step(ops.to_ids(pos[lo:hi]), ops.to_ids(neg[lo * k: hi * k]))
loss = tr.pop_loss()
state_rows = tr.part["acc_own"].shape[0] if (tr.part is not None and tr.part["acc_own"] is not None) else (0 if tr.ent_acc is None else tr.ent_acc.shape[0])
np.savez(os.environ["OEA_OUT"] + "/part_w%d_r%d.npz" % (world, rank), ent=ent.raw(), rel=rel.raw(), loss=loss, state_rows=state_rows,
         xbytes=tr.exchange_bytes_per_step(), scratch_clean=int(not bool((tr.ws[: tr.ws.numel() - 8 * 4096] != 0).any().item())))
if world > 1:
    dist.barrier()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(tmp_path, world, opt):
    env = dict(os.environ, OEA_ROOT=ROOT, OEA_OUT=str(tmp_path), OEA_PORT=str(_free_port()), WORLD_SIZE=str(world), OEA_OPT=opt)
    procs = [subprocess.Popen([sys.executable, "-c", WORKER], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(str(tmp_path), "part_w%d_r%d.npz" % (world, r))) for r in range(world)]


@pytest.mark.parametrize("opt", ["Adagrad", "SGD"])
def test_partitioned_step_equals_single_process(tmp_path, opt, capsys):
    from _tol import assert_rows_close
    single = _launch(tmp_path, 1, opt)[0]
    for world in (2, 4):
        ranks = _launch(tmp_path, world, opt)
        for r in ranks[1:]:
            assert np.array_equal(r["ent"], ranks[0]["ent"]) and np.array_equal(r["rel"], ranks[0]["rel"])    # replicas: same bits
        with capsys.disabled():
            assert_rows_close(ranks[0]["ent"], single["ent"], "%s, %d ranks vs 1 process, entity table after 4 steps" % (opt, world))
            assert_rows_close(ranks[0]["rel"], single["rel"], "%s, %d ranks vs 1 process, relation table" % (opt, world))
        total = sum(float(r["loss"]) for r in ranks) if False else float(ranks[0]["loss"])
        assert abs(total - float(single["loss"])) <= 1e-5 * abs(float(single["loss"]))     # pop_loss all-reduces the per-rank losses
        assert all(int(r["scratch_clean"]) == 1 for r in ranks)
        if opt == "Adagrad":
            assert all(int(r["state_rows"]) == -(-1003 // world) for r in ranks)           # 1/G of the optimiser state per rank
        with capsys.disabled():
            print("exchange bytes per step and rank at world %d: %d" % (world, int(ranks[0]["xbytes"])))


COMM_WORKER = r'''
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.environ["OEA_ROOT"])
from openea_amd import ops
from openea_amd._lib import check
lib = ops.lib()
torch.cuda.set_device(0)
uid = (C.c_char * 128)()
check(lib.oea_comm_unique_id(uid))
comm = C.c_void_p()
check(lib.oea_comm_init(uid, 0, 1, C.byref(comm)))
assert lib.oea_comm_rank(comm) == 0 and lib.oea_comm_size(comm) == 1
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.arange(12, dtype=torch.float32, device="cuda").view(3, 4).contiguous()
y = torch.empty_like(x)
check(lib.oea_allgather_rows(comm, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), 3, 4, st))
z = torch.empty(12, dtype=torch.float32, device="cuda")
check(lib.oea_comm_reduce_scatter_f32(comm, C.c_void_p(x.data_ptr()), C.c_void_p(z.data_ptr()), 12, st))
i = torch.tensor([5, 7], dtype=torch.int64, device="cuda")
check(lib.oea_allreduce_i64(comm, C.c_void_p(i.data_ptr()), 2, st))
d = torch.tensor([0.25], dtype=torch.float64, device="cuda")
check(lib.oea_allreduce_f64(comm, C.c_void_p(d.data_ptr()), 1, st))
f = torch.tensor([1.5, 2.5], dtype=torch.float32, device="cuda")
check(lib.oea_allreduce_f32(comm, C.c_void_p(f.data_ptr()), 2, st))
torch.cuda.synchronize()
assert torch.equal(x, y) and torch.equal(x.view(-1), z) and i.tolist() == [5, 7] and d.item() == 0.25 and f.tolist() == [1.5, 2.5]
check(lib.oea_comm_destroy(comm))
print("COMM_OK")
'''


def test_comm_group_of_the_c_abi_single_rank():
    """oea_comm_* (RCCL resolved with dlopen): communicator of one rank on the box's GPU -- every collective of the group
    runs and is the identity.  (More ranks need more GPUs: RCCL refuses two ranks on one device; the N > 1 exchange of the
    step is covered through torch.distributed above and by the driver's multi-GPU bench.)"""
    p = subprocess.run([sys.executable, "-c", COMM_WORKER], env=dict(os.environ, OEA_ROOT=ROOT), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=180)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "COMM_OK" in out, out[-3000:]


EPOCH_COMM_WORKER = r'''
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ["OEA_ROOT"])
from openea_amd import ops
from openea_amd.models.trainer import EmbeddingTable, TripleTrainer
torch.cuda.set_device(0)
rng = np.random.RandomState(7)
n_ent, n_rel, d, B, k, steps = 1003, 19, 40, 900, 3, 5
ent_h = (rng.standard_normal((n_ent, d)) / np.sqrt(d)).astype(np.float32)
rel_h = (rng.standard_normal((n_rel, d)) / np.sqrt(d)).astype(np.float32)
pos = np.stack([rng.randint(0, n_ent, steps * B), rng.randint(0, n_rel, steps * B), rng.randint(0, n_ent, steps * B)], 1).astype(np.int32)
neg = np.repeat(pos, k, 0)
neg[:, 2] = rng.randint(0, n_ent, len(neg))
offsets = (np.arange(steps + 1) * B).astype(np.int64)
offsets[3] -= 37                                                  # ragged batches
splits = np.full(steps, B // 2, np.int64)
dev = torch.device("cuda:0")
pos_d, neg_d = ops.to_ids(pos), ops.to_ids(neg)
off_d, spl_d = torch.from_numpy(offsets).to(dev), torch.from_numpy(splits).to(dev)
res = {}
for mode in ("plain", "comm"):
    for opt in ("Adagrad", "SGD"):
        cfg = ops.make_step_cfg(loss="limited", loss_norm="L2", pos_margin=0.01, neg_margin=2.0, balance=0.2, optimizer=opt, lr=0.02,
                                neg_group_k=k)
        ent, rel = EmbeddingTable(ent_h, True, "e"), EmbeddingTable(rel_h, True, "r")
        tr = TripleTrainer(ent, rel, cfg, opt)
        tr.count_steps(steps)
        if mode == "plain":
            ops.triple_epoch(ent.var, tr.ent_acc, rel.var, tr.rel_acc, d, pos_d, offsets, splits, k, None, None, 1, 0, neg_d, None, tr.cfg,
                             tr.ws, tr.loss, off_d, spl_d)
            acc = tr.ent_acc
        else:
            comm = ops.comm_single_or_none()
            bufs = ops.part_buffers(n_ent, n_rel, ent.ld, 1, dev, adagrad=opt == "Adagrad")
            ops.triple_epoch_comm(comm, ent.var, bufs["acc_own"], rel.var, tr.rel_acc, d, pos_d, offsets, splits, k, None, None, 1, 0, neg_d,
                                  None, tr.cfg, tr.ws, tr.loss, off_d, spl_d, bufs)
            torch.cuda.synchronize()
            ops.check(ops.lib().oea_comm_destroy(comm))
            acc = bufs["acc_own"]
        torch.cuda.synchronize()
        res[mode + opt] = (ent.raw(), rel.raw(), None if acc is None else acc.cpu().numpy(), float(tr.loss.item()))
for opt in ("Adagrad", "SGD"):
    a, b = res["plain" + opt], res["comm" + opt]
    for x, y, name in ((a[0], b[0], "entity table"), (a[1], b[1], "relation table"), (a[2], b[2], "accumulator")):
        if x is None:
            continue
        err = float(np.abs(x - y).max() / max(np.abs(x).max(), 1e-30))
        assert err <= 2e-6, (opt, name, err)
    assert abs(a[3] - b[3]) <= 1e-6 * abs(a[3]), (opt, a[3], b[3])
    assert float(np.abs(a[0] - ent_h).max()) > 1e-3               # the epoch did move the tables
print("EPOCH_COMM_OK")
'''


def test_partitioned_epoch_from_one_c_call_single_rank():
    """oea_triple_epoch_range_comm: the steps of an epoch under the entity-id partition (GRAD -> pack -> RCCL reduce-scatter +
    all-reduce -> owned apply -> all-gather -> unpack) enqueued by ONE call over the C ABI's own communicator.  With the one
    rank this box can give RCCL every collective is the identity and the result must be the single-GPU epoch call's (tables,
    Adagrad state, loss; ragged batches); more ranks need more GPUs (the step-by-step protocol it strings together is covered
    with 2 and 4 ranks through torch.distributed above)."""
    p = subprocess.run([sys.executable, "-c", EPOCH_COMM_WORKER], env=dict(os.environ, OEA_ROOT=ROOT), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "EPOCH_COMM_OK" in out, out[-3000:]


TRAINER_COMM_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["OEA_ROOT"])
torch.cuda.set_device(0)
from openea_amd import ops
from openea_amd.models.trainer import EmbeddingTable, RelationTripleEpochs, TripleTrainer
from openea_amd.modules.load.synth import make_kgs
kgs = make_kgs("small", mode="swapping", seed=0)
rng = np.random.RandomState(2)
d, k = 32, 4
ent_h = (rng.standard_normal((kgs.entities_num, d)) / np.sqrt(d)).astype(np.float32)
rel_h = (rng.standard_normal((kgs.relations_num, d)) / np.sqrt(d)).astype(np.float32)
out = {}
for mode in ("plain", "c_epoch"):
    group = None
    if mode == "c_epoch":
        os.environ["OEA_DP_C_EPOCH"] = "1"
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["OEA_PORT"], rank=0, world_size=1)
        group = dist.group.WORLD
    cfg = ops.make_step_cfg(loss="limited", loss_norm="L2", pos_margin=0.01, neg_margin=2.0, balance=0.2, optimizer="Adagrad", lr=0.02,
                            neg_group_k=k)
    ent, rel = EmbeddingTable(ent_h, True, "e"), EmbeddingTable(rel_h, True, "r")
    tr = TripleTrainer(ent, rel, cfg, "Adagrad", dist_group=group)
    assert (tr.part is not None and tr.comm is not None) == (mode == "c_epoch")
    ep = RelationTripleEpochs(kgs, 700, k, seed=3)
    n = sum(ep.run_epoch(tr) for _ in range(3))
    torch.cuda.synchronize()
    out[mode] = (ent.raw(), rel.raw(), tr.pop_loss(), n)
a, b = out["plain"], out["c_epoch"]
assert a[3] == b[3] and a[3] > 0
for x, y in ((a[0], b[0]), (a[1], b[1])):
    assert float(np.abs(x - y).max() / np.abs(x).max()) <= 1e-5, float(np.abs(x - y).max())
assert abs(a[2] - b[2]) <= 1e-5 * abs(a[2])
assert float(np.abs(a[0] - ent_h).max()) > 1e-3
print("TRAINER_COMM_OK")
'''


def test_trainer_runs_the_partitioned_epoch_from_one_c_call():
    """OEA_DP_C_EPOCH=1: TripleTrainer builds the C ABI's communicator over its process group and RelationTripleEpochs hands the
    epoch to oea_triple_epoch_range_comm (sampler ahead on the side stream, device shuffle, ranges, loss read-back as in the
    single-GPU path).  One rank (what RCCL can get on this box): three epochs equal the plain single-GPU trainer's."""
    p = subprocess.run([sys.executable, "-c", TRAINER_COMM_WORKER], env=dict(os.environ, OEA_ROOT=ROOT, OEA_PORT=str(_free_port())),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "TRAINER_COMM_OK" in out, out[-3000:]


EPOCH_RANKS_WORKER = r'''
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ["OEA_ROOT"])
world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
group = None
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["OEA_PORT"], rank=rank, world_size=world)
    group = dist.group.WORLD
torch.cuda.set_device(0)
from openea_amd import ops
from openea_amd.models.trainer import EmbeddingTable, RelationTripleEpochs, TripleTrainer, refresh_neighbours
from openea_amd.modules.load.synth import make_kgs
kgs = make_kgs("small", mode="swapping", seed=0)
rng = np.random.RandomState(2)
d, k = 36, 4
ent_h = (rng.standard_normal((kgs.entities_num, d)) / np.sqrt(d)).astype(np.float32)
rel_h = (rng.standard_normal((kgs.relations_num, d)) / np.sqrt(d)).astype(np.float32)
opt = os.environ["OEA_OPT"]
cfg = ops.make_step_cfg(loss="limited", loss_norm="L2", pos_margin=0.01, neg_margin=2.0, balance=0.2, optimizer=opt, lr=0.02,
                        neg_group_k=k)
ent, rel = EmbeddingTable(ent_h, True, "e"), EmbeddingTable(rel_h, True, "r")
exchange = os.environ.get("OEA_EXCHANGE", "step")
tr = TripleTrainer(ent, rel, cfg, opt, dist_group=group, exchange=exchange if world > 1 else None)
if world > 1:
    assert tr.part is not None and tr.comm is not None and tr.comm.callbacks      # the one-call epoch is the default path
    assert tr.exchange == exchange
    tr.comm.profile_begin()
ep = RelationTripleEpochs(kgs, 700, k, seed=3, rank=rank, world=world)
n = 0
n_epochs, refresh = int(os.environ["OEA_EPOCHS"]), os.environ["OEA_REFRESH"] == "1"
chunk = int(os.environ.get("OEA_CHUNK", "0"))            # > 0: the epoch in ranges of that many steps (partial ranges of the C call)
S = len(ep.batches.splits)
for e in range(n_epochs):
    if e == 2 and refresh:            # a truncated-sampling refresh in between, as BootEA / AlignE do
        nbr1 = refresh_neighbours(ent, kgs.kg1.entities_list, 40)
        nbr2 = refresh_neighbours(ent, kgs.kg2.entities_list, 40)
        ep.set_neighbours(nbr1, nbr2)
    if chunk:
        done = 0
        while done < S:
            c = min(chunk, S - done)
            n += ep.run_steps(tr, c)
            done += c
    else:
        n += ep.run_epoch(tr)
ep.check()
torch.cuda.synchronize()
phases = None
if world > 1:
    phases, steps = tr.comm.profile_end()
    assert steps == n_epochs * len(ep.batches.splits) and all(v >= 0 for v in phases.values()) and phases["grad"] > 0, (phases, steps)
loss = tr.pop_loss()
np.savez(os.environ["OEA_OUT"] + "/ep_w%d_r%d.npz" % (world, rank), ent=ent.raw(), rel=rel.raw(), loss=loss, n=n,
         det=int(ops.deterministic()), halo=np.asarray(tr.halo_stats, np.int64), xbytes=tr.exchange_bytes_per_step(),
         rows=kgs.entities_num, ld=ent.ld, steps=n_epochs * S)
if world > 1:
    dist.barrier()
'''


def _launch_epochs(tmp_path, world, opt, det, exchange="step", chunk=0):
    # fixed point: five epochs with a neighbour refresh in between, compared BIT FOR BIT.  fp32 atomics: the single-GPU job is
    # not reproducible run to run (the order of the atomics), a flipped hinge or a neighbour set that differs by one entity
    # after the refresh grows into 5e-3 per row within three epochs -- two epochs without the refresh are held to 1e-4
    env = dict(os.environ, OEA_ROOT=ROOT, OEA_OUT=str(tmp_path), OEA_PORT=str(_free_port()), WORLD_SIZE=str(world), OEA_OPT=opt,
               OEA_STEP_DETERMINISTIC="1" if det else "0", OEA_EPOCHS="5" if det else "2", OEA_REFRESH="1" if det else "0",
               OEA_EXCHANGE=exchange, OEA_CHUNK=str(chunk))
    env.pop("OEA_DP_C_EPOCH", None)
    procs = [subprocess.Popen([sys.executable, "-c", EPOCH_RANKS_WORKER], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    # read NOW: a later launch with the same world size writes the same file names (np.load is lazy)
    return [dict(np.load(os.path.join(str(tmp_path), "ep_w%d_r%d.npz" % (world, r)))) for r in range(world)]


@pytest.mark.parametrize("det", [False, True], ids=["fp32_atomics", "fixed_point"])
@pytest.mark.parametrize("opt", ["Adagrad", "SGD"])
def test_one_call_partitioned_epochs_with_2_and_4_ranks_equal_the_single_process_job(tmp_path, opt, det, capsys):
    """The DEFAULT data-parallel path of the translational models: TripleTrainer + RelationTripleEpochs hand every epoch to
    oea_triple_epoch_range_comm (GRAD -> pack -> reduce-scatter + relation all-reduce -> owned apply -> all-gather -> unpack,
    one C call per epoch).  2 and 4 ranks share this box's GPU, so the communicator runs over host callbacks on the gloo group
    (RCCL needs a GPU per rank; a multi-GPU node takes the RCCL branch of the same entry points).  Replicas hold the same
    bits; the tables equal the single-process job's within the north-star tolerance (fp32 atomics: two epochs) -- and BIT FOR
    BIT after five epochs with a neighbour refresh in between in the fixed-point build (libopenea_hip_det.so: integer sums
    have no order)."""
    from _tol import assert_rows_close
    single = _launch_epochs(tmp_path, 1, opt, det)[0]
    assert int(single["det"]) == int(det)
    again = _launch_epochs(tmp_path, 1, opt, det)[0]
    if det:       # the single-GPU job itself: two runs, the same bits
        assert np.array_equal(again["ent"], single["ent"]) and np.array_equal(again["rel"], single["rel"])
        assert float(again["loss"]) == float(single["loss"])
    for world in (2, 4):
        ranks = _launch_epochs(tmp_path, world, opt, det)
        for r in ranks[1:]:
            assert np.array_equal(r["ent"], ranks[0]["ent"]) and np.array_equal(r["rel"], ranks[0]["rel"])
        assert sum(int(r["n"]) for r in ranks) == int(single["n"])
        with capsys.disabled():
            assert_rows_close(ranks[0]["ent"], single["ent"], "%s%s, %d ranks (one C call per epoch) vs 1 process, entity table"
                              % (opt, " fixed point" if det else "", world))
            assert_rows_close(ranks[0]["rel"], single["rel"], "relation table")
        assert abs(float(ranks[0]["loss"]) - float(single["loss"])) <= 1e-5 * abs(float(single["loss"]))
        if det:
            assert np.array_equal(ranks[0]["ent"], single["ent"]) and np.array_equal(ranks[0]["rel"], single["rel"]), \
                "fixed-point build: the %d-rank job must equal the single-process job bit for bit" % world


@pytest.mark.parametrize("det,opt,chunk", [(True, "Adagrad", 0), (True, "SGD", 3), (False, "Adagrad", 0)],
                         ids=["fixed_point-Adagrad", "fixed_point-SGD-ranges_of_3", "fp32_atomics-Adagrad"])
def test_boundary_row_exchange_with_2_and_4_ranks_equals_the_single_process_job(tmp_path, det, opt, chunk, capsys):
    """dp_exchange = 'halo' (oea_triple_epoch_range_halo): the partitioned step moving only the rows a step's batch refers to -- row
    lists derived on every rank from the epoch's positives and negatives, all-to-all of the gradient rows to their owners,
    all-to-all of the rows the next step's readers refer to, one dense all-gather at the end of every call.  2 and 4 ranks on this
    box's GPU (host callbacks over gloo, point-to-point messages): replicas hold the same bits, the tables are the single-process
    job's BIT FOR BIT in the fixed-point build (five epochs, a neighbour refresh in between, also when an epoch is run in ranges of
    3 steps: pulls between ranges, presampled negatives) and within the north-star tolerance with fp32 atomics; and the bytes a
    rank moves per step are a fraction of the dense protocol's."""
    from _tol import assert_rows_close
    single = _launch_epochs(tmp_path, 1, opt, det)[0]
    for world in (2, 4):
        ranks = _launch_epochs(tmp_path, world, opt, det, exchange="halo", chunk=chunk)
        for r in ranks[1:]:
            assert np.array_equal(r["ent"], ranks[0]["ent"]) and np.array_equal(r["rel"], ranks[0]["rel"])
        assert sum(int(r["n"]) for r in ranks) == int(single["n"])
        with capsys.disabled():
            assert_rows_close(ranks[0]["ent"], single["ent"], "halo exchange, %s%s, %d ranks vs 1 process, entity table"
                              % (opt, " fixed point" if det else "", world))
            assert_rows_close(ranks[0]["rel"], single["rel"], "relation table")
        assert abs(float(ranks[0]["loss"]) - float(single["loss"])) <= 1e-5 * abs(float(single["loss"]))
        if det:
            assert np.array_equal(ranks[0]["ent"], single["ent"]) and np.array_equal(ranks[0]["rel"], single["rel"]), \
                "fixed-point build: the %d-rank halo job must equal the single-process job bit for bit" % world
        for r in ranks:
            pushed, pulled, max_rows, steps = (int(x) for x in r["halo"])
            assert steps == int(r["steps"]) and pushed > 0 and pulled > 0 and 0 < max_rows <= int(r["rows"])
        rows, ld = int(ranks[0]["rows"]), int(ranks[0]["ld"])
        dense = (world - 1) / world * rows * (2 * ld + 1) * (8 if det else 4)          # reduce-scatter + all-gather of every owned row
        with capsys.disabled():
            print("halo exchange at world %d: %.0f bytes per step and rank (dense protocol: %.0f)" % (world, float(ranks[0]["xbytes"]), dense))
        assert float(ranks[0]["xbytes"]) < dense


@pytest.mark.parametrize("world,k", [(2, 4), (4, 10), (8, 10), (3, 0)])
def test_halo_plan_equals_the_oracle_restatement(world, k):
    """oea_halo_plan (halo_mark_kernel + halo_compact_kernel: what sizes every message of the boundary-row exchange and names the
    rows) against oracle/np_oracle.py:halo_plan on ragged batches (the last one short, one EMPTY), Zipf-headed positives, entries of
    the negatives that are not corruptions of their positive, a world size that does not divide the table, k = 0: the counts
    [steps, world, world] and every row list (ascending local indices per owner) are equal."""
    torch = pytest.importorskip("torch")
    from openea_amd import ops
    from oracle import np_oracle as orc
    ops.lib()
    rng = np.random.RandomState(100 * world + k)
    n_ent, n_rel = 7001, 37
    offsets = np.array([0, 1500, 1500, 3200, 5000, 5303], np.int64)            # step 1 is empty, step 4 short
    n = int(offsets[-1])
    w = 1.0 / np.arange(1, n_ent + 1) ** 0.9
    pos = np.stack([rng.choice(n_ent, n, p=w / w.sum()), rng.randint(0, n_rel, n), rng.randint(0, n_ent, n)], 1).astype(np.int32)
    neg = None
    if k:
        neg = np.repeat(pos, k, 0)
        flip = rng.rand(len(neg)) < 0.5
        neg[flip, 0] = rng.randint(0, n_ent, int(flip.sum()))
        neg[~flip, 2] = rng.randint(0, n_ent, int((~flip).sum()))
        neg[7] = (n_ent - 1, 3, n_ent - 2)                                       # not a corruption of its positive; the table's last rows
    pos_d = ops.to_ids(pos)
    neg_d = ops.to_ids(neg) if k else None
    counts, lists = ops.halo_plan(pos_d, neg_d, k, offsets, n_ent, world)
    ref_counts, ref_lists = orc.halo_plan(pos, neg, k, offsets, n_ent, world)
    assert np.array_equal(counts.astype(np.int64), ref_counts)
    for s in range(len(offsets) - 1):
        for r in range(world):
            at = 0
            for o in range(world):
                c = int(counts[s, r, o])
                assert np.array_equal(lists[s, r, at:at + c].astype(np.int64), ref_lists[(s, r, o)]), (s, r, o)
                at += c
    # a sub-range of the steps plans the same lists
    c2, l2 = ops.halo_plan(pos_d, neg_d, k, offsets, n_ent, world, step_range=(2, 5))
    assert np.array_equal(c2, counts[2:5])
    for s in range(3):
        for r in range(world):
            m = int(c2[s, r].sum())
            assert np.array_equal(l2[s, r, :m], lists[2 + s, r, :m])
