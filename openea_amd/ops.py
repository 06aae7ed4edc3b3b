"""Thin typed wrappers over the C ABI (include/openea_hip.h).

PyTorch is plumbing here: device allocations (``torch.empty(..., device='cuda')``), the
current HIP stream and ``torch.distributed``.  All arithmetic happens inside
libopenea_hip.so; nothing in this module computes on the CPU.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import LOSS_KIND, METRIC, OPT_KIND, OpenEAHipError, RotateCfg, StepCfg, check


def lib():
    return _lib.load(require_device=True)


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pad4(d):
    return (int(d) + 3) // 4 * 4


def device(index=None):
    lib()
    return torch.device('cuda', torch.cuda.current_device() if index is None else index)


# -------------------------------------------------------------------------------------------
# tables
# -------------------------------------------------------------------------------------------


def to_table(array, ld=None, dev=None):
    """host [n, d] (numpy / list) -> device fp32 [n, ld] zero-padded (ld % 4 == 0)."""
    a = np.ascontiguousarray(np.asarray(array, dtype=np.float32))
    assert a.ndim == 2
    n, d = a.shape
    ld = pad4(d) if ld is None else ld
    t = torch.zeros((n, ld), dtype=torch.float32, device=dev or device())
    if n:
        t[:, :d].copy_(torch.from_numpy(a), non_blocking=False)
    return t


def to_vec(array, dev=None):
    """host 1-D float array -> device fp32 vector."""
    a = np.ascontiguousarray(np.asarray(array, dtype=np.float32)).reshape(-1)
    return torch.from_numpy(a).to(dev or device())


def to_ids(array, dev=None):
    a = np.ascontiguousarray(np.asarray(array, dtype=np.int32))
    return torch.from_numpy(a).to(dev or device())


def gather_rows(table, dim, ids, normalize=False, out_ld=None):
    """tf.nn.embedding_lookup(l2_normalize?(table), ids) -> device [n, out_ld]."""
    n = ids.numel()
    out_ld = pad4(dim) if out_ld is None else out_ld
    out = torch.empty((n, out_ld), dtype=torch.float32, device=table.device)
    check(lib().oea_gather_rows(_p(table), dim, table.shape[1], _p(ids), n, int(bool(normalize)),
                                _p(out), out_ld, _stream()))
    return out


def normalize_rows_(table, dim, sklearn=True):
    check(lib().oea_normalize_rows(_p(table), table.shape[0], dim, table.shape[1], int(bool(sklearn)), _stream()))
    return table


# -------------------------------------------------------------------------------------------
# translational step
# -------------------------------------------------------------------------------------------


SCORE_TRANSE, SCORE_TRANSH, SCORE_TRANSD = 0, 1, 2


def make_step_cfg(loss='limited', loss_norm='L2', margin=0.0, pos_margin=0.0, neg_margin=0.0,
                  balance=1.0, ent_l2_norm=True, rel_l2_norm=True, optimizer='Adagrad', lr=0.01,
                  neg_group_k=0, normal=None, normal_acc=None, transfer_bases=None):
    """neg_group_k = k when the negatives are laid out as the device sampler writes them
    (neg[p*k:(p+1)*k] corrupt pos p); 0 for arbitrary lists.
    normal (+ normal_acc for Adagrad): device [n_rel, ld] normal_vector table -> TransH scoring.
    transfer_bases = (n_ent, n_rel): TransD scoring on stacked tables (rows [n, 2n) hold the transfer vectors)."""
    assert normal is None or transfer_bases is None
    score = SCORE_TRANSH if normal is not None else (SCORE_TRANSD if transfer_bases is not None else SCORE_TRANSE)
    eb, rb = transfer_bases if transfer_bases is not None else (0, 0)
    # tf.train defaults: AdamOptimizer(beta1=0.9, beta2=0.999, epsilon=1e-8), AdadeltaOptimizer(rho=0.95, epsilon=1e-8)
    b1 = 0.95 if optimizer == 'Adadelta' else 0.9
    cfg = StepCfg(LOSS_KIND[loss], 1 if loss_norm == 'L1' else 0, float(margin), float(pos_margin),
                  float(neg_margin), float(balance), int(bool(ent_l2_norm)), int(bool(rel_l2_norm)),
                  OPT_KIND[optimizer], float(lr), int(neg_group_k), score,
                  None if normal is None else normal.data_ptr(), None if normal_acc is None else normal_acc.data_ptr(),
                  int(eb), int(rb), b1, 0.999, 1e-8, 0)
    cfg._keep = (normal, normal_acc)
    return cfg


def scratch_dtype():
    """element type of the step's gradient scratch, its touched flags and the partition's exchange buffers: fp32, or int64
    fixed point in the deterministic build (OEA_STEP_DETERMINISTIC=1 -> libopenea_hip_det.so, csrc/common.h)"""
    return torch.int64 if lib().oea_step_scratch_elem_bytes() == 8 else torch.float32


def deterministic():
    return lib().oea_step_scratch_elem_bytes() == 8


def step_workspace(n_ent, n_rel, ld, dev=None):
    nbytes = lib().oea_step_workspace_bytes(n_ent, n_rel, ld)
    return torch.zeros(nbytes, dtype=torch.uint8, device=dev or device())


PHASE_BOTH, PHASE_GRAD, PHASE_APPLY = 0, 1, 2


def triple_step(ent, ent_acc, rel, rel_acc, dim, pos, neg, cfg, workspace, loss_accum, phase=PHASE_BOTH):
    """One optimiser step in place; the batch loss is added to `loss_accum` (device f64[1]).
    phase: PHASE_GRAD / PHASE_APPLY split the step at its exchange point (data parallelism)."""
    n_neg = 0 if neg is None else neg.shape[0]
    check(lib().oea_triple_step_phase(_p(ent), _p(ent_acc), ent.shape[0], _p(rel), _p(rel_acc), rel.shape[0],
                                      dim, ent.shape[1], _p(pos), pos.shape[0], _p(neg), n_neg,
                                      C.byref(cfg), _p(workspace), _p(loss_accum), int(phase), _stream()))


# ---- RotatE (fp64) --------------------------------------------------------------------------------------------
def to_table64(array, dev=None):
    """host [n, d] -> device fp64 [n, ld] zero-padded (ld % 4 == 0)."""
    a = np.ascontiguousarray(np.asarray(array, dtype=np.float64))
    n, d = a.shape
    t = torch.zeros((n, pad4(d)), dtype=torch.float64, device=dev or device())
    if n:
        t[:, :d].copy_(torch.from_numpy(a))
    return t


def make_rotate_cfg(gamma, dim, ent_l2_norm=True, rel_l2_norm=False, optimizer='Adam', lr=0.01, beta1=0.9, beta2=0.999,
                    eps=1e-8, epsilon=2.0):
    """bootea_rotate.py:29-33,90: embedding_range = (gamma + epsilon) / dim, phase = rel / (embedding_range / pi)."""
    rng = (float(gamma) + float(epsilon)) / dim
    return RotateCfg(float(gamma), 3.14159265358979323846 / rng, float(lr), float(beta1), float(beta2), float(eps), 0,
                     int(bool(ent_l2_norm)), int(bool(rel_l2_norm)), OPT_KIND[optimizer], 0)


def rotate_workspace(n_ent, n_rel, ld, dev=None):
    return torch.zeros(lib().oea_rotate_workspace_bytes(n_ent, n_rel, ld), dtype=torch.uint8, device=dev or device())


def rotate_exchange_view(workspace, n_ent, n_rel, ld):
    """fp64 view of the workspace prefix that data-parallel ranks sum between PHASE_GRAD and PHASE_APPLY."""
    n = lib().oea_rotate_exchange_doubles(n_ent, n_rel, ld)
    return workspace[: n * 8].view(torch.float64)


def rotate_state(table, optimizer):
    """optimiser state of one fp64 table: Adagrad accumulator (0.1), Adam [m ; v] (zeros), None for SGD."""
    if optimizer == 'Adagrad':
        return torch.full_like(table, 0.1)
    if optimizer == 'Adam':
        return torch.zeros((2,) + tuple(table.shape), dtype=torch.float64, device=table.device)
    return None


def rotate_step(ent, ent_state, rel, rel_state, dim, pos, neg, neg_group_k, cfg, workspace, loss_accum, phase=PHASE_BOTH):
    """One RotatE optimiser step in place (ent: fp64 [2E, ld], re rows then im rows).  cfg.t must hold the 1-based
    step count of this optimiser when it is Adam."""
    n_neg = 0 if neg is None else neg.shape[0]
    check(lib().oea_rotate_step(_p(ent), _p(ent_state), ent.shape[0] // 2, _p(rel), _p(rel_state), rel.shape[0], dim,
                                ent.shape[1], _p(pos), pos.shape[0], _p(neg), n_neg, int(neg_group_k), C.byref(cfg),
                                _p(workspace), _p(loss_accum), int(phase), _stream()))


def rotate_lookup(ent, dim, ids, part_norm=True, sum_norm=False):
    """fp32 [n, pad4(dim)] block of l2n?(l2n?(re[ids]) + l2n?(im[ids])) for the evaluation kernels."""
    n = ent.shape[0] // 2 if ids is None else ids.numel()
    out = torch.empty((n, pad4(dim)), dtype=torch.float32, device=ent.device)
    check(lib().oea_rotate_lookup(_p(ent), ent.shape[0] // 2, dim, ent.shape[1], _p(ids), n, int(bool(part_norm)),
                                  int(bool(sum_norm)), _p(out), out.shape[1], _stream()))
    return out


def step_scatter_ent_rows(workspace, n_ent, n_rel, ld, ids, src):
    """grad_scratch[ids[i]] += src[i] (+ touched flags): gradients w.r.t. normalised entity rows
    computed outside the fused step (MTransE mapping loss)."""
    check(lib().oea_step_scatter_ent_rows(_p(workspace), n_ent, n_rel, ld, _p(ids), ids.numel(), _p(src),
                                          src.shape[1], _stream()))


def mapping_step(ent, dim, ent_l2_norm, ids1, ids2, mapping, mapping_acc, alpha, lr, optimizer, workspace, n_ent, n_rel,
                 loss_accum, work=None):
    """MTransE mapping step (oea_mapping_step): updates `mapping` in place and adds the entity-row gradients into
    the step workspace's scratch; follow with triple_step(..., empty, phase=PHASE_APPLY)."""
    n = ids1.numel()
    need = lib().oea_mapping_workspace_floats(n, ent.shape[1], dim)
    if work is None or work.numel() < need:
        work = torch.empty(need, dtype=torch.float32, device=ent.device)
    eg, et = C.c_void_p(), C.c_void_p()
    check(lib().oea_step_entity_scratch(_p(workspace), n_ent, n_rel, ent.shape[1], C.byref(eg), C.byref(et)))
    check(lib().oea_mapping_step(_p(ent), ent.shape[1], dim, int(bool(ent_l2_norm)), _p(ids1), _p(ids2), n, _p(mapping),
                                 _p(mapping_acc), float(alpha), float(lr), OPT_KIND[optimizer], eg, et, _p(work),
                                 _p(loss_accum), _stream()))
    return work


def mapping_epoch(ent, ent_acc, rel, rel_acc, dim, ent_l2_norm, batches, mapping, mapping_acc, alpha, lr, optimizer, cfg, workspace,
                  mapping_loss, step_loss, work=None):
    """a whole MTransE mapping epoch with one call (oea_mapping_epoch): batches device int32 [steps, 2, n]."""
    steps, _, n = batches.shape
    need = lib().oea_mapping_workspace_floats(n, ent.shape[1], dim)
    if work is None or work.numel() < need:
        work = torch.empty(need, dtype=torch.float32, device=ent.device)
    check(lib().oea_mapping_epoch(_p(ent), _p(ent_acc), ent.shape[0], _p(rel), _p(rel_acc), rel.shape[0], dim, ent.shape[1],
                                  int(bool(ent_l2_norm)), _p(batches), int(steps), int(n), _p(mapping), _p(mapping_acc), float(alpha),
                                  float(lr), OPT_KIND[optimizer], C.byref(cfg), _p(workspace), _p(work), _p(mapping_loss),
                                  _p(step_loss), _stream()))
    return work


def part_rows_per_rank(n_ent, world):
    return int(lib().oea_part_rows_per_rank(int(n_ent), int(world)))


def part_pack(workspace, n_ent, n_rel, ld, world, send, rel_x):
    check(lib().oea_part_pack(_p(workspace), n_ent, n_rel, ld, world, _p(send), _p(rel_x), _stream()))


def part_apply(ent, acc_own, rel, rel_acc, world, rank, own, rel_x, upd, cfg, workspace, n_items, loss_accum):
    check(lib().oea_part_apply(_p(ent), _p(acc_own), ent.shape[0], _p(rel), _p(rel_acc), rel.shape[0], ent.shape[1], world, rank,
                               _p(own), _p(rel_x), _p(upd), C.byref(cfg), _p(workspace), int(n_items), _p(loss_accum), _stream()))


def part_unpack(ent, world, rank, all_rows):
    check(lib().oea_part_unpack(_p(ent), ent.shape[0], ent.shape[1], world, rank, _p(all_rows), _stream()))


def step_exchange_view(workspace, n_ent, n_rel, ld):
    """view (fp32, or int64 in the deterministic build) of the workspace region (gradient scratch + touched flags) that
    data-parallel ranks sum with one all-reduce."""
    n = lib().oea_step_exchange_floats(n_ent, n_rel, ld)
    dt = scratch_dtype()
    return workspace[: dt.itemsize * n].view(dt)


def step_entity_flags(workspace, n_ent, n_rel, ld):
    """the entity rows' touched flags inside the step workspace (a view; scratch element type): nonzero = the row received gradient
    through the atomic scratch in the GRAD phase that just ran"""
    eg, et = C.c_void_p(), C.c_void_p()
    check(lib().oea_step_entity_scratch(_p(workspace), int(n_ent), int(n_rel), int(ld), C.byref(eg), C.byref(et)))
    dt = scratch_dtype()
    off = et.value - workspace.data_ptr()
    return workspace[off: off + dt.itemsize * int(n_ent)].view(dt)


def step_normal_views(workspace, n_ent, n_rel, ld):
    """TransH under the entity-id partition: fp32 views of the normal-vector gradient scratch [n_rel * ld] and its touched
    flags [n_rel] inside the step workspace (summed over the ranks after the GRAD phase, include/openea_hip.h)."""
    g_off, t_off = C.c_int64(0), C.c_int64(0)
    check(lib().oea_step_normal_scratch(int(n_ent), int(n_rel), int(ld), C.byref(g_off), C.byref(t_off)))
    dt = scratch_dtype()
    grad = workspace[g_off.value: g_off.value + dt.itemsize * n_rel * ld].view(dt)
    touched = workspace[t_off.value: t_off.value + dt.itemsize * n_rel].view(dt)
    return grad, touched


def step_apply_normals(n_ent, n_rel, ld, cfg, workspace):
    check(lib().oea_step_apply_normals(int(n_ent), int(n_rel), int(ld), C.byref(cfg), _p(workspace), _stream()))


def profile_begin(stride=1):
    check(lib().oea_profile_begin(int(stride)))


def profile_end(group=4):
    ms = (C.c_double * (group - 1))()
    n = C.c_int32(0)
    check(lib().oea_profile_end(group, ms, C.byref(n)))
    return [float(x) for x in ms], int(n.value)


# -------------------------------------------------------------------------------------------
# sampler
# -------------------------------------------------------------------------------------------


def tripleset_build(triples):
    """device int32 [n,3] -> device uint64 table (viewed as int64)."""
    n = triples.shape[0]
    if n:   # key layout (csrc/common.h:pack_triple): head 24 | relation 16 | tail 24 bits -- larger ids would alias
        mx = triples.max(dim=0).values.cpu()
        if int(mx[0]) >= 1 << 24 or int(mx[2]) >= 1 << 24 or int(mx[1]) >= 1 << 16 or int(triples.min().item()) < 0:
            raise OpenEAHipError("triple ids out of range for the packed membership keys "
                                 "(entity ids < 16,777,216, relation ids < 65,536)")
    cap = lib().oea_tripleset_capacity(n)
    table = torch.empty(cap, dtype=torch.int64, device=triples.device)
    check(lib().oea_tripleset_build(_p(triples), n, _p(table), cap, _stream()))
    return table


def sample_negatives(pos, k, table, entity_list, ent_pos=None, nbr=None, seed=0, step=0,
                     pos_offset=0, max_try=10, out=None, err_flag=None):
    n_pos = pos.shape[0]
    if out is None:
        out = torch.empty((n_pos * k, 3), dtype=torch.int32, device=pos.device)
    if err_flag is None:
        err_flag = torch.zeros(1, dtype=torch.int32, device=pos.device)
    nbr_k = 0 if nbr is None else nbr.shape[1]
    check(lib().oea_sample_negatives(_p(pos), n_pos, k, _p(table), table.numel(), _p(entity_list),
                                     entity_list.numel(), _p(ent_pos), _p(nbr), nbr_k, int(seed),
                                     int(step), int(pos_offset), int(max_try), _p(out), _p(err_flag),
                                     _stream()))
    return out, err_flag


def sample_negatives_replay(pos, k, table, entity_list, replay, ent_pos=None, nbr=None, max_try=10):
    """the sampler kernel fed from a recorded run of the reference (oea_sample_negatives_replay): replay int32
    [n_pos, max_try, 1 + k] = per round the side bit and the candidate-list positions random.sample returned"""
    n_pos = pos.shape[0]
    out = torch.zeros((n_pos * k, 3), dtype=torch.int32, device=pos.device)
    err_flag = torch.zeros(1, dtype=torch.int32, device=pos.device)
    nbr_k = 0 if nbr is None else nbr.shape[1]
    assert tuple(replay.shape) == (n_pos, max_try, 1 + k) and replay.dtype == torch.int32 and replay.is_contiguous()
    check(lib().oea_sample_negatives_replay(_p(pos), n_pos, k, _p(table), table.numel(), _p(entity_list), entity_list.numel(),
                                            _p(ent_pos), _p(nbr), nbr_k, int(max_try), _p(replay), _p(out), _p(err_flag), _stream()))
    return out, err_flag


def tripleset_filter(triples, capacity):
    """the "certainly absent" bit array over the triples of a membership table of `capacity` slots (oea_tripleset_filter_build)
    -> device int32 [capacity / 4] (8 x capacity bits)"""
    bits = int(lib().oea_tripleset_filter_bits(int(capacity)))
    filt = torch.empty(bits // 32, dtype=torch.int32, device=triples.device)
    check(lib().oea_tripleset_filter_build(_p(triples), triples.shape[0], _p(filt), bits, _stream()))
    return filt


def sampler_side(table, entity_list, ent_pos, nbr, filt=None):
    """pack one KG's sampler state for sample_negatives_pair (keeps the tensors alive)."""
    side = _lib.SamplerSide(table.data_ptr(), table.numel(), entity_list.data_ptr(),
                            ent_pos.data_ptr() if ent_pos is not None else None,
                            nbr.data_ptr() if nbr is not None else None, entity_list.numel(),
                            0 if nbr is None else nbr.shape[1],
                            filt.data_ptr() if filt is not None else None, 0 if filt is None else 32 * filt.numel())
    side._keep = (table, entity_list, ent_pos, nbr, filt)
    return side


def sample_negatives_pair(pos, n_split, k, side0, side1, seed, step, pos_offset, out, err_flag, max_try=10):
    """one launch for pos_batch1 + pos_batch2 (batch.py:36-45): rows [0, n_split) against side0."""
    check(lib().oea_sample_negatives_pair(_p(pos), pos.shape[0], int(n_split), k, C.byref(side0), C.byref(side1),
                                          int(seed), int(step), int(pos_offset), int(max_try), _p(out), _p(err_flag),
                                          _stream()))
    return out


def sample_negatives_epoch(pos_all, offsets_dev, splits_dev, steps, k, side0, side1, seed, step_base, out, err_flag, max_try=10):
    """negatives of every batch of an epoch in one launch (on the CURRENT torch stream)."""
    check(lib().oea_sample_negatives_epoch(_p(pos_all), pos_all.shape[0], _p(offsets_dev), _p(splits_dev), int(steps), int(k),
                                           C.byref(side0), C.byref(side1), int(seed), int(step_base), int(max_try), _p(out),
                                           _p(err_flag), _stream()))
    return out


def triple_epoch(ent, ent_acc, rel, rel_acc, dim, pos_all, offsets, splits, k, side0, side1, seed, step_base,
                 neg_buf, err_flag, cfg, workspace, loss_accum, offsets_dev=None, splits_dev=None, step_range=None, shard=(0, 1),
                 plan=None):
    """Enqueue every step of an epoch (or steps [lo, hi) of it: step_range) with one call (offsets / splits: host int64
    numpy arrays; their device copies enable sampling the whole epoch ahead in one launch -- neg_buf then covers the
    epoch).  step_base: Philox step of the epoch's step 0.  shard = (rank, world): this rank's contiguous share of every
    batch, trained on the local tables (dp_exchange = 'epoch')."""
    steps = len(splits)
    lo, hi = (0, steps) if step_range is None else step_range
    if plan is not None and tuple(shard) == (0, 1):
        # the gathered-sum plan of the epoch (step_plan_buffer / step_plan_build; include/openea_hip.h): plan = (buffer, built)
        check(lib().oea_triple_epoch_range_plan(_p(ent), _p(ent_acc), ent.shape[0], _p(rel), _p(rel_acc), rel.shape[0], dim,
                                                ent.shape[1], _p(pos_all), offsets.ctypes.data_as(C.c_void_p),
                                                splits.ctypes.data_as(C.c_void_p), steps, int(lo), int(hi), int(k),
                                                C.byref(side0) if side0 is not None else None,
                                                C.byref(side1) if side1 is not None else None, int(seed), int(step_base),
                                                _p(neg_buf), _p(err_flag), C.byref(cfg), _p(workspace), _p(loss_accum),
                                                _p(offsets_dev), _p(splits_dev), _p(plan[0]), plan[0].numel(), int(bool(plan[1])),
                                                _stream()))
        return
    check(lib().oea_triple_epoch_range_shard(_p(ent), _p(ent_acc), ent.shape[0], _p(rel), _p(rel_acc), rel.shape[0], dim,
                                       ent.shape[1], _p(pos_all), offsets.ctypes.data_as(C.c_void_p),
                                       splits.ctypes.data_as(C.c_void_p), steps, int(lo), int(hi), int(k),
                                       C.byref(side0) if side0 is not None else None,
                                       C.byref(side1) if side1 is not None else None, int(seed), int(step_base),
                                       _p(neg_buf), _p(err_flag), C.byref(cfg), _p(workspace), _p(loss_accum),
                                       _p(offsets_dev), _p(splits_dev), int(shard[0]), int(shard[1]), _stream()))


def epoch_layout(triples, n1, n2, slot, seed, epoch, dall, workspace=None):
    """an epoch's shuffle of both triple lists + the batch layout gather (oea_epoch_layout) on the current stream -> workspace"""
    n = int(n1) + int(n2)
    if workspace is None:
        workspace = torch.empty(lib().oea_epoch_layout_bytes(n), dtype=torch.uint8, device=triples.device)
    check(lib().oea_epoch_layout(_p(triples), int(n1), int(n2), _p(slot), slot.numel(), int(seed) & 0xFFFFFFFFFFFFFFFF, int(epoch) & 0xFFFFFFFF,
                                 _p(dall), _p(workspace), workspace.numel(), _stream()))
    return workspace


def step_plan_supported(cfg, n_ent, n_rel, ld, k):
    """would an epoch under cfg run on the gathered-sum plan (oea_step_plan_supported)?"""
    return bool(lib().oea_step_plan_supported(C.byref(cfg), int(n_ent), int(n_rel), int(ld), int(k)))


def step_plan_buffer(n_total, steps, max_batch, n_ent, ld, dev=None):
    """workspace of one epoch's plan (oea_step_plan_bytes)"""
    nbytes = lib().oea_step_plan_bytes(int(n_total), int(steps), int(max_batch), int(n_ent), int(ld))
    return torch.empty(nbytes, dtype=torch.uint8, device=dev or device())


def step_plan_build(pos_all, neg_all, k, offsets_dev, n_total, steps, max_batch, n_ent, ld, plan):
    """sort the epoch's (step, row) references on the current stream (oea_step_plan_build): no allocation, no host read"""
    check(lib().oea_step_plan_build(_p(pos_all), _p(neg_all), int(k), _p(offsets_dev), int(n_total), int(steps), int(max_batch),
                                    int(n_ent), int(ld), _p(plan), plan.numel(), _stream()))


def step_plan_arrays(plan, n_total, steps, max_batch, n_ent, ld):
    """host copies of a built plan (tests): dict(vals uint32 [2 n_total], ukeys uint64 [n_unique], uoff uint32 [n_unique + 1],
    step_first int32 [steps + 1], row_bits, pflags uint32 [n_total])"""
    off = (C.c_int64 * 9)()
    check(lib().oea_step_plan_offsets(int(n_total), int(steps), int(max_batch), int(n_ent), int(ld), C.cast(off, C.c_void_p)))
    raw = plan.cpu().numpy()
    m = 2 * int(n_total)
    nu = int(raw[off[3]: off[3] + 4].view(np.int32)[0])
    return dict(vals=raw[off[0]: off[0] + 4 * m].view(np.uint32).copy(), ukeys=raw[off[1]: off[1] + 8 * nu].view(np.uint64).copy(),
                uoff=raw[off[2]: off[2] + 4 * (nu + 1)].view(np.uint32).copy(),
                step_first=raw[off[4]: off[4] + 4 * (steps + 1)].view(np.int32).copy(), row_bits=int(off[6]), n_unique=nu,
                pflags=raw[off[8]: off[8] + 4 * int(n_total)].view(np.uint32).copy())


def step_plan_stats(plan, n_total, steps, max_batch, n_ent, ld):
    """what a built plan looks like (bench detail / experiments): positives outside the rule, references to hub rows, rows and
    entries per step"""
    a = step_plan_arrays(plan, n_total, steps, max_batch, n_ent, ld)
    last = int(a["step_first"][steps])                       # distinct keys of real steps
    cnt = np.diff(a["uoff"].astype(np.int64))[:last]
    listed = int(a["uoff"][last])                            # references of positives inside the rule
    hub = cnt > 8
    return dict(positives=int(n_total), outside_rule=int(n_total - listed // 2), rows_per_step=float(last) / max(steps, 1),
                refs_per_row_mean=float(cnt.mean()) if last else 0.0, refs_per_row_max=int(cnt.max()) if last else 0,
                hub_rows_per_step=float(hub.sum()) / max(steps, 1), refs_to_hub_rows=int(cnt[hub].sum()),
                refs_listed=listed)


def comm_single_or_none():
    """a one-rank communicator of the C ABI (tests): (handle, destroy)"""
    uid = (C.c_char * 128)()
    check(lib().oea_comm_unique_id(uid))
    comm = C.c_void_p()
    check(lib().oea_comm_init(uid, 0, 1, C.byref(comm)))
    return comm


def part_buffers(n_ent, n_rel, ld, world, dev, adagrad=True):
    """the exchange buffers of the partitioned step (include/openea_hip.h) + the optimiser state of the owned rows"""
    rpr = lib().oea_part_rows_per_rank(int(n_ent), int(world))
    chunk = rpr * (ld + 1)
    f = dict(dtype=torch.float32, device=dev)
    g = dict(dtype=scratch_dtype(), device=dev)         # the gradients travel in the scratch's own type
    return dict(rpr=rpr, chunk=chunk, send=torch.empty(world * chunk, **g), own=torch.empty(chunk, **g),
                rel_x=torch.empty(n_rel * (ld + 1), **g), upd=torch.empty((rpr, ld), **f), all=torch.empty((world, rpr, ld), **f),
                acc_own=torch.full((rpr, ld), 0.1, **f) if adagrad else None)


def triple_epoch_comm(comm, ent, acc_own, rel, rel_acc, dim, pos_all, offsets, splits, k, side0, side1, seed, step_base, neg_buf,
                      err_flag, cfg, workspace, loss_accum, offsets_dev, splits_dev, bufs, step_range=None):
    """steps [lo, hi) of a data-parallel epoch under the entity-id partition from ONE C call over the C ABI's communicator
    (oea_triple_epoch_range_comm); bufs = part_buffers(...)."""
    offsets, splits = np.ascontiguousarray(offsets, np.int64), np.ascontiguousarray(splits, np.int64)   # what the C side reads
    steps = len(splits)
    lo, hi = (0, steps) if step_range is None else step_range
    check(lib().oea_triple_epoch_range_comm(comm, _p(ent), _p(acc_own), ent.shape[0], _p(rel), _p(rel_acc), rel.shape[0], dim,
                                            ent.shape[1], _p(pos_all), offsets.ctypes.data_as(C.c_void_p),
                                            splits.ctypes.data_as(C.c_void_p), steps, int(lo), int(hi), int(k),
                                            C.byref(side0) if side0 is not None else None,
                                            C.byref(side1) if side1 is not None else None, int(seed), int(step_base),
                                            _p(neg_buf), _p(err_flag), C.byref(cfg), _p(workspace), _p(loss_accum),
                                            _p(offsets_dev), _p(splits_dev), _p(bufs['send']), _p(bufs['own']), _p(bufs['rel_x']),
                                            _p(bufs['upd']), _p(bufs['all']), _stream()))


def halo_buffers(n_ent, ld, world, steps, max_batch, k, dev):
    """workspace + the two exchange buffers of the boundary-row exchange (oea_triple_epoch_range_halo)"""
    wsb = lib().oea_halo_workspace_bytes(int(n_ent), int(world), int(steps), int(max_batch), int(k))
    xb = lib().oea_halo_buffer_bytes(int(n_ent), int(world), int(max_batch), int(k), int(ld))
    u8 = dict(dtype=torch.uint8, device=dev)
    return dict(ws=torch.empty(wsb, **u8), a=torch.empty(xb, **u8), b=torch.empty(xb, **u8), key=(int(steps), int(max_batch), int(k)))


def halo_plan(pos_all, neg_all, k, offsets, n_ent, world, step_range=None, with_lists=True):
    """which rows every rank's share of every step refers to, per owner (oea_halo_plan) -> (counts int32 [steps, world, world],
    lists int32 [steps, world, cap] or None): what sizes the messages of the boundary-row exchange"""
    offsets = np.ascontiguousarray(offsets, np.int64)      # ONE array for the host pointer and the device copy
    steps = len(offsets) - 1
    lo, hi = (0, steps) if step_range is None else step_range
    max_batch = int(np.diff(offsets).max()) if steps else 0
    dev = pos_all.device
    ws = torch.empty(lib().oea_halo_workspace_bytes(int(n_ent), int(world), hi - lo, max_batch, int(k)), dtype=torch.uint8, device=dev)
    off_dev = torch.from_numpy(offsets).to(dev)
    counts = np.zeros((hi - lo, world, world), np.int32)
    cap = C.c_int64(0)
    check(lib().oea_halo_plan(_p(pos_all), _p(neg_all), int(k), offsets.ctypes.data_as(C.c_void_p), _p(off_dev), steps, int(lo), int(hi),
                              int(n_ent), int(world), _p(ws), ws.numel(), counts.ctypes.data_as(C.c_void_p), None, C.byref(cap), _stream()))
    lists = None
    if with_lists:
        lists = np.zeros((hi - lo, world, int(cap.value)), np.int32)
        check(lib().oea_halo_plan(_p(pos_all), _p(neg_all), int(k), offsets.ctypes.data_as(C.c_void_p), _p(off_dev), steps, int(lo), int(hi),
                                  int(n_ent), int(world), _p(ws), ws.numel(), counts.ctypes.data_as(C.c_void_p),
                                  lists.ctypes.data_as(C.c_void_p), C.byref(cap), _stream()))
    return counts, lists


def triple_epoch_halo(comm, ent, acc_own, rel, rel_acc, dim, pos_all, offsets, splits, k, side0, side1, seed, step_base, neg_buf,
                      err_flag, cfg, workspace, loss_accum, offsets_dev, splits_dev, bufs, halo, step_range=None):
    """triple_epoch_comm with the boundary-row exchange (oea_triple_epoch_range_halo); bufs = part_buffers(...), halo =
    halo_buffers(...) -> (bytes pushed, bytes pulled, largest rows sent in a step, steps) of this rank"""
    offsets, splits = np.ascontiguousarray(offsets, np.int64), np.ascontiguousarray(splits, np.int64)   # what the C side reads
    steps = len(splits)
    lo, hi = (0, steps) if step_range is None else step_range
    stats = (C.c_int64 * 4)()
    check(lib().oea_triple_epoch_range_halo(comm, _p(ent), _p(acc_own), ent.shape[0], _p(rel), _p(rel_acc), rel.shape[0], dim,
                                            ent.shape[1], _p(pos_all), offsets.ctypes.data_as(C.c_void_p),
                                            splits.ctypes.data_as(C.c_void_p), steps, int(lo), int(hi), int(k),
                                            C.byref(side0) if side0 is not None else None,
                                            C.byref(side1) if side1 is not None else None, int(seed), int(step_base),
                                            _p(neg_buf), _p(err_flag), C.byref(cfg), _p(workspace), _p(loss_accum),
                                            _p(offsets_dev), _p(splits_dev), _p(halo['ws']), halo['ws'].numel(), _p(halo['a']),
                                            _p(halo['b']), halo['a'].numel(), _p(bufs['rel_x']), _p(bufs['upd']), _p(bufs['all']),
                                            C.cast(stats, C.c_void_p), _stream()))
    return tuple(int(x) for x in stats)


def sample_link_negatives(n_pos, k, pos_links=None, ents1=None, ents2=None, nbr1=None, row1=None, nbr2=None, row2=None,
                          exclude=None, seed=0, step=0, scratch=None):
    """AliNet.generate_input_batch negatives on the device -> (pairs int32 [m, 2], valid fp32 [m]).
    uniform: ents1 / ents2 (device int32 lists); truncated: pos_links [n_pos, 2] + nbr1/nbr2 [rows, nbr_k] + row maps.
    exclude: tripleset over (e1, 0, e2) (tripleset_build) or None."""
    truncated = nbr1 is not None
    m = (2 if truncated else 1) * k * n_pos
    dev = (nbr1 if truncated else ents1).device
    pairs = torch.empty((m, 2), dtype=torch.int32, device=dev)
    valid = torch.empty(m, dtype=torch.float32, device=dev)
    cap = 2
    while cap < 2 * max(m, 1):
        cap *= 2
    if scratch is None or scratch[0].numel() < cap:
        scratch = (torch.empty(cap, dtype=torch.int64, device=dev), torch.empty(cap, dtype=torch.int32, device=dev))
    check(lib().oea_sample_link_negatives(_p(pos_links), n_pos, k, _p(ents1), 0 if ents1 is None else ents1.numel(),
                                          _p(ents2), 0 if ents2 is None else ents2.numel(), _p(nbr1), _p(row1), _p(nbr2),
                                          _p(row2), nbr1.shape[1] if truncated else 0, _p(exclude),
                                          0 if exclude is None else exclude.numel(), int(seed), int(step), _p(pairs),
                                          _p(valid), _p(scratch[0]), _p(scratch[1]), cap, _stream()))
    return pairs, valid, scratch


def greedy_matching(left, right, weight):
    """HOST numpy arrays -> bool mask of the edges kept by the weight-descending one-to-one selection."""
    left = np.ascontiguousarray(left, np.int32)
    right = np.ascontiguousarray(right, np.int32)
    weight = np.ascontiguousarray(weight, np.float32)
    sel = np.zeros(len(left), np.uint8)
    check(_lib.load(require_device=False).oea_greedy_matching(left.ctypes.data_as(C.c_void_p), right.ctypes.data_as(C.c_void_p),
                                                              weight.ctypes.data_as(C.c_void_p), len(left),
                                                              sel.ctypes.data_as(C.c_void_p)))
    return sel.astype(bool)


def pair_dots(e1, e2, dim, ii, jj):
    """device: out[i] = <e1[ii[i]], e2[jj[i]]>."""
    out = torch.empty(ii.numel(), dtype=torch.float32, device=e1.device)
    check(lib().oea_pair_dots(_p(e1), e1.shape[1], _p(e2), e2.shape[1], dim, _p(ii), _p(jj), ii.numel(), _p(out), _stream()))
    return out


# -------------------------------------------------------------------------------------------
# neighbour search / evaluation
# -------------------------------------------------------------------------------------------


def topk_inner(q, c, dim, k, id_map=None, ws_bytes=None):
    nq, nc = q.shape[0], c.shape[0]
    full = lib().oea_topk_workspace_bytes(nq, nc)
    # default: strips of <= 2 GB.  Measured at 100,000 x 100,000, k = 2,000: 64 MB strips 169 ms, 192 MB (Infinity-
    # Cache resident) 77 ms, 1.5 GB 51 ms, 12 GB 46 ms -- thousands of rows per launch (a full wave of workgroups
    # for both kernels, few launch tails) matter more than keeping the strip in the 256 MB cache.
    # Long candidate lists (nc >= 32,768, nq >= 4,096) take the strip-free path (csrc/topk.hip): ~55 KB of workspace per
    # query row (sample strip + survivor lists); 8 GB covers 100,000 queries in one pass.
    big = nc >= int(os.environ.get("OEA_TOPK_LISTS_MIN", "32768")) and nq >= 4096
    ws_bytes = min(full, max((8 << 30) if big else (2 << 30), 128 * ((nc + 31) // 32 * 32) * 4)) if ws_bytes is None else min(full, ws_bytes)
    if q.data_ptr() == c.data_ptr() and nq == nc:
        # one KG's entities against themselves: the symmetric search (upper-triangle tiles only) wants its lists for ALL rows
        # at once (~0.3 MB per row at k / n = 2 %); taken when it fits in a third of the free memory
        need = int(lib().oea_topk_sym_workspace_bytes(nq, k))
        if need and need <= torch.cuda.mem_get_info(q.device)[0] // 3:
            ws_bytes = max(ws_bytes, need)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
    out = torch.empty((nq, k), dtype=torch.int32, device=q.device)
    check(lib().oea_topk_inner(_p(q), nq, q.shape[1], _p(c), nc, c.shape[1], dim, k, _p(id_map), _p(out),
                               _p(ws), ws_bytes, _stream()))
    return out


def topk_rows(s, k, id_map=None, nc=None):
    """k largest of the first nc (default: all) columns of every row of a device similarity strip
    [n, ld] -> int32 [n, k]."""
    out = torch.empty((s.shape[0], k), dtype=torch.int32, device=s.device)
    check(lib().oea_topk_rows(_p(s), s.shape[0], s.shape[1] if nc is None else nc, s.stride(0), k, _p(id_map), _p(out),
                              _stream()))
    return out


def row_rank_select(vals, k, largest, ids=None, want_sel=True, want_kth=False):
    """the k best of every row of a short [n, c <= 1,024] fp32 / fp64 matrix by ranking (ties: earlier column) -> (selected columns
    int32 [n, k] in ascending column order, mapped through ids [n, c] when given; k-th best value [n]) (oea_row_rank_select_*)"""
    assert vals.dim() == 2 and vals.is_contiguous() and vals.dtype in (torch.float32, torch.float64)
    n, c = vals.shape
    sel = torch.empty((n, k), dtype=torch.int32, device=vals.device) if want_sel else None
    kth = torch.empty(n, dtype=vals.dtype, device=vals.device) if want_kth else None
    fn = lib().oea_row_rank_select_f32 if vals.dtype == torch.float32 else lib().oea_row_rank_select_f64
    if ids is not None:
        assert ids.dtype == torch.int32 and ids.is_contiguous() and ids.shape == vals.shape
    check(fn(_p(vals), n, c, vals.stride(0), int(k), int(bool(largest)), _p(ids), 0 if ids is None else ids.stride(0), _p(sel), _p(kth),
             _stream()))
    return sel, kth


def eval_bf16_enabled(n1, n2):
    """the certified bf16 prefilter (oea_rank_eval_bf16) takes the inner-product evaluation (with or without CSLS means) from
    3e8 pairs (~17,000^2) on: its six launches and the exact fix-up cost ~0.1 ms, which the 10,500 test pairs of the 15K
    datasets do not win back inside greedy_alignment (bench r04k: 34.7 vs 38.5 M pairs/s), the 70,000 of the 100K datasets do
    (11.6 vs 7.5 M pairs/s).  OEA_EVAL_BF16=0 keeps the fp32 sweep, OEA_EVAL_BF16_MIN_PAIRS moves the limit."""
    lim = float(os.environ.get('OEA_EVAL_BF16_MIN_PAIRS', '3e8'))
    return os.environ.get('OEA_EVAL_BF16', '1')[:1] != '0' and tile_glds() and n1 * n2 >= lim


def rank_eval(e1, e2, dim, metric='inner', csls_r=None, csls_c=None, gold_offset=0, allow_bf16=True):
    """gold of query row i is candidate row gold_offset + i (row-sharded evaluation passes its lo).  allow_bf16=False: the fp32
    sweep whatever the size (rank_eval_bf16's fallback on a record overflow -- an explicit argument instead of a process-global
    re-entrancy flag)."""
    n1, n2 = e1.shape[0], e2.shape[0]
    assert n1 + gold_offset <= n2, "gold of row i is column gold_offset + i <= n2"
    if metric == 'inner' and n1 > 0 and allow_bf16 and eval_bf16_enabled(n1, n2):
        return rank_eval_bf16(e1, e2, dim, gold_offset, csls_r=csls_r, csls_c=csls_c)
    if metric == 'manhattan' and n1 > 0 and n2 >= 2048 and os.environ.get('OEA_L1_EVAL', 'grid') != 'f64':
        return rank_eval_l1_grid(e1, e2, dim, gold_offset, csls_r=csls_r, csls_c=csls_c)
    ws = torch.empty(lib().oea_rank_workspace_bytes(n1), dtype=torch.uint8, device=e1.device)
    rank = torch.empty(n1, dtype=torch.int32, device=e1.device)
    argmax = torch.empty(n1, dtype=torch.int32, device=e1.device)
    check(lib().oea_rank_eval(_p(e1), n1, e1.shape[1], _p(e2), n2, e2.shape[1], dim, METRIC[metric],
                              _p(csls_r), _p(csls_c), int(gold_offset), _p(rank), _p(argmax), _p(ws), _stream()))
    return rank, argmax


class L1Grid:
    """both tables of a manhattan evaluation on ONE 16-bit grid over their common range (oea_quantize_rows_u16), the error
    bound of a grid distance, and -- when they fit a third of the free memory -- the strips of grid distances of the query
    blocks, kept so that the CSLS evaluation (row means, then ranks) computes them once."""

    def __init__(self, e1, e2, dim, block_bytes=2 << 30):
        self.e1, self.e2, self.dim = e1, e2, dim
        lo1, hi1 = torch.aminmax(e1[:, :dim])
        lo2, hi2 = torch.aminmax(e2[:, :dim])
        lo, hi = min(float(lo1), float(lo2)), max(float(hi1), float(hi2))
        self.step = max(hi - lo, 1e-30) / 65535.0
        self.q1 = quantize_rows_u16(e1, dim, lo, 1.0 / self.step)
        self.q2 = self.q1 if e2 is e1 else quantize_rows_u16(e2, dim, lo, 1.0 / self.step)
        self.err = (dim * 1.02 + 1.0) * self.step      # half a step per operand and column, + the fp32 rounding of the grid map
        self.ld = (e2.shape[0] + 31) // 32 * 32
        self.rows_per = int(max(128, min(e1.shape[0], block_bytes // (4 * self.ld))))
        self.strips = {}                               # r0 -> strip of query rows [r0, r0 + rows) against all of e2

    def blocks(self):
        n1 = self.e1.shape[0]
        return [(r0, min(self.rows_per, n1 - r0)) for r0 in range(0, n1, self.rows_per)]

    def strip(self, r0, rows, keep=False):
        s = self.strips.get(r0)
        if s is None:
            s = l1_u16_strip(self.q1[r0: r0 + rows], self.q2)
            if keep:
                self.strips[r0] = s
        return s

    def can_keep(self):
        need = 4 * self.ld * self.e1.shape[0]
        return need <= torch.cuda.mem_get_info(self.e1.device)[0] // 3


def pair_l1_sim(q, table, dim, cand):
    """exact manhattan similarities float(1 - d) of the candidate lists cand int32 [nq, c] (sequential fp64 chain) -> fp32 [nq, c]"""
    nq, c = cand.shape
    out = torch.empty((nq, c), dtype=torch.float32, device=q.device)
    check(lib().oea_pair_l1_sim(_p(q), nq, q.shape[1], _p(table), table.shape[0], table.shape[1], dim, _p(cand), c, _p(out), _stream()))
    return out


def l1_grid_topk_means(q_tab, c_tab, qq, qc, dim, k, step, err, margin=32, block_bytes=2 << 30, keep=None, stats=None):
    """mean of the k largest manhattan similarities of every row of q_tab against c_tab (calculate_nearest_k,
    similarity.py:80-83) WITHOUT the fp64 distance of every pair: the k + margin nearest on the 16-bit grid, their exact
    similarities (sequential fp64 chain, rounded to fp32 like sim()), the mean by oea_row_topk_mean on that short list --
    the same values summed in the same order as over the whole row.  Every list is CERTIFIED: no entity outside it can be
    nearer than its farthest member's lower bound, which must not beat the k-th exact value; rows that fail take the
    all-pairs fp64 strip.  keep: an L1Grid whose strips this pass should leave behind (q_tab = its e1)."""
    n, nc = q_tab.shape[0], c_tab.shape[0]
    ld = (nc + 31) // 32 * 32
    rows_per = keep.rows_per if keep is not None else int(max(128, min(n, block_bytes // (4 * ld))))
    out = torch.empty(n, dtype=torch.float32, device=q_tab.device)
    c = min(k + margin, nc)
    bad = torch.zeros(n, dtype=torch.bool, device=q_tab.device)
    for r0 in range(0, n, rows_per):
        rows = min(rows_per, n - r0)
        strip = keep.strip(r0, rows, keep=True) if keep is not None else l1_u16_strip(qq[r0: r0 + rows], qc)
        cand = topk_rows(strip, c, nc=nc)                                            # the c smallest grid distances
        worst = -torch.gather(strip, 1, cand.to(torch.int64)).amin(dim=1).to(torch.float64)
        qb = q_tab[r0: r0 + rows]
        sims = pair_l1_sim(qb, c_tab, dim, cand)
        out[r0: r0 + rows] = row_topk_mean(sims, k)
        _, kth = row_rank_select(sims, k, True, want_sel=False, want_kth=True)       # the k-th largest exact similarity of the list
        # a non-member's distance is at least worst * step - err (+ 4 steps: float rounding of large grid sums)
        bound = 1.0 - (worst * step - (err + 4.0 * step))
        if c < nc:
            bad[r0: r0 + rows] = ~(bound <= kth.to(torch.float64))
        del strip
    redo = torch.nonzero(bad).reshape(-1)                    # ONE host read per call (was one per block of rows)
    redone = int(redo.numel())
    for b0 in range(0, redone, 4096):                        # uncertified rows: the all-pairs fp64 strip
        idx = redo[b0: b0 + 4096]
        s = sim_matrix(q_tab.index_select(0, idx).contiguous(), c_tab, dim, 'manhattan')
        out[idx] = row_topk_mean(s, k)
        del s
    if stats is not None:
        stats['uncertified'] = stats.get('uncertified', 0) + redone
    return out


def csls_means_l1_grid(e1, e2, dim, k, cols=True):
    """CSLS means of the manhattan metric from grid distances -> (r [n1], c [n2] or None, grid): the L1Grid keeps the query
    strips (when they fit) for rank_eval_l1_grid(..., grid=grid)."""
    grid = L1Grid(e1, e2, dim)
    keep = grid if (cols and grid.can_keep()) else None        # the strips are kept for the single-process rank pass only
    r = l1_grid_topk_means(e1, e2, grid.q1, grid.q2, dim, k, grid.step, grid.err, keep=keep)
    c = l1_grid_topk_means(e2, e1, grid.q2, grid.q1, dim, k, grid.step, grid.err) if cols else None
    return r, c, grid


def sim_bf16_matrix(e1, e2, dim):
    """the bf16 prefilter's approximate inner products (tests / timing) -> fp32 [n1, n2]"""
    n1, n2 = e1.shape[0], e2.shape[0]
    out = torch.empty((n1, n2), dtype=torch.float32, device=e1.device)
    check(lib().oea_sim_bf16_matrix(_p(e1), n1, e1.shape[1], _p(e2), n2, e2.shape[1], dim, _p(out), n2, _stream()))
    return out


def rank_eval_bf16(e1, e2, dim, gold_offset=0, stats=None, csls_r=None, csls_c=None):
    """rank_eval(metric='inner') through the certified bf16 prefilter (oea_rank_eval_bf16[_csls]): the same rank / argmax, the
    matrix work at 3/16 of the fp32 matrix time.  Falls back to the fp32 sweep when the record buffer overflows (one host read
    of the status word).  stats: optional dict, receives 'records' and 'fallback'."""
    n1 = e1.shape[0]
    ws = torch.empty(lib().oea_rank_eval_bf16_workspace_bytes(n1, dim), dtype=torch.uint8, device=e1.device)
    rank = torch.empty(n1, dtype=torch.int32, device=e1.device)
    argmax = torch.empty(n1, dtype=torch.int32, device=e1.device)
    status = torch.zeros(2, dtype=torch.int32, device=e1.device)
    check(lib().oea_rank_eval_bf16_csls(_p(e1), n1, e1.shape[1], _p(e2), e2.shape[0], e2.shape[1], dim, _p(csls_r), _p(csls_c),
                                        int(gold_offset), _p(rank), _p(argmax), _p(status), _p(ws), _stream()))
    st = status.cpu().numpy()
    if stats is not None:
        stats['records'], stats['fallback'] = int(st[1]), bool(st[0])
    if st[0]:
        return rank_eval(e1, e2, dim, 'inner', csls_r, csls_c, gold_offset=gold_offset, allow_bf16=False)
    return rank, argmax


def rank_eval_metrics_bf16(e1, e2, dim, top_k, gold_offset=0, stats=None, csls_r=None, csls_c=None):
    """rank_eval_metrics through the certified bf16 prefilter: six launches + ONE device->host copy that carries the metrics and
    the sweep's status -> (rank, argmax, hits, rank_sum, rr_sum), or None when the record buffer overflowed (the caller takes
    the fp32 sweep)."""
    n1, nk = e1.shape[0], len(top_k)
    ws = torch.empty(lib().oea_rank_eval_bf16_workspace_bytes(n1, dim), dtype=torch.uint8, device=e1.device)
    rank = torch.empty(n1, dtype=torch.int32, device=e1.device)
    argmax = torch.empty(n1, dtype=torch.int32, device=e1.device)
    buf = torch.empty(nk + 4, dtype=torch.int64, device=e1.device)
    tk = (C.c_int32 * nk)(*[int(k) for k in top_k])
    check(lib().oea_rank_eval_metrics_bf16(_p(e1), n1, e1.shape[1], _p(e2), e2.shape[0], e2.shape[1], dim, _p(csls_r), _p(csls_c),
                                           int(gold_offset), tk, nk, _p(rank), _p(argmax), C.c_void_p(buf.data_ptr()), _p(ws), _stream()))
    host = buf.cpu().numpy()
    if stats is not None:
        stats['records'], stats['fallback'] = int(host[nk + 3]), bool(host[nk + 2])
    if host[nk + 2]:
        return None
    return rank, argmax, [int(x) for x in host[:nk]], int(host[nk]), float(host[nk + 1:nk + 2].view(np.float64)[0])


def rank_eval_l1_grid(e1, e2, dim, gold_offset=0, block_bytes=2 << 30, csls_r=None, csls_c=None, grid=None):
    """rank_eval(metric='manhattan') without the fp64 distance of every pair: 16-bit grid distances of all pairs
    (oea_l1_u16_strip, blocks of query rows), then oea_rank_l1_grid_rows[_csls] -- exact similarities only where the grid
    leaves a doubt.  Same ranks and nearest candidates as the all-pairs fp64 kernel (tested), with or without CSLS means."""
    n1, n2 = e1.shape[0], e2.shape[0]
    if grid is None or grid.e1.data_ptr() != e1.data_ptr() or grid.e2.data_ptr() != e2.data_ptr() or grid.e1.shape != e1.shape:
        grid = L1Grid(e1, e2, dim, block_bytes)
    step, err, ld = grid.step, grid.err, grid.ld
    rank = torch.empty(n1, dtype=torch.int32, device=e1.device)
    argmax = torch.empty(n1, dtype=torch.int32, device=e1.device)
    n_exact = torch.zeros(1, dtype=torch.int32, device=e1.device)
    csls = csls_r is not None
    for r0, rows in grid.blocks():
        strip = grid.strip(r0, rows)
        if csls:
            check(lib().oea_rank_l1_grid_rows_csls(_p(strip), rows, r0, n2, ld, _p(e1), e1.shape[1], _p(e2), e2.shape[1], dim,
                                                   int(gold_offset), float(step), float(err), _p(csls_r), _p(csls_c), _p(rank),
                                                   _p(argmax), _p(n_exact), _stream()))
        else:
            check(lib().oea_rank_l1_grid_rows(_p(strip), rows, r0, n2, ld, _p(e1), e1.shape[1], _p(e2), e2.shape[1], dim,
                                              int(gold_offset), float(step), float(err), _p(rank), _p(argmax), _p(n_exact), _stream()))
        del strip
        grid.strips.pop(r0, None)
        r1 = r0 + rows
        if r0 == 0 and r1 < n1 and int(n_exact.item()) > rows // 64:
            # the grid is useless for this table (outliers stretch its range: most candidates are within the error bound of
            # the gold distance and whole rows fall back to exact pairs): the rest through the all-pairs fp64 kernel
            rest = n1 - r1
            ws = torch.empty(lib().oea_rank_workspace_bytes(rest), dtype=torch.uint8, device=e1.device)
            rk, am = torch.empty(rest, dtype=torch.int32, device=e1.device), torch.empty(rest, dtype=torch.int32, device=e1.device)
            check(lib().oea_rank_eval(_p(e1[r1:]), rest, e1.shape[1], _p(e2), n2, e2.shape[1], dim, METRIC['manhattan'],
                                      _p(csls_r[r1:].contiguous()) if csls else None, _p(csls_c), int(gold_offset) + r1, _p(rk), _p(am),
                                      _p(ws), _stream()))
            rank[r1:], argmax[r1:] = rk, am
            grid.strips.clear()
            break
    return rank, argmax


def tile_glds():
    """the similarity tiles run on packed operands staged by LDS-DMA (default; OEA_TILE_GLDS=0 selects the register-staged
    pipeline, kept for the bit-exactness test between the two)"""
    import os
    return os.environ.get("OEA_TILE_GLDS", "1")[:1] != "0"


def rank_eval_metrics(e1, e2, dim, top_k, csls_r=None, csls_c=None, gold_offset=0):
    """inner-product evaluation in two launches (oea_rank_eval_metrics) + ONE device->host copy ->
    (rank int32 [n1] device, argmax int32 [n1] device, hits counts list[int], rank_sum int, rr_sum float)."""
    n1, n2 = e1.shape[0], e2.shape[0]
    assert n1 + gold_offset <= n2, "gold of row i is column gold_offset + i <= n2"
    nk = len(top_k)
    ws = torch.empty(lib().oea_rank_eval_metrics_workspace_bytes(n1), dtype=torch.uint8, device=e1.device)
    rank = torch.empty(n1, dtype=torch.int32, device=e1.device)
    argmax = torch.empty(n1, dtype=torch.int32, device=e1.device)
    buf = torch.empty(nk + 2, dtype=torch.int64, device=e1.device)     # hits[nk], rank_sum, rr bits: all written by the kernel
    tk = (C.c_int32 * nk)(*[int(k) for k in top_k])
    check(lib().oea_rank_eval_metrics(_p(e1), n1, e1.shape[1], _p(e2), n2, e2.shape[1], dim, _p(csls_r), _p(csls_c),
                                      int(gold_offset), tk, nk, _p(rank), _p(argmax), C.c_void_p(buf.data_ptr()),
                                      C.c_void_p(buf.data_ptr() + 8 * (nk + 1)), _p(ws), _stream()))
    host = buf.cpu().numpy()
    return rank, argmax, [int(x) for x in host[:nk]], int(host[nk]), float(host[nk + 1:nk + 2].view(np.float64)[0])


def rank_rows(s, gold_idx):
    """rank of column gold_idx[i] in row i of a device similarity block + row argmax -> (int32[n], int32[n])."""
    n = s.shape[0]
    rank = torch.empty(n, dtype=torch.int32, device=s.device)
    argmax = torch.empty(n, dtype=torch.int32, device=s.device)
    check(lib().oea_rank_rows(_p(s), n, s.shape[1], s.stride(0), _p(gold_idx), _p(rank), _p(argmax), _stream()))
    return rank, argmax


def rank_metrics(rank, top_k):
    """-> (hits counts list[int], rank_sum int, rr_sum float) with ONE device->host copy."""
    nk = len(top_k)
    tk = (C.c_int32 * nk)(*[int(k) for k in top_k])
    buf = torch.zeros(nk + 2, dtype=torch.int64, device=rank.device)   # hits[nk], rank_sum, rr bits
    check(lib().oea_rank_metrics(_p(rank), rank.numel(), tk, nk, C.c_void_p(buf.data_ptr()),
                                 C.c_void_p(buf.data_ptr() + 8 * nk), C.c_void_p(buf.data_ptr() + 8 * (nk + 1)),
                                 _stream()))
    host = buf.cpu().numpy()
    hits = [int(x) for x in host[:nk]]
    return hits, int(host[nk]), float(host[nk + 1:nk + 2].view(np.float64)[0])


def sim_matrix(e1, e2, dim, metric='inner', pad=False):
    """-> device fp32 [n1, n2]; pad=True: [n1, ld] with ld = n2 rounded up to 32 (the row layout that
    topk_rows reads; columns n2.. are left unwritten)."""
    n1, n2 = e1.shape[0], e2.shape[0]
    ld = (n2 + 31) // 32 * 32 if pad else n2
    out = torch.empty((n1, ld), dtype=torch.float32, device=e1.device)
    check(lib().oea_sim_matrix(_p(e1), n1, e1.shape[1], _p(e2), n2, e2.shape[1], dim, METRIC[metric],
                               _p(out), ld, _stream()))
    return out


def quantize_rows_u16(table, dim, lo, inv_step):
    """-> int16-typed device tensor [n, ldq] holding u16 grid points (ldq = dim rounded up to 8; pad columns 0)."""
    n = table.shape[0]
    ldq = (dim + 7) // 8 * 8
    out = torch.empty((n, ldq), dtype=torch.int16, device=table.device)
    check(lib().oea_quantize_rows_u16(_p(table), n, table.shape[1], dim, float(lo), float(inv_step), _p(out), ldq, _stream()))
    return out


def l1_u16_strip(qq, qc, pad=True):
    """-> device fp32 [nq, ld]: minus the integer L1 distance of every (query, candidate) pair of two u16 row sets."""
    nq, nc = qq.shape[0], qc.shape[0]
    ld = (nc + 31) // 32 * 32 if pad else nc
    out = torch.empty((nq, ld), dtype=torch.float32, device=qq.device)
    check(lib().oea_l1_u16_strip(_p(qq), nq, _p(qc), nc, qq.shape[1], _p(out), ld, _stream()))
    return out


def pair_l1_f64(q, table, dim, cand):
    """exact fp64 L1 distances of the candidate lists cand int32 [nq, c] -> fp64 [nq, c]."""
    nq, c = cand.shape
    out = torch.empty((nq, c), dtype=torch.float64, device=q.device)
    check(lib().oea_pair_l1_f64(_p(q), nq, q.shape[1], _p(table), table.shape[0], table.shape[1], dim, _p(cand), c, _p(out),
                                _stream()))
    return out


def row_topk_mean(s, k):
    out = torch.empty(s.shape[0], dtype=torch.float32, device=s.device)
    check(lib().oea_row_topk_mean(_p(s), s.shape[0], s.shape[1], s.stride(0), k, _p(out), _stream()))
    return out


def csls_means(e1, e2, dim, k):
    """one-sweep CSLS means (oea_csls_means) -> (r [n1], c [n2]) or None when the shape is not covered"""
    n1, n2 = e1.shape[0], e2.shape[0]
    nbytes = lib().oea_csls_means_workspace_bytes(n1, n2, int(k))
    if nbytes == 0 or not _lib.tile_glds():
        return None
    ws = torch.empty(nbytes, dtype=torch.uint8, device=e1.device)
    r = torch.empty(n1, dtype=torch.float32, device=e1.device)
    c = torch.empty(n2, dtype=torch.float32, device=e1.device)
    check(lib().oea_csls_means(_p(e1), n1, e1.shape[1], _p(e2), n2, e2.shape[1], dim, int(k), _p(r), _p(c), _p(ws), nbytes, _stream()))
    return r, c


def csls_apply_(s, r, c):
    check(lib().oea_csls_apply(_p(s), s.shape[0], s.shape[1], s.stride(0), _p(r), _p(c), _stream()))
    return s


# -------------------------------------------------------------------------------------------
# graph builders (csrc/graph_build.hip): triples in, sorted COO out (host numpy: the operands are cut into CSR / chunk
# layouts by the constructors of models/graph_ops.py afterwards)
# -------------------------------------------------------------------------------------------


def _triples_dev(triples, dev=None):
    """list / set iteration order / ndarray [n, 3] -> device int32 [n, 3]"""
    if isinstance(triples, torch.Tensor):
        return triples.to(dtype=torch.int32).contiguous()
    if isinstance(triples, np.ndarray):
        a = np.ascontiguousarray(triples, dtype=np.int32).reshape(-1, 3)
    else:
        a = np.fromiter((x for tr in triples for x in tr), np.int32, count=3 * len(triples)).reshape(-1, 3)
    return torch.from_numpy(a).to(dev or device())


def _coo_out(cap, dev, val_dtype):
    return (torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev),
            torch.empty(cap, dtype=val_dtype, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))


def build_unweighted_adj(triples, n_ent):
    """alinet.py:155-181 -> (row, col, val fp64) numpy, sorted by (row, col)."""
    tri = _triples_dev(triples)
    n = tri.shape[0]
    cap = 2 * n + int(n_ent)
    row, col, val, nnz = _coo_out(cap, tri.device, torch.float64)
    check(lib().oea_build_unweighted_adj(_p(tri), n, int(n_ent), _p(row), _p(col), _p(val), cap, _p(nnz), _stream()))
    m = int(nnz.item())
    return row[:m].cpu().numpy(), col[:m].cpu().numpy(), val[:m].cpu().numpy()


def build_weighted_adj(triples, n_ent, n_rel, raw=False):
    """gcn_align.py:610-664 + 566-578 -> dict(r2f, r2if [n_rel], support = (row, col, val), adj = (row, col, val) if raw)."""
    tri = _triples_dev(triples)
    n = tri.shape[0]
    dev = tri.device
    cap = 2 * n + int(n_ent)
    r2f = torch.zeros(int(n_rel), dtype=torch.float64, device=dev)
    r2if = torch.zeros(int(n_rel), dtype=torch.float64, device=dev)
    row, col, val, nnz = _coo_out(cap, dev, torch.float64)
    a_row, a_col, a_val, a_nnz = _coo_out(cap, dev, torch.float64) if raw else (None, None, None, None)
    check(lib().oea_build_weighted_adj(_p(tri), n, int(n_ent), int(n_rel), _p(r2f), _p(r2if), _p(a_row), _p(a_col), _p(a_val),
                                       _p(a_nnz), _p(row), _p(col), _p(val), cap, _p(nnz), _stream()))
    m = int(nnz.item())
    out = dict(r2f=r2f.cpu().numpy(), r2if=r2if.cpu().numpy(),
               support=(row[:m].cpu().numpy(), col[:m].cpu().numpy(), val[:m].cpu().numpy()))
    if raw:
        ma = int(a_nnz.item())
        out["adj"] = (a_row[:ma].cpu().numpy(), a_col[:ma].cpu().numpy(), a_val[:ma].cpu().numpy())
    return out


def build_primal_adj(triples, n_ent):
    """rdgcn.py:45-72 -> (row, col, val fp32) numpy."""
    tri = _triples_dev(triples)
    n = tri.shape[0]
    cap = 2 * n + int(n_ent)
    row, col, val, nnz = _coo_out(cap, tri.device, torch.float32)
    check(lib().oea_build_primal_adj(_p(tri), n, int(n_ent), _p(row), _p(col), _p(val), cap, _p(nnz), _stream()))
    m = int(nnz.item())
    return row[:m].cpu().numpy(), col[:m].cpu().numpy(), val[:m].cpu().numpy()


def build_dual_adj(triples, n_rel):
    """rdgcn.py:17-42 + 268-277 -> device fp32 [n_rel, n_rel]."""
    tri = _triples_dev(triples)
    out = torch.empty((int(n_rel), int(n_rel)), dtype=torch.float32, device=tri.device)
    check(lib().oea_build_dual_adj(_p(tri), tri.shape[0], int(n_rel), _p(out), _stream()))
    return out


def build_2hop(triples, full_triples, n_ent, n_rel, n_cut=5):
    """alinet.py:250-287 -> (int64 [m, 3] numpy sorted by (h, r, t), stats = the reference's four log counts)."""
    tri = _triples_dev(triples)
    full = tri if full_triples is None else _triples_dev(full_triples)
    n = tri.shape[0]
    stats = (C.c_int64 * 4)()
    n_out = torch.zeros(1, dtype=torch.int64, device=tri.device)
    cap = max(4 * n, 1 << 16)
    while True:
        out = torch.empty((cap, 3), dtype=torch.int32, device=tri.device)
        rc = lib().oea_build_2hop(_p(tri), n, _p(full), full.shape[0], int(n_ent), int(n_rel), int(n_cut), _p(out), cap, _p(n_out),
                                  stats, _stream())
        m = int(n_out.item())
        if rc != 0 and m > cap:            # the count is known now: once more with room for it
            cap = m
            continue
        check(rc)
        return out[:m].cpu().numpy().astype(np.int64), [int(x) for x in stats]


# -------------------------------------------------------------------------------------------
# graph aggregate
# -------------------------------------------------------------------------------------------


def csr_split(indptr, threshold=768, chunk=512, dev=None, row_range=None):
    """host CSR row pointer -> oea_csr_split for rows with more than `threshold` nonzeros (None if there
    are none): each such row is cut into chunks of `chunk` nonzeros.  row_range=(lo, hi): only rows of
    that block, numbered relative to lo (a rank's block of a row-sharded aggregate)."""
    indptr = np.asarray(indptr, np.int64)
    lens = np.diff(indptr)
    lo, hi = row_range if row_range is not None else (0, len(lens))
    rows = np.flatnonzero(lens[lo:hi] > threshold)
    if len(rows) == 0:
        return None
    c_row, c_e0, c_e1, first = [], [], [], []
    for r in rows:
        first.append(len(c_row))
        for e0 in range(int(indptr[lo + r]), int(indptr[lo + r + 1]), chunk):
            c_row.append(r)
            c_e0.append(e0)
            c_e1.append(min(e0 + chunk, int(indptr[lo + r + 1])))
    first.append(len(c_row))
    t = [to_ids(np.asarray(a, np.int32), dev) for a in (c_row, c_e0, c_e1, rows, first)]
    tickets = torch.zeros(len(rows), dtype=torch.int32, device=t[0].device)      # last-arriver tickets of the hub rows (self-resetting)
    sp_ = _lib.CsrSplit(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), len(c_row), len(rows), int(threshold),
                        t[4].data_ptr(), None, 0, tickets.data_ptr())
    sp_._keep = t + [tickets]
    return sp_


def _split_partials(split, ld, dev):
    """grow-only buffer for the hub chunks' partial sums (stream-ordered reuse by the calls on this operand)"""
    need = int(split.n_chunks) * int(ld)
    if split.partials_floats < need:
        buf = torch.empty(need, dtype=torch.float32, device=dev)
        split._partials = buf
        split.partials, split.partials_floats = buf.data_ptr(), need
    return split


def spmm_csr(rowptr, colidx, vals, x, dim, act=0, mask_from=None, out=None, split=None):
    n_rows = rowptr.numel() - 1
    if out is None:
        out = torch.empty((n_rows, x.shape[1]), dtype=torch.float32, device=x.device)
    if split is not None:
        _split_partials(split, out.shape[1], x.device)
    check(lib().oea_spmm_csr(_p(rowptr), _p(colidx), _p(vals), n_rows, _p(x), dim, x.shape[1], int(act),
                             _p(mask_from), _p(out), out.shape[1], C.byref(split) if split is not None else None,
                             _stream()))
    return out


def align_loss_l1(out_emb, dim, ill, k, gamma, neg_left, neg_right, neg2_left, neg2_right, grad, loss_accum):
    check(lib().oea_align_loss_l1(_p(out_emb), out_emb.shape[0], dim, out_emb.shape[1], _p(ill), ill.shape[0],
                                  k, float(gamma), _p(neg_left), _p(neg_right), _p(neg2_left), _p(neg2_right),
                                  _p(grad), _p(loss_accum), _stream()))


def sgd_rows_(w, grad_t, dim, normalize, lr):
    check(lib().oea_sgd_rows(_p(w), _p(grad_t), w.shape[0], dim, w.shape[1], int(bool(normalize)), float(lr), _stream()))
    return w


# -------------------------------------------------------------------------------------------
# sparse graph attention + dense Adam
# -------------------------------------------------------------------------------------------


ATTN_ALPHA, ATTN_AGGREGATE, ATTN_DZ, ATTN_DV = 1, 2, 1, 2


def attn_graph(sub_ptr, sub_seg, seg_sub_ptr, seg_row, colidx, agg_rowptr, agg_colidx, agg_edge, t_rowptr, t_row, t_edge,
               agg_split=None, t_split=None, seg_range=None, sub_range=None, agg_rows=None, agg_slots=None, t_rows=None,
               t_slots=None):
    """pack the device arrays of an attention graph (keeps them alive).  agg_edge None: the edge order is the
    aggregate's CSR order.  The *_range / *_rows / *_slots arguments restrict the ranges a call works on (a rank's blocks
    of a row-sharded job); default: the whole graph."""
    n_sub, n_seg = sub_seg.numel(), seg_row.numel()
    n_agg, n_t, nnz = agg_rowptr.numel() - 1, t_rowptr.numel() - 1, colidx.numel()
    s0, s1 = sub_range if sub_range is not None else (0, n_sub)
    g0, g1 = seg_range if seg_range is not None else (0, n_seg)
    a0, a1 = agg_rows if agg_rows is not None else (0, n_agg)
    as0, as1 = agg_slots if agg_slots is not None else (0, nnz)
    t0, t1 = t_rows if t_rows is not None else (0, n_t)
    ts0, ts1 = t_slots if t_slots is not None else (0, nnz)
    g = _lib.AttnGraph(sub_ptr.data_ptr(), sub_seg.data_ptr(), seg_sub_ptr.data_ptr(), seg_row.data_ptr(), colidx.data_ptr(),
                       n_sub, n_seg, nnz, agg_rowptr.data_ptr(), agg_colidx.data_ptr(),
                       agg_edge.data_ptr() if agg_edge is not None else None, n_agg,
                       C.pointer(agg_split) if agg_split is not None else None,
                       t_rowptr.data_ptr(), t_row.data_ptr(), t_edge.data_ptr(), n_t,
                       C.pointer(t_split) if t_split is not None else None,
                       int(s0), int(s1), int(g0), int(g1), int(a0), int(a1), int(as0), int(as1), int(t0), int(t1), int(ts0), int(ts1))
    g._keep = (sub_ptr, sub_seg, seg_sub_ptr, seg_row, colidx, agg_rowptr, agg_colidx, agg_edge, t_rowptr, t_row, t_edge,
               agg_split, t_split)
    g._splits = (agg_split, t_split)
    return g


def _attn_ws(g, ld, dev):
    for sp_ in g._splits:
        if sp_ is not None:
            _split_partials(sp_, ld, dev)
    return torch.empty(lib().oea_sparse_attn_workspace_floats(C.byref(g)), dtype=torch.float32, device=dev)


def sparse_attn_fwd(g, z, v, dim, slope, n_rows, out=None, alpha=None, phases=ATTN_ALPHA | ATTN_AGGREGATE):
    """-> (out [n_rows, ld], alpha [nnz]); every row / edge of the graph's ranges is written (no zero-fill needed for a
    whole-graph call; a sharded caller passes its own out / alpha)."""
    if out is None:
        out = torch.empty((n_rows, v.shape[1]), dtype=torch.float32, device=v.device)
    if alpha is None:
        alpha = torch.empty_like(z)
    ws = _attn_ws(g, v.shape[1], v.device)
    check(lib().oea_sparse_attn_fwd(C.byref(g), _p(z), _p(v), dim, v.shape[1], float(slope), _p(out), _p(alpha), _p(ws),
                                    int(phases), _stream()))
    return out, alpha


def sparse_attn_bwd(g, z, v, alpha, dout, dim, slope, dz=None, dv=None, phases=ATTN_DZ | ATTN_DV):
    """-> (dz [nnz], dv [n_cols, ld])."""
    if dz is None:
        dz = torch.empty_like(z)
    if dv is None:
        dv = torch.empty_like(v)
    ws = _attn_ws(g, v.shape[1], v.device)
    check(lib().oea_sparse_attn_bwd(C.byref(g), _p(z), _p(v), _p(alpha), _p(dout), dim, v.shape[1], float(slope), _p(dz),
                                    _p(dv), _p(ws), int(phases), _stream()))
    return dz, dv


def sparse_attn_dz_(g, z, alpha, dalpha, slope, ld):
    """d alpha per edge (overwritten) -> d z per edge, over the graph's softmax groups (oea_sparse_attn_dz)"""
    ws = _attn_ws(g, ld, z.device)
    check(lib().oea_sparse_attn_dz(C.byref(g), _p(z), _p(alpha), _p(dalpha), float(slope), _p(ws), _stream()))
    return dalpha


def adam_dense_(param, grad, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    check(lib().oea_adam_dense(_p(param), _p(grad), _p(m), _p(v), param.numel(), float(lr), float(beta1), float(beta2),
                               float(eps), int(t), _stream()))
    return param


# -------------------------------------------------------------------------------------------
# fused row-wise glue of the GNN approaches (csrc/gnn_fused.hip)
# -------------------------------------------------------------------------------------------


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def _i32_array(vals):
    return (C.c_int32 * len(vals))(*[int(v) for v in vals])


def concat_l2n_fwd(xs):
    """xs: list of <= 4 contiguous [n, d_i] fp32 tensors -> (emb [n, pad4(sum d)], inv_blk [n, 4], inv_all [n])."""
    n = xs[0].shape[0]
    dims = [x.shape[1] for x in xs]
    out = torch.empty((n, pad4(sum(dims))), dtype=torch.float32, device=xs[0].device)
    inv_blk = torch.empty((n, 4), dtype=torch.float32, device=out.device)
    inv_all = torch.empty(n, dtype=torch.float32, device=out.device)
    check(lib().oea_concat_l2n_fwd(_ptr_array(xs), _i32_array(dims), _i32_array(dims), len(xs), n, _p(out), out.shape[1],
                                   _p(inv_blk), _p(inv_all), _stream()))
    return out, inv_blk, inv_all


def concat_l2n_bwd(dims, z, dz, inv_blk, inv_all):
    n = z.shape[0]
    dxs = [torch.empty((n, d), dtype=torch.float32, device=z.device) for d in dims]
    check(lib().oea_concat_l2n_bwd(_ptr_array(dxs), _i32_array(dims), _i32_array(dims), len(dims), n, _p(z), _p(dz), z.shape[1],
                                   _p(inv_blk), _p(inv_all), _stream()))
    return dxs


def pair_loss_l2_fwd(emb, dim, pairs, n_pos, weight, margin, balance):
    """pairs int32 [m, 2] (the first n_pos positive) -> (terms [m], coef [m])."""
    m = pairs.shape[0]
    coef = torch.empty(m, dtype=torch.float32, device=emb.device)
    terms = torch.empty(m, dtype=torch.float32, device=emb.device)
    check(lib().oea_pair_loss_l2_fwd(_p(emb), emb.shape[0], dim, emb.shape[1], _p(pairs), m, int(n_pos), _p(weight), float(margin),
                                     float(balance), _p(coef), _p(terms), _stream()))
    return terms, coef


def pair_rows_csr(pairs, n_rows):
    """the endpoints of `pairs` (device int [m, 2]) grouped by row, slots in pair order (oea_pair_rows_build: one stable
    device sort + a scan, no host synchronisation) -> (rowptr int32 [n_rows + 1], other int32 [2 m], slot_pair int32 [2 m])
    for pair_grad_rows."""
    pairs = pairs.to(torch.int32).contiguous()
    m = pairs.shape[0]
    rowptr = torch.empty(n_rows + 1, dtype=torch.int32, device=pairs.device)
    other = torch.empty(2 * m, dtype=torch.int32, device=pairs.device)
    slot_pair = torch.empty(2 * m, dtype=torch.int32, device=pairs.device)
    check(lib().oea_pair_rows_build(_p(pairs), m, int(n_rows), _p(rowptr), _p(other), _p(slot_pair), _stream()))
    return rowptr, other, slot_pair


def pair_grad_rows(emb, dim, rowptr, other, slot_pair, coef, gscale=None, norm=2, out=None):
    """grad[r] = gscale * sum over row r's pair slots of 2 coef (e_r - e_other) (norm 2) | coef sign(e_r - e_other) (norm 1)."""
    grad = torch.empty_like(emb) if out is None else out
    check(lib().oea_pair_grad_rows(_p(emb), emb.shape[0], dim, emb.shape[1], _p(rowptr), _p(other), _p(slot_pair), _p(coef),
                                   _p(gscale), int(norm), _p(grad), _stream()))
    return grad


def align_loss_l1_coef(out_emb, dim, ill, k, gamma, neg_left, neg_right, neg2_left, neg2_right, loss_accum, coef=None):
    """the L1 hinge without its gradient -> coef [t + 2 t k] (signed pair coefficients for pair_grad_rows(norm=1))."""
    t = ill.shape[0]
    if coef is None:
        coef = torch.empty(t + 2 * t * k, dtype=torch.float32, device=out_emb.device)
    check(lib().oea_align_loss_l1_coef(_p(out_emb), out_emb.shape[0], dim, out_emb.shape[1], _p(ill), t, k, float(gamma),
                                       _p(neg_left), _p(neg_right), _p(neg2_left), _p(neg2_right), None, _p(loss_accum), _p(coef),
                                       _stream()))
    return coef


def highway_fwd(a, b, p, gamma, beta):
    out = torch.empty_like(a)
    check(lib().oea_highway_fwd(_p(a), _p(b), _p(p), _p(gamma), _p(beta), a.shape[0], a.shape[1], _p(out), _stream()))
    return out


def highway_bwd(a, b, p, gamma, beta, out, gout):
    n, d = a.shape
    da, db, dp = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    parts = torch.empty((lib().oea_colsum_blocks(n), 2, d), dtype=torch.float32, device=a.device)
    check(lib().oea_highway_bwd(_p(a), _p(b), _p(p), _p(gamma), _p(beta), _p(out), _p(gout), n, d, _p(da), _p(db), _p(dp), _p(parts),
                                _stream()))
    sums = parts.sum(0)
    return da, db, dp, sums[0], sums[1]


def gemm_tn(a, b):
    """a^T @ b for two tall row-major fp32 matrices with the same number of rows -> [a.shape[1], b.shape[1]]
    (oea_gemm_tn_f32; shapes it does not take -- widths not multiples of 4 -- go to the library)."""
    m, k1 = a.shape
    k2 = b.shape[1]
    if k1 % 4 or k2 % 4 or not (a.is_contiguous() and b.is_contiguous()) or a.data_ptr() % 16 or b.data_ptr() % 16:
        return a.t() @ b
    out = torch.empty((k1, k2), dtype=torch.float32, device=a.device)
    n_ws = lib().oea_gemm_tn_workspace_floats(m, k1, k2)
    ws = torch.empty(n_ws, dtype=torch.float32, device=a.device) if n_ws else None
    check(lib().oea_gemm_tn_f32(_p(a), k1, k1, _p(b), k2, k2, m, _p(out), k2, _p(ws), _stream()))
    return out


def gemm_tn_sharded(a, b, rank, world, allgather_blocks):
    """a^T @ b in a job whose ranks share the work by ROW CHUNKS (include/openea_hip.h: oea_gemm_tn_partial / _reduce): this
    rank's chunks, one all-gather of the chunk partials (allgather_blocks(full [chunks, k1*k2], bounds)), then all chunks
    added in chunk order -- bit-identical with gemm_tn on one process.  Falls back to the replicated product when the shape
    does not take the kernel or has fewer chunks than ranks."""
    m, k1 = a.shape
    k2 = b.shape[1]
    if (world <= 1 or m == 0 or k1 % 4 or k2 % 4 or not (a.is_contiguous() and b.is_contiguous()) or a.data_ptr() % 16
            or b.data_ptr() % 16):
        return gemm_tn(a, b)
    chunks, rpc = C.c_int32(0), C.c_int64(0)
    check(lib().oea_gemm_tn_plan(m, k1, k2, C.byref(chunks), C.byref(rpc)))
    chunks = chunks.value
    if chunks < world:
        return gemm_tn(a, b)
    bounds = [chunks * r // world for r in range(world + 1)]
    ws = torch.empty((chunks, k1 * k2), dtype=torch.float32, device=a.device)
    check(lib().oea_gemm_tn_partial(_p(a), k1, k1, _p(b), k2, k2, m, bounds[rank], bounds[rank + 1], _p(ws), _stream()))
    allgather_blocks(ws, bounds)
    out = torch.empty((k1, k2), dtype=torch.float32, device=a.device)
    check(lib().oea_gemm_tn_reduce(_p(ws), chunks, k1, k2, _p(out), k2, _stream()))
    return out


def sigmoid_mix_fwd(a, b, p, bias):
    out = torch.empty_like(a)
    check(lib().oea_sigmoid_mix_fwd(_p(a), _p(b), _p(p), _p(bias), a.shape[0], a.shape[1], _p(out), _stream()))
    return out


def sigmoid_mix_bwd(a, b, p, bias, gout, b_relu=False):
    """-> da, db, dp, dbias (db gated by b > 0 when b is a relu's output)"""
    n, d = a.shape
    da, db, dp = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    parts = torch.empty((lib().oea_colsum_blocks(n), d), dtype=torch.float32, device=a.device)
    check(lib().oea_sigmoid_mix_bwd(_p(a), _p(b), _p(p), _p(bias), _p(gout), n, d, int(bool(b_relu)), _p(da), _p(db), _p(dp),
                                    _p(parts), _stream()))
    return da, db, dp, parts.sum(0)


def relu_axpy_fwd(x, y, alpha):
    out = torch.empty_like(x)
    check(lib().oea_relu_axpy_fwd(_p(x), _p(y), float(alpha), x.numel(), _p(out), _stream()))
    return out


def relu_axpy_bwd(y, gout, alpha):
    dy = torch.empty_like(y)
    check(lib().oea_relu_axpy_bwd(_p(y), _p(gout), float(alpha), y.numel(), _p(dy), _stream()))
    return dy


def colsum_prod(x, y):
    """column sums of x * y -> fp32 [d] (fixed order: row blocks, then blocks)"""
    n, d = x.shape
    parts = torch.empty((lib().oea_colsum_blocks(n), d), dtype=torch.float32, device=x.device)
    check(lib().oea_colsum_prod(_p(x), _p(y), n, d, _p(parts), _stream()))
    return parts.sum(0)


def bias_tanh_fwd(x, bias):
    y = torch.empty_like(x)
    check(lib().oea_bias_tanh_fwd(_p(x), _p(bias), x.shape[0], x.shape[1], _p(y), _stream()))
    return y


def bias_tanh_bwd(y, gy):
    n, d = y.shape
    gx = torch.empty_like(y)
    parts = torch.empty((lib().oea_colsum_blocks(n), d), dtype=torch.float32, device=y.device)
    check(lib().oea_bias_tanh_bwd(_p(y), _p(gy), n, d, _p(gx), _p(parts), _stream()))
    return gx, parts.sum(0)


def segment_sum(vals, order, seg_ptr):
    """out[s] = sum of vals[order[e]] over e in [seg_ptr[s], seg_ptr[s + 1]) (order may be None) -> fp32 [n_seg]."""
    n_seg = seg_ptr.numel() - 1
    out = torch.empty(n_seg, dtype=torch.float32, device=vals.device)
    check(lib().oea_segment_sum_f32(_p(vals), _p(order), _p(seg_ptr), n_seg, _p(out), _stream()))
    return out
