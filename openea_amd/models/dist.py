"""Multi-GPU layer of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on MI355X; "gloo" in the CPU tests).  The reference has no distributed code at all.

What shards and how:

* alignment evaluation and neighbour search -- independent QUERY rows: rank r owns the contiguous
  row block [lo_r, hi_r) of the queries, the candidate block is replicated (it is an embedding
  lookup of a replicated table), no data-path collective; the integer metrics are summed with
  one small all-reduce, per-row outputs (argmax / neighbour ids) are all-gathered.
* translational step -- entity rows are OWNED cyclically by id (owner = id mod G); every rank keeps a read copy
  of the table, scores its slice of the batch, the packed gradients are reduce-scattered to their owners,
  the owners run the optimiser on their rows (1/G of the state and of the work) and the updated rows are
  all-gathered (models/trainer.py: TripleTrainer._step_partitioned).  The small replicated steps (MTransE's
  mapping step, BootEA's alignment step) and the projected scores keep one dense all-reduce.

Everything here is index arithmetic + collectives on whatever device the tensors live on, so it
is exercised on CPU with gloo (tests/test_dist_cpu.py); the compute callbacks default to the HIP
kernels.
"""
import numpy as np
import torch
import torch.distributed as dist


def world(group=None):
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _staged(t, group):
    """gloo (CPU tests, and the N-processes-on-one-GPU wiring tests) has no device collectives: stage on the host"""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def shard_range(n, rank, world_size):
    """contiguous, balanced block [lo, hi) of n rows owned by `rank` (sizes differ by at most 1)."""
    return n * rank // world_size, n * (rank + 1) // world_size


def shard_batch(n, n_split, rank, world_size):
    """this rank's share of a (pos_batch1 + pos_batch2) batch of n rows whose first n_split rows are
    KG1's: -> (lo, hi, local_split).  `lo` doubles as the Philox positive offset, so the union of all
    ranks' negatives equals the single-GPU draw for the same batch."""
    lo, hi = shard_range(n, rank, world_size)
    return lo, hi, min(max(n_split - lo, 0), hi - lo)


def shard_sizes(n, world_size):
    return [shard_range(n, r, world_size)[1] - shard_range(n, r, world_size)[0] for r in range(world_size)]


def allgather_rows(local, n_total, group=None):
    """local: this rank's row block [hi-lo, ...] -> the full [n_total, ...] tensor on every rank.
    ONE all_gather_into_tensor of blocks padded to the largest block (no per-rank list, no torch.cat)."""
    rank, ws = world(group)
    if ws == 1:
        return local
    sizes = shard_sizes(n_total, ws)
    m = max(sizes)
    stage = _staged(local, group)
    dev = "cpu" if stage else local.device
    if local.shape[0] == m and not stage:
        pad = local.contiguous()
    else:
        pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
        pad[: local.shape[0]] = local
    out = torch.empty((ws * m,) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)     # concatenation along dim 0
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.view((ws, m) + tuple(local.shape[1:]))
    if all(sz == m for sz in sizes):
        return out.reshape((ws * m,) + tuple(local.shape[1:])).to(local.device)
    return torch.cat([out[r, :sz] for r, sz in enumerate(sizes)], dim=0).to(local.device)


def balanced_bounds(indptr, world_size):
    """row-block boundaries [b_0=0, ..., b_W=n] of a CSR matrix with (almost) equal NONZEROS per block:
    ids are frequency-ordered (read.py:64-79), so equal row counts would put every hub on rank 0."""
    indptr = np.asarray(indptr, np.int64)
    n, nnz = len(indptr) - 1, int(indptr[-1])
    b = [0]
    for r in range(1, world_size):
        # first row whose prefix reaches r/W of the nonzeros (+ rows, so that empty rows spread too)
        target = (nnz + n) * r / world_size
        w = indptr[: n + 1] + np.arange(n + 1)
        b.append(int(min(max(np.searchsorted(w, target), b[-1]), n)))
    b.append(n)
    return b


def allgather_blocks(full, bounds, group=None):
    """`full` [n, ...]: rank r has written rows [bounds[r], bounds[r+1]); afterwards every rank has every
    block (in place; ONE all_gather_into_tensor of blocks padded to the largest)."""
    rank, ws = world(group)
    if ws == 1:
        return full
    sizes = [bounds[r + 1] - bounds[r] for r in range(ws)]
    m = max(sizes)
    stage = _staged(full, group)
    dev = "cpu" if stage else full.device
    pad = torch.zeros((m,) + tuple(full.shape[1:]), dtype=full.dtype, device=dev)
    pad[: sizes[rank]] = full[bounds[rank]: bounds[rank + 1]]
    out = torch.empty((ws * m,) + tuple(full.shape[1:]), dtype=full.dtype, device=dev)
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.view((ws, m) + tuple(full.shape[1:]))
    for r in range(ws):
        if r != rank and sizes[r]:
            full[bounds[r]: bounds[r + 1]] = out[r, : sizes[r]].to(full.device)
    return full


def _collective_device(backend):
    """where a tensor handed to the group's collectives lives ("nccl" = RCCL moves device memory)"""
    return "cuda" if backend == "nccl" else "cpu"


class CAbiComm:
    """The C ABI's communicator (include/openea_hip.h: oea_comm_*) over the ranks of `group`, for the one-call partitioned
    epoch (oea_triple_epoch_range_comm).

    * backend "nccl" (one GPU per rank -- a multi-GPU node): RCCL through the library's own dlopen, rank 0's 128-byte id
      travels through torch.distributed.
    * any other backend (gloo: the CPU group, and N ranks sharing ONE GPU in the build pool's tests), or
      OEA_COMM_CALLBACKS=1: oea_comm_init_callbacks -- every collective of the C call comes back to `_collective`, which
      stages the device buffer through the host (oea_copy_to_host / _from_host, ordered on the call's stream) and runs the
      torch.distributed collective on the host copy.  Same call path, same protocol, slower wire."""

    def __init__(self, group=None, device_buffers=()):
        """device_buffers: the tensors the C call will hand to the collectives (the partition's send / own / rel_x / upd /
        all buffers, the TransH normal views): with a device-capable backend the callback back end runs the group's
        collective directly on them (no staging) -- the fallback when the library's own RCCL communicator cannot be made."""
        import ctypes as C
        import os
        import sys
        from .. import _lib, ops
        from .._lib import check
        self.group = group
        self.rank, self.world = world(group)
        self.lib = lib = ops.lib()
        self.handle = C.c_void_p()
        self._bufs = {int(t.data_ptr()): t for t in device_buffers if t is not None}
        backend = dist.get_backend(group) if self.world > 1 else "nccl"
        self.device_collectives = self.world > 1 and backend == "nccl"
        self.callbacks = (self.world > 1 and backend != "nccl") or os.environ.get("OEA_COMM_CALLBACKS") == "1"
        if not self.callbacks:
            # every rank must take the same branch: the outcome of the RCCL set-up is agreed on through the group
            # (every rank runs the SAME sequence of group operations whatever fails locally -- rank 0 broadcasts its id
            # or None, every rank treats None as "no RCCL", oea_comm_init runs only with a valid id, the MIN all-reduce decides)
            ok = 1
            box = [None]
            if self.rank == 0:
                try:
                    uid = (C.c_char * 128)()
                    check(lib.oea_comm_unique_id(uid))
                    box = [bytes(uid.raw)]
                except Exception as e:        # noqa: BLE001 -- e.g. librccl not loadable: the others must still get an answer
                    self._note_no_rccl(e)
            if self.world > 1:
                src = dist.get_global_rank(group, 0) if group is not None and group is not dist.group.WORLD else 0
                dist.broadcast_object_list(box, src=src, group=group)
            if box[0] is None:
                ok = 0
            else:
                try:
                    buf = (C.c_char * 128).from_buffer_copy(box[0])
                    check(lib.oea_comm_init(buf, self.rank, self.world, C.byref(self.handle)))
                except Exception as e:        # noqa: BLE001 -- e.g. ranks sharing a device
                    ok = 0
                    self._note_no_rccl(e)
            if self.world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device=_collective_device(backend))
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                ok_all = int(flag.item())
            else:
                ok_all = ok
            if ok_all:
                return
            if ok:                            # made here but not everywhere: drop it
                lib.oea_comm_destroy(self.handle)
                self.handle = C.c_void_p()
            self.callbacks = True
        self._fn = _lib.COMM_CALLBACK(self._collective)            # kept alive with the object
        check(lib.oea_comm_init_callbacks(self.rank, self.world, C.cast(self._fn, C.c_void_p), None, C.byref(self.handle)))
        self._fn_a2a = _lib.COMM_ALLTOALLV_CALLBACK(self._alltoallv)
        check(lib.oea_comm_set_alltoallv(self.handle, C.cast(self._fn_a2a, C.c_void_p)))

    @staticmethod
    def _note_no_rccl(e):
        import sys
        print("[openea_amd] the C ABI's RCCL communicator could not be made (%s): collectives of the one-call epoch go "
              "through torch.distributed callbacks" % str(e)[:200], file=sys.stderr)

    def close(self):
        """oea_comm_destroy (the RCCL communicator and its staging buffers) after the work enqueued on the device has drained;
        idempotent.  The EXPLICIT way out (TripleTrainer.close()): ncclCommDestroy with work in flight, or after the runtime /
        process group is gone, can hang or crash where no try/except reaches."""
        h, self.handle = getattr(self, "handle", None), None
        if h is not None and getattr(h, "value", None):
            try:
                import torch
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                self.lib.oea_comm_destroy(h)
            except Exception:            # noqa: BLE001
                pass

    def __del__(self):
        # garbage collection is not a safe place for ncclCommDestroy: at interpreter shutdown the HIP / RCCL runtime or the process
        # group may already be torn down.  Late collection leaks the communicator (the process is ending); close() destroys it.
        import sys
        if sys is None or sys.is_finalizing():
            self.handle = None
            return
        try:
            import torch.distributed as dist
            if not dist.is_initialized():
                self.handle = None
                return
        except Exception:                # noqa: BLE001
            self.handle = None
            return
        self.close()

    _NP = {0: np.float32, 1: np.float64, 2: np.int64}

    def _collective(self, user, op, send, recv, count, dtype, stream):
        """oea_comm_callback: returns 0 on success (an exception cannot cross the C frame: it is printed and reported as 1)"""
        try:
            from .._lib import COMM_ALLGATHER, COMM_ALLREDUCE, check
            if self.device_collectives and int(send) in self._bufs and int(recv) in self._bufs:
                # a device-capable group (nccl): the collective on the registered device tensors, ordered on torch's current
                # stream (the stream the C call enqueues on)
                n = int(count)
                src, dst = self._bufs[int(send)].view(-1), self._bufs[int(recv)].view(-1)
                if op == COMM_ALLREDUCE:
                    dist.all_reduce(dst[:n], op=dist.ReduceOp.SUM, group=self.group)
                elif op == COMM_ALLGATHER:
                    dist.all_gather_into_tensor(dst[: n * self.world], src[:n], group=self.group)
                else:
                    dist.reduce_scatter_tensor(dst[:n], src[: n * self.world], op=dist.ReduceOp.SUM, group=self.group)
                return 0
            npdt = self._NP[int(dtype)]
            n_in = int(count) * (self.world if op != COMM_ALLGATHER and op != COMM_ALLREDUCE else 1)
            host = np.empty(n_in, npdt)
            check(self.lib.oea_copy_to_host(send, host.ctypes.data, host.nbytes, stream))
            t = torch.from_numpy(host)
            if self.world == 1:
                out = t
            elif op == COMM_ALLREDUCE:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                out = t
            elif op == COMM_ALLGATHER:
                out = torch.empty(int(count) * self.world, dtype=t.dtype)
                dist.all_gather_into_tensor(out, t, group=self.group)
            else:
                out = torch.empty(int(count), dtype=t.dtype)
                dist.reduce_scatter_tensor(out, t, op=dist.ReduceOp.SUM, group=self.group)
            o = out.numpy()
            check(self.lib.oea_copy_from_host(recv, o.ctypes.data, o.nbytes, stream))
            return 0
        except Exception:            # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    def _alltoallv(self, user, send, send_counts, send_displs, recv, recv_counts, recv_displs, dtype, stream):
        """oea_comm_alltoallv_callback over torch.distributed point-to-point messages, staged through the host (gloo; N ranks on
        one GPU): peer p gets send[send_displs[p] : + send_counts[p]] (elements)"""
        try:
            from .._lib import check
            npdt = np.dtype(self._NP[int(dtype)])
            w, me = self.world, self.rank
            sc = [int(send_counts[p]) for p in range(w)]
            sd = [int(send_displs[p]) for p in range(w)]
            rc = [int(recv_counts[p]) for p in range(w)]
            rd = [int(recv_displs[p]) for p in range(w)]
            outs, ins, reqs = {}, {}, []
            for p in range(w):
                if sc[p] > 0:
                    h = np.empty(sc[p], npdt)
                    check(self.lib.oea_copy_to_host(int(send) + sd[p] * npdt.itemsize, h.ctypes.data, h.nbytes, stream))
                    outs[p] = torch.from_numpy(h)
                if rc[p] > 0:
                    ins[p] = torch.empty(rc[p], dtype=torch.from_numpy(np.empty(0, npdt)).dtype)
            for p in range(w):               # post every receive, then the sends (any order completes)
                if p != me and p in ins:
                    reqs.append(dist.irecv(ins[p], src=self._global(p), group=self.group))
            for p in range(w):
                if p != me and p in outs:
                    reqs.append(dist.isend(outs[p], dst=self._global(p), group=self.group))
            if me in ins:
                ins[me].copy_(outs[me])
            for r in reqs:
                r.wait()
            for p, t in ins.items():
                o = t.numpy()
                check(self.lib.oea_copy_from_host(int(recv) + rd[p] * npdt.itemsize, o.ctypes.data, o.nbytes, stream))
            return 0
        except Exception:            # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    def _global(self, p):
        return dist.get_global_rank(self.group, p) if self.group is not None and self.group is not dist.group.WORLD else p

    def profile_begin(self):
        from .._lib import check
        check(self.lib.oea_comm_profile_begin(self.handle))

    def profile_end(self):
        """-> ({phase: ms summed over the recorded steps}, steps)"""
        import ctypes as C
        from .._lib import COMM_PHASES, check
        ms = (C.c_double * len(COMM_PHASES))()
        n = C.c_int32(0)
        check(self.lib.oea_comm_profile_end(self.handle, ms, C.byref(n)))
        return {k: float(v) for k, v in zip(COMM_PHASES, ms)}, int(n.value)

    def description(self):
        return ("host callbacks over torch.distributed '%s' (staged through the host)" % dist.get_backend(self.group)
                if self.callbacks else "RCCL (the C ABI's own communicator, librccl through dlopen)")


def c_abi_comm(group=None, device_buffers=()):
    return CAbiComm(group, device_buffers)


def allreduce_sum_(t, group=None):
    _, ws = world(group)
    if ws > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def reduce_scatter_(out, inp, group=None):
    """out [chunk] = sum over ranks of inp[rank * chunk : (rank + 1) * chunk]  (one RCCL reduce-scatter)"""
    rank, ws = world(group)
    if ws == 1:
        out.copy_(inp)
        return out
    if _staged(inp, group):
        o, i = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.reduce_scatter_tensor(o, i, op=dist.ReduceOp.SUM, group=group)
        out.copy_(o)
    else:
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group)
    return out


def all_gather_into_(out, inp, group=None):
    """out [G, ...] <- every rank's inp [...]  (one RCCL all-gather, no list / cat copies)"""
    rank, ws = world(group)
    if ws == 1:
        out.view(inp.shape).copy_(inp)
        return out
    flat = (out.shape[0] * out.shape[1],) + tuple(out.shape[2:])                 # [G, rows, ...] viewed as the dim-0 concatenation
    if _staged(inp, group):
        o = torch.empty(flat, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu().contiguous(), group=group)
        out.view(flat).copy_(o)
    else:
        dist.all_gather_into_tensor(out.view(flat), inp.contiguous(), group=group)
    return out


def sync_replicated_(t, group=None):
    """A quantity every rank computes in FULL from replicated inputs (the losses / dense layers of the GNN approaches)
    still differs between ranks in the last bits -- fp32 atomics sum in another order in every process -- and an
    optimiser would let the replicas drift apart.  Averaging it across the ranks (one all-reduce) gives every rank
    the same bits again.  No-op on a single rank."""
    _, ws = world(group)
    if ws > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t /= ws
    return t


def sharded_rank_metrics(e1, e2, dim, top_k, rank_fn, group=None):
    """Row-sharded greedy_alignment core.  e1 [n1, ld] / e2 [n2, ld]: full (replicated) query and
    candidate blocks.  rank_fn(e1_block, e2, dim, gold_offset) -> (rank int32 [m], argmax int32 [m])
    for the query block whose gold columns start at gold_offset.
    Returns (hits counts list[int], rank_sum int, rr_sum float, argmax int32 [n1]) -- identical on
    every rank and identical to the unsharded result (integers exactly; rr_sum to fp64 roundoff)."""
    rank, ws = world(group)
    n1 = e1.shape[0]
    lo, hi = shard_range(n1, rank, ws)
    rk, am = rank_fn(e1[lo:hi], e2, dim, lo)
    rk64 = rk.to(torch.int64)
    nk = len(top_k)
    ints = torch.zeros(nk + 1, dtype=torch.int64, device=rk.device)
    for i, k in enumerate(top_k):
        ints[i] = (rk64 < k).sum()
    ints[nk] = (rk64 + 1).sum()
    rr = (1.0 / (rk64 + 1).to(torch.float64)).sum().reshape(1)
    allreduce_sum_(ints, group)
    allreduce_sum_(rr, group)
    argmax = allgather_rows(am, n1, group)
    host = ints.cpu().numpy()
    return [int(x) for x in host[:nk]], int(host[nk]), float(rr.item()), argmax


def sharded_neighbours(embeds, dim, entity_ids, k, topk_fn, group=None):
    """Row-sharded truncated-neighbour refresh: rank r searches the neighbours of entity rows
    [lo, hi) among ALL rows; the [N, k] table is all-gathered so every rank's sampler has it.
    topk_fn(q_block, candidates, dim, k, id_map) -> int32 [m, k]."""
    rank, ws = world(group)
    n = embeds.shape[0]
    lo, hi = shard_range(n, rank, ws)
    local = topk_fn(embeds[lo:hi], embeds, dim, k, entity_ids)
    return allgather_rows(local, n, group)
