"""Graph operators for the GNN approaches: the HIP aggregate / attention kernels wrapped as
``torch.autograd.Function`` so that the dense parts of AliNet / RDGCN (plain library GEMMs,
activations) can be chained with them.  torch is the tape here, not the arithmetic of the sparse
ops: every sparse forward / backward below is a call into libopenea_hip.so.
"""
import numpy as np
import scipy.sparse as sp
import torch

from .. import ops


def split_ranges(ptr, max_len, drop_empty=False):
    """cut every range [ptr[i], ptr[i+1]) into consecutive pieces of at most max_len edges (empty
    ranges keep one empty piece unless drop_empty).  The pieces tile [ptr[0], ptr[-1]) in order.
    -> (piece_ptr [n_pieces+1], piece_owner [n_pieces], owner_piece_ptr [n+1])."""
    ptr = np.asarray(ptr, np.int64)
    lens = np.diff(ptr)
    n_pieces = np.maximum((lens + max_len - 1) // max_len, 0 if drop_empty else 1)
    owner = np.repeat(np.arange(len(lens)), n_pieces)
    owner_ptr = np.concatenate([[0], np.cumsum(n_pieces)]).astype(np.int64)
    within = np.arange(len(owner)) - owner_ptr[owner]
    start = ptr[owner] + within * max_len
    piece_ptr = np.concatenate([start, ptr[-1:]]).astype(np.int64)
    return piece_ptr, owner.astype(np.int64), owner_ptr


class CsrOperand:
    """One sparse matrix as a device CSR operand of the aggregate  y = act(A . x).
    Under torch.distributed (world > 1) the aggregate is ROW-SHARDED: every rank computes the block of
    output rows it owns (blocks balanced by nonzeros) from the full, replicated x and the blocks are
    all-gathered -- one exchange per layer; the transposed operand does the same for the
    backward, so no reduce-scatter is needed."""

    def __init__(self, a, dev):
        from . import dist as mdist
        a = sp.csr_matrix(a, dtype=np.float32)
        a.sum_duplicates()
        a.sort_indices()
        self.shape, self.nnz = a.shape, a.nnz
        self.indptr_host = a.indptr
        self.rowptr, self.colidx, self.vals = ops.to_ids(a.indptr, dev), ops.to_ids(a.indices, dev), ops.to_vec(a.data, dev)
        self.rank, self.world = mdist.world()
        self.bounds = mdist.balanced_bounds(a.indptr, self.world)
        self.lo, self.hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.split = ops.csr_split(a.indptr, dev=dev, row_range=(self.lo, self.hi) if self.world > 1 else None)

    def apply(self, x, dim, act=0, mask_from=None):
        if self.world == 1:
            return ops.spmm_csr(self.rowptr, self.colidx, self.vals, x, dim, act=act, mask_from=mask_from, split=self.split)
        from . import dist as mdist
        lo, hi = self.lo, self.hi
        out = torch.empty((self.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
        if hi > lo:
            ops.spmm_csr(self.rowptr[lo: hi + 1], self.colidx, self.vals, x, dim, act=act,
                         mask_from=None if mask_from is None else mask_from[lo:hi], out=out[lo:hi], split=self.split)
        return mdist.allgather_blocks(out, self.bounds)


class EdgeGraph:
    """A sparse [n_rows, n_cols] operator given as an ordered edge list (rows, cols, vals), kept in
    the order the reference would feed it to TF, plus the derived device structures:

    * CSR / transposed CSR (row-sorted) for the plain aggregate (tf.sparse_tensor_dense_matmul);
    * softmax segments over the ORDERED edge list: grouping='row' -> one segment per row (the
      mathematically intended tf.sparse_softmax), grouping='runs' -> maximal runs of consecutive
      edges with equal row (what TF1's CPU kernel does on non-canonical order);
    * the transposed edge list (per column: output row + edge id) for the attention backward.
    """

    SUB = 256          # edges per sub-segment / column chunk

    def __init__(self, rows, cols, vals, shape, dev, grouping='row'):
        rows = np.asarray(rows, np.int64)
        cols = np.asarray(cols, np.int64)
        vals = np.asarray(vals, np.float32)
        self.shape = tuple(shape)
        self.nnz = len(rows)
        self.dev = dev
        a = sp.csr_matrix((vals, (rows, cols)), shape=shape)
        a.sum_duplicates()
        a.sort_indices()
        at = sp.csr_matrix(a.T)
        at.sort_indices()
        self.fwd, self.bwd = CsrOperand(a, dev), CsrOperand(at, dev)      # A and A^T (row-sharded under torch.distributed)
        # ---- attention structures over the ordered edge list ------------------------------------
        if grouping == 'row':
            order = np.lexsort((cols, rows))                       # canonical row-major order
            rows_o, cols_o, vals_o = rows[order], cols[order], vals[order]
        elif grouping in ('runs', 'reorder'):
            rows_o, cols_o, vals_o = rows, cols, vals              # as fed
        else:
            raise ValueError(grouping)
        self.grouping = grouping
        if grouping == 'reorder':
            self._init_reorder(rows, cols, dev)
        change = np.flatnonzero(np.diff(rows_o)) + 1 if self.nnz else np.zeros(0, np.int64)
        seg_start = np.concatenate([[0], change]).astype(np.int64) if self.nnz else np.zeros(0, np.int64)
        seg_ptr = np.concatenate([seg_start, [self.nnz]]).astype(np.int64)
        seg_row = rows_o[seg_start] if self.nnz else np.zeros(0, np.int64)
        self.seg_ptr_host, self.seg_row_host = seg_ptr, seg_row
        self.unique_rows = len(np.unique(seg_row)) == len(seg_row) if self.nnz else True
        self.e_rows = torch.from_numpy(rows_o).to(dev)             # int64: torch index ops on edge vectors
        self.e_cols = torch.from_numpy(cols_o).to(dev)
        self.e_colidx = ops.to_ids(cols_o, dev)
        self.e_vals = ops.to_vec(vals_o, dev)
        # sub-segments of at most SUB edges (balanced work on power-law degrees)
        sub_ptr, sub_seg, seg_sub_ptr = split_ranges(seg_ptr, self.SUB)
        t_order = np.lexsort((rows_o, cols_o))                     # edges grouped by column
        counts = np.bincount(cols_o, minlength=shape[1]) if self.nnz else np.zeros(shape[1], np.int64)
        t_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        t_sub_ptr, t_sub_col, _ = split_ranges(t_ptr, self.SUB, drop_empty=True)
        self.attn = ops.attn_graph(ops.to_ids(sub_ptr, dev), ops.to_ids(sub_seg, dev), ops.to_ids(seg_sub_ptr, dev),
                                   ops.to_ids(seg_row, dev), self.e_colidx, ops.to_ids(t_sub_ptr, dev),
                                   ops.to_ids(t_sub_col, dev), ops.to_ids(rows_o[t_order], dev), ops.to_ids(t_order, dev),
                                   self.unique_rows, len(t_sub_col) > int((counts > 0).sum()))
        # ---- row-sharded attention under torch.distributed (one exchange per operator output) --------
        # rank r owns a block of SEGMENTS (= output rows, balanced by edges; canonical order makes its edges one
        # contiguous range) for out / alpha / dz, and a block of COLUMNS (balanced by incoming edges) for dv.
        from . import dist as mdist
        rank, ws = mdist.world()
        self.shard = None
        if ws > 1 and self.unique_rows and grouping == 'row' and self.nnz > 0:
            sb = mdist.balanced_bounds(seg_ptr, ws)
            s_lo, s_hi = sb[rank], sb[rank + 1]
            e_lo, e_hi = int(seg_ptr[s_lo]), int(seg_ptr[s_hi])
            l_sub_ptr, l_sub_seg, l_seg_sub_ptr = split_ranges(seg_ptr[s_lo: s_hi + 1] - e_lo, self.SUB)
            if s_hi == s_lo:
                l_sub_ptr, l_sub_seg, l_seg_sub_ptr = np.zeros(1, np.int64), np.zeros(0, np.int64), np.zeros(1, np.int64)
            row_bounds = [0] + [int(seg_row[sb[r]]) if sb[r] < len(seg_row) else shape[0] for r in range(1, ws)] + [shape[0]]
            edge_bounds = [int(seg_ptr[b]) for b in sb]
            cb = mdist.balanced_bounds(t_ptr, ws)
            c_lo, c_hi = cb[rank], cb[rank + 1]
            t_lo, t_hi = int(t_ptr[c_lo]), int(t_ptr[c_hi])
            lt_sub_ptr, lt_sub_col, _ = split_ranges(t_ptr[c_lo: c_hi + 1] - t_lo, self.SUB, drop_empty=True)
            empty = ops.to_ids(np.zeros(0, np.int32), dev)
            one0 = ops.to_ids(np.zeros(1, np.int32), dev)
            seg_part = ops.attn_graph(ops.to_ids(l_sub_ptr, dev), ops.to_ids(l_sub_seg, dev), ops.to_ids(l_seg_sub_ptr, dev),
                                      ops.to_ids(seg_row[s_lo:s_hi], dev), ops.to_ids(cols_o[e_lo:e_hi], dev),
                                      one0, empty, empty, empty, True, False)
            lcounts = counts[c_lo:c_hi]
            t_part = ops.attn_graph(one0, empty, one0, empty, empty, ops.to_ids(lt_sub_ptr, dev),
                                    ops.to_ids(lt_sub_col + c_lo, dev), ops.to_ids(rows_o[t_order][t_lo:t_hi], dev),
                                    ops.to_ids(t_order[t_lo:t_hi], dev), True, len(lt_sub_col) > int((lcounts > 0).sum()))
            self.shard = dict(seg=seg_part, t=t_part, e_lo=e_lo, e_hi=e_hi, row_bounds=row_bounds, edge_bounds=edge_bounds,
                              col_bounds=cb)


def _pattern_csr(major, minor, n_major, dev):
    """CSR of an edge pattern that keeps duplicates apart: (rowptr, colidx, edge id of every CSR slot)"""
    order = np.lexsort((minor, major))
    ptr = np.concatenate([[0], np.cumsum(np.bincount(major, minlength=n_major))]).astype(np.int32)
    return ops.to_ids(ptr, dev), ops.to_ids(minor[order], dev), torch.from_numpy(order.astype(np.int64)).to(dev), ptr


def _init_reorder(self, rows, cols, dev):
    """grouping='reorder' (tests/golden/tf_shim.py, third reading of tf.sparse_softmax): the softmax runs over the rows of
    the CANONICALLY sorted tensor and its p-th output value is attached to the p-th index of the tensor as fed."""
    perm = np.lexsort((cols, rows))                                            # sorted position -> edge as fed
    self.r_perm = torch.from_numpy(perm.astype(np.int64)).to(dev)
    self.r_seg = torch.from_numpy(rows[perm]).to(dev)                           # softmax group of every sorted position
    self.r_fwd = _pattern_csr(rows, cols, self.shape[0], dev)                   # out = P . v,  P[rows[p], cols[p]] = alpha_sorted[p]
    self.r_bwd = _pattern_csr(cols, rows, self.shape[1], dev)                   # dv = P^T . dout
    self.r_rows32, self.r_cols32 = ops.to_ids(rows, dev), ops.to_ids(cols, dev)
    self.r_split = (ops.csr_split(self.r_fwd[3], dev=dev), ops.csr_split(self.r_bwd[3], dev=dev))


EdgeGraph._init_reorder = _init_reorder


class ReorderAttnFn(torch.autograd.Function):
    """sparse attention under grouping='reorder': a per-row softmax of the sorted logits (1-D torch segment ops: plumbing
    on [nnz] vectors) whose values are re-attached to the as-fed pattern; the aggregate, its transpose and the per-edge
    dots dout[row] . v[col] are the HIP kernels (oea_spmm_csr, oea_pair_dots)."""

    @staticmethod
    def forward(ctx, z, v, g, slope):
        v = v.contiguous()
        zs = torch.nn.functional.leaky_relu(z[g.r_perm], slope)
        mx = torch.full((g.shape[0],), -torch.inf, device=z.device).scatter_reduce(0, g.r_seg, zs, 'amax')
        e = torch.exp(zs - mx[g.r_seg])
        alpha = e / torch.zeros(g.shape[0], device=z.device).index_add_(0, g.r_seg, e)[g.r_seg]
        rowptr, colidx, slot_edge, _ = g.r_fwd
        out = ops.spmm_csr(rowptr, colidx, alpha[slot_edge].contiguous(), v, v.shape[1], split=g.r_split[0])
        ctx.g, ctx.slope = g, slope
        ctx.save_for_backward(z, v, alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, v, alpha = ctx.saved_tensors
        g, dout = ctx.g, dout.contiguous()
        rowptr, colidx, slot_edge, _ = g.r_bwd
        dv = ops.spmm_csr(rowptr, colidx, alpha[slot_edge].contiguous(), dout, dout.shape[1], split=g.r_split[1])
        dalpha = ops.pair_dots(dout, v, v.shape[1], g.r_rows32, g.r_cols32)         # position p: the edge as fed
        s = torch.zeros(g.shape[0], device=z.device).index_add_(0, g.r_seg, alpha * dalpha)
        dzs = alpha * (dalpha - s[g.r_seg])
        zs = z[g.r_perm]
        dzs = torch.where(zs > 0, dzs, dzs * ctx.slope)
        dz = torch.empty_like(z)
        dz[g.r_perm] = dzs
        return dz, dv, None, None


class SpmmFn(torch.autograd.Function):
    """y = A . x (tf.sparse_tensor_dense_matmul); backward dx = A^T . dy -- both oea_spmm_csr."""

    @staticmethod
    def forward(ctx, x, graph):
        ctx.graph = graph
        x = x.contiguous()
        return graph.fwd.apply(x, x.shape[1])

    @staticmethod
    def backward(ctx, dy):
        g = ctx.graph
        dy = dy.contiguous()
        return g.bwd.apply(dy, dy.shape[1]), None


class SparseAttnFn(torch.autograd.Function):
    """out = sparse_softmax(leaky_relu(z)) . v over the graph's segments (oea_sparse_attn_fwd/bwd)."""

    @staticmethod
    def forward(ctx, z, v, graph, slope):
        z = z.contiguous()
        v = v.contiguous()
        sh = graph.shard
        if sh is None:
            out, alpha = ops.sparse_attn_fwd(graph.attn, z, v, v.shape[1], slope, graph.shape[0])
        else:       # own segments only, then one all-gather of the output rows (and of alpha, for the dv half of the backward)
            from . import dist as mdist
            out, a_loc = ops.sparse_attn_fwd(sh['seg'], z[sh['e_lo']: sh['e_hi']].contiguous(), v, v.shape[1], slope, graph.shape[0])
            out = mdist.allgather_blocks(out, sh['row_bounds'])
            alpha = torch.empty_like(z)
            alpha[sh['e_lo']: sh['e_hi']] = a_loc
            mdist.allgather_blocks(alpha, sh['edge_bounds'])
        ctx.graph, ctx.slope = graph, slope
        ctx.save_for_backward(z, v, alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, v, alpha = ctx.saved_tensors
        g = ctx.graph
        dout = dout.contiguous()
        sh = g.shard
        if sh is None:
            dz, dv = ops.sparse_attn_bwd(g.attn, z, v, alpha, dout, v.shape[1], ctx.slope)
            return dz, dv, None, None
        from . import dist as mdist
        lo, hi = sh['e_lo'], sh['e_hi']
        dz_loc, _ = ops.sparse_attn_bwd(sh['seg'], z[lo:hi].contiguous(), v, alpha[lo:hi].contiguous(), dout, v.shape[1], ctx.slope)
        dz = torch.empty_like(z)
        dz[lo:hi] = dz_loc
        mdist.allgather_blocks(dz, sh['edge_bounds'])
        _, dv = ops.sparse_attn_bwd(sh['t'], z, v, alpha, dout, v.shape[1], ctx.slope)      # dv rows of this rank's column block
        mdist.allgather_blocks(dv, sh['col_bounds'])
        return dz, dv, None, None


def spmm(graph, x):
    return SpmmFn.apply(x, graph)


def sparse_attention(graph, z, v, slope=0.2):
    if getattr(graph, "grouping", "row") == 'reorder':
        return ReorderAttnFn.apply(z, v, graph, slope)
    return SparseAttnFn.apply(z, v, graph, slope)


class TFAdam:
    """tf.train.AdamOptimizer over a list of dense device parameters (oea_adam_dense)."""

    def __init__(self, params, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def step(self):
        from . import dist as mdist
        self.t += 1
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad.contiguous()
            mdist.sync_replicated_(g)                   # torch.distributed: replicas must apply the same bits
            ops.adam_dense_(p.data, g, m, v, self.lr, self.t, self.b1, self.b2, self.eps)
            p.grad = None


class DenseSGD:
    """tf.train.GradientDescentOptimizer over dense device parameters (p -= lr * grad; oea_sgd_rows without the
    normalisation pull-back)."""

    def __init__(self, params, lr):
        self.params, self.lr = list(params), lr

    def step(self):
        from . import dist as mdist
        for p in self.params:
            if p.grad is None:
                continue
            g = p.grad.contiguous()
            mdist.sync_replicated_(g)
            w = p.data.view(-1, p.shape[-1]) if p.dim() > 1 else p.data.view(1, -1)
            gg = g.view_as(w)
            if w.shape[1] % 4 == 0:
                ops.sgd_rows_(w, gg, w.shape[1], False, self.lr)
            else:                                   # oea_sgd_rows wants 16-byte rows: flatten to one padded row is not
                p.data.add_(g, alpha=-self.lr)      # possible in place; tiny parameters take the element-wise add
            p.grad = None
