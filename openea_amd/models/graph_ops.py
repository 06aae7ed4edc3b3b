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
        elif grouping == 'runs':
            rows_o, cols_o, vals_o = rows, cols, vals              # as fed
        else:
            raise ValueError(grouping)
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
        out, alpha = ops.sparse_attn_fwd(graph.attn, z, v, v.shape[1], slope, graph.shape[0])
        ctx.graph, ctx.slope = graph, slope
        ctx.save_for_backward(z, v, alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, v, alpha = ctx.saved_tensors
        g = ctx.graph
        dz, dv = ops.sparse_attn_bwd(g.attn, z, v, alpha, dout.contiguous(), v.shape[1], ctx.slope)
        return dz, dv, None, None


def spmm(graph, x):
    return SpmmFn.apply(x, graph)


def sparse_attention(graph, z, v, slope=0.2):
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
        self.t += 1
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            ops.adam_dense_(p.data, p.grad.contiguous(), m, v, self.lr, self.t, self.b1, self.b2, self.eps)
            p.grad = None
