"""Graph operators for the GNN approaches: the HIP aggregate / attention kernels wrapped as
``torch.autograd.Function`` so that the dense parts of AliNet / RDGCN (plain library GEMMs,
activations) can be chained with them.  torch is the tape here, not the arithmetic of the sparse
ops: every sparse forward / backward below is a call into libopenea_hip.so.
"""
import numpy as np
import scipy.sparse as sp
import torch

from .. import ops


class EdgeGraph:
    """A sparse [n_rows, n_cols] operator given as an ordered edge list (rows, cols, vals), kept in
    the order the reference would feed it to TF, plus the derived device structures:

    * CSR / transposed CSR (row-sorted) for the plain aggregate (tf.sparse_tensor_dense_matmul);
    * softmax segments over the ORDERED edge list: grouping='row' -> one segment per row (the
      mathematically intended tf.sparse_softmax), grouping='runs' -> maximal runs of consecutive
      edges with equal row (what TF1's CPU kernel does on non-canonical order);
    * the transposed edge list (per column: output row + edge id) for the attention backward.
    """

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
        self.rowptr, self.colidx, self.vals = ops.to_ids(a.indptr, dev), ops.to_ids(a.indices, dev), ops.to_vec(a.data, dev)
        self.t_rowptr, self.t_colidx, self.t_vals = ops.to_ids(at.indptr, dev), ops.to_ids(at.indices, dev), ops.to_vec(at.data, dev)
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
        self.seg_ptr = ops.to_ids(np.concatenate([seg_start, [self.nnz]]), dev)
        self.seg_row = ops.to_ids(rows_o[seg_start] if self.nnz else np.zeros(0), dev)
        self.unique_rows = len(np.unique(rows_o[seg_start])) == len(seg_start) if self.nnz else True
        self.e_rows = torch.from_numpy(rows_o).to(dev)             # int64: torch index ops on edge vectors
        self.e_cols = torch.from_numpy(cols_o).to(dev)
        self.e_colidx = ops.to_ids(cols_o, dev)
        self.e_vals = ops.to_vec(vals_o, dev)
        t_order = np.lexsort((rows_o, cols_o))                     # edges grouped by column
        counts = np.bincount(cols_o, minlength=shape[1]) if self.nnz else np.zeros(shape[1], np.int64)
        self.t_ptr = ops.to_ids(np.concatenate([[0], np.cumsum(counts)]), dev)
        self.t_row = ops.to_ids(rows_o[t_order], dev)
        self.t_edge = ops.to_ids(t_order, dev)


class SpmmFn(torch.autograd.Function):
    """y = A . x (tf.sparse_tensor_dense_matmul); backward dx = A^T . dy -- both oea_spmm_csr."""

    @staticmethod
    def forward(ctx, x, graph):
        ctx.graph = graph
        x = x.contiguous()
        return ops.spmm_csr(graph.rowptr, graph.colidx, graph.vals, x, x.shape[1])

    @staticmethod
    def backward(ctx, dy):
        g = ctx.graph
        dy = dy.contiguous()
        return ops.spmm_csr(g.t_rowptr, g.t_colidx, g.t_vals, dy, dy.shape[1]), None


class SparseAttnFn(torch.autograd.Function):
    """out = sparse_softmax(leaky_relu(z)) . v over the graph's segments (oea_sparse_attn_fwd/bwd)."""

    @staticmethod
    def forward(ctx, z, v, graph, slope):
        z = z.contiguous()
        v = v.contiguous()
        out, alpha = ops.sparse_attn_fwd(graph.seg_ptr, graph.seg_row, graph.e_colidx, z, v, v.shape[1], slope,
                                         graph.unique_rows, graph.shape[0])
        ctx.graph, ctx.slope = graph, slope
        ctx.save_for_backward(z, v, alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, v, alpha = ctx.saved_tensors
        g = ctx.graph
        dz, dv = ops.sparse_attn_bwd(g.seg_ptr, g.seg_row, g.e_colidx, z, v, alpha, dout.contiguous(), v.shape[1],
                                     ctx.slope, g.t_ptr, g.t_row, g.t_edge)
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
