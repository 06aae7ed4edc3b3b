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

    SUB = 256          # edges per softmax sub-segment
    HUB, CHUNK = 768, 512      # aggregate rows / columns with more than HUB slots are summed in chunks of CHUNK (oea_csr_split)

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
        # ---- attention structures over the ordered edge list: built on first use (most graphs only aggregate) ----
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
        self._rows_o, self._cols_o = rows_o, cols_o
        change = np.flatnonzero(np.diff(rows_o)) + 1 if self.nnz else np.zeros(0, np.int64)
        seg_start = np.concatenate([[0], change]).astype(np.int64) if self.nnz else np.zeros(0, np.int64)
        self.seg_ptr_host = np.concatenate([seg_start, [self.nnz]]).astype(np.int64)
        self.seg_row_host = rows_o[seg_start] if self.nnz else np.zeros(0, np.int64)
        self.unique_rows = len(np.unique(self.seg_row_host)) == len(self.seg_row_host) if self.nnz else True
        self.e_rows = torch.from_numpy(rows_o).to(dev)             # int64: torch index ops on edge vectors
        self.e_cols = torch.from_numpy(cols_o).to(dev)
        self.e_colidx = ops.to_ids(cols_o, dev)
        self.e_vals = ops.to_vec(vals_o, dev)
        self._attn = None
        self._shard = None
        self._cut = (self.SUB, self.HUB, self.CHUNK)           # as set when the graph was made (the build is lazy)

    @property
    def single_edge_groups(self):
        """every softmax group is one edge (alpha = 1, d z = 0 exactly: csrc/sparse_attn.hip runs no softmax kernel then)"""
        return self.grouping != 'reorder' and self.nnz > 0 and len(self.seg_row_host) == self.nnz

    @property
    def attn(self):
        if self._attn is None:
            self._build_attn()
        return self._attn

    @property
    def shard(self):
        """the bounds of this rank's blocks when the job is row-sharded (torch.distributed, world > 1), else None"""
        if self._attn is None:
            self._build_attn()
        return self._shard

    def _build_attn(self):
        """softmax segments cut into sub-segments + the aggregate as a CSR over the OUTPUT rows (slots in edge order inside
        a row: the fixed summation order) + the transposed CSR for dv -> oea_attn_graph.
        Row-sharded attention under torch.distributed (one exchange per operator output), BOTH groupings:
        rank r owns a block of SEGMENTS (balanced by edges: a contiguous edge range in either order) for the softmax
        statistics, alpha and dz; a block of OUTPUT ROWS (balanced by nonzeros) for the aggregate; a block of COLUMNS
        (balanced by incoming edges) for dv.  Every rank holds the whole graph, the C call gets its ranges."""
        from . import dist as mdist
        dev, nnz, shape = self.dev, self.nnz, self.shape
        rows_o, cols_o, seg_ptr, seg_row = self._rows_o, self._cols_o, self.seg_ptr_host, self.seg_row_host
        sub_ptr, sub_seg, seg_sub_ptr = split_ranges(seg_ptr, self._cut[0])
        canonical = nnz == 0 or bool(np.all(np.diff(rows_o) >= 0))
        agg_order = np.arange(nnz, dtype=np.int64) if canonical else np.argsort(rows_o, kind="stable")
        agg_rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows_o, minlength=shape[0]))]).astype(np.int64)
        t_order = np.argsort(cols_o, kind="stable")                # edges grouped by column, edge order inside a column
        t_rowptr = np.concatenate([[0], np.cumsum(np.bincount(cols_o, minlength=shape[1]))]).astype(np.int64)
        rank, ws = mdist.world()
        kw = {}
        a_rng = t_rng = None
        if ws > 1 and nnz > 0:
            sb = mdist.balanced_bounds(seg_ptr, ws)
            ab = mdist.balanced_bounds(agg_rowptr, ws)
            cb = mdist.balanced_bounds(t_rowptr, ws)
            a_rng, t_rng = (ab[rank], ab[rank + 1]), (cb[rank], cb[rank + 1])
            kw = dict(seg_range=(sb[rank], sb[rank + 1]), sub_range=(int(seg_sub_ptr[sb[rank]]), int(seg_sub_ptr[sb[rank + 1]])),
                      agg_rows=a_rng, agg_slots=(int(agg_rowptr[a_rng[0]]), int(agg_rowptr[a_rng[1]])),
                      t_rows=t_rng, t_slots=(int(t_rowptr[t_rng[0]]), int(t_rowptr[t_rng[1]])))
            self._shard = dict(edge_bounds=[int(seg_ptr[b]) for b in sb], row_bounds=ab, col_bounds=cb)
        self._attn = ops.attn_graph(
            ops.to_ids(sub_ptr, dev), ops.to_ids(sub_seg, dev), ops.to_ids(seg_sub_ptr, dev), ops.to_ids(seg_row, dev), self.e_colidx,
            ops.to_ids(agg_rowptr, dev), self.e_colidx if canonical else ops.to_ids(cols_o[agg_order], dev),
            None if canonical else ops.to_ids(agg_order, dev),
            ops.to_ids(t_rowptr, dev), ops.to_ids(rows_o[t_order], dev), ops.to_ids(t_order, dev),
            agg_split=ops.csr_split(agg_rowptr, self._cut[1], self._cut[2], dev=dev, row_range=a_rng),
            t_split=ops.csr_split(t_rowptr, self._cut[1], self._cut[2], dev=dev, row_range=t_rng),
            **kw)


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
    # the softmax groups: the rows of the sorted tensor, as the segment structures of csrc/sparse_attn.hip (statistics, alpha
    # and d z only -- the aggregate, its transpose and the dots run over the AS-FED pattern below)
    self.r_sorted = EdgeGraph(rows[perm], cols[perm], np.ones(len(rows), np.float32), self.shape, dev, grouping='row')
    self.r_fwd = _pattern_csr(rows, cols, self.shape[0], dev)                   # out = P . v,  P[rows[p], cols[p]] = alpha_sorted[p]
    self.r_bwd = _pattern_csr(cols, rows, self.shape[1], dev)                   # dv = P^T . dout
    self.r_rows32, self.r_cols32 = ops.to_ids(rows, dev), ops.to_ids(cols, dev)
    self.r_split = (ops.csr_split(self.r_fwd[3], dev=dev), ops.csr_split(self.r_bwd[3], dev=dev))


EdgeGraph._init_reorder = _init_reorder


class ReorderAttnFn(torch.autograd.Function):
    """sparse attention under grouping='reorder': the per-row softmax of the SORTED logits is csrc/sparse_attn.hip's segment
    softmax over the sorted pattern's rows (oea_sparse_attn_fwd, OEA_ATTN_ALPHA) and its backward oea_sparse_attn_dz; the values
    are attached position by position to the as-fed pattern, over which the aggregate, its transpose (oea_spmm_csr) and the
    per-edge dots dout[row] . v[col] (oea_pair_dots) run.  torch only permutes the [nnz] logit / gradient vectors."""

    @staticmethod
    def forward(ctx, z, v, g, slope):
        v = v.contiguous()
        zs = z[g.r_perm].contiguous()
        gs = g.r_sorted.attn
        _, alpha = ops.sparse_attn_fwd(gs, zs, v, v.shape[1], slope, g.shape[0], out=zs, phases=ops.ATTN_ALPHA)   # (out unused)
        sh = g.r_sorted.shard
        if sh is not None:                         # row-sharded job: this rank normalised its block of groups
            from . import dist as mdist
            mdist.allgather_blocks(alpha, sh['edge_bounds'])
        rowptr, colidx, slot_edge, _ = g.r_fwd
        out = ops.spmm_csr(rowptr, colidx, alpha[slot_edge].contiguous(), v, v.shape[1], split=g.r_split[0])
        ctx.g, ctx.slope = g, slope
        ctx.save_for_backward(zs, v, alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        zs, v, alpha = ctx.saved_tensors
        g, dout = ctx.g, dout.contiguous()
        rowptr, colidx, slot_edge, _ = g.r_bwd
        dv = ops.spmm_csr(rowptr, colidx, alpha[slot_edge].contiguous(), dout, dout.shape[1], split=g.r_split[1])
        dalpha = ops.pair_dots(dout, v, v.shape[1], g.r_rows32, g.r_cols32)         # position p: the edge as fed
        dzs = ops.sparse_attn_dz_(g.r_sorted.attn, zs, alpha, dalpha, ctx.slope, v.shape[1])
        sh = g.r_sorted.shard
        if sh is not None:
            from . import dist as mdist
            mdist.allgather_blocks(dzs, sh['edge_bounds'])
        dz = torch.empty_like(dzs)
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
    """out = sparse_softmax(leaky_relu(z)) . v over the graph's segments (oea_sparse_attn_fwd/bwd; no atomics: two runs
    give identical bits).  Row-sharded job: statistics / alpha / dz on this rank's block of segments, the aggregate on
    its block of output rows, dv on its block of columns; one all-gather per output."""

    @staticmethod
    def forward(ctx, z, v, graph, slope):
        z = z.contiguous()
        v = v.contiguous()
        sh = graph.shard
        if sh is None:
            out, alpha = ops.sparse_attn_fwd(graph.attn, z, v, v.shape[1], slope, graph.shape[0])
        else:
            from . import dist as mdist
            out = torch.empty((graph.shape[0], v.shape[1]), dtype=torch.float32, device=v.device)
            alpha = torch.empty_like(z)
            ops.sparse_attn_fwd(graph.attn, z, v, v.shape[1], slope, graph.shape[0], out=out, alpha=alpha, phases=ops.ATTN_ALPHA)
            mdist.allgather_blocks(alpha, sh['edge_bounds'])
            ops.sparse_attn_fwd(graph.attn, z, v, v.shape[1], slope, graph.shape[0], out=out, alpha=alpha, phases=ops.ATTN_AGGREGATE)
            mdist.allgather_blocks(out, sh['row_bounds'])
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
        dz, dv = torch.empty_like(z), torch.empty_like(v)
        ops.sparse_attn_bwd(g.attn, z, v, alpha, dout, v.shape[1], ctx.slope, dz=dz, dv=dv, phases=ops.ATTN_DZ)
        mdist.allgather_blocks(dz, sh['edge_bounds'])
        ops.sparse_attn_bwd(g.attn, z, v, alpha, dout, v.shape[1], ctx.slope, dz=dz, dv=dv, phases=ops.ATTN_DV)
        mdist.allgather_blocks(dv, sh['col_bounds'])
        return dz, dv, None, None


class ConcatL2NormFn(torch.autograd.Function):
    """l2n(concat(l2n(x_0), ..., l2n(x_{k-1}))) in one pass each way (oea_concat_l2n_fwd / bwd); output [n, pad4(sum d)]."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [x.contiguous() for x in xs]
        out, inv_blk, inv_all = ops.concat_l2n_fwd(xs)
        ctx.dims = [x.shape[1] for x in xs]
        ctx.save_for_backward(out, inv_blk, inv_all)
        return out

    @staticmethod
    def backward(ctx, dz):
        out, inv_blk, inv_all = ctx.saved_tensors
        return tuple(ops.concat_l2n_bwd(ctx.dims, out, dz.contiguous(), inv_blk, inv_all))


class PairLossFn(torch.autograd.Function):
    """sum ||e_i - e_j||^2 over the positive links + balance * sum w relu(margin - ||e_i - e_j||^2) over the negative ones
    (alinet.py:828-850) -> scalar.  Forward: one wave per pair; backward: the pairs' endpoints grouped by embedding row
    (one stable device sort per batch), one wave per row adds its active pairs in slot order -- no atomics."""

    @staticmethod
    def forward(ctx, emb, dim, pos, neg, weight, margin, balance, side=None):
        emb = emb.contiguous()
        pairs = torch.cat([pos[:, :2], neg[:, :2]]).to(torch.int32).contiguous()
        terms, coef = ops.pair_loss_l2_fwd(emb, dim, pairs, pos.shape[0], weight, margin, balance)
        ctx.dim = dim
        ctx.save_for_backward(emb, pairs, coef)
        loss = terms.sum()
        ctx.side = None
        if side is not None:
            # a second loss on FEW rows of emb (AliNet's relation loss, alinet.py:852-866: 2 x 20,000 of 200,000 rows of width
            # 1,200): evaluated on a gathered copy, its row gradients are added into the dense gradient this Function returns
            # anyway -- as two more autograd branches each cost a zero-filled [E, d] tensor, a scatter and a full-size add
            idx, fn = side
            rows = emb.index_select(0, idx).detach().requires_grad_(True)
            with torch.enable_grad():
                side_loss = fn(rows)
            ctx.side = (idx, rows, side_loss)
            loss = loss + side_loss.detach()
        return loss

    @staticmethod
    def backward(ctx, gloss):
        emb, pairs, coef = ctx.saved_tensors
        rowptr, other, slot_pair = ops.pair_rows_csr(pairs, emb.shape[0])
        g = gloss.to(torch.float32).reshape(1).contiguous()
        grad = ops.pair_grad_rows(emb, ctx.dim, rowptr, other, slot_pair, coef, g, norm=2)
        if ctx.side is not None:
            idx, rows, side_loss = ctx.side
            (g_rows,) = torch.autograd.grad(side_loss, rows)
            grad.index_put_((idx,), g_rows * g, accumulate=True)      # duplicates summed in sorted order (deterministic)
            ctx.side = None
        return grad, None, None, None, None, None, None, None


class HighwayFn(torch.autograd.Function):
    """out = tanh(b' (1 - gate) + a' gate), gate = relu(tanh(p)), a' / b' = a / b through the layer's BatchNorm affine
    (alinet.py:597-622), one pass each way (oea_highway_fwd / bwd)."""

    @staticmethod
    def forward(ctx, a, b, p, gamma, beta):
        a, b, p = a.contiguous(), b.contiguous(), p.contiguous()
        out = ops.highway_fwd(a, b, p, gamma.contiguous(), beta.contiguous())
        ctx.save_for_backward(a, b, p, gamma, beta, out)
        return out

    @staticmethod
    def backward(ctx, gout):
        a, b, p, gamma, beta, out = ctx.saved_tensors
        return ops.highway_bwd(a, b, p, gamma.contiguous(), beta.contiguous(), out, gout.contiguous())


class BiasTanhFn(torch.autograd.Function):
    """tanh(x + bias) (alinet.py:583-590), one pass each way."""

    @staticmethod
    def forward(ctx, x, bias):
        y = ops.bias_tanh_fwd(x.contiguous(), bias.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return ops.bias_tanh_bwd(y, gy.contiguous())


def weight_grad(x, dy):
    """dW = x^T dy of a dense layer in front of a sparse aggregate.  Under torch.distributed the row chunks of the product
    are shared by the ranks (ops.gemm_tn_sharded: this rank's chunk partials, one all-gather of [chunks, d_in * d_out], all
    chunks added in chunk order = the single-process bits); OEA_DP_DENSE_DW=0 keeps the replicated product."""
    import os
    from . import dist as mdist
    rank, world = mdist.world()
    if world > 1 and os.environ.get("OEA_DP_DENSE_DW", "1") != "0":
        return ops.gemm_tn_sharded(x, dy, rank, world, mdist.allgather_blocks)
    return ops.gemm_tn(x, dy)


class DenseFn(torch.autograd.Function):
    """y = x W (+ bias row) for a tall x [E, d_in]: forward and dx on the library's NN / NT kernels (85-105 TFLOP/s at these
    shapes), the weight gradient dW = x^T dy on oea_gemm_tn_f32 (the library's TN kernels run it at 44-57)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return torch.addmm(bias, x, w) if bias is not None else x @ w

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ w.t() if ctx.needs_input_grad[0] else None
        dw = weight_grad(x.contiguous(), dy) if ctx.needs_input_grad[1] else None
        return dx, dw, (dy.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None)


def dense(x, w, bias=None):
    return DenseFn.apply(x, w, bias)


class DiagHighwayFn(torch.autograd.Function):
    """One GCN block of RDGCN (rdgcn.py:184-191, 250-256, 334-337):  h = relu(M (x * w0)),  gate = sigmoid(x W + b),
    out = gate h + (1 - gate) x.  The aggregate carries the relu, the gate / mix and their gradients are one kernel each way
    (oea_sigmoid_mix_*: its db is already the gradient of the relu's input), the two GEMMs are the library's."""

    @staticmethod
    def forward(ctx, x, w0, kernel_gate, bias_gate, graph):
        x = x.contiguous()
        d = x.shape[1]
        h = graph.fwd.apply(x * w0, d, act=1)
        p = x @ kernel_gate
        ctx.graph = graph
        ctx.save_for_backward(x, w0, kernel_gate, bias_gate, h, p)
        return ops.sigmoid_mix_fwd(x, h, p, bias_gate.contiguous())

    @staticmethod
    def backward(ctx, gout):
        x, w0, kernel_gate, bias_gate, h, p = ctx.saved_tensors
        da, dh_pre, dp, dbias = ops.sigmoid_mix_bwd(x, h, p, bias_gate.contiguous(), gout.contiguous(), b_relu=True)
        dxs = ctx.graph.bwd.apply(dh_pre, x.shape[1])
        dx = torch.addmm(da, dp, kernel_gate.t())
        dx.addcmul_(dxs, w0)
        return dx, ops.colsum_prod(dxs, x).reshape(w0.shape), weight_grad(x, dp), dbias, None


class ReluAxpyFn(torch.autograd.Function):
    """x + alpha relu(y) (RDGCN's residual around an attention layer, rdgcn.py:330-333), one pass each way."""

    @staticmethod
    def forward(ctx, x, y, alpha):
        x, y = x.contiguous(), y.contiguous()
        ctx.alpha = alpha
        ctx.save_for_backward(y)
        return ops.relu_axpy_fwd(x, y, alpha)

    @staticmethod
    def backward(ctx, gout):
        (y,) = ctx.saved_tensors
        gout = gout.contiguous()
        return gout, ops.relu_axpy_bwd(y, gout, ctx.alpha), None


class GatherFewFn(torch.autograd.Function):
    """z = src[idx] for a 1-D src with FEW rows and many gathers (RDGCN: one logit per relation, gathered per attention
    edge); backward = one wave per source row adds its edges' gradients in a fixed order (oea_segment_sum_f32) instead of
    torch's sorted index_put walk over ~800 duplicates per row."""

    @staticmethod
    def forward(ctx, src, idx, plan):
        ctx.plan, ctx.n = plan, src.shape[0]
        return src[idx]

    @staticmethod
    def backward(ctx, g):
        order, chunk_ptr, row_chunk_ptr = ctx.plan
        return ops.segment_sum(ops.segment_sum(g.contiguous(), order, chunk_ptr), None, row_chunk_ptr), None, None


def gather_few_plan(idx, n_rows, chunk=2048):
    """(order int32 [len(idx)], chunk_ptr int32 [n_chunks + 1], row_chunk_ptr int32 [n_rows + 1]): the gather positions grouped by
    source row, position order kept, every row's run cut into chunks of <= `chunk` positions (one wave sums a chunk, then one
    wave a row's chunk sums: a relation of RDGCN's 100K graphs holds 10^5 edges -- summed by ONE wave the call took 1.3 ms)."""
    idx64 = idx.to(torch.int64)
    order = torch.argsort(idx64, stable=True).to(torch.int32).contiguous()
    counts = torch.bincount(idx64, minlength=n_rows)
    seg_ptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=idx.device)
    seg_ptr[1:] = torch.cumsum(counts, 0)
    per_row = (counts + chunk - 1) // chunk
    row_chunk_ptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=idx.device)
    row_chunk_ptr[1:] = torch.cumsum(per_row, 0)
    n_chunks = int(row_chunk_ptr[-1])
    row_of = torch.repeat_interleave(torch.arange(n_rows, device=idx.device), per_row)
    local = torch.arange(n_chunks, device=idx.device) - row_chunk_ptr[:-1][row_of]
    start = seg_ptr[:-1][row_of] + local * chunk
    chunk_ptr = torch.cat([start, seg_ptr[-1:]])                   # a chunk ends where the next begins (rows are contiguous)
    return order, chunk_ptr.to(torch.int32).contiguous(), row_chunk_ptr.to(torch.int32).contiguous()


def diag_highway(x, w0, kernel_gate, bias_gate, graph):
    return DiagHighwayFn.apply(x, w0, kernel_gate, bias_gate, graph)


def relu_axpy(x, y, alpha):
    return ReluAxpyFn.apply(x, y, float(alpha))


def gather_few(src, idx, plan):
    return GatherFewFn.apply(src, idx, plan)


def concat_l2n(xs):
    return ConcatL2NormFn.apply(*xs)


def pair_loss(emb, dim, pos, neg, weight, margin, balance, side=None):
    """side = (idx int64 [m], fn): + fn(emb[idx]), a loss on few rows whose gradient joins the pair loss's dense one"""
    return PairLossFn.apply(emb, dim, pos, neg, weight, margin, balance, side)


def highway_gate(a, b, p, gamma, beta):
    return HighwayFn.apply(a, b, p, gamma, beta)


def bias_tanh(x, bias):
    return BiasTanhFn.apply(x, bias)


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
    """tf.train.GradientDescentOptimizer over dense device parameters: p -= lr * grad (oea_sgd_rows without the
    normalisation pull-back).  Any shape: the parameter is viewed as rows of 256 floats (the tail, if any, as one more
    zero-padded row), so 1-D biases and widths that are not a multiple of 4 take the same kernel."""

    ROW = 256

    def __init__(self, params, lr):
        self.params, self.lr = list(params), lr

    def step(self):
        from . import dist as mdist
        for p in self.params:
            if p.grad is None:
                continue
            g = p.grad.contiguous()
            mdist.sync_replicated_(g)
            w = p.data if p.data.is_contiguous() else p.data.contiguous()
            n, row = w.numel(), self.ROW
            full = n // row * row
            if full:
                ops.sgd_rows_(w.view(-1)[:full].view(-1, row), g.view(-1)[:full].view(-1, row), row, False, self.lr)
            if n > full:
                tail_w = torch.zeros(row, dtype=w.dtype, device=w.device)
                tail_g = torch.zeros(row, dtype=w.dtype, device=w.device)
                tail_w[: n - full] = w.view(-1)[full:]
                tail_g[: n - full] = g.view(-1)[full:]
                ops.sgd_rows_(tail_w.view(1, row), tail_g.view(1, row), row, False, self.lr)
                w.view(-1)[full:] = tail_w[: n - full]
            if w.data_ptr() != p.data.data_ptr():
                p.data.copy_(w)
            p.grad = None
