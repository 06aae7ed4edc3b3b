"""RDGCN (mirror of openea/approaches/rdgcn.py); BASELINE.json config 5.

Relation-aware dual-graph convolution: a dual graph over the relations (dense R x R attention,
R ~ 700) interacts twice with the primal entity graph through a sparse attention aggregate whose
per-edge logit is a learned scalar of the edge's relation (rdgcn.py:202-215); two diagonal-weight
GCN layers with highway gates follow; L1 margin loss over seed links with hard negatives mined by
L1 nearest-neighbour search every 10 epochs (rdgcn.py:75-87, 466-490); Adam.

Device kernels: the primal sparse attention (csrc/sparse_attn.hip), the GCN aggregate and the
per-relation head/tail averages (csrc/spmm.hip), the L1 hinge (oea_align_loss_l1), hard-negative
mining (fp64 L1 similarity strip + bucket select: oea_sim_matrix + oea_topk_rows), Adam
(csrc/optim.hip).  The dual graph is tiny and dense: plain library GEMMs + softmax.

Reproduced quirks: `get_mat` compares head with RELATION id and increments
degree[relation id] (rdgcn.py:49-51).  The entity input is the summed word vectors of the entity
names (rdgcn.py:415-464) when the `word_embed` file exists (`_get_desc_input`); without the file the
default here is the reference's own alternative `get_input_layer` (glorot, rdgcn.py:280-282).
TF1 op semantics restated, not executed; the composition of Layer.build() is pinned by the reference's own code run
under a numpy stand-in (tests/golden/tf_graphs.npz).
"""
import math
import os
import time

import numpy as np
import torch

from .. import ops
from ..models.basic_model import BasicModel
from ..modules.base.optimizers import generate_optimizer
from ..models.graph_ops import (diag_highway, gather_few, gather_few_plan, relu_axpy, EdgeGraph, TFAdam, sparse_attention,
                                spmm)
from ..modules.finding.evaluation import early_stop, test, valid
from ..modules.load import read as rd


# ---- host-side structures -----------------------------------------------------------------------
def rfunc(triple_list, ent_num, rel_num):
    """rdgcn.py:17-42: head/tail entity sets per relation, and r_mat = one edge (h, t) per triple in
    triple_list order carrying the relation id (duplicated (h, t) pairs stay separate edges)."""
    head, tail = {}, {}
    r_mat_ind, r_mat_val = [], []
    for h, r, t in triple_list:
        head.setdefault(r, set()).add(h)
        tail.setdefault(r, set()).add(t)
        r_mat_ind.append((h, t))
        r_mat_val.append(r)
    return head, tail, np.asarray(r_mat_ind, np.int64).reshape(-1, 2), np.asarray(r_mat_val, np.int64)


def get_mat(triple_list, ent_num):
    """rdgcn.py:45-59 including the head-vs-relation comparison and degree[relation] increment."""
    degree = [1] * ent_num
    pos = {}
    for h, r, t in triple_list:
        if h != r:
            degree[h] += 1
            degree[r] += 1
        if h == t:
            continue
        if (h, t) not in pos:
            pos[(h, t)] = 1
            pos[(t, h)] = 1
    for i in range(ent_num):
        pos[(i, i)] = 1
    return pos, degree


def get_sparse_tensor(triple_list, ent_num):
    """rdgcn.py:63-72 on top of get_mat (:45-59): M[sec, fir] = 1 / sqrt(deg[fir]) / sqrt(deg[sec]) over the symmetric
    closure of the (h, t) pairs plus the diagonal, with get_mat's degree rule (a triple whose head id differs from its
    RELATION id adds one to degree[head] and to degree[relation id]).  Same entries and the same fp64 arithmetic as the
    dictionary walk of get_mat, as numpy set operations (5.7 s -> 0.5 s at the 100K shape); entries come back sorted."""
    tri = np.fromiter((x for tr in triple_list for x in tr), np.int64, count=3 * len(triple_list)).reshape(-1, 3)
    h, r, t = tri[:, 0], tri[:, 1], tri[:, 2]
    n = int(ent_num)
    counted = h != r
    degree = 1 + np.bincount(h[counted], minlength=n) + np.bincount(r[counted], minlength=n)
    off = h != t
    keys = np.unique(np.concatenate([h[off] * n + t[off], t[off] * n + h[off], np.arange(n, dtype=np.int64) * (n + 1)]))
    fir, sec = keys // n, keys % n
    vals = 1.0 / np.sqrt(degree[fir].astype(np.float64)) / np.sqrt(degree[sec].astype(np.float64))
    return sec, fir, vals.astype(np.float32)


def dual_adjacency(head, tail, count_r):
    """rdgcn.py:268-277: Jaccard overlap of head sets + of tail sets, dense [R, R].  The R^2 python set
    intersections of the reference (minutes at 100K) are one sparse incidence product here: |A & B| = (I I^T)[a, b],
    |A | B| = |A| + |B| - |A & B|; same doubles, same float32 result."""
    import scipy.sparse as sp

    def jaccard(sets):
        rows = np.fromiter((r for r, s_ in sets.items() for _ in s_), np.int64)
        cols = np.fromiter((e for s_ in sets.values() for e in s_), np.int64)
        n_cols = int(cols.max()) + 1 if len(cols) else 1
        inc = sp.csr_matrix((np.ones(len(rows), np.float64), (rows, cols)), shape=(count_r, n_cols))
        inter = np.asarray((inc @ inc.T).todense(), np.float64)
        size = np.asarray(inc.sum(1)).reshape(-1)
        union = size[:, None] + size[None, :] - inter
        with np.errstate(divide='ignore', invalid='ignore'):
            return np.where(union > 0, inter / union, 0.0)
    return (jaccard(head) + jaccard(tail)).astype(np.float32)


def glorot(rng, shape, dev):
    r = math.sqrt(6.0 / (shape[0] + shape[1]))
    return torch.from_numpy(rng.uniform(-r, r, shape).astype(np.float32)).to(dev).requires_grad_(True)


class AlignLossL1(torch.autograd.Function):
    """get_loss (rdgcn.py:293-315) = GCN-Align's align_loss: forward + gradient in one HIP kernel."""

    @staticmethod
    def forward(ctx, out, ill, k, gamma, negs):
        out = out.contiguous()
        grad = torch.zeros_like(out)
        loss = torch.zeros(1, dtype=torch.float64, device=out.device)
        ops.align_loss_l1(out, out.shape[1], ill, k, gamma, negs[0], negs[1], negs[2], negs[3], grad, loss)
        ctx.save_for_backward(grad)
        return loss.to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, dloss):
        (grad,) = ctx.saved_tensors
        return grad * dloss, None, None, None, None


class AlignLossL1Rows(torch.autograd.Function):
    """get_loss without atomics: the hinge kernel writes one signed coefficient per pair, every output row then adds its own
    pairs in a fixed order (oea_align_loss_l1_coef + oea_pair_grad_rows; the row lists are built once per redraw of the
    negatives, every 10 epochs) -- reproducible bits."""

    @staticmethod
    def forward(ctx, out, ill, k, gamma, negs, pairs):
        out = out.contiguous()
        loss = torch.zeros(1, dtype=torch.float64, device=out.device)
        coef = ops.align_loss_l1_coef(out, out.shape[1], ill, k, gamma, negs[0], negs[1], negs[2], negs[3], loss)
        ctx.pairs = pairs
        ctx.save_for_backward(out, coef)
        return loss.to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, dloss):
        out, coef = ctx.saved_tensors
        grad = ops.pair_grad_rows(out, out.shape[1], *ctx.pairs, coef, gscale=dloss.reshape(1).contiguous(), norm=1)
        return grad, None, None, None, None, None


def align_pair_rows(ill, k, negs, n_rows):
    """the endpoints of the hinge's pairs grouped by row: pair a < t = link a, pair t + a 2k + i = negative i of link a
    (i < k: (neg_left, neg_right), i >= k: (neg2_left, neg2_right)) -- the coefficient layout of oea_align_loss_l1_coef"""
    nl, nr, n2l, n2r = negs
    t = ill.shape[0]
    neg_pairs = torch.stack([torch.stack([nl.view(t, k), n2l.view(t, k)], 1).reshape(-1),
                             torch.stack([nr.view(t, k), n2r.view(t, k)], 1).reshape(-1)], 1)
    return ops.pair_rows_csr(torch.cat([ill.to(neg_pairs.dtype), neg_pairs]), n_rows)


class Layer:
    """rdgcn.py:162-338."""

    def __init__(self, args, kgs, embedding, dev, seed=0, attn_grouping='runs'):
        self.args, self.dev = args, dev
        self.dim = args.dim
        self.gamma, self.k, self.alpha, self.beta = args.gamma, args.neg_triple_num, args.alpha, args.beta
        self.ILL = np.array(kgs.train_links)
        self.ill_dev = ops.to_ids(self.ILL.astype(np.int32), dev)
        self.triple_list = kgs.kg1.relation_triples_list + kgs.kg2.relation_triples_list
        self.rel_num, self.ent_num = kgs.relations_num, kgs.entities_num
        rng = np.random.RandomState(seed)
        if getattr(args, 'graph_builders', 'device') == 'host':           # the python / numpy restatements (same outputs)
            self.head, self.tail, r_ind, r_val = rfunc(self.triple_list, self.ent_num, self.rel_num)
            self.count_r = len(self.head)
            rows, cols, vals = get_sparse_tensor(self.triple_list, self.ent_num)
            hr = np.array([(r, h) for r, hs in self.head.items() for h in hs], np.int64).reshape(-1, 2)
            tr = np.array([(r, t) for r, ts in self.tail.items() for t in ts], np.int64).reshape(-1, 2)
            dual = torch.from_numpy(dual_adjacency(self.head, self.tail, self.count_r)).to(dev)
        else:
            # device builders (csrc/graph_build.hip): primal adjacency and the Jaccard overlaps; the per-relation head /
            # tail sets are the distinct (relation, entity) pairs of the triple table
            tri = np.fromiter((x for t in self.triple_list for x in t), np.int64, count=3 * len(self.triple_list)).reshape(-1, 3)
            r_ind, r_val = tri[:, [0, 2]], tri[:, 1]
            n_r = int(tri[:, 1].max()) + 1 if len(tri) else 0
            hr = np.unique(tri[:, 1] * self.ent_num + tri[:, 0])
            tr = np.unique(tri[:, 1] * self.ent_num + tri[:, 2])
            hr = np.stack([hr // self.ent_num, hr % self.ent_num], 1)
            tr = np.stack([tr // self.ent_num, tr % self.ent_num], 1)
            self.count_r = len(np.unique(tri[:, 1]))
            assert self.count_r == n_r, "relation ids must be dense (rdgcn.py:268-277 indexes the overlap matrix by id)"
            self.head = self.tail = None
            rows, cols, vals = ops.build_primal_adj(tri, self.ent_num)
            dual = ops.build_dual_adj(tri, self.count_r)
        self.M = EdgeGraph(rows, cols, vals, (self.ent_num, self.ent_num), dev)
        self.r_graph = EdgeGraph(r_ind[:, 0], r_ind[:, 1], np.ones(len(r_ind), np.float32),
                                 (self.ent_num, self.ent_num), dev, grouping=attn_grouping)
        # relation id of every attention edge, in the graph's (possibly re-ordered) edge order
        order = np.lexsort((r_ind[:, 1], r_ind[:, 0])) if attn_grouping == 'row' else np.arange(len(r_ind))
        self.edge_rel = torch.from_numpy(np.ascontiguousarray(r_val[order])).to(dev)
        # compute_r (rdgcn.py:258-266): per-relation mean of its head / tail entity embeddings
        hcnt = np.bincount(hr[:, 0], minlength=self.count_r).astype(np.float32)
        tcnt = np.bincount(tr[:, 0], minlength=self.count_r).astype(np.float32)
        self.head_mean = EdgeGraph(hr[:, 0], hr[:, 1], 1.0 / hcnt[hr[:, 0]], (self.count_r, self.ent_num), dev)
        self.tail_mean = EdgeGraph(tr[:, 0], tr[:, 1], 1.0 / tcnt[tr[:, 0]], (self.count_r, self.ent_num), dev)
        self.dual_A = dual
        self.dual_bias = -1e9 * (1.0 - (self.dual_A > 0).float())
        d = self.dim
        # ---- variables (creation order of rdgcn.py:317-338) ---------------------------------------
        if embedding is not None:
            self.primal_X_0 = torch.tensor(np.asarray(embedding, np.float32), device=dev, requires_grad=True)
        else:
            self.primal_X_0 = glorot(rng, (self.ent_num, d), dev)                 # get_input_layer
        p = {}
        p['sa_w'] = glorot(rng, (2 * d, d), dev)                                  # add_self_att_layer: conv1d(dim, 1), no bias
        p['sa_f1'], p['sa_b1'] = glorot(rng, (d, 1), dev), torch.zeros(1, device=dev, requires_grad=True)
        p['sa_f2'], p['sa_b2'] = glorot(rng, (d, 1), dev), torch.zeros(1, device=dev, requires_grad=True)
        p['pa1_w'], p['pa1_b'] = glorot(rng, (2 * d, 1), dev), torch.zeros(1, device=dev, requires_grad=True)
        p['da_w'], p['da_b'] = glorot(rng, (2 * d, d), dev), torch.zeros(d, device=dev, requires_grad=True)
        p['da_f1'], p['da_b1'] = glorot(rng, (d, 1), dev), torch.zeros(1, device=dev, requires_grad=True)
        p['da_f2'], p['da_b2'] = glorot(rng, (d, 1), dev), torch.zeros(1, device=dev, requires_grad=True)
        p['pa2_w'], p['pa2_b'] = glorot(rng, (2 * d, 1), dev), torch.zeros(1, device=dev, requires_grad=True)
        p['diag1'] = torch.ones((1, d), device=dev, requires_grad=True)           # add_diag_layer init=ones
        p['hw1_w'], p['hw1_b'] = glorot(rng, (d, d), dev), torch.zeros(d, device=dev, requires_grad=True)
        p['diag2'] = torch.ones((1, d), device=dev, requires_grad=True)
        p['hw2_w'], p['hw2_b'] = glorot(rng, (d, d), dev), torch.zeros(d, device=dev, requires_grad=True)
        self.p = p

    def params(self):
        return [self.primal_X_0] + list(self.p.values())

    # ---- layers --------------------------------------------------------------------------------------
    def compute_r(self, inlayer):
        return torch.cat([spmm(self.head_mean, inlayer), spmm(self.tail_mean, inlayer)], dim=-1)

    def _dense_att(self, in_fts, f1, b1, f2, b2, values):
        logits = (in_fts @ f1 + b1) + (in_fts @ f2 + b2).t()
        logits = self.dual_A * logits
        coefs = torch.softmax(torch.nn.functional.leaky_relu(logits, 0.2) + self.dual_bias, dim=-1)
        return torch.relu(coefs @ values)

    def add_self_att_layer(self, inlayer):
        """rdgcn.py:233-248."""
        p = self.p
        return self._dense_att(inlayer @ p['sa_w'], p['sa_f1'], p['sa_b1'], p['sa_f2'], p['sa_b2'], inlayer)

    def add_dual_att_layer(self, inlayer, inlayer2):
        """rdgcn.py:217-231."""
        p = self.p
        return self._dense_att(inlayer2 @ p['da_w'] + p['da_b'], p['da_f1'], p['da_b1'], p['da_f2'], p['da_b2'], inlayer)

    def add_sparse_att_layer(self, inlayer, dual_layer, w, b, relu=True):
        """rdgcn.py:202-215: logit of an edge = conv1d(dual feature of its relation).  relu=False: the caller applies it."""
        dual_transform = (dual_layer @ w + b).reshape(-1)
        if getattr(self, "_rel_plan", None) is None:
            self._rel_plan = gather_few_plan(self.edge_rel, dual_transform.shape[0])
        z = gather_few(dual_transform, self.edge_rel, self._rel_plan)      # backward: chunked wave sums per relation (fixed order)
        out = sparse_attention(self.r_graph, z, inlayer, slope=0.2)
        return torch.relu(out) if relu else out

    def add_diag_layer(self, inlayer, w0):
        """rdgcn.py:184-191; tf.nn.dropout(inlayer, 1 - dropout) in front (in the graph: also when evaluating)."""
        rate = float(getattr(self.args, "dropout", 0.0) or 0.0)
        if rate > 0.0:
            inlayer = torch.nn.functional.dropout(inlayer, p=rate, training=True)
        return torch.relu(spmm(self.M, inlayer * w0))

    @staticmethod
    def highway(layer1, layer2, kernel_gate, bias_gate):
        """rdgcn.py:250-256."""
        gate = torch.sigmoid(layer1 @ kernel_gate + bias_gate)
        return gate * layer2 + (1.0 - gate) * layer1

    def gcn_block(self, x, w0, kernel_gate, bias_gate):
        """highway(x, add_diag_layer(x)) (rdgcn.py:334-337): fused (models/graph_ops.py:DiagHighwayFn) unless dropout is on"""
        if float(getattr(self.args, "dropout", 0.0) or 0.0) > 0.0 or os.environ.get("OEA_RDGCN_FUSED", "1") == "0":
            return self.highway(x, self.add_diag_layer(x, w0), kernel_gate, bias_gate)
        return diag_highway(x, w0, kernel_gate, bias_gate, self.M)

    def forward(self):
        """rdgcn.py:317-337."""
        p = self.p
        x0 = self.primal_X_0
        fused = os.environ.get("OEA_RDGCN_FUSED", "1") != "0"
        dual_H_1 = self.add_self_att_layer(self.compute_r(x0))
        if fused:
            x1 = relu_axpy(x0, self.add_sparse_att_layer(x0, dual_H_1, p['pa1_w'], p['pa1_b'], relu=False), self.alpha)
        else:
            x1 = x0 + self.alpha * self.add_sparse_att_layer(x0, dual_H_1, p['pa1_w'], p['pa1_b'])
        dual_H_2 = self.add_dual_att_layer(dual_H_1, self.compute_r(x1))
        if fused:
            x2 = relu_axpy(x0, self.add_sparse_att_layer(x1, dual_H_2, p['pa2_w'], p['pa2_b'], relu=False), self.beta)
        else:
            x2 = x0 + self.beta * self.add_sparse_att_layer(x1, dual_H_2, p['pa2_w'], p['pa2_b'])
        g1 = self.gcn_block(x2, p['diag1'], p['hw1_w'], p['hw1_b'])
        return self.gcn_block(g1, p['diag2'], p['hw2_w'], p['hw2_b'])

    def loss(self, out, negs):
        """get_loss (rdgcn.py:293-315).  Default: row-grouped gradient (no atomics); `OEA_RDGCN_LOSS=atomic`: one kernel with
        fp32 atomics (bits depend on the arrival order)."""
        if os.environ.get("OEA_RDGCN_LOSS", "rows") == "atomic":
            return AlignLossL1.apply(out, self.ill_dev, self.k, float(self.gamma), negs)
        key = tuple((x.data_ptr(), x._version) for x in negs)
        if getattr(self, "_pairs_key", None) != key:
            self._pairs = align_pair_rows(self.ill_dev, self.k, negs, out.shape[0])
            self._pairs_key, self._pairs_keep = key, negs
        return AlignLossL1Rows.apply(out, self.ill_dev, self.k, float(self.gamma), negs, self._pairs)


def get_neg(ill_ids, output_layer, dim, k, exact_strip=False, margin=32, prefilter=None, stats=None):
    """rdgcn.py:75-87: the k L1-nearest entities of every seed entity among ALL entities (the seed
    itself included, as in the reference) -> device int32 [t*k], ascending ids per seed.

    The reference ranks fp64 `cdist` values.  Here a cheap distance of every (seed, entity) pair ranks k + margin candidates
    per seed and their EXACT fp64 distances pick the k nearest (ties: smaller id) -- the selection the reference makes as long
    as the cheap ranking keeps the true k nearest among its first k + margin.
    prefilter='u16' (default, `OEA_L1_PREFILTER`): the rows on a common 16-bit grid over the table's range, integer L1
      distances (`oea_l1_u16_strip`: a quarter of the fp32 kernel's vector instructions).  grid distance and true distance
      differ by at most dim * step, so every list is CERTIFIED: it is accepted only if the worst candidate's lower bound lies
      above the k-th exact distance; the seeds that fail (none on the shapes tested) go through the all-pairs fp64 path.
    prefilter='f32': fp32 L1 distances (accumulation error ~1e-5 of a distance, not certified).
    exact_strip=True: every pair in fp64 (sim_valu_store_kernel, the bits of scipy's cdist), rounded to fp32 for the select.
    stats: optional dict, receives 'uncertified' (number of seeds redone)."""
    q = ops.gather_rows(output_layer, dim, ill_ids)
    n = output_layer.shape[0]
    if exact_strip or k + margin >= n:
        s = ops.sim_matrix(q, output_layer, dim, 'manhattan', pad=True)        # 1 - cityblock distance, fp64 inside
        return ops.topk_rows(s, k, nc=n).reshape(-1)
    prefilter = prefilter or os.environ.get('OEA_L1_PREFILTER', 'u16')
    bound = None
    if prefilter == 'u16':
        lo, hi = torch.aminmax(output_layer[:, :dim])
        lo, hi = float(lo), float(hi)
        step = max(hi - lo, 1e-30) / 65535.0
        qt = ops.quantize_rows_u16(output_layer, dim, lo, 1.0 / step)
        s = ops.l1_u16_strip(qt.index_select(0, ill_ids.to(torch.int64)), qt)      # minus the grid distance in steps
        del qt
    else:
        s = ops.sim_matrix(q, output_layer, dim, 'manhattan_f32', pad=True)
    cand = ops.topk_rows(s, k + margin, nc=n)                                    # ascending ids
    if prefilter == 'u16':
        # no entity outside the list is nearer on the grid than the list's farthest member; (dim + 4) steps cover both
        # quantisations per column (0.5 step each, + the fp32 rounding of the grid map) and the float rounding of large sums
        worst = -torch.gather(s, 1, cand.to(torch.int64)).amin(dim=1).to(torch.float64)
        bound = worst * step - (dim * 1.02 + 4.0) * step
    del s
    d64 = ops.pair_l1_f64(q, output_layer, dim, cand)
    # k smallest distances, ties -> smaller id (cand is ascending), ids ascending: ranking kernel instead of argsort + sort
    sel, kth = ops.row_rank_select(d64.contiguous(), k, False, ids=cand.contiguous(), want_kth=bound is not None)
    if bound is not None:
        redo = torch.nonzero(~(bound > kth)).reshape(-1)
        if stats is not None:
            stats['uncertified'] = int(redo.numel())
        if redo.numel():
            s = ops.sim_matrix(q.index_select(0, redo).contiguous(), output_layer, dim, 'manhattan', pad=True)
            sel[redo] = ops.topk_rows(s, k, nc=n)
    return sel.reshape(-1).contiguous()


def read_word_vectors(file_path):
    """fastText .vec text file (header line, then `word v1 ... vd`) -> (words, [n + 1, d] matrix whose LAST row is
    the zero vector of unknown words) -- rdgcn.py:424-433."""
    words, vecs = [], []
    with open(file_path, 'r', encoding='utf-8') as f:
        f.readline()                                   # "N d" header (rdgcn.py:426: w[1:])
        for line in f:
            parts = line.rstrip('\n').split(' ')
            if len(parts) < 2:
                continue
            words.append(parts[0])
            vecs.append(np.asarray([x for x in parts[1:] if x != ''], dtype=np.float64))
    mat = np.stack(vecs, axis=0)
    return words, np.append(mat, np.zeros((1, mat.shape[1])), axis=0)


def name_vectors(names_by_entity, entities_num, words, word_em, default_length=4):
    """rdgcn.py:421-462: a name = its first 4 space-separated words after punctuation is removed; every word maps to
    its vector (unknown words and padding to the zero vector, see below); the entity input is the SUM of the 4.
    Padding quirk kept: the reference pads with the id of the LAST entry of its word table, which is the
    unknown-word id when any name word is unknown and otherwise the last vocabulary word."""
    import re
    import string
    punct = re.compile('[{}]+'.format(re.escape(string.punctuation)))
    index = {w: i for i, w in enumerate(words)}
    un_logged_id = len(words)
    split = {e: punct.sub('', n).split(' ') for e, n in names_by_entity.items()}
    any_unknown = any(w not in index for ws in split.values() for w in ws)
    pad_id = un_logged_id if any_unknown else len(words) - 1
    ids = np.full((entities_num, default_length), un_logged_id, np.int64)
    for e, ws in split.items():
        row = [index.get(w, un_logged_id) for w in ws] + [pad_id] * default_length
        ids[e] = row[:default_length]
    return word_em[ids].sum(axis=1), ids


class RDGCN(BasicModel):
    def __init__(self):
        super().__init__()
        self.word_embed = '../../datasets/wiki-news-300d-1M.vec'
        self.local_name_vectors = None
        self.attn_grouping = 'runs'

    def init(self):
        self.dev = ops.device()
        if self.local_name_vectors is None:
            # rdgcn.py:358,424: the name vectors ARE the model's input (accuracy rests on them); the reference raises
            # FileNotFoundError when the word-vector file is missing.  Random initialisation must be asked for.
            if os.path.exists(self.word_embed):
                _, _, self.local_name_vectors = self._get_desc_input()
            elif not getattr(self.args, 'random_name_init', False):
                raise FileNotFoundError("word vectors %r not found (RDGCN initialises the entity features from them, "
                                        "rdgcn.py:424); set args.random_name_init = True for a glorot-random input"
                                        % self.word_embed)
        self.gcn_model = Layer(self.args, self.kgs, self.local_name_vectors, self.dev, seed=self._seed,
                               attn_grouping=getattr(self.args, 'attn_grouping', self.attn_grouping))   # 'row' | 'runs' 
        self.optimizer = generate_optimizer(None, self.args.learning_rate, var_list=list(self.gcn_model.params()), opt='Adam')     # rdgcn.py:332

    def _get_local_name_by_name_triple(self, name_attribute_list=None):
        """rdgcn.py:366-413: entity id -> name = value of a name attribute if the dataset has one, else the local
        part of the URI with '_' -> ' '."""
        if name_attribute_list is None:
            if 'D_Y' in self.args.training_data:
                name_attribute_list = {'skos:prefLabel', 'http://dbpedia.org/ontology/birthName'}
            elif 'D_W' in self.args.training_data:
                name_attribute_list = {'http://www.wikidata.org/entity/P373', 'http://www.wikidata.org/entity/P1476'}
            else:
                name_attribute_list = {}
        triples = []
        for h, a, v in self.kgs.kg1.local_attribute_triples_set | self.kgs.kg2.local_attribute_triples_set:
            v = v.strip('"')
            if v.endswith('"@eng'):
                v = v.rstrip('"@eng')
            triples.append((h, a, v))
        id_ent_dict = {}
        for kg in (self.kgs.kg1, self.kgs.kg2):
            for e, e_id in (kg.entities_id_dict or {}).items():
                id_ent_dict[e_id] = e
        name_ids = set()
        for kg in (self.kgs.kg1, self.kgs.kg2):
            for a, a_id in (kg.attributes_id_dict or {}).items():
                if a in name_attribute_list:
                    name_ids.add(a_id)
        local_name_dict = {}
        for e, a, v in triples:
            if a in name_ids:
                local_name_dict[e] = v
        for e in self.kgs.kg1.entities_set | self.kgs.kg2.entities_set:
            if e not in local_name_dict:
                local_name_dict[e] = str(id_ent_dict[e]).split('/')[-1].replace('_', ' ')
        return [(e, -1, n) for e, n in local_name_dict.items()]

    def _get_desc_input(self):
        """rdgcn.py:415-464 -> (word_em, e_desc_input ids [E, 4], name_embeds [E, d])."""
        start = time.time()
        names = {e: n for e, _, n in self._get_local_name_by_name_triple()}
        words, word_em = read_word_vectors(self.word_embed)
        name_embeds, ids = name_vectors(names, self.kgs.entities_num, words, word_em)
        print('generating desc input costs time: {:.4f}s'.format(time.time() - start))
        return word_em, ids, name_embeds

    def _output(self):
        with torch.no_grad():
            return self.gcn_model.forward().contiguous()

    def training(self):
        """rdgcn.py:466-499."""
        neg_num = self.args.neg_triple_num
        train_links = np.array(self.kgs.train_links)
        dev, d = self.dev, self.args.dim
        neg_left = ops.to_ids(np.repeat(train_links[:, 0], neg_num).astype(np.int32), dev)
        neg2_right = ops.to_ids(np.repeat(train_links[:, 1], neg_num).astype(np.int32), dev)
        ill1 = ops.to_ids(train_links[:, 0].astype(np.int32), dev)
        ill2 = ops.to_ids(train_links[:, 1].astype(np.int32), dev)
        negs = None
        for i in range(1, self.args.max_epoch + 1):
            start = time.time()
            if i % 10 == 1:
                output = self._output()
                neg2_left = get_neg(ill2, output, d, neg_num)
                neg_right = get_neg(ill1, output, d, neg_num)
                negs = (neg_left, neg_right, neg2_left, neg2_right)
            out = self.gcn_model.forward()
            loss = self.gcn_model.loss(out, negs)
            loss.backward()
            self.optimizer.step()
            if i % 10 == 0 or i == 1:
                print('epoch {}, avg. relation triple loss: {:.4f}, cost time: {:.4f}s'.format(i, float(loss.item()),
                                                                                               time.time() - start))
            if i >= self.args.start_valid and i % self.args.eval_freq == 0:
                flag = self.valid_(self.args.stop_metric)
                self.flag1, self.flag2, self.early_stop = early_stop(self.flag1, self.flag2, flag)
                if self.early_stop or i == self.args.max_epoch:
                    break

    def _pick(self, emb, ids):
        out = ops.gather_rows(emb, self.args.dim, ops.to_ids(np.asarray(ids, np.int32), self.dev))
        out.oea_dim = self.args.dim
        return out

    def valid_(self, stop_metric):
        """rdgcn.py:519-527."""
        emb = self._output()
        hits1_12, mrr_12 = valid(self._pick(emb, self.kgs.valid_entities1),
                                 self._pick(emb, self.kgs.valid_entities2 + self.kgs.test_entities2), None,
                                 self.args.top_k, self.args.test_threads_num, metric=self.args.eval_metric)
        return hits1_12 if stop_metric == 'hits1' else mrr_12

    def test(self, save=True):
        """rdgcn.py:501-512."""
        emb = self._output()
        e1, e2 = self._pick(emb, self.kgs.test_entities1), self._pick(emb, self.kgs.test_entities2)
        rest_12, _, _ = test(e1, e2, None, self.args.top_k, self.args.test_threads_num, metric=self.args.eval_metric,
                             normalize=self.args.eval_norm, csls_k=0, accurate=True)
        test(e1, e2, None, self.args.top_k, self.args.test_threads_num, metric=self.args.eval_metric,
             normalize=self.args.eval_norm, csls_k=self.args.csls, accurate=True)
        if save:
            rd.save_results(self.out_folder, [(self.kgs.test_entities1[i], self.kgs.test_entities2[j]) for i, j in rest_12])

    def save(self):
        rd.save_embeddings(self.out_folder, self.kgs, self._output()[:, :self.args.dim].cpu().numpy(), None, None,
                           mapping_mat=None)

    def run(self):
        t = time.time()
        self.training()
        print("training finish")
        print("Training ends. Total time = {:.3f} s.".format(time.time() - t))
