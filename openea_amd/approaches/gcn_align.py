"""GCN-Align (mirror of openea/approaches/gcn_align.py); BASELINE.json config 3.

Two 2-layer GCNs trained full-batch with SGD on an L1 alignment hinge: the structure model SE
(featureless: the layer-1 weight [E, se_dim] *is* the embedding table) and the attribute model
AE (sparse 0/1 attribute features x W).  Host side (one-off, scipy): functionality-weighted
adjacency (gcn_align.py:610-664) and its symmetric normalisation (gcn_align.py:566-578).
Device side, per epoch and per model (csrc/spmm.hip):

    T   = l2_normalize(W)                       trunc_normal returns the normalised tensor (:52-56)
    H1  = relu(A . X)        X = T (SE) | F . T (AE, F = sparse features)   GraphConvolution (:239-267)
    out = A . H1                                                           (:259, no weights, identity)
    loss, d out = align_loss(out, ILL, negatives)                          (:298-320)
    d H1 = A^T . d out  (gated by H1 > 0),  d X = A^T . d H1,  d T = F^T . d X (AE)
    W  -= lr * (d T through the normalisation)                             GradientDescentOptimizer (:511)

i.e. 4 (SE) / 6 (AE) CSR aggregates per epoch, no atomics in the aggregates, fixed summation
order.  TF1 op semantics are restated, not executed; the composition of the unit's graph is pinned by the reference's own
GCN_Align_Unit code run under a numpy stand-in (tests/golden/tf_graphs.npz).
"""
import math
import time

import numpy as np
import scipy.sparse as sp
import torch

from .. import ops
from ..models.basic_model import BasicModel
from ..models.graph_ops import CsrOperand
from ..modules.base.initializers import truncated_normal_host
from ..modules.finding.evaluation import early_stop, test, valid
from ..modules.load import read as rd
from ..modules.utils.util import merge_dic


# ---- host-side graph construction ---------------------------------------------------------------
def load_attr(ent_num, kgs):
    """gcn_align.py:89-109: 0/1 matrix over the 70% most frequent attributes."""
    cnt = {}
    entity_attributes_dict = merge_dic(kgs.kg1.entity_attributes_dict, kgs.kg2.entity_attributes_dict)
    for _, vs in entity_attributes_dict.items():
        for v in vs:
            cnt[v] = cnt.get(v, 0) + 1
    fre = sorted(cnt, key=cnt.get, reverse=True)
    num = int(0.7 * len(cnt))
    attr2id = {fre[i]: i for i in range(num)}
    rows, cols = [], []
    for ent, vs in entity_attributes_dict.items():
        for v in vs:
            if v in attr2id:
                rows.append(ent)
                cols.append(attr2id[v])
    data = np.ones(len(rows), np.float32)
    return sp.csr_matrix((data, (rows, cols)), shape=(ent_num, max(num, 0)), dtype=np.float32)


class GCN_Utils:
    def __init__(self, args, kgs):
        self.args = args
        self.kgs = kgs

    @staticmethod
    def func(triples):
        """gcn_align.py:610-624: r2f[r] = #distinct heads of r / #triples of r."""
        head, cnt = {}, {}
        for h, r, t in triples:
            cnt[r] = cnt.get(r, 0) + 1
            head.setdefault(r, set()).add(h)
        return {r: len(head[r]) / cnt[r] for r in cnt}

    @staticmethod
    def ifunc(triples):
        """gcn_align.py:626-640: r2if[r] = #distinct tails of r / #triples of r."""
        tail, cnt = {}, {}
        for h, r, t in triples:
            cnt[r] = cnt.get(r, 0) + 1
            tail.setdefault(r, set()).add(t)
        return {r: len(tail[r]) / cnt[r] for r in cnt}

    def get_weighted_adj(self, e, KG):
        """gcn_align.py:642-664: M[(h,t)] += max(ifun(r), .3), M[(t,h)] += max(fun(r), .3);
        COO with row = second key, col = first key."""
        r2f, r2if = self.func(KG), self.ifunc(KG)
        M = {}
        for h, r, t in KG:
            if h == t:
                continue
            M[(h, t)] = M.get((h, t), 0.0) + max(r2if[r], 0.3)
            M[(t, h)] = M.get((t, h), 0.0) + max(r2f[r], 0.3)
        keys = list(M.keys())
        row = [k[1] for k in keys]
        col = [k[0] for k in keys]
        data = [M[k] for k in keys]
        return sp.coo_matrix((data, (row, col)), shape=(e, e))

    @staticmethod
    def normalize_adj(adj):
        """gcn_align.py:566-573: adj.dot(D^-1/2).transpose().dot(D^-1/2)."""
        adj = sp.coo_matrix(adj)
        rowsum = np.array(adj.sum(1))
        with np.errstate(divide='ignore'):
            d_inv_sqrt = np.power(rowsum, -0.5).flatten()
        d_inv_sqrt[np.isinf(d_inv_sqrt)] = 0.
        d_mat_inv_sqrt = sp.diags(d_inv_sqrt)
        return adj.dot(d_mat_inv_sqrt).transpose().dot(d_mat_inv_sqrt).tocoo()

    def preprocess_adj(self, adj):
        """gcn_align.py:575-578 -> scipy COO (fp64 values; cast to fp32 when fed, :719)."""
        return self.normalize_adj(adj + sp.eye(adj.shape[0]))

    def load_data(self, attr):
        triples = self.kgs.kg1.relation_triples_list + self.kgs.kg2.relation_triples_list
        adj = self.get_weighted_adj(self.kgs.entities_num, triples)
        train = np.array(self.kgs.train_links)
        return adj, attr, train

    def load_data_device(self, attr):
        """load_data + preprocess_adj with the functionality weights, the weighted adjacency and its normalisation built on
        the device (oea_build_weighted_adj) -> (adj, support, attr, train); same entries and fp64 values as the host
        functions above (sums of duplicate pairs in triple order)."""
        triples = self.kgs.kg1.relation_triples_list + self.kgs.kg2.relation_triples_list
        e = self.kgs.entities_num
        n_rel = max(int(self.kgs.relations_num), 1 + max((r for _, r, _ in triples), default=0))
        b = ops.build_weighted_adj(triples, e, n_rel, raw=True)
        ar, ac, av = b["adj"]
        sr, sc, sv = b["support"]
        adj = sp.coo_matrix((av, (ar, ac)), shape=(e, e))
        support = sp.coo_matrix((sv, (sr, sc)), shape=(e, e))
        return adj, support, attr, np.array(self.kgs.train_links)


class DeviceCSR:
    """CSR + transposed CSR of a sparse matrix, fp32, on the device (models/graph_ops.py:CsrOperand: the
    aggregates are row-sharded + all-gathered when the job runs under torch.distributed)."""

    def __init__(self, mat, dev):
        a = sp.csr_matrix(mat, dtype=np.float32)
        self.shape = a.shape
        self.nnz = a.nnz
        self.fwd, self.bwd = CsrOperand(a, dev), CsrOperand(sp.csr_matrix(a.T, dtype=np.float32), dev)
        self.rowptr = self.fwd.rowptr

    def mm(self, x, dim, act=0, mask_from=None):
        return self.fwd.apply(x, dim, act=act, mask_from=mask_from)

    def tmm(self, x, dim, mask_from=None):
        return self.bwd.apply(x, dim, mask_from=mask_from)


class GCN_Align_Unit:
    """gcn_align.py:498-539: GraphConvolution(relu, trunc_normal weight) -> GraphConvolution(identity,
    no weight) + align_loss + GradientDescentOptimizer."""

    def __init__(self, args, adj: DeviceCSR, weight_rows, output_dim, ILL, features: DeviceCSR = None, seed=0):
        self.args = args
        self.adj = adj
        self.features = features              # None = featureless (SE)
        self.dim = output_dim
        dev = adj.rowptr.device
        rng = np.random.RandomState(seed)
        # trunc_normal(shape): truncated_normal(stddev = 1/sqrt(shape[0])) then l2_normalize rows (:52-56)
        self.W = ops.to_table(truncated_normal_host(rng, (weight_rows, output_dim), 1.0 / math.sqrt(weight_rows)), dev=dev)
        self.row_ids = torch.arange(weight_rows, dtype=torch.int32, device=dev)
        self.ILL = ops.to_ids(np.asarray(ILL, np.int32).reshape(-1, 2), dev)
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self.outputs = None

    def forward(self):
        d = self.dim
        T = ops.gather_rows(self.W, d, self.row_ids, normalize=True)
        x = T if self.features is None else self.features.mm(T, d)
        H1 = self.adj.mm(x, d, act=1)
        out = self.adj.mm(H1, d)
        return T, H1, out

    def train_step(self, negs):
        """one full-batch epoch: forward, L1 hinge, backward, SGD.  Returns nothing; loss accumulates.
        Single process: ONE C call enqueues the epoch's kernels (oea_gcn_unit_epoch) -- driven op by op from Python the
        epoch was bound by the host at the 15K shapes.  Under torch.distributed the aggregates are row-sharded with an
        all-gather per layer (models/graph_ops.py:CsrOperand), op by op."""
        from ..models import dist as mdist
        if mdist.world()[1] == 1:
            return self._train_step_fused(negs)
        d = self.dim
        T, H1, out = self.forward()
        nl, nr, n2l, n2r = negs
        k = self.args.neg_triple_num
        pairs = self._pair_lists(negs, out.shape[0])
        if pairs is not None:
            self._coef = ops.align_loss_l1_coef(out, d, self.ILL, k, self.args.gamma, nl, nr, n2l, n2r, self.loss, getattr(self, "_coef", None))
            g_out = ops.pair_grad_rows(out, d, *pairs, self._coef, norm=1)
        else:
            g_out = torch.zeros_like(out)
            ops.align_loss_l1(out, d, self.ILL, k, self.args.gamma, nl, nr, n2l, n2r, g_out, self.loss)
        g_pre1 = self.adj.tmm(g_out, d, mask_from=H1)           # relu gate fused
        g_x = self.adj.tmm(g_pre1, d)
        g_T = g_x if self.features is None else self.features.tmm(g_x, d)
        mdist.sync_replicated_(g_T)                     # torch.distributed: keep the replicas on the same bits
        ops.sgd_rows_(self.W, g_T, d, True, self.args.learning_rate)
        self.outputs = out

    def _pair_lists(self, negs, n_rows):
        """k <= 16 (GCN-Align's 5): the negatives stay for 10 epochs (gcn_align.py:740-755), so the endpoints of the hinge's
        pairs are grouped by row once per redraw; every epoch the hinge kernel only writes the pairs' coefficients and each
        row adds its pairs in a fixed order -- no atomics, reproducible bits.  Larger k (RDGCN's 125): the atomic kernel."""
        k = self.args.neg_triple_num
        if k > 16:
            return None
        key = tuple(x.data_ptr() for x in negs)
        if getattr(self, "_pairs_key", None) != key or getattr(self, "_pairs_ver", None) != tuple(x._version for x in negs):
            nl, nr, n2l, n2r = negs
            t = self.ILL.shape[0]
            neg_pairs = torch.stack([torch.stack([nl.view(t, k), n2l.view(t, k)], 1).reshape(-1),
                                     torch.stack([nr.view(t, k), n2r.view(t, k)], 1).reshape(-1)], 1)     # [t, 2, k]: index a 2k + i
            self._pair_csr = ops.pair_rows_csr(torch.cat([self.ILL.to(neg_pairs.dtype), neg_pairs]), n_rows)
            self._pairs_key, self._pairs_ver = key, tuple(x._version for x in negs)
            self._pairs_keep = negs                              # the device pointers in the key stay valid
        return self._pair_csr

    def _train_step_fused(self, negs):
        import ctypes as C
        key = tuple(x.data_ptr() for x in negs)
        if getattr(self, "_unit_key", None) != key or self._pairs_ver != tuple(x._version for x in negs):
            self._build_unit(negs)                                   # once per redraw of the negatives (every 10 epochs)
            self._unit_key = key
        ops.check(self._epoch_fn(C.byref(self._unit), self.W.data_ptr(), C.byref(self._cb), self.loss.data_ptr(), ops._stream()))
        self.outputs = self._bufs["out"]

    def _build_unit(self, negs):
        """the argument block of oea_gcn_unit_epoch: operands, negatives, pair lists, buffers (allocated once)"""
        import ctypes as C
        from .. import _lib
        d, dev = self.dim, self.W.device
        ld = self.W.shape[1]
        n = self.adj.shape[0]
        k = self.args.neg_triple_num
        if getattr(self, "_bufs", None) is None:
            def buf(rows):
                return torch.empty((rows, ld), dtype=torch.float32, device=dev)
            feat = self.features is not None
            self._bufs = dict(t=buf(self.W.shape[0]), x=buf(n) if feat else None, h1=buf(n), out=buf(n), g_out=buf(n), g_pre1=buf(n),
                              g_x=buf(n), g_t=buf(self.W.shape[0]) if feat else None,
                              coef=torch.empty(self.ILL.shape[0] * (1 + 2 * k), dtype=torch.float32, device=dev))
            self._cb = _lib.GcnUnitBuffers(*[None if self._bufs[nm] is None else self._bufs[nm].data_ptr()
                                             for nm in ("t", "x", "h1", "out", "g_out", "g_pre1", "g_x", "g_t", "coef")])
            self._epoch_fn = ops.lib().oea_gcn_unit_epoch
        pairs = self._pair_lists(negs, n)
        u = _lib.GcnUnit()
        ops_list = [("a_", self.adj.fwd), ("at_", self.adj.bwd)]
        if self.features is not None:
            ops_list += [("f_", self.features.fwd), ("ft_", self.features.bwd)]
        for pre, op in ops_list:
            setattr(u, pre + "rowptr", op.rowptr.data_ptr())
            setattr(u, pre + "colidx", op.colidx.data_ptr())
            setattr(u, pre + "vals", op.vals.data_ptr())
            if op.split is not None:
                ops._split_partials(op.split, ld, dev)
                setattr(u, pre + "split", C.pointer(op.split))
        nl, nr, n2l, n2r = negs
        u.row_ids, u.ill = self.row_ids.data_ptr(), self.ILL.data_ptr()
        u.neg_left, u.neg_right, u.neg2_left, u.neg2_right = nl.data_ptr(), nr.data_ptr(), n2l.data_ptr(), n2r.data_ptr()
        if pairs is not None:
            u.pair_rowptr, u.pair_other, u.pair_slot = (x.data_ptr() for x in pairs)
        u.n, u.w_rows, u.t = n, self.W.shape[0], self.ILL.shape[0]
        u.dim, u.ld, u.k = d, ld, k
        u.gamma, u.lr = float(self.args.gamma), float(self.args.learning_rate)
        self._unit = u
        self._unit_keep = negs
        if pairs is None:
            self._pairs_ver = tuple(x._version for x in negs)

    def pop_loss(self):
        v = float(self.loss.item())
        self.loss.zero_()
        return v


class GCN_Align(BasicModel):
    def __init__(self):
        super().__init__()
        self.attr = None
        self.vec_ae = None
        self.vec_se = None
        self.model_ae = None
        self.model_se = None

    def init(self):
        assert self.args.alignment_module == 'mapping'
        assert self.args.neg_triple_num > 1
        assert self.args.learning_rate >= 0.01
        dev = ops.device()
        self.utils = GCN_Utils(self.args, self.kgs)
        self.attr = load_attr(self.kgs.entities_num, self.kgs)
        self.e = self.kgs.entities_num
        if getattr(self.args, 'graph_builders', 'device') == 'host':       # the python restatements (same outputs)
            self.adj, self.ae_input, self.train = self.utils.load_data(self.attr)
            self.support = DeviceCSR(self.utils.preprocess_adj(self.adj), dev)
        else:
            self.adj, support, self.ae_input, self.train = self.utils.load_data_device(self.attr)
            self.support = DeviceCSR(support, dev)
        self.model_ae = None
        if self.attr.shape[1] > 0:
            self.model_ae = GCN_Align_Unit(self.args, self.support, self.attr.shape[1], self.args.ae_dim, self.train,
                                           features=DeviceCSR(self.attr, dev), seed=self._seed + 1)
        self.model_se = GCN_Align_Unit(self.args, self.support, self.e, self.args.se_dim, self.train, seed=self._seed)

    def _negatives(self, rng, train_num, neg_num, dev):
        """gcn_align.py:740-755: neg_left / neg2_right repeat the seed ids, the others are redrawn."""
        return ops.to_ids(rng.choice(self.e, train_num * neg_num).astype(np.int32), dev)

    def train_embeddings(self, loss=None, optimizer=None, output=None):
        """gcn_align.py:737-785."""
        neg_num = self.args.neg_triple_num
        train_links = np.array(self.kgs.train_links)
        train_num = len(train_links)
        dev = self.support.rowptr.device
        neg_left = ops.to_ids(np.repeat(train_links[:, 0], neg_num).astype(np.int32), dev)
        neg2_right = ops.to_ids(np.repeat(train_links[:, 1], neg_num).astype(np.int32), dev)
        neg2_left = neg_right = None
        rng = np.random.RandomState(self._seed + 7)
        last_report, t_report = 0, time.time()
        for i in range(1, self.args.max_epoch + 1):
            if i % 10 == 1:
                neg2_left = self._negatives(rng, train_num, neg_num, dev)
                neg_right = self._negatives(rng, train_num, neg_num, dev)
            negs = (neg_left, neg_right, neg2_left, neg2_right)
            batch_loss = 0.0
            if self.model_ae is not None:
                self.model_ae.train_step(negs)
            self.model_se.train_step(negs)
            if i % 10 == 0 or i == 1:          # the reference prints every epoch; one sync per 10 here
                # the device accumulator holds the SUM over the epochs since the last read-back: report the per-epoch
                # mean (comparable with the reference's per-epoch value); the read-back synchronises, so the time is
                # the wall time of those epochs divided by their number
                n_ep = i - last_report
                batch_loss = (self.model_se.pop_loss() + (self.model_ae.pop_loss() if self.model_ae else 0.0)) / n_ep
                now = time.time()
                print('epoch {}, avg. relation triple loss: {:.4f}, cost time: {:.4f}s'.format(i, batch_loss, (now - t_report) / n_ep))
                last_report, t_report = i, now
            if i >= self.args.start_valid and i % self.args.eval_freq == 0:
                flag = self.valid_(self.args.stop_metric)
                self.flag1, self.flag2, self.early_stop = early_stop(self.flag1, self.flag2, flag)
                if self.early_stop or i == self.args.max_epoch:
                    break
        self.vec_se = self.model_se.forward()[2]
        self.vec_ae = self.model_ae.forward()[2] if self.model_ae is not None else None
        return self.vec_se, self.vec_ae

    def _embeddings(self, se, ae):
        """gcn_align.py:787-792 / 810-815: concat[beta * se, (1 - beta) * ae] (test_method 'sa')."""
        d_se = self.args.se_dim
        if self.args.test_method == "sa" and ae is not None:
            beta, d_ae = self.args.beta, self.args.ae_dim
            emb = torch.zeros((se.shape[0], ops.pad4(d_se + d_ae)), dtype=torch.float32, device=se.device)
            emb[:, :d_se] = se[:, :d_se] * beta
            emb[:, d_se:d_se + d_ae] = ae[:, :d_ae] * (1.0 - beta)
            return emb, d_se + d_ae
        return se, d_se

    def _pick(self, emb, dim, ids):
        out = ops.gather_rows(emb, dim, ops.to_ids(np.asarray(ids, np.int32), emb.device))
        out.oea_dim = dim
        return out

    def valid_(self, stop_metric):
        """gcn_align.py:808-822 (valid() defaults: csls 0, normalize False, quick mode)."""
        emb, dim = self._embeddings(self.model_se.forward()[2], self.model_ae.forward()[2] if self.model_ae else None)
        embeds1 = self._pick(emb, dim, self.kgs.valid_entities1)
        embeds2 = self._pick(emb, dim, self.kgs.valid_entities2 + self.kgs.test_entities2)
        hits1_12, mrr_12 = valid(embeds1, embeds2, None, self.args.top_k, self.args.test_threads_num,
                                 metric=self.args.eval_metric)
        return hits1_12 if stop_metric == 'hits1' else mrr_12

    def test(self, save=True):
        """gcn_align.py:787-802."""
        emb, dim = self._embeddings(self.vec_se, self.vec_ae)
        embeds1 = self._pick(emb, dim, self.kgs.test_entities1)
        embeds2 = self._pick(emb, dim, self.kgs.test_entities2)
        rest_12, _, _ = test(embeds1, embeds2, None, self.args.top_k, self.args.test_threads_num,
                             metric=self.args.eval_metric, normalize=self.args.eval_norm, csls_k=0, accurate=True)
        test(embeds1, embeds2, None, self.args.top_k, self.args.test_threads_num,
             metric=self.args.eval_metric, normalize=self.args.eval_norm, csls_k=self.args.csls, accurate=True)
        if save:
            ent_ids_rest_12 = [(self.kgs.test_entities1[i], self.kgs.test_entities2[j]) for i, j in rest_12]
            rd.save_results(self.out_folder, ent_ids_rest_12)

    def save(self):
        """gcn_align.py:804-806."""
        ent_embeds = self.vec_se[:, :self.args.se_dim].cpu().numpy()
        attr_embeds = self.vec_ae[:, :self.args.ae_dim].cpu().numpy() if self.vec_ae is not None else None
        rd.save_embeddings(self.out_folder, self.kgs, ent_embeds, None, attr_embeds, mapping_mat=None)

    def run(self):
        t = time.time()
        self.train_embeddings()
        print("Training ends. Total time = {:.3f} s.".format(time.time() - t))
