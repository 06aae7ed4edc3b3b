// rotate_step.hip -- the RotatE step of BootEA_RotatE (approaches/bootea_rotate.py:50-109,148-158) in fp64 for gfx950.
//
// Variables (bootea_rotate.py:50-57, all tf.float64): re_ent_embeds / im_ent_embeds [E, d] (l2-normalised per row when
// args.ent_l2_norm: init_embeddings returns l2_normalize(variable), initializers.py:26) and rel_embeds [R, d] (phases).
// Score of a triple (bootea_rotate.py:59-70, 86-96):
//     theta = rel[r] * (pi / embedding_range),   (a, b) = (h_re + i h_im) * e^{i theta} - (t_re + i t_im),
//     dist = sum_d sqrt(a_d^2 + b_d^2),   positives: gamma - dist,  negatives: dist - gamma,
//     loss = - sum log sigmoid(score)  =  sum softplus(dist+ - gamma) + sum softplus(gamma - dist-)   (bootea_rotate.py:72-81)
// and the alignment loss is the positive half alone (bootea_rotate.py:148-158).
//
// Layout: the two entity tables are STACKED in one [2E, ld] fp64 array (rows [0, E) real, [E, 2E) imaginary parts: same
// l2_norm flag, same optimiser), the relation phases are [R, ld].  Kernel 1 (rotate_triples): one G-lane group per
// chunk of up to 16 triples of a positive's family (itself + its k negatives, the sampler's layout; free lists run one
// triple per group): cos / sin of the shared relation row are evaluated once per group, the relation-row gradient of
// the group leaves as one row of atomics into one of kRelCopies scratch copies, so do the rows of the family's own
// head and tail; the corrupted entities' rows leave per triple (global_atomic_add_f64).  Kernel 2 (rotate_apply)
// visits EVERY row: TF's AdamOptimizer moves all rows of a variable every step (dense gradient through l2_normalize
// for the entity tables; _apply_sparse decays m, v and updates the whole variable for the raw relation table), so there
// are no touched flags -- Adagrad / SGD rows with a zero gradient do not move anyway.
// d(sqrt(a^2+b^2)) at a = b = 0 is taken as 0 (TF yields NaN there).
#include "common.h"

namespace {

using oea::group_sum_d;

constexpr int kMaxBlocks = 4096;
constexpr int kRelCopies = 8;
constexpr int kChunk = 16;         // triples of one positive's family per lane group (k = 10: the whole family)

struct RotWs {
    double *ent_grad;      // [2E, ld]  w.r.t. the normalised rows
    double *rel_grad;      // copy 0 [R, ld]
    double *rel_extra;     // copies 1.. [kRelCopies-1][R, ld]
    double *partials;      // [kMaxBlocks]
    int64_t rel_stride;
    __device__ __forceinline__ double *rel_copy(int64_t c) const { return c == 0 ? rel_grad : rel_extra + (c - 1) * rel_stride; }
};

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

static size_t ws_layout(int64_t n_ent2, int64_t n_rel, int32_t ld, void *base, RotWs *ws) {
    size_t off = 0;
    char *b = static_cast<char *>(base);
    auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return b ? b + o : nullptr; };
    double *eg = (double *)take(sizeof(double) * (size_t)n_ent2 * ld);
    double *rg = (double *)take(sizeof(double) * (size_t)n_rel * ld);
    double *rx = (double *)take(sizeof(double) * (size_t)n_rel * ld * (kRelCopies - 1));
    double *pp = (double *)take(sizeof(double) * kMaxBlocks);
    if (ws) { ws->ent_grad = eg; ws->rel_grad = rg; ws->rel_extra = rx; ws->partials = pp; ws->rel_stride = n_rel * (int64_t)ld; }
    return off;
}

template <int G, int IT>
struct RowD {
    double v[IT];
};
template <int G, int IT>
__device__ __forceinline__ void load_row(const double *__restrict__ base, int ld, int lane, RowD<G, IT> &r) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = it * G + lane;
        r.v[it] = c < ld ? base[c] : 0.0;
    }
}
template <int G, int IT>
__device__ __forceinline__ double sumsq(const RowD<G, IT> &r) {
    double s = 0.0;
#pragma unroll
    for (int it = 0; it < IT; ++it) s += r.v[it] * r.v[it];
    return group_sum_d<G>(s);
}
template <int G, int IT>
__device__ __forceinline__ void normalize(RowD<G, IT> &r, int on) {
    if (!on) return;
    const double inv = 1.0 / sqrt(fmax(sumsq<G, IT>(r), 1e-12));
#pragma unroll
    for (int it = 0; it < IT; ++it) r.v[it] *= inv;
}
template <int G, int IT>
__device__ __forceinline__ void atomic_row(double *__restrict__ dst, int ld, int lane, const RowD<G, IT> &g) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = it * G + lane;
        if (c < ld && g.v[it] != 0.0) unsafeAtomicAdd(dst + c, g.v[it]);
    }
}
__device__ __forceinline__ double softplus_(double x) { return x > 0.0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
__device__ __forceinline__ double sigmoid_(double x) { return 1.0 / (1.0 + exp(-x)); }

template <int G, int IT>
__device__ __forceinline__ void phase_row(const double *__restrict__ rel, int r, int ld, int lane, const oea_rotate_cfg &cfg,
                                          RowD<G, IT> &c, RowD<G, IT> &s) {
    RowD<G, IT> yr;
    load_row<G, IT>(rel + (int64_t)r * ld, ld, lane, yr);
    normalize<G, IT>(yr, cfg.rel_l2_norm);
#pragma unroll
    for (int it = 0; it < IT; ++it) sincos(yr.v[it] * cfg.phase_scale, &s.v[it], &c.v[it]);
}

// the gradient rows of a family's own head and tail, summed in registers over its triples (a corruption keeps one
// side of its positive) and sent once: 24 instead of 44 rows of fp64 atomics per positive at k = 10
template <int G, int IT>
struct FamilyAcc {
    int h0, t0;
    RowD<G, IT> h_re, h_im, t_re, t_im;
};

// one triple: loss term returned, entity gradients scattered (or kept in `fam`), d loss / d theta ADDED to gth
template <int G, int IT, bool ACC>
__device__ __forceinline__ double rotate_triple(const double *__restrict__ ent, int64_t E, int ld, int lane, int h, int t,
                                                bool is_pos, const RowD<G, IT> &c, const RowD<G, IT> &s,
                                                const oea_rotate_cfg &cfg, const RotWs &ws, RowD<G, IT> &gth,
                                                FamilyAcc<G, IT> &fam) {
    RowD<G, IT> rh, ih, rt, it_;
    load_row<G, IT>(ent + (int64_t)h * ld, ld, lane, rh);
    load_row<G, IT>(ent + (E + h) * ld, ld, lane, ih);
    load_row<G, IT>(ent + (int64_t)t * ld, ld, lane, rt);
    load_row<G, IT>(ent + (E + t) * ld, ld, lane, it_);
    normalize<G, IT>(rh, cfg.ent_l2_norm);
    normalize<G, IT>(ih, cfg.ent_l2_norm);
    normalize<G, IT>(rt, cfg.ent_l2_norm);
    normalize<G, IT>(it_, cfg.ent_l2_norm);
    RowD<G, IT> a, b;
    double dist = 0.0;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        a.v[it] = rh.v[it] * c.v[it] - ih.v[it] * s.v[it] - rt.v[it];
        b.v[it] = rh.v[it] * s.v[it] + ih.v[it] * c.v[it] - it_.v[it];
        const double n = sqrt(a.v[it] * a.v[it] + b.v[it] * b.v[it]);
        dist += n;
        const double inv = n > 0.0 ? 1.0 / n : 0.0;
        a.v[it] *= inv;          // d dist / d a
        b.v[it] *= inv;          // d dist / d b
    }
    dist = group_sum_d<G>(dist);
    const double x = is_pos ? dist - cfg.gamma : cfg.gamma - dist;
    const double coef = is_pos ? sigmoid_(x) : -sigmoid_(x);          // d loss / d dist
    RowD<G, IT> g1, g2;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const double da = coef * a.v[it], db = coef * b.v[it];
        g1.v[it] = da * c.v[it] + db * s.v[it];                        // d/d h_re
        g2.v[it] = db * c.v[it] - da * s.v[it];                        // d/d h_im
        gth.v[it] += da * (-rh.v[it] * s.v[it] - ih.v[it] * c.v[it]) + db * (rh.v[it] * c.v[it] - ih.v[it] * s.v[it]);
        a.v[it] = -da;                                                 // d/d t_re
        b.v[it] = -db;                                                 // d/d t_im
    }
    if (ACC && h == fam.h0) {
#pragma unroll
        for (int it = 0; it < IT; ++it) { fam.h_re.v[it] += g1.v[it]; fam.h_im.v[it] += g2.v[it]; }
    } else {
        atomic_row<G, IT>(ws.ent_grad + (int64_t)h * ld, ld, lane, g1);
        atomic_row<G, IT>(ws.ent_grad + (E + h) * ld, ld, lane, g2);
    }
    if (ACC && t == fam.t0) {
#pragma unroll
        for (int it = 0; it < IT; ++it) { fam.t_re.v[it] += a.v[it]; fam.t_im.v[it] += b.v[it]; }
    } else {
        atomic_row<G, IT>(ws.ent_grad + (int64_t)t * ld, ld, lane, a);
        atomic_row<G, IT>(ws.ent_grad + (E + t) * ld, ld, lane, b);
    }
    return softplus_(x);
}

template <int G, int IT>
__global__ __launch_bounds__(256) void rotate_triples(const double *__restrict__ ent, int64_t E, const double *__restrict__ rel,
                                                      int ld, const int32_t *__restrict__ pos, int64_t n_pos,
                                                      const int32_t *__restrict__ neg, int64_t n_neg, int k,
                                                      oea_rotate_cfg cfg, RotWs ws) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    // grouped layout: the 1 + k triples of a positive are dealt to `chunks` groups of <= kChunk triples; every chunk
    // evaluates cos / sin of the shared relation row itself and sends its own rows of relation / head / tail gradient.
    // (Measured at the 15K shape, k = 10, rocprofv3: 222 us per step with chunks of 3, ~190 us with the whole family in
    // one group, ~130 us with the family rows summed in registers: the kernel is bound by its fp64 atomics -- 22 M per
    // step before, 12 M after -- not by the number of resident waves.)
    const int chunks = k > 0 ? (k + 1 + kChunk - 1) / kChunk : 1;
    const int64_t items = k > 0 ? n_pos * chunks : n_pos + n_neg;
    double loss_local = 0.0;
    for (int64_t item = grp; item < items; item += ngrp) {
        const int64_t fam = k > 0 ? item / chunks : item;                 // the positive this group works for
        const int j0 = k > 0 ? (int)(item % chunks) * kChunk : 0;
        const int j1 = k > 0 ? min(j0 + kChunk, k + 1) : 1;
        const bool free_neg = k == 0 && item >= n_pos;
        const int32_t *lead = free_neg ? neg + 3 * (item - n_pos) : pos + 3 * fam;
        const int r = lead[1];
        RowD<G, IT> c, s, gth;
        phase_row<G, IT>(rel, r, ld, lane, cfg, c, s);
        constexpr bool ACC = IT <= 4;                   // 4 more rows of fp64 accumulators: not at the 256-VGPR shapes
        FamilyAcc<G, IT> facc;
        facc.h0 = lead[0];
        facc.t0 = lead[2];
#pragma unroll
        for (int it = 0; it < IT; ++it) gth.v[it] = facc.h_re.v[it] = facc.h_im.v[it] = facc.t_re.v[it] = facc.t_im.v[it] = 0.0;
        double l = 0.0;
        for (int j = j0; j < j1; ++j) {
            const int32_t *tr = j == 0 ? lead : neg + 3 * (fam * k + j - 1);
            const bool is_pos = j == 0 && !free_neg;
            if (tr[1] == r) {
                l += rotate_triple<G, IT, ACC>(ent, E, ld, lane, tr[0], tr[2], is_pos, c, s, cfg, ws, gth, facc);
            } else {                                    // not a corruption of this positive: its own relation row
                RowD<G, IT> c2, s2, g2;
                phase_row<G, IT>(rel, tr[1], ld, lane, cfg, c2, s2);
#pragma unroll
                for (int it = 0; it < IT; ++it) g2.v[it] = 0.0;
                l += rotate_triple<G, IT, ACC>(ent, E, ld, lane, tr[0], tr[2], is_pos, c2, s2, cfg, ws, g2, facc);
#pragma unroll
                for (int it = 0; it < IT; ++it) g2.v[it] *= cfg.phase_scale;
                atomic_row<G, IT>(ws.rel_copy(item % kRelCopies) + (int64_t)tr[1] * ld, ld, lane, g2);
            }
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) gth.v[it] *= cfg.phase_scale;       // theta = y_r * phase_scale
        atomic_row<G, IT>(ws.rel_copy(item % kRelCopies) + (int64_t)r * ld, ld, lane, gth);
        if (ACC) {
            atomic_row<G, IT>(ws.ent_grad + (int64_t)facc.h0 * ld, ld, lane, facc.h_re);
            atomic_row<G, IT>(ws.ent_grad + (E + facc.h0) * ld, ld, lane, facc.h_im);
            atomic_row<G, IT>(ws.ent_grad + (int64_t)facc.t0 * ld, ld, lane, facc.t_re);
            atomic_row<G, IT>(ws.ent_grad + (E + facc.t0) * ld, ld, lane, facc.t_im);
        }
        if (lane == 0) loss_local += l;
    }
    __shared__ double sred[4];
    const double w = oea::wave_sum_d(loss_local);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) ws.partials[blockIdx.x] = sred[0] + sred[1] + sred[2] + sred[3];
}

// data parallel: fold relation copies 1.. into copy 0 so that [ent_grad | rel_grad] is the exchanged prefix
__global__ void rotate_fold_kernel(RotWs ws, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double sum = 0.0;
#pragma unroll
        for (int cpy = 0; cpy < kRelCopies - 1; ++cpy) {
            const double v = ws.rel_extra[cpy * ws.rel_stride + i];
            if (v != 0.0) { sum += v; ws.rel_extra[cpy * ws.rel_stride + i] = 0.0; }
        }
        if (sum != 0.0) ws.rel_grad[i] += sum;
    }
}

template <int G, int IT>
__global__ __launch_bounds__(256) void rotate_apply(double *__restrict__ ent, double *__restrict__ ent_state, int64_t n_ent2,
                                                    double *__restrict__ rel, double *__restrict__ rel_state, int64_t n_rel,
                                                    int ld, oea_rotate_cfg cfg, double lr_t, RotWs ws, int n_partials,
                                                    double *__restrict__ loss_accum, int copies_folded) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t row_all = grp; row_all < n_ent2 + n_rel; row_all += ngrp) {
        const bool is_rel = row_all < n_rel;
        const int64_t row = is_rel ? row_all : row_all - n_rel;
        const int64_t rows = is_rel ? n_rel : n_ent2;
        double *v = (is_rel ? rel : ent) + row * ld;
        double *st = is_rel ? rel_state : ent_state;
        double *g = (is_rel ? ws.rel_grad : ws.ent_grad) + row * ld;
        const int on = is_rel ? cfg.rel_l2_norm : cfg.ent_l2_norm;
        RowD<G, IT> rv, rg;
        load_row<G, IT>(v, ld, lane, rv);
        load_row<G, IT>(g, ld, lane, rg);
        if (is_rel && !copies_folded) {
            for (int cp = 1; cp < kRelCopies; ++cp) {
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int c = it * G + lane;
                    if (c < ld) {
                        const double x = ws.rel_copy(cp)[row * ld + c];
                        if (x != 0.0) { rg.v[it] += x; ws.rel_copy(cp)[row * ld + c] = 0.0; }
                    }
                }
            }
        }
        double inv = 1.0, ydg = 0.0;
        if (on) {
            const double ss = sumsq<G, IT>(rv);
            inv = 1.0 / sqrt(fmax(ss, 1e-12));
            double dot = 0.0;
#pragma unroll
            for (int it = 0; it < IT; ++it) dot += rv.v[it] * rg.v[it];
            dot = group_sum_d<G>(dot) * inv;
            ydg = ss > 1e-12 ? dot : 0.0;
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * G + lane;
            if (c < ld) {
                const double gv = on ? (rg.v[it] - rv.v[it] * inv * ydg) * inv : rg.v[it];
                const int64_t e = row * ld + c;
                if (cfg.opt_kind == OEA_OPT_ADAM) {          // tf.train.AdamOptimizer: every row, every step
                    const double m = cfg.beta1 * st[e] + (1.0 - cfg.beta1) * gv;
                    const double vv = cfg.beta2 * st[rows * ld + e] + (1.0 - cfg.beta2) * gv * gv;
                    st[e] = m;
                    st[rows * ld + e] = vv;
                    v[c] = rv.v[it] - lr_t * m / (sqrt(vv) + cfg.eps);
                } else if (cfg.opt_kind == OEA_OPT_ADAGRAD) {
                    const double a = st[e] + gv * gv;
                    st[e] = a;
                    v[c] = rv.v[it] - cfg.lr * gv / sqrt(a);
                } else {
                    v[c] = rv.v[it] - cfg.lr * gv;
                }
                g[c] = 0.0;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        double s = 0.0;
        for (int i = threadIdx.x; i < n_partials; i += 64) s += ws.partials[i];
        s = oea::wave_sum_d(s);
        if (threadIdx.x == 0) *loss_accum += s;
    }
}

// out[i] = f32( [l2n]( part(re[ids[i]]) + part(im[ids[i]]) ) ), part = l2_normalize when part_norm: the embeddings the
// evaluation / bootstrapping / neighbour search read (bootea_rotate.py:111-146,160-167)
template <int G, int IT>
__global__ __launch_bounds__(256) void rotate_lookup_kernel(const double *__restrict__ ent, int64_t E, int ld,
                                                            const int32_t *__restrict__ ids, int64_t n, int part_norm,
                                                            int sum_norm, float *__restrict__ out, int out_ld) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t i = grp; i < n; i += ngrp) {
        const int64_t id = ids ? ids[i] : i;
        RowD<G, IT> re, im;
        load_row<G, IT>(ent + id * ld, ld, lane, re);
        load_row<G, IT>(ent + (E + id) * ld, ld, lane, im);
        normalize<G, IT>(re, part_norm);
        normalize<G, IT>(im, part_norm);
#pragma unroll
        for (int it = 0; it < IT; ++it) re.v[it] += im.v[it];
        normalize<G, IT>(re, sum_norm);
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * G + lane;
            if (c < out_ld) out[i * out_ld + c] = c < ld ? (float)re.v[it] : 0.f;
        }
        for (int c = IT * G + lane; c < out_ld; c += G) out[i * out_ld + c] = 0.f;     // pad columns of the fp32 block
    }
}

template <int G, int IT>
int launch_rotate(double *ent, double *ent_state, int64_t E, double *rel, double *rel_state, int64_t n_rel, int32_t ld,
                  const int32_t *pos, int64_t n_pos, const int32_t *neg, int64_t n_neg, int k, const oea_rotate_cfg &cfg,
                  const RotWs &ws, double *loss_accum, int phase, hipStream_t st) {
    const int block = 256, gpb = block / G;
    const int64_t items = k > 0 ? n_pos * ((k + 1 + kChunk - 1) / kChunk) : n_pos + n_neg;
    const int nb1 = (int)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(items, gpb), 1), kMaxBlocks);
    if (phase != OEA_PHASE_APPLY && items > 0) {
        rotate_triples<G, IT><<<nb1, block, 0, st>>>(ent, E, rel, ld, pos, n_pos, neg, n_neg, k, cfg, ws);
        if (phase == OEA_PHASE_GRAD)
            rotate_fold_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(n_rel * (int64_t)ld, 256), 1024), 256, 0, st>>>(
                ws, n_rel * (int64_t)ld);
    }
    if (phase != OEA_PHASE_GRAD) {
        double lr_t = cfg.lr;
        if (cfg.opt_kind == OEA_OPT_ADAM)
            lr_t = cfg.lr * sqrt(1.0 - pow(cfg.beta2, (double)cfg.t)) / (1.0 - pow(cfg.beta1, (double)cfg.t));
        const int nb2 = (int)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(2 * E + n_rel, gpb), 1), 16384);
        rotate_apply<G, IT><<<nb2, block, 0, st>>>(ent, ent_state, 2 * E, rel, rel_state, n_rel, ld, cfg, lr_t, ws,
                                                   items > 0 ? nb1 : 0, loss_accum, phase == OEA_PHASE_APPLY);
    }
    return 0;
}

}  // namespace

extern "C" {

size_t oea_rotate_workspace_bytes(int64_t n_ent, int64_t n_rel, int32_t ld) { return ws_layout(2 * n_ent, n_rel, ld, nullptr, nullptr); }

size_t oea_rotate_exchange_doubles(int64_t n_ent, int64_t n_rel, int32_t ld) {
    RotWs ws;
    ws_layout(2 * n_ent, n_rel, ld, reinterpret_cast<void *>(256), &ws);
    return (size_t)(reinterpret_cast<char *>(ws.rel_extra) - reinterpret_cast<char *>(256)) / sizeof(double);
}

int oea_rotate_step(double *ent, double *ent_state, int64_t n_ent, double *rel, double *rel_state, int64_t n_rel,
                    int32_t dim, int32_t ld, const int32_t *pos, int64_t n_pos, const int32_t *neg, int64_t n_neg,
                    int32_t neg_group_k, const oea_rotate_cfg *cfg, void *workspace, double *loss_accum, int32_t phase,
                    void *stream) {
    OEA_REQUIRE(phase >= OEA_PHASE_BOTH && phase <= OEA_PHASE_APPLY, "phase");
    OEA_REQUIRE(ent && rel && (pos || n_pos == 0) && cfg && workspace && loss_accum, "null pointer");
    OEA_REQUIRE(dim > 0 && dim <= ld && ld % 2 == 0 && ld <= 512, "0 < dim <= ld <= 512, ld even");
    OEA_REQUIRE(n_pos >= 0 && n_neg >= 0 && (neg || n_neg == 0), "neg == NULL needs n_neg == 0");
    OEA_REQUIRE(neg_group_k >= 0 && (neg_group_k == 0 || n_neg == n_pos * (int64_t)neg_group_k),
                "neg_group_k > 0 needs n_neg == n_pos * neg_group_k");
    OEA_REQUIRE(cfg->opt_kind >= OEA_OPT_SGD && cfg->opt_kind <= OEA_OPT_ADAM, "opt_kind");
    OEA_REQUIRE(cfg->opt_kind == OEA_OPT_SGD || (ent_state && rel_state), "Adagrad / Adam need their state arrays");
    OEA_REQUIRE(cfg->opt_kind != OEA_OPT_ADAM || cfg->t >= 1, "Adam: t >= 1");
    RotWs ws;
    ws_layout(2 * n_ent, n_rel, ld, workspace, &ws);
    hipStream_t st = oea::as_stream(stream);
#define OEA_ROT(G, IT) launch_rotate<G, IT>(ent, ent_state, n_ent, rel, rel_state, n_rel, ld, pos, n_pos, neg, n_neg, neg_group_k, *cfg, ws, loss_accum, phase, st)
    if (ld <= 32) OEA_ROT(32, 1);
    else if (ld <= 64) OEA_ROT(32, 2);
    else if (ld <= 128) OEA_ROT(32, 4);
    else if (ld <= 256) OEA_ROT(64, 4);
    else OEA_ROT(64, 8);
#undef OEA_ROT
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_rotate_lookup(const double *ent, int64_t n_ent, int32_t dim, int32_t ld, const int32_t *ids, int64_t n,
                      int32_t part_norm, int32_t sum_norm, float *out, int32_t out_ld, void *stream) {
    OEA_REQUIRE(ent && out && dim > 0 && dim <= ld && ld <= 512 && out_ld >= dim, "shapes");
    if (n == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
#define OEA_LOOK(G, IT) rotate_lookup_kernel<G, IT><<<(unsigned)std::min<int64_t>(oea::ceil_div(n, 256 / G), 16384), 256, 0, st>>>( \
        ent, n_ent, ld, ids, n, part_norm, sum_norm, out, out_ld)
    if (ld <= 32) OEA_LOOK(32, 1);
    else if (ld <= 64) OEA_LOOK(32, 2);
    else if (ld <= 128) OEA_LOOK(32, 4);
    else if (ld <= 256) OEA_LOOK(64, 4);
    else OEA_LOOK(64, 8);
#undef OEA_LOOK
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
