// triple_step.hip -- fused translational step for gfx950.
//
// Replaces one session.run([triple_loss, triple_optimizer]) of the reference
// (models/basic_model.py:222-232): 6x tf.nn.embedding_lookup on l2_normalize(table)
// (basic_model.py:89-94, modules/base/initializers.py:26), the translational losses
// (modules/base/losses.py:15-73, approaches/bootea.py:197), TF autodiff through the gather
// and the normalisation, duplicate-row summation, and Adagrad / SGD
// (modules/base/optimizers.py:4-20).
//
// Kernel 1 (triple_fwd_bwd): one G-lane group per triple (G*4*IT >= ld), each lane owns
//   float4 slices of the h, r, t rows: coalesced 16-B loads, __shfl_xor butterflies for
//   the three squared norms and the score, analytic dL/d(delta), hardware fp32 atomics
//   (global_atomic_add_f32) into a dense gradient scratch w.r.t. the NORMALISED rows.
//   Triples whose hinge is inactive issue no atomics at all.
// Kernel 2 (apply_rows): one group per table row; touched rows pull the summed gradient
//   back through the normalisation (g - y (y.g)) / |v| and apply Adagrad / SGD, then zero
//   their scratch row, so the scratch is clean for the next step.
//
// HBM / cache traffic per scored triple: 3 rows read (12*ld B) + up to 3 rows of atomics.
#include "common.h"

namespace {

using oea::group_sum;

struct StepWs {
    float *ent_grad, *rel_grad;
    float *ent_touched, *rel_touched;   // 1.0f = row received gradient (float so one SUM all-reduce covers grads + flags)
    double *partials;   // [kMaxBlocks]
};
constexpr int kMaxBlocks = 4096;

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

static size_t ws_layout(int64_t n_ent, int64_t n_rel, int32_t ld, void *base, StepWs *ws) {
    size_t off = 0;
    char *b = static_cast<char *>(base);
    auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return b ? b + o : nullptr; };
    float *eg = (float *)take(sizeof(float) * (size_t)n_ent * ld);
    float *rg = (float *)take(sizeof(float) * (size_t)n_rel * ld);
    float *et = (float *)take(sizeof(float) * (size_t)n_ent);
    float *rt = (float *)take(sizeof(float) * (size_t)n_rel);
    double *pp = (double *)take(sizeof(double) * kMaxBlocks);
    if (ws) { ws->ent_grad = eg; ws->rel_grad = rg; ws->ent_touched = et; ws->rel_touched = rt; ws->partials = pp; }
    return off;
}

template <int G, int IT>
struct Row {
    float4 v[IT];
};

template <int G, int IT>
__device__ __forceinline__ void load_row(const float *__restrict__ base, int ld, int lane, Row<G, IT> &r) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = (it * G + lane) * 4;
        r.v[it] = c < ld ? oea::ld4(base + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int G, int IT>
__device__ __forceinline__ float sumsq(const Row<G, IT> &r) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it)
        s += r.v[it].x * r.v[it].x + r.v[it].y * r.v[it].y + r.v[it].z * r.v[it].z + r.v[it].w * r.v[it].w;
    return group_sum<G>(s);
}

// delta = yh + yr - yt on normalised rows; returns the score (sum |d| or sum d^2).
template <int G, int IT>
__device__ __forceinline__ float score_triple(const float *__restrict__ ent, const float *__restrict__ rel,
                                              int ld, int lane, int h, int r, int t, int ent_norm,
                                              int rel_norm, int l1, Row<G, IT> &delta) {
    Row<G, IT> vh, vr, vt;
    load_row<G, IT>(ent + (int64_t)h * ld, ld, lane, vh);
    load_row<G, IT>(rel + (int64_t)r * ld, ld, lane, vr);
    load_row<G, IT>(ent + (int64_t)t * ld, ld, lane, vt);
    float ih = 1.f, ir = 1.f, itl = 1.f;
    if (ent_norm) {
        ih = rsqrtf(fmaxf(sumsq<G, IT>(vh), 1e-12f));
        itl = rsqrtf(fmaxf(sumsq<G, IT>(vt), 1e-12f));
    }
    if (rel_norm) ir = rsqrtf(fmaxf(sumsq<G, IT>(vr), 1e-12f));
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        float4 d;
        d.x = vh.v[it].x * ih + vr.v[it].x * ir - vt.v[it].x * itl;
        d.y = vh.v[it].y * ih + vr.v[it].y * ir - vt.v[it].y * itl;
        d.z = vh.v[it].z * ih + vr.v[it].z * ir - vt.v[it].z * itl;
        d.w = vh.v[it].w * ih + vr.v[it].w * ir - vt.v[it].w * itl;
        delta.v[it] = d;
        s += l1 ? (fabsf(d.x) + fabsf(d.y) + fabsf(d.z) + fabsf(d.w)) : (d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w);
    }
    return group_sum<G>(s);
}

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// scatter coef * ds/d(delta) into the gradient scratch: +h, +r, -t.
template <int G, int IT>
__device__ __forceinline__ void scatter_grad(float *__restrict__ eg, float *__restrict__ rg,
                                             float *__restrict__ et, float *__restrict__ rt, int ld,
                                             int lane, int h, int r, int t, float coef, int l1,
                                             const Row<G, IT> &delta) {
    float *gh = eg + (int64_t)h * ld, *gr = rg + (int64_t)r * ld, *gt = eg + (int64_t)t * ld;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = (it * G + lane) * 4;
        if (c < ld) {
            const float4 d = delta.v[it];
            float g[4];
            if (l1) { g[0] = coef * sgn(d.x); g[1] = coef * sgn(d.y); g[2] = coef * sgn(d.z); g[3] = coef * sgn(d.w); }
            else { const float c2 = 2.f * coef; g[0] = c2 * d.x; g[1] = c2 * d.y; g[2] = c2 * d.z; g[3] = c2 * d.w; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (g[q] != 0.f) {
                    oea::atomic_add_f32(gh + c + q, g[q]);
                    oea::atomic_add_f32(gr + c + q, g[q]);
                    oea::atomic_add_f32(gt + c + q, -g[q]);
                }
            }
        }
    }
    if (lane == 0) { et[h] = 1.f; et[t] = 1.f; rt[r] = 1.f; }
}

__device__ __forceinline__ float softplusf_(float x) { return x > 0.f ? x + log1pf(expf(-x)) : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

template <int G, int IT>
__global__ __launch_bounds__(256) void triple_fwd_bwd(
    const float *__restrict__ ent, const float *__restrict__ rel, int ld, const int32_t *__restrict__ pos,
    int64_t n_pos, const int32_t *__restrict__ neg, int64_t n_neg, oea_step_cfg cfg, StepWs ws) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    const bool margin = cfg.loss_kind == OEA_LOSS_MARGIN;
    const int64_t items = margin ? n_pos : n_pos + n_neg;
    double loss_local = 0.0;

    for (int64_t item = grp; item < items; item += ngrp) {
        if (margin) {
            // losses.py:15-27: sum relu(margin + s+ - s-), pos i paired with neg i
            const int32_t *tp = pos + 3 * item, *tn = neg + 3 * item;
            const int ph = tp[0], pr = tp[1], pt = tp[2], nh = tn[0], nr = tn[1], nt = tn[2];
            Row<G, IT> dp, dn;
            const float sp = score_triple<G, IT>(ent, rel, ld, lane, ph, pr, pt, cfg.ent_l2_norm, cfg.rel_l2_norm, cfg.l1, dp);
            const float sn = score_triple<G, IT>(ent, rel, ld, lane, nh, nr, nt, cfg.ent_l2_norm, cfg.rel_l2_norm, cfg.l1, dn);
            const float x = cfg.margin + sp - sn;
            if (x > 0.f) {
                if (lane == 0) loss_local += (double)x;
                scatter_grad<G, IT>(ws.ent_grad, ws.rel_grad, ws.ent_touched, ws.rel_touched, ld, lane, ph, pr, pt, 1.f, cfg.l1, dp);
                scatter_grad<G, IT>(ws.ent_grad, ws.rel_grad, ws.ent_touched, ws.rel_touched, ld, lane, nh, nr, nt, -1.f, cfg.l1, dn);
            }
            continue;
        }
        const bool is_pos = item < n_pos;
        const int32_t *tr = is_pos ? pos + 3 * item : neg + 3 * (item - n_pos);
        const int h = tr[0], r = tr[1], t = tr[2];
        Row<G, IT> delta;
        const float s = score_triple<G, IT>(ent, rel, ld, lane, h, r, t, cfg.ent_l2_norm, cfg.rel_l2_norm, cfg.l1, delta);
        float coef = 0.f, l = 0.f;
        switch (cfg.loss_kind) {
        case OEA_LOSS_LIMITED:  // losses.py:53-55
            if (is_pos) { const float x = s - cfg.pos_margin; if (x > 0.f) { l = x; coef = 1.f; } }
            else { const float x = cfg.neg_margin - s; if (x > 0.f) { l = cfg.balance * x; coef = -cfg.balance; } }
            break;
        case OEA_LOSS_LOGISTIC:  // losses.py:70-72
            if (is_pos) { l = softplusf_(s); coef = sigmoidf_(s); }
            else { l = softplusf_(-s); coef = -sigmoidf_(-s); }
            break;
        case OEA_LOSS_POSITIVE:  // losses.py:38
            l = s; coef = 1.f;
            break;
        case OEA_LOSS_ALIGN:  // bootea.py:197: -log sigmoid(-s) = softplus(s)
            l = softplusf_(s); coef = sigmoidf_(s);
            break;
        default: break;
        }
        if (lane == 0) loss_local += (double)l;
        if (coef != 0.f)
            scatter_grad<G, IT>(ws.ent_grad, ws.rel_grad, ws.ent_touched, ws.rel_touched, ld, lane, h, r, t, coef, cfg.l1, delta);
    }

    // block-level loss partial (fixed reduction tree -> the partial is deterministic)
    __shared__ double sred[4];
    double w = oea::wave_sum_d(loss_local);
    const int wid = threadIdx.x / 64;
    if ((threadIdx.x & 63) == 0) sred[wid] = w;
    __syncthreads();
    if (threadIdx.x == 0) ws.partials[blockIdx.x] = sred[0] + sred[1] + sred[2] + sred[3];
}

// One G-lane group per table row (entity rows first, then relation rows).
template <int G, int IT>
__global__ __launch_bounds__(256) void apply_rows(float *__restrict__ ent, float *__restrict__ ent_acc,
                                                  int64_t n_ent, float *__restrict__ rel,
                                                  float *__restrict__ rel_acc, int64_t n_rel, int ld,
                                                  oea_step_cfg cfg, StepWs ws, int n_partials,
                                                  double *__restrict__ loss_accum) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t row_all = grp; row_all < n_ent + n_rel; row_all += ngrp) {
        const bool is_rel = row_all >= n_ent;
        const int64_t row = is_rel ? row_all - n_ent : row_all;
        float *touched = is_rel ? ws.rel_touched : ws.ent_touched;
        if (touched[row] == 0.f) continue;
        float *v = (is_rel ? rel : ent) + row * ld;
        float *acc = (is_rel ? rel_acc : ent_acc) + row * ld;
        float *g = (is_rel ? ws.rel_grad : ws.ent_grad) + row * ld;
        const int on = is_rel ? cfg.rel_l2_norm : cfg.ent_l2_norm;
        Row<G, IT> rv, rg;
        load_row<G, IT>(v, ld, lane, rv);
        load_row<G, IT>(g, ld, lane, rg);
        float inv = 1.f, ydg = 0.f;
        if (on) {
            const float ss = sumsq<G, IT>(rv);
            inv = rsqrtf(fmaxf(ss, 1e-12f));
            float dot = 0.f;
#pragma unroll
            for (int it = 0; it < IT; ++it)
                dot += rv.v[it].x * rg.v[it].x + rv.v[it].y * rg.v[it].y + rv.v[it].z * rg.v[it].z + rv.v[it].w * rg.v[it].w;
            dot = group_sum<G>(dot) * inv;           // y . g
            ydg = ss > 1e-12f ? dot : 0.f;
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * G + lane) * 4;
            if (c < ld) {
                const float vv[4] = {rv.v[it].x, rv.v[it].y, rv.v[it].z, rv.v[it].w};
                const float gg[4] = {rg.v[it].x, rg.v[it].y, rg.v[it].z, rg.v[it].w};
                float nv[4], na[4];
                float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cfg.opt_kind == OEA_OPT_ADAGRAD) a4 = oea::ld4(acc + c);
                const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float gv = on ? (gg[q] - vv[q] * inv * ydg) * inv : gg[q];
                    if (cfg.opt_kind == OEA_OPT_ADAGRAD) {
                        na[q] = aa[q] + gv * gv;
                        nv[q] = vv[q] - cfg.lr * gv / sqrtf(na[q]);
                    } else {
                        na[q] = 0.f;
                        nv[q] = vv[q] - cfg.lr * gv;
                    }
                }
                oea::st4(v + c, make_float4(nv[0], nv[1], nv[2], nv[3]));
                if (cfg.opt_kind == OEA_OPT_ADAGRAD) oea::st4(acc + c, make_float4(na[0], na[1], na[2], na[3]));
                oea::st4(g + c, make_float4(0.f, 0.f, 0.f, 0.f));
            }
        }
        if (lane == 0) touched[row] = 0.f;
    }
    // fixed-order reduction of the loss partials by one wave of block 0
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        double s = 0.0;
        for (int i = threadIdx.x; i < n_partials; i += 64) s += ws.partials[i];
        s = oea::wave_sum_d(s);
        if (threadIdx.x == 0) *loss_accum += s;
    }
}

template <int G, int IT>
int launch_step(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel,
                int32_t ld, const int32_t *pos, int64_t n_pos, const int32_t *neg, int64_t n_neg,
                const oea_step_cfg &cfg, const StepWs &ws, double *loss_accum, int phase, hipStream_t st) {
    const int block = 256, gpb = block / G;
    const int64_t items = cfg.loss_kind == OEA_LOSS_MARGIN ? n_pos : n_pos + n_neg;
    const int nb1 = (int)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(items, gpb), 1), kMaxBlocks);
    oea::prof_mark(st);
    if (phase != OEA_PHASE_APPLY)
        triple_fwd_bwd<G, IT><<<nb1, block, 0, st>>>(ent, rel, ld, pos, n_pos, neg, n_neg, cfg, ws);
    oea::prof_mark(st);
    const int nb2 = (int)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(n_ent + n_rel, gpb), 1), 16384);
    if (phase != OEA_PHASE_GRAD)
        apply_rows<G, IT><<<nb2, block, 0, st>>>(ent, ent_acc, n_ent, rel, rel_acc, n_rel, ld, cfg, ws, nb1, loss_accum);
    oea::prof_mark(st);
    return 0;
}

}  // namespace

extern "C" {

size_t oea_step_workspace_bytes(int64_t n_ent, int64_t n_rel, int32_t ld) {
    return ws_layout(n_ent, n_rel, ld, nullptr, nullptr);
}

size_t oea_step_exchange_floats(int64_t n_ent, int64_t n_rel, int32_t ld) {
    StepWs ws;
    ws_layout(n_ent, n_rel, ld, reinterpret_cast<void *>(256), &ws);   // fake base: only offsets matter
    return (size_t)(reinterpret_cast<char *>(ws.partials) - reinterpret_cast<char *>(256)) / sizeof(float);
}

int oea_triple_step(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc,
                    int64_t n_rel, int32_t dim, int32_t ld, const int32_t *pos, int64_t n_pos,
                    const int32_t *neg, int64_t n_neg, const oea_step_cfg *cfg, void *workspace,
                    double *loss_accum, void *stream) {
    return oea_triple_step_phase(ent, ent_acc, n_ent, rel, rel_acc, n_rel, dim, ld, pos, n_pos, neg, n_neg, cfg,
                                 workspace, loss_accum, OEA_PHASE_BOTH, stream);
}

int oea_triple_step_phase(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc,
                          int64_t n_rel, int32_t dim, int32_t ld, const int32_t *pos, int64_t n_pos,
                          const int32_t *neg, int64_t n_neg, const oea_step_cfg *cfg, void *workspace,
                          double *loss_accum, int32_t phase, void *stream) {
    OEA_REQUIRE(phase >= OEA_PHASE_BOTH && phase <= OEA_PHASE_APPLY, "phase");
    OEA_REQUIRE(ent && rel && pos && cfg && workspace && loss_accum, "null pointer");
    OEA_REQUIRE(ld % 4 == 0 && dim <= ld && dim > 0, "ld % 4 == 0 and dim <= ld");
    OEA_REQUIRE(n_pos >= 0 && n_neg >= 0 && (neg || n_neg == 0), "neg == NULL needs n_neg == 0");
    OEA_REQUIRE(cfg->loss_kind >= OEA_LOSS_MARGIN && cfg->loss_kind <= OEA_LOSS_ALIGN, "loss_kind");
    OEA_REQUIRE(cfg->opt_kind == OEA_OPT_SGD || cfg->opt_kind == OEA_OPT_ADAGRAD, "opt_kind");
    OEA_REQUIRE(cfg->opt_kind != OEA_OPT_ADAGRAD || (ent_acc && rel_acc), "Adagrad needs accumulators");
    if (cfg->loss_kind == OEA_LOSS_MARGIN) OEA_REQUIRE(n_neg == n_pos, "margin loss pairs pos i with neg i");
    if (cfg->loss_kind == OEA_LOSS_POSITIVE || cfg->loss_kind == OEA_LOSS_ALIGN)
        OEA_REQUIRE(n_neg == 0, "positive-only loss takes no negatives");
    if (n_pos + n_neg == 0) return OEA_OK;
    StepWs ws;
    ws_layout(n_ent, n_rel, ld, workspace, &ws);
    hipStream_t st = oea::as_stream(stream);
#define OEA_STEP(G, IT) launch_step<G, IT>(ent, ent_acc, n_ent, rel, rel_acc, n_rel, ld, pos, n_pos, neg, n_neg, *cfg, ws, loss_accum, phase, st)
    if (ld <= 64) OEA_STEP(16, 1);
    else if (ld <= 128) OEA_STEP(32, 1);
    else if (ld <= 256) OEA_STEP(64, 1);
    else if (ld <= 512) OEA_STEP(64, 2);
    else if (ld <= 1280) OEA_STEP(64, 5);
    else { oea::set_error("dim %d > 1280 unsupported", dim); return OEA_EUNSUPPORTED; }
#undef OEA_STEP
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
