// triple_step.hip -- fused translational step for gfx950.
//
// Replaces one session.run([triple_loss, triple_optimizer]) of the reference
// (models/basic_model.py:222-232): 6x tf.nn.embedding_lookup on l2_normalize(table)
// (basic_model.py:89-94, modules/base/initializers.py:26), the translational losses
// (modules/base/losses.py:15-73, approaches/bootea.py:197), TF autodiff through the gather
// and the normalisation, duplicate-row summation, and Adagrad / SGD
// (modules/base/optimizers.py:4-20).
//
// Layout: a G-lane group (G = 32 for ld <= 128, else 64) owns one work item; lane l holds
// elements l, l+G, l+2G, ... of a row, so every load AND every atomic instruction of the
// group covers one contiguous 128-B / 256-B segment of a table row.
//
// Kernel 1a (triple_grouped): one group per POSITIVE and its k negatives, used when the
//   negatives come from oea_sample_negatives (neg[p*k+s] corrupts head or tail of pos p,
//   batch.py:89-119).  h, r, t are loaded and normalised once; each negative costs one more
//   row; the gradients of h, r, t are accumulated in registers and leave as ONE row of
//   atomics each: (3+k) row reads and (3+k) rows of atomics per positive instead of
//   3(1+k) -- and the hot relation rows (a few hundred rows shared by the whole batch) see
//   (1+k)x fewer same-address atomics.
// Kernel 1b (triple_generic): one group per triple for arbitrary (pos, neg) lists, and the
//   margin loss (pos i paired with neg i).
//   Both: hardware fp32 atomics (global_atomic_add_f32) into a dense gradient scratch w.r.t.
//   the NORMALISED rows; triples whose hinge is inactive issue no atomics.
// Kernel 2 (apply_rows): one group per table row; touched rows pull the summed gradient
//   back through the normalisation (g - y (y.g)) / |v| and apply Adagrad / SGD, then zero
//   their scratch row, so the scratch is clean for the next step.
//
// Algorithmic bytes per scored triple: 3 rows read + 3 rows of gradient = 24*d.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "step_plan.h"

namespace {

using oea::group_sum;

using oea::flag_t;
using oea::grad_t;      // float, or int64 fixed point in the deterministic build (common.h)

struct StepWs {
    grad_t *ent_grad, *rel_grad;        // rel_grad: copy 0 of the relation scratch [n_rel][ld]
    grad_t *rel_extra;                  // copies 1 .. kRelCopies-1, [kRelCopies-1][n_rel][ld]
    int64_t rel_copy_stride;            // n_rel * ld
    flag_t *ent_touched, *rel_touched;  // 1 = row received gradient (same type as the gradients: one SUM all-reduce covers both)
    grad_t *nrm_grad, *nrm_extra;       // TransH normal vectors: copy 0 / copies 1.. (same shapes as the relation scratch)
    flag_t *nrm_touched;
    double *partials;                   // [kMaxBlocks]
    // layout: [ent_grad | rel_grad (copy 0) | nrm_grad (copy 0) | ent_touched | rel_touched | nrm_touched] is the
    // contiguous prefix that data-parallel ranks all-reduce (the extra copies are folded into copy 0 first);
    // [rel_extra | nrm_extra | partials] follow.
    __device__ __forceinline__ grad_t *rel_copy(int64_t c) const {
        return c == 0 ? rel_grad : rel_extra + (c - 1) * rel_copy_stride;
    }
    __device__ __forceinline__ grad_t *nrm_copy(int64_t c) const {
        return c == 0 ? nrm_grad : nrm_extra + (c - 1) * rel_copy_stride;
    }
};
constexpr int kMaxBlocks = 4096;
// Relation rows are few (a few hundred) and shared by the whole batch: with one scratch row per
// relation the hottest relation took ~750 same-address atomics per step (19 of 45 us, measured).
// The relation scratch is therefore replicated kRelCopies times; a work item adds into copy
// (item index % kRelCopies) and apply_rows sums the copies.
constexpr int kRelCopies = 16;

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

static size_t ws_layout(int64_t n_ent, int64_t n_rel, int32_t ld, void *base, StepWs *ws) {
    size_t off = 0;
    char *b = static_cast<char *>(base);
    auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return b ? b + o : nullptr; };
    grad_t *eg = (grad_t *)take(sizeof(grad_t) * (size_t)n_ent * ld);
    grad_t *rg = (grad_t *)take(sizeof(grad_t) * (size_t)n_rel * ld);
    grad_t *ng = (grad_t *)take(sizeof(grad_t) * (size_t)n_rel * ld);
    flag_t *et = (flag_t *)take(sizeof(flag_t) * (size_t)n_ent);
    flag_t *rt = (flag_t *)take(sizeof(flag_t) * (size_t)n_rel);
    flag_t *nt = (flag_t *)take(sizeof(flag_t) * (size_t)n_rel);
    grad_t *rx = (grad_t *)take(sizeof(grad_t) * (size_t)n_rel * ld * (kRelCopies - 1));
    grad_t *nx = (grad_t *)take(sizeof(grad_t) * (size_t)n_rel * ld * (kRelCopies - 1));
    double *pp = (double *)take(sizeof(double) * kMaxBlocks);
    if (ws) {
        ws->rel_copy_stride = n_rel * (int64_t)ld; ws->ent_grad = eg; ws->rel_grad = rg; ws->rel_extra = rx;
        ws->ent_touched = et; ws->rel_touched = rt; ws->nrm_grad = ng; ws->nrm_extra = nx; ws->nrm_touched = nt; ws->partials = pp;
    }
    return off;
}

// ---- lane-strided row fragments ------------------------------------------------------------------
template <int G, int IT>
struct Row {
    float v[IT];
};

// TIGHT (G == 32 only): the dispatch gives ld > (IT - 1) * 32, so only the last fragment needs the bound.  Used by the
// grouped scoring kernel (18.8 vs 19.2 us at the 15K shape, 64 vs 75 us at the 100K shape); apply_rows is SLOWER with it
// (13.2 vs 11.3 us, 83 vs 64 us: the optimiser's three row streams schedule worse as unconditional loads) and keeps the bound.
template <int G, int IT, bool TIGHT = false>
__device__ __forceinline__ void load_row(const float *__restrict__ base, int ld, int lane, Row<G, IT> &r) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = it * G + lane;
        r.v[it] = ((TIGHT && G == 32 && it < IT - 1) || c < ld) ? base[c] : 0.f;
    }
}
// a row of the gradient scratch: raw elements (exact sums of the relation copies), then ONE conversion to fp32
template <int G, int IT>
__device__ __forceinline__ void load_grad_raw(const grad_t *__restrict__ base, int ld, int lane, grad_t (&q)[IT]) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = it * G + lane;
        q[it] = c < ld ? base[c] : (grad_t)0;
    }
}
template <int G, int IT>
__device__ __forceinline__ void load_grad_row(const grad_t *__restrict__ base, int ld, int lane, Row<G, IT> &r) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = it * G + lane;
        r.v[it] = c < ld ? oea::grad_val(base[c]) : 0.f;
    }
}
template <int G, int IT>
__device__ __forceinline__ float sumsq(const Row<G, IT> &r) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) s += r.v[it] * r.v[it];
    return group_sum<G>(s);
}
// y = l2_normalize(v) when `on` (tf.nn.l2_normalize: v * rsqrt(max(sum v^2, 1e-12)))
template <int G, int IT>
__device__ __forceinline__ void normalize(Row<G, IT> &r, int on) {
    if (!on) return;
    const float inv = rsqrtf(fmaxf(sumsq<G, IT>(r), 1e-12f));
#pragma unroll
    for (int it = 0; it < IT; ++it) r.v[it] *= inv;
}
template <int G, int IT>
__device__ __forceinline__ float score(const Row<G, IT> &yh, const Row<G, IT> &yr, const Row<G, IT> &yt, int l1,
                                       Row<G, IT> &delta) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const float d = yh.v[it] + yr.v[it] - yt.v[it];
        delta.v[it] = d;
        s += l1 ? fabsf(d) : d * d;
    }
    return group_sum<G>(s);
}
__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// g = coef * ds/d(delta)
template <int G, int IT>
__device__ __forceinline__ void dscore(const Row<G, IT> &delta, float coef, int l1, Row<G, IT> &g) {
#pragma unroll
    for (int it = 0; it < IT; ++it) g.v[it] = l1 ? coef * sgn(delta.v[it]) : 2.f * coef * delta.v[it];
}
// SKIPZERO: elements whose gradient is exactly 0 issue no atomic (L1 norm: sgn(0); a per-element branch) -- the grouped
// kernel's L2 paths turn it off (a zero there is a measure-zero event and the branches cost more than the atomics)
template <int G, int IT, bool SKIPZERO = true>
__device__ __forceinline__ void atomic_row(grad_t *__restrict__ dst, int ld, int lane, const Row<G, IT> &g, float sign) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = it * G + lane;
        const float v = sign * g.v[it];
        if (((G == 32 && it < IT - 1) || c < ld) && (!SKIPZERO || v != 0.f)) oea::grad_add(dst + c, v);
    }
}

__device__ __forceinline__ float softplusf_(float x) { return x > 0.f ? x + log1pf(expf(-x)) : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// dL/ds and the loss term of ONE triple for the per-triple losses (not margin).
__device__ __forceinline__ void triple_coef_kind(int loss_kind, const oea_step_cfg &cfg, bool is_pos, float s, float &coef,
                                                 float &l) {
    coef = 0.f; l = 0.f;
    switch (loss_kind) {
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
}
__device__ __forceinline__ void triple_coef(const oea_step_cfg &cfg, bool is_pos, float s, float &coef, float &l) {
    triple_coef_kind(cfg.loss_kind, cfg, is_pos, s, coef, l);
}

__device__ __forceinline__ void block_loss_partial(double loss_local, double *partials) {
    // fixed reduction tree -> the partial is deterministic
    __shared__ double sred[4];
    const double w = oea::wave_sum_d(loss_local);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = sred[0] + sred[1] + sred[2] + sred[3];
}

// ---- kernel 1b: arbitrary triple lists + margin pairs --------------------------------------------
template <int G, int IT>
__global__ __launch_bounds__(256) void triple_generic(
    const float *__restrict__ ent, const float *__restrict__ rel, int ld, const int32_t *__restrict__ pos,
    int64_t n_pos, const int32_t *__restrict__ neg, int64_t n_neg, oea_step_cfg cfg, StepWs ws) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    const bool margin = cfg.loss_kind == OEA_LOSS_MARGIN;
    const int64_t items = margin ? n_pos : n_pos + n_neg;
    double loss_local = 0.0;
    for (int64_t item = grp; item < items; item += ngrp) {
        const bool is_pos = margin || item < n_pos;
        const int32_t *tr = is_pos ? pos + 3 * item : neg + 3 * (item - n_pos);
        const int h = tr[0], r = tr[1], t = tr[2];
        Row<G, IT> yh, yr, yt, delta, g;
        load_row<G, IT>(ent + (int64_t)h * ld, ld, lane, yh);
        load_row<G, IT>(rel + (int64_t)r * ld, ld, lane, yr);
        load_row<G, IT>(ent + (int64_t)t * ld, ld, lane, yt);
        normalize<G, IT>(yh, cfg.ent_l2_norm);
        normalize<G, IT>(yr, cfg.rel_l2_norm);
        normalize<G, IT>(yt, cfg.ent_l2_norm);
        const float s = score<G, IT>(yh, yr, yt, cfg.l1, delta);
        float coef, l;
        if (margin) {
            // losses.py:15-27: sum relu(margin + s+ - s-), pos i paired with neg i
            const int32_t *tn = neg + 3 * item;
            const int nh = tn[0], nr = tn[1], nt = tn[2];
            Row<G, IT> zh, zr, zt, dn;
            load_row<G, IT>(ent + (int64_t)nh * ld, ld, lane, zh);
            load_row<G, IT>(rel + (int64_t)nr * ld, ld, lane, zr);
            load_row<G, IT>(ent + (int64_t)nt * ld, ld, lane, zt);
            normalize<G, IT>(zh, cfg.ent_l2_norm);
            normalize<G, IT>(zr, cfg.rel_l2_norm);
            normalize<G, IT>(zt, cfg.ent_l2_norm);
            const float sn = score<G, IT>(zh, zr, zt, cfg.l1, dn);
            const float x = cfg.margin + s - sn;
            if (x <= 0.f) continue;
            if (lane == 0) loss_local += (double)x;
            dscore<G, IT>(dn, -1.f, cfg.l1, g);
            atomic_row<G, IT>(ws.ent_grad + (int64_t)nh * ld, ld, lane, g, 1.f);
            atomic_row<G, IT>(ws.rel_copy(item % kRelCopies) + (int64_t)nr * ld, ld, lane, g, 1.f);
            atomic_row<G, IT>(ws.ent_grad + (int64_t)nt * ld, ld, lane, g, -1.f);
            if (lane == 0) { ws.ent_touched[nh] = 1.f; ws.ent_touched[nt] = 1.f; ws.rel_touched[nr] = 1.f; }
            coef = 1.f;
        } else {
            triple_coef(cfg, is_pos, s, coef, l);
            if (lane == 0) loss_local += (double)l;
            if (coef == 0.f) continue;
        }
        dscore<G, IT>(delta, coef, cfg.l1, g);
        atomic_row<G, IT>(ws.ent_grad + (int64_t)h * ld, ld, lane, g, 1.f);
        atomic_row<G, IT>(ws.rel_copy(item % kRelCopies) + (int64_t)r * ld, ld, lane, g, 1.f);
        atomic_row<G, IT>(ws.ent_grad + (int64_t)t * ld, ld, lane, g, -1.f);
        if (lane == 0) { ws.ent_touched[h] = 1.f; ws.ent_touched[t] = 1.f; ws.rel_touched[r] = 1.f; }
    }
    block_loss_partial(loss_local, ws.partials);
}

// ---- kernel 1a: one group per positive + its k negatives -------------------------------------------
// The k corrupted rows are requested KC at a time BEFORE any of them is consumed, so a group pays
// one memory latency per chunk instead of one per negative (the first version walked the
// negatives with a prefetch depth of one and was latency-bound: 38 us for 5,000 x 11 triples).
// score one (h, r, t) as an independent triple (entries of a grouped batch that are not corruptions of
// their positive): 3 gathers, 3 atomics.
template <int G, int IT>
__device__ __forceinline__ double score_independent(const float *__restrict__ ent, const float *__restrict__ rel, int ld,
                                                    int lane, int64_t item, int ch, int cr, int ct, bool is_pos,
                                                    const oea_step_cfg &cfg, const StepWs &ws, int lk, int l1) {
    Row<G, IT> zh, zr, zt, delta, g;
    load_row<G, IT>(ent + (int64_t)ch * ld, ld, lane, zh);
    load_row<G, IT>(rel + (int64_t)cr * ld, ld, lane, zr);
    load_row<G, IT>(ent + (int64_t)ct * ld, ld, lane, zt);
    normalize<G, IT>(zh, cfg.ent_l2_norm);
    normalize<G, IT>(zr, cfg.rel_l2_norm);
    normalize<G, IT>(zt, cfg.ent_l2_norm);
    const float s = score<G, IT>(zh, zr, zt, l1, delta);
    float coef, l;
    triple_coef_kind(lk, cfg, is_pos, s, coef, l);
    if (coef != 0.f) {
        dscore<G, IT>(delta, coef, l1, g);
        atomic_row<G, IT>(ws.ent_grad + (int64_t)ch * ld, ld, lane, g, 1.f);
        atomic_row<G, IT>(ws.rel_copy(item % kRelCopies) + (int64_t)cr * ld, ld, lane, g, 1.f);
        atomic_row<G, IT>(ws.ent_grad + (int64_t)ct * ld, ld, lane, g, -1.f);
        if (lane == 0) { ws.ent_touched[ch] = 1.f; ws.ent_touched[ct] = 1.f; ws.rel_touched[cr] = 1.f; }
    }
    return (double)l;
}

// The step is latency-bound (a few thousand short waves, ~1,800 instructions each), so the kernel is
// organised around DEPENDENT ROUND TRIPS, two per positive: (1) the ids -- the positive's three and
// its negatives' 3k, fetched by the lanes of the group in one coalesced load each and handed round
// with cross-lane reads; (2) every row the group needs, 3 + k gathers issued back to back.  The
// arithmetic after that is branch-free per negative so the k reduction chains interleave.
// Occupancy: a batch of 5,000 positives is 2,500 waves; at 2 waves/SIMD the chip holds 2,048, and the 452
// late-comers doubled the kernel's critical path (25.9 us).  Capping the registers at 3 waves/SIMD (a few
// spilled dwords for IT >= 2) lets every wave start at once: 19.4 us.
// Measured and dropped (gpurun_out r02f, same binary, env-selected): 4 waves / SIMD (<= 128 VGPRs, ~40 spill accesses)
// 20.1 vs 17.8 us at the 15K shape and 90 vs 65 us at the 100K shape; no per-element zero test in front of the atomics
// (100 fewer VALU instructions) 19.4 vs 17.8 us -- the test skips WHOLE rows often (a positive inside its margin with only
// tail-corrupted negatives active leaves gt all zero).
template <int G, int IT, int LOSS, int L1>
__global__ __launch_bounds__(256, (IT <= 4 ? 3 : 2)) void triple_grouped(
    const float *__restrict__ ent, const float *__restrict__ rel, int ld, const int32_t *__restrict__ pos,
    int64_t n_pos, const int32_t *__restrict__ neg, int k, oea_step_cfg cfg, StepWs ws) {
    constexpr int KC = IT <= 4 ? 10 : (IT <= 8 ? 4 : 2);      // negatives in flight: KC * IT row registers
    static_assert(3 * KC <= G, "ids of a chunk fit one register across the group");
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    // LOSS / L1 >= 0: the loss kind and the norm are compile-time constants (one compact code path per kind: with the
    // run-time switch every call site carried the exp / log1p expansions of the logistic losses, 12,000 instructions)
    const int lk = LOSS >= 0 ? LOSS : cfg.loss_kind;
    const int l1 = L1 >= 0 ? L1 : cfg.l1;
    double loss_local = 0.0;
    for (int64_t p = grp; p < n_pos; p += ngrp) {
        const int32_t *ng = neg + (int64_t)p * k * 3;
        // ---- round trip 1: ids -------------------------------------------------------------------------
        const int pid = lane < 3 ? pos[3 * p + lane] : 0;
        int nid = lane < 3 * min(KC, k) ? ng[lane] : 0;
        const int h = __shfl(pid, 0, G), r = __shfl(pid, 1, G), t = __shfl(pid, 2, G);
        Row<G, IT> yh, yr, yt, delta, g, gh, gr, gt;
        load_row<G, IT, true>(ent + (int64_t)h * ld, ld, lane, yh);
        load_row<G, IT, true>(rel + (int64_t)r * ld, ld, lane, yr);
        load_row<G, IT, true>(ent + (int64_t)t * ld, ld, lane, yt);
        double lsum = 0.0;
        bool any = false;
        for (int base = 0; base < k; base += KC) {
            if (base > 0) nid = lane < 3 * min(KC, k - base) ? ng[3 * base + lane] : 0;
            int ce[KC];                                                // the corrupted entity of slot j
            unsigned tailm = 0, okm = 0, slow = 0;                     // per-slot flags as bit masks (registers are tight)
            Row<G, IT> yc[KC];
            // ---- round trip 2: the corrupted rows (and, first time round, the positive's) ----------------
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                const int ch = __shfl(nid, 3 * j, G), cr = __shfl(nid, 3 * j + 1, G), ct = __shfl(nid, 3 * j + 2, G);
                const bool valid = base + j < k;
                const bool tl = ch == h;                               // tail corrupted (or neg == pos): uses yh, yr
                const bool ok = valid && cr == r && (tl || ct == t);
                tailm |= tl ? 1u << j : 0u;
                okm |= ok ? 1u << j : 0u;
                slow |= (valid && !ok) ? 1u << j : 0u;                 // entries that are not corruptions of this positive
                ce[j] = valid ? (tl ? ct : ch) : h;
                load_row<G, IT, true>(ent + (int64_t)ce[j] * ld, ld, lane, yc[j]);
            }
            if (base == 0) {
                normalize<G, IT>(yh, cfg.ent_l2_norm);
                normalize<G, IT>(yr, cfg.rel_l2_norm);
                normalize<G, IT>(yt, cfg.ent_l2_norm);
                const float s = score<G, IT>(yh, yr, yt, l1, delta);
                float coef, l;
                triple_coef_kind(lk, cfg, true, s, coef, l);
                lsum = (double)l;
                dscore<G, IT>(delta, coef, l1, g);
#pragma unroll
                for (int it = 0; it < IT; ++it) { gh.v[it] = g.v[it]; gr.v[it] = g.v[it]; gt.v[it] = -g.v[it]; }
                any = coef != 0.f;
            }
            if (slow) {                                                // rare: one copy of the code, ids re-read across lanes
#pragma unroll 1
                for (int j = 0; j < KC; ++j)
                    if ((slow >> j) & 1u)
                        lsum += score_independent<G, IT>(ent, rel, ld, lane, p, __shfl(nid, 3 * j, G), __shfl(nid, 3 * j + 1, G),
                                                         __shfl(nid, 3 * j + 2, G), false, cfg, ws, lk, l1);
            }
            float sc[KC];
#pragma unroll
            for (int j = 0; j < KC; ++j) {                             // branch-free: k independent chains
                normalize<G, IT>(yc[j], cfg.ent_l2_norm);
                float s = 0.f;
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const bool tl = (tailm >> j) & 1u;
                    const float a = tl ? yh.v[it] : yc[j].v[it], b = tl ? yc[j].v[it] : yt.v[it];
                    const float d = a + yr.v[it] - b;
                    yc[j].v[it] = d;                                    // the row is not needed again: keep delta in place
                    s += l1 ? fabsf(d) : d * d;
                }
                sc[j] = group_sum<G>(s);
            }
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                if (!((okm >> j) & 1u)) continue;
                const bool tl = (tailm >> j) & 1u;
                float coef, l;
                triple_coef_kind(lk, cfg, false, sc[j], coef, l);
                lsum += (double)l;
                if (coef != 0.f) {
                    any = true;
                    dscore<G, IT>(yc[j], coef, l1, g);
                    if (tl) {
#pragma unroll
                        for (int it = 0; it < IT; ++it) { gh.v[it] += g.v[it]; gr.v[it] += g.v[it]; }
                    } else {
#pragma unroll
                        for (int it = 0; it < IT; ++it) { gr.v[it] += g.v[it]; gt.v[it] -= g.v[it]; }
                    }
                    atomic_row<G, IT>(ws.ent_grad + (int64_t)ce[j] * ld, ld, lane, g, tl ? -1.f : 1.f);
                    if (lane == 0) ws.ent_touched[ce[j]] = 1.f;
                }
            }
        }
        if (k <= 0) {       // no negatives: the positive alone
            normalize<G, IT>(yh, cfg.ent_l2_norm);
            normalize<G, IT>(yr, cfg.rel_l2_norm);
            normalize<G, IT>(yt, cfg.ent_l2_norm);
            const float s = score<G, IT>(yh, yr, yt, l1, delta);
            float coef, l;
            triple_coef_kind(lk, cfg, true, s, coef, l);
            lsum = (double)l;
            dscore<G, IT>(delta, coef, l1, g);
#pragma unroll
            for (int it = 0; it < IT; ++it) { gh.v[it] = g.v[it]; gr.v[it] = g.v[it]; gt.v[it] = -g.v[it]; }
            any = coef != 0.f;
        }
        if (any) {
            atomic_row<G, IT>(ws.ent_grad + (int64_t)h * ld, ld, lane, gh, 1.f);
            atomic_row<G, IT>(ws.rel_copy(p % kRelCopies) + (int64_t)r * ld, ld, lane, gr, 1.f);
            atomic_row<G, IT>(ws.ent_grad + (int64_t)t * ld, ld, lane, gt, 1.f);
            if (lane == 0) { ws.ent_touched[h] = 1.f; ws.ent_touched[t] = 1.f; ws.rel_touched[r] = 1.f; }
        }
        if (lane == 0) loss_local += lsum;
    }
    block_loss_partial(loss_local, ws.partials);
}

// ---- kernel 1a'', round 6: one WAVE per positive ----------------------------------------------------------------------------------------
// triple_grouped puts two positives on a wave (one per 32-lane half).  Everything that depends on the ids -- row addresses, "is this
// negative a tail or a head corruption", "is its hinge active" -- is then per-LANE state: 64-bit address arithmetic on the VALU for every
// row, ids handed round by ds_bpermute, every conditional atomic under its own exec mask (of its 2,900 instructions 1,100 are scalar
// bookkeeping, 220 of them s_nop), 168 registers = three waves per SIMD.  Here a wave owns ONE positive:
//   * ids arrive by SCALAR loads (the wave index is uniform: s_load_dwordx16 + x8 + x4 + x2 for the 30 ids of 10 negatives);
//   * rows move through BUFFER instructions: one resource per table, the row's byte offset in the instruction's scalar offset (one
//     s_mul_i32 per row), lane * 4 in the vector offset, fragment * 256 in the immediate -- no address arithmetic on the VALU at all.
//     The lanes whose column of the LAST fragment lies past the row's end carry an out-of-range vector offset for that fragment:
//     the addressing hardware returns 0 for their loads and drops their atomics -- no exec masks, no clamps, no selects;
//   * all control flow is wave-uniform (s_cbranch_scc / vcc): an inactive hinge skips its atomics without touching exec;
//   * the sampler corrupts ONE side per round (batch.py:101-107), so the k negatives of a positive are almost always all tail
//     corruptions or all head corruptions: one uniform test picks a select-free loop (tail: d = (h + r) - c with h + r computed
//     once; head: d = (c + r) - t), and gh == gr (tail) / gt == -gr (head) bit for bit, so ONE accumulator serves both rows.
//     Positives with mixed sides, or with entries that are not corruptions of them at all, are scored as 1 + k independent triples
//     (score_independent: the same loss and gradient, three rows per triple);
//   * lane-strided fragments of 64 columns: ceil(ld / 64) registers per row (2 at d = 100, packed-fp32 instructions) -> ~64 registers.
// The scalar unit is shared by the CU's four SIMDs (one instruction per cycle): the scalar work per positive (~250 instructions)
// matters as much as the vector work (~400).
// Arithmetic per element as in triple_grouped (normalise, a + r - b, hinge coefficients, accumulation in slot order); the row
// reductions run over 64 lanes instead of 32, so sums differ from it in the last bit (both are held to the oracle at 1e-4).
// LIMITED loss, both tables normalised, k <= 10, ld <= 256; everything else stays on triple_grouped.
#ifdef OEA_DET_SCRATCH
constexpr int kGradShift = 3;
#else
constexpr int kGradShift = 2;
#endif
constexpr unsigned kBufFlags = 0x00020000u;                       // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)
// Resources span 2 GB from the table's base; row byte offsets (scalar offset) stay below 1 GB (launch rule), so an offset of 3 GB in the
// vector register is out of range whether or not the hardware counts the scalar offset in its range check, and the sum never wraps.
constexpr int kBufRecords = (int)0x80000000u;
constexpr int kBufOob = (int)0xC0000000u;

__device__ __forceinline__ void buf_grad_add(__amdgpu_buffer_rsrc_t rs, float v, int voff, int soff, int imm) {
#ifdef OEA_DET_SCRATCH
    const long long q = oea::to_fixed(v);
    const int vo = voff + imm;
    // s_nop 4: the hazard recogniser does not look inside inline assembly, and a VMEM instruction must not read a scalar register a
    // VALU instruction (v_readlane of a spilled scalar, v_readfirstlane) wrote fewer than 5 wait states earlier
    asm volatile("s_nop 4\n\tbuffer_atomic_add_x2 %0, %1, %2, %3 offen" ::"v"(q), "v"(vo), "s"(rs), "s"(soff) : "memory");
#else
    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rs, voff + imm, soff, 0);
#endif
}
__device__ __forceinline__ void buf_flag_set(__amdgpu_buffer_rsrc_t rs, int row) {       // every lane stores the same word: one request
#ifdef OEA_DET_SCRATCH
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 one = {1u, 0u};
    __builtin_amdgcn_raw_buffer_store_b64(one, rs, 0, row * 8, 0);
#else
    __builtin_amdgcn_raw_buffer_store_b32(0x3f800000u, rs, 0, row * 4, 0);
#endif
}
// `vt` = this lane's vector offset for the LAST fragment: its column's byte offset where the column exists, kBufOob where it does not
// (>= num_records under either reading of the range check: the access reads 0 / the atomic is dropped)
template <int IT>
__device__ __forceinline__ void wave_load_row(__amdgpu_buffer_rsrc_t rs, int soff, int v4, int vt, Row<64, IT> &r) {
#pragma unroll
    for (int it = 0; it < IT; ++it)
        r.v[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, it < IT - 1 ? v4 + it * 256 : vt, soff, 0));
}
template <int IT>
__device__ __forceinline__ void wave_atomic_row(__amdgpu_buffer_rsrc_t rs, int soff, int vg, int vgt, const Row<64, IT> &g, float sign) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        if (it < IT - 1) buf_grad_add(rs, sign * g.v[it], vg, soff, it * (64 << kGradShift));
        else buf_grad_add(rs, sign * g.v[it], vgt, soff, 0);
    }
}
__device__ __forceinline__ float uniform_f(float x) {       // a value every lane holds -> scalar register (uniform branches on it)
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

// y = l2_normalize(v) with the sum of squares as a scalar (tf.nn.l2_normalize: v * rsqrt(max(sum v^2, 1e-12)))
template <int IT>
__device__ __forceinline__ void wave_normalize(Row<64, IT> &r) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) s += r.v[it] * r.v[it];
    const float inv = rsqrtf(fmaxf(oea::wave_sum_uniform(s), 1e-12f));
#pragma unroll
    for (int it = 0; it < IT; ++it) r.v[it] *= inv;
}

// PLAN (step_plan.h): the positive's two gradient rows A = gacc, B = gpos leave as PLAIN stores into contrib[2 p + {0, 1}] (the
// epoch's plan tells the optimiser kernel which entity rows add which of them, with which sign, in which order); only the relation
// row and the corrupted rows of active negatives still go through the atomic scratch.
template <int IT, int L1, int KT, bool PLAN>
__global__ __launch_bounds__(512, (IT <= 2 ? (KT == 10 ? 8 : 7) : 4)) void triple_wave(
    const float *__restrict__ ent, const float *__restrict__ rel, int ld, const int32_t *__restrict__ pos,
    int64_t n_pos, const int32_t *__restrict__ neg, int k, oea_step_cfg cfg, StepWs ws, int dbg, float *__restrict__ contrib,
    const uint32_t *__restrict__ pflags) {
    constexpr int G = 64, KC = 10;
    if (KT > 0) k = KT;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = blockDim.x >> 6;                                          // 8 (ld <= 128) or 4 waves: launch_step's groups per block
    const int64_t w0 = (int64_t)blockIdx.x * wpb + wv, nw = (int64_t)gridDim.x * wpb;
    const int row_b = ld * 4, row_g = ld << kGradShift;                       // bytes per table row / scratch row
    // dbg (OEA_STEP_WAVE_DBG, experiments only): bit 0 = an empty resource for the scratch (every atomic is issued and dropped), bit 1 =
    // empty resources for the tables (every row load is issued and returns 0), bit 2 = no touched flags
    const __amdgpu_buffer_rsrc_t rs_ent = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ent), 0, (dbg & 2) ? 0 : kBufRecords, kBufFlags);
    const __amdgpu_buffer_rsrc_t rs_rel = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(rel), 0, (dbg & 2) ? 0 : kBufRecords, kBufFlags);
    const __amdgpu_buffer_rsrc_t rs_eg = __builtin_amdgcn_make_buffer_rsrc(ws.ent_grad, 0, (dbg & 1) ? 0 : kBufRecords, kBufFlags);
    const __amdgpu_buffer_rsrc_t rs_et = __builtin_amdgcn_make_buffer_rsrc(ws.ent_touched, 0, (dbg & 4) ? 0 : kBufRecords, kBufFlags);
    const int v4 = lane * 4, vg = lane << kGradShift;
    const int c_last = (IT - 1) * 64 + lane;                                  // this lane's column in the last fragment
    const int vt = c_last < ld ? c_last * 4 : kBufOob, vgt = c_last < ld ? c_last << kGradShift : kBufOob;
    const float c_neg = L1 ? -cfg.balance : -2.f * cfg.balance;               // dL/dd of an active negative = c_neg * (d or sgn d)
    double loss_local = 0.0;
    for (int64_t p = w0; p < n_pos; p += nw) {
        const int32_t *pp = pos + 3 * p;
        const int32_t *ng = neg + p * k * 3;
        const int h = pp[0], r = pp[1], t = pp[2];
        Row<G, IT> yh, yr, yt;
        wave_load_row<IT>(rs_ent, h * row_b, v4, vt, yh);
        wave_load_row<IT>(rs_rel, r * row_b, v4, vt, yr);
        wave_load_row<IT>(rs_ent, t * row_b, v4, vt, yt);
        int ids[3 * KC];
        if (KT == KC || k == KC) {                                      // constant offsets -> wide scalar loads
#pragma unroll
            for (int q = 0; q < 3 * KC; ++q) ids[q] = ng[q];
        } else {
#pragma unroll
            for (int q = 0; q < 3 * KC; ++q) ids[q] = q < 3 * k ? ng[q] : 0;
        }
        // every entry a corruption of this positive, and all on one side?
        unsigned bad = 0, xh = 0, xt = 0;
#pragma unroll
        for (int j = 0; j < KC; ++j)
            if (KT == KC || j < k) {
                const unsigned a = (unsigned)(ids[3 * j] ^ h), b = (unsigned)(ids[3 * j + 2] ^ t), c = (unsigned)(ids[3 * j + 1] ^ r);
                bad |= c | min(a, b);
                xh |= a;
                xt |= b;
            }
        if (bad | min(xh, xt)) {                                        // rare: 1 + k independent triples, one copy of the code
            double ls = score_independent<G, IT>(ent, rel, ld, lane, p, h, r, t, true, cfg, ws, OEA_LOSS_LIMITED, L1);
#pragma unroll 1
            for (int j = 0; j < k; ++j)
                ls += score_independent<G, IT>(ent, rel, ld, lane, p, ng[3 * j], ng[3 * j + 1], ng[3 * j + 2], false, cfg, ws, OEA_LOSS_LIMITED, L1);
            loss_local += ls;
            continue;
        }
        const bool tails = xh == 0;                                     // (k == 0: vacuously)
        int ce[KC];
        Row<G, IT> yc[KC];
#pragma unroll
        for (int j = 0; j < KC; ++j)
            if (KT == KC || j < k) {
                ce[j] = tails ? ids[3 * j + 2] : ids[3 * j];
                wave_load_row<IT>(rs_ent, ce[j] * row_b, v4, vt, yc[j]);
            }
        // ---- the positive ------------------------------------------------------------------------------------------------------
        wave_normalize<IT>(yh);
        wave_normalize<IT>(yr);
        wave_normalize<IT>(yt);
        Row<G, IT> u, gacc, gpos;
        float sp = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            u.v[it] = yh.v[it] + yr.v[it];
            const float d = u.v[it] - yt.v[it];
            gpos.v[it] = d;
            sp += L1 ? fabsf(d) : d * d;
        }
        const float xp = oea::wave_sum_uniform(sp) - cfg.pos_margin;  // losses.py:53: relu(s+ - pos_margin)
        const bool cpos = xp > 0.f;
        float lsum = cpos ? xp : 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            gpos.v[it] = cpos ? (L1 ? sgn(gpos.v[it]) : 2.f * gpos.v[it]) : 0.f;
            gacc.v[it] = gpos.v[it];
        }
        // ---- the negatives: scores first (k independent chains), then hinges in slot order -------------------------------------
        float sc[KC];
        if (tails) {
#pragma unroll
            for (int j = 0; j < KC; ++j)
                if (KT == KC || j < k) {
                    wave_normalize<IT>(yc[j]);
                    float s = 0.f;
#pragma unroll
                    for (int it = 0; it < IT; ++it) {
                        const float d = u.v[it] - yc[j].v[it];          // (h + r) - t'
                        yc[j].v[it] = d;
                        s += L1 ? fabsf(d) : d * d;
                    }
                    sc[j] = oea::wave_sum_uniform(s);
                }
        } else {
#pragma unroll
            for (int j = 0; j < KC; ++j)
                if (KT == KC || j < k) {
                    wave_normalize<IT>(yc[j]);
                    float s = 0.f;
#pragma unroll
                    for (int it = 0; it < IT; ++it) {
                        const float d = yc[j].v[it] + yr.v[it] - yt.v[it];   // (h' + r) - t
                        yc[j].v[it] = d;
                        s += L1 ? fabsf(d) : d * d;
                    }
                    sc[j] = oea::wave_sum_uniform(s);
                }
        }
        bool anyneg = false;
        const float sign_c = tails ? -1.f : 1.f;                        // d/d(corrupted row): -g (tail) / +g (head)
#pragma unroll
        for (int j = 0; j < KC; ++j)
            if (KT == KC || j < k) {
                const float xn = cfg.neg_margin - sc[j];     // losses.py:54: balance * relu(neg_margin - s-)
                if (xn > 0.f) {
                    lsum += cfg.balance * xn;
                    anyneg = true;
                    Row<G, IT> g;
#pragma unroll
                    for (int it = 0; it < IT; ++it) {
                        g.v[it] = L1 ? c_neg * sgn(yc[j].v[it]) : c_neg * yc[j].v[it];
                        gacc.v[it] += g.v[it];
                    }
                    wave_atomic_row<IT>(rs_eg, ce[j] * row_g, vg, vgt, g, sign_c);
                    buf_flag_set(rs_et, ce[j]);
                }
            }
        // ---- the positive's rows: tail side gh = gr = gacc, gt = -gpos; head side gr = gacc, gt = -gacc, gh = gpos -----------------
        if (PLAN) {
            const __amdgpu_buffer_rsrc_t rs_cb = __builtin_amdgcn_make_buffer_rsrc(contrib, 0, kBufRecords, kBufFlags);
            const int so = (int)p * 2 * row_b;                          // rows 2 p (A) and 2 p + 1 (B), always written: the plan reads them
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int vo = it < IT - 1 ? v4 + it * 256 : vt;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gacc.v[it]), rs_cb, vo, so, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gpos.v[it]), rs_cb, vo, so + row_b, 0);
            }
        }
        if (cpos | anyneg) {
            const int copy = (int)(p % kRelCopies);
            grad_t *rgb = copy == 0 ? ws.rel_grad : ws.rel_extra + (copy - 1) * ws.rel_copy_stride;
            const __amdgpu_buffer_rsrc_t rs_rg = __builtin_amdgcn_make_buffer_rsrc(rgb, 0, (dbg & 1) ? 0 : kBufRecords, kBufFlags);
            const __amdgpu_buffer_rsrc_t rs_rt = __builtin_amdgcn_make_buffer_rsrc(ws.rel_touched, 0, (dbg & 4) ? 0 : kBufRecords, kBufFlags);
            wave_atomic_row<IT>(rs_rg, r * row_g, vg, vgt, gacc, 1.f);
            buf_flag_set(rs_rt, r);
            // without a plan both rows go through the atomic scratch; with one only the rows the plan marks as hubs of this step
            const unsigned via_atomics = PLAN ? pflags[p] : 3u;         // bit 0: head row, bit 1: tail row
            const bool h_full = tails, t_full = !tails;                  // the row that receives the whole sum (the other: the positive's term)
            if ((via_atomics & 1u) && (h_full || cpos)) {
                wave_atomic_row<IT>(rs_eg, h * row_g, vg, vgt, h_full ? gacc : gpos, 1.f);
                buf_flag_set(rs_et, h);
            }
            if ((via_atomics & 2u) && (t_full || cpos)) {
                wave_atomic_row<IT>(rs_eg, t * row_g, vg, vgt, t_full ? gacc : gpos, -1.f);
                buf_flag_set(rs_et, t);
            }
        }
        loss_local += (double)lsum;
    }
    // one partial per workgroup, fixed order
    __shared__ double sred[8];
    if (lane == 0) sred[wv] = loss_local;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < wpb; ++w) s += sred[w];
        ws.partials[blockIdx.x] = s;
    }
}

// ---- TransH (approaches/bootea_transh.py:58-96) ---------------------------------------------------------------
// h' = h - (h.n) n, t' = t - (t.n) n with n = l2n(l2n(normal[r])); s = |h' + r - t'|.  With P = I - n n^T:
//   d/dh = P g,  d/dt = -P g,  d/dr = g,  d/dn = ((t.n) - (h.n)) g + (g.n) (t - h)        (g = dL/d delta).
// A separate, plainer kernel than triple_grouped (one group per positive and its k negatives, negatives
// two at a time): the TransE kernel's register budget stays what it is.
template <int G, int IT>
__device__ __forceinline__ float dot(const Row<G, IT> &a, const Row<G, IT> &b) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) s += a.v[it] * b.v[it];
    return group_sum<G>(s);
}
template <int G, int IT>
__device__ __forceinline__ void load_normal(const float *__restrict__ nrm, int r, int ld, int lane, Row<G, IT> &yn) {
    load_row<G, IT>(nrm + (int64_t)r * ld, ld, lane, yn);
    normalize<G, IT>(yn, 1);          // the variable is created with is_l2_norm=True ...
    normalize<G, IT>(yn, 1);          // ... and _calc normalises the looked-up row again
}
// one full triple with its own relation (positives of a batch without negatives; entries of a grouped batch
// that are not corruptions of their positive)
template <int G, int IT>
__device__ double transh_independent(const float *__restrict__ ent, const float *__restrict__ rel, int ld, int lane,
                                     int64_t item, int h, int r, int t, bool is_pos, const oea_step_cfg &cfg,
                                     const StepWs &ws) {
    Row<G, IT> yh, yr, yt, yn, delta, g;
    load_row<G, IT>(ent + (int64_t)h * ld, ld, lane, yh);
    load_row<G, IT>(rel + (int64_t)r * ld, ld, lane, yr);
    load_row<G, IT>(ent + (int64_t)t * ld, ld, lane, yt);
    load_normal<G, IT>(cfg.normal, r, ld, lane, yn);
    normalize<G, IT>(yh, cfg.ent_l2_norm);
    normalize<G, IT>(yr, cfg.rel_l2_norm);
    normalize<G, IT>(yt, cfg.ent_l2_norm);
    const float ah = dot<G, IT>(yh, yn), at = dot<G, IT>(yt, yn);
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const float d = (yh.v[it] - ah * yn.v[it]) + yr.v[it] - (yt.v[it] - at * yn.v[it]);
        delta.v[it] = d;
        s += cfg.l1 ? fabsf(d) : d * d;
    }
    s = group_sum<G>(s);
    float coef, l;
    triple_coef(cfg, is_pos, s, coef, l);
    if (coef != 0.f) {
        dscore<G, IT>(delta, coef, cfg.l1, g);
        const float gdn = dot<G, IT>(g, yn);
        Row<G, IT> pg, gn;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            pg.v[it] = g.v[it] - gdn * yn.v[it];
            gn.v[it] = (at - ah) * g.v[it] + gdn * (yt.v[it] - yh.v[it]);
        }
        atomic_row<G, IT>(ws.ent_grad + (int64_t)h * ld, ld, lane, pg, 1.f);
        atomic_row<G, IT>(ws.ent_grad + (int64_t)t * ld, ld, lane, pg, -1.f);
        atomic_row<G, IT>(ws.rel_copy(item % kRelCopies) + (int64_t)r * ld, ld, lane, g, 1.f);
        atomic_row<G, IT>(ws.nrm_copy(item % kRelCopies) + (int64_t)r * ld, ld, lane, gn, 1.f);
        if (lane == 0) { ws.ent_touched[h] = 1.f; ws.ent_touched[t] = 1.f; ws.rel_touched[r] = 1.f; ws.nrm_touched[r] = 1.f; }
    }
    return (double)l;
}

template <int G, int IT>
__global__ __launch_bounds__(256) void triple_transh_grouped(
    const float *__restrict__ ent, const float *__restrict__ rel, int ld, const int32_t *__restrict__ pos,
    int64_t n_pos, const int32_t *__restrict__ neg, int k, oea_step_cfg cfg, StepWs ws) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    double loss_local = 0.0;
    for (int64_t p = grp; p < n_pos; p += ngrp) {
        const int h = pos[3 * p], r = pos[3 * p + 1], t = pos[3 * p + 2];
        const int32_t *ng = neg + (int64_t)p * k * 3;
        Row<G, IT> yh, yr, yt, yn, ph, pt, g, gh, gr, gt, gn;
        load_row<G, IT>(ent + (int64_t)h * ld, ld, lane, yh);
        load_row<G, IT>(rel + (int64_t)r * ld, ld, lane, yr);
        load_row<G, IT>(ent + (int64_t)t * ld, ld, lane, yt);
        load_normal<G, IT>(cfg.normal, r, ld, lane, yn);
        normalize<G, IT>(yh, cfg.ent_l2_norm);
        normalize<G, IT>(yr, cfg.rel_l2_norm);
        normalize<G, IT>(yt, cfg.ent_l2_norm);
        const float ah = dot<G, IT>(yh, yn), at = dot<G, IT>(yt, yn);
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            ph.v[it] = yh.v[it] - ah * yn.v[it];
            pt.v[it] = yt.v[it] - at * yn.v[it];
            const float d = ph.v[it] + yr.v[it] - pt.v[it];
            g.v[it] = d;
            s += cfg.l1 ? fabsf(d) : d * d;
        }
        s = group_sum<G>(s);
        float coef, l;
        triple_coef(cfg, true, s, coef, l);
        double lsum = (double)l;
        bool any = coef != 0.f;
        {
            Row<G, IT> d0 = g;
            dscore<G, IT>(d0, coef, cfg.l1, g);
            const float gdn = dot<G, IT>(g, yn);
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const float pg = g.v[it] - gdn * yn.v[it];
                gh.v[it] = pg; gt.v[it] = -pg; gr.v[it] = g.v[it];
                gn.v[it] = (at - ah) * g.v[it] + gdn * (yt.v[it] - yh.v[it]);
            }
        }
        for (int j0 = 0; j0 < k; j0 += 2) {
            int ch[2], cr[2], ct[2];
            Row<G, IT> yc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = j0 + u < k ? j0 + u : j0;
                ch[u] = ng[3 * j]; cr[u] = ng[3 * j + 1]; ct[u] = ng[3 * j + 2];
                load_row<G, IT>(ent + (int64_t)((ch[u] == h) ? ct[u] : ch[u]) * ld, ld, lane, yc[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (j0 + u >= k) continue;
                const bool tail = ch[u] == h;
                if (!(cr[u] == r && (tail || ct[u] == t))) {              // not a corruption of this positive
                    lsum += transh_independent<G, IT>(ent, rel, ld, lane, p, ch[u], cr[u], ct[u], false, cfg, ws);
                    continue;
                }
                const int ce = tail ? ct[u] : ch[u];
                normalize<G, IT>(yc[u], cfg.ent_l2_norm);
                const float ac = dot<G, IT>(yc[u], yn);
                Row<G, IT> delta;
                s = 0.f;
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const float pc = yc[u].v[it] - ac * yn.v[it];
                    const float d = tail ? ph.v[it] + yr.v[it] - pc : pc + yr.v[it] - pt.v[it];
                    delta.v[it] = d;
                    s += cfg.l1 ? fabsf(d) : d * d;
                }
                s = group_sum<G>(s);
                triple_coef(cfg, false, s, coef, l);
                lsum += (double)l;
                if (coef == 0.f) continue;
                any = true;
                dscore<G, IT>(delta, coef, cfg.l1, g);
                const float gdn = dot<G, IT>(g, yn);
                Row<G, IT> pg;
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    pg.v[it] = g.v[it] - gdn * yn.v[it];
                    gr.v[it] += g.v[it];
                    if (tail) {
                        gh.v[it] += pg.v[it];
                        gn.v[it] += (ac - ah) * g.v[it] + gdn * (yc[u].v[it] - yh.v[it]);
                    } else {
                        gt.v[it] -= pg.v[it];
                        gn.v[it] += (at - ac) * g.v[it] + gdn * (yt.v[it] - yc[u].v[it]);
                    }
                }
                atomic_row<G, IT>(ws.ent_grad + (int64_t)ce * ld, ld, lane, pg, tail ? -1.f : 1.f);
                if (lane == 0) ws.ent_touched[ce] = 1.f;
            }
        }
        if (any) {
            atomic_row<G, IT>(ws.ent_grad + (int64_t)h * ld, ld, lane, gh, 1.f);
            atomic_row<G, IT>(ws.ent_grad + (int64_t)t * ld, ld, lane, gt, 1.f);
            atomic_row<G, IT>(ws.rel_copy(p % kRelCopies) + (int64_t)r * ld, ld, lane, gr, 1.f);
            atomic_row<G, IT>(ws.nrm_copy(p % kRelCopies) + (int64_t)r * ld, ld, lane, gn, 1.f);
            if (lane == 0) { ws.ent_touched[h] = 1.f; ws.ent_touched[t] = 1.f; ws.rel_touched[r] = 1.f; ws.nrm_touched[r] = 1.f; }
        }
        if (lane == 0) loss_local += lsum;
    }
    block_loss_partial(loss_local, ws.partials);
}

// ---- projected scores, one group per work item (models/trans/transh.py:16-51, models/trans/transd.py:16-57) ------
// The plain ModelFamily members: margin loss on pairs (pos i, neg i) or a per-triple loss on arbitrary lists, rows
// projected before the translation.
//   TransH  h' = h - (h.n) n                        n = l2n(l2n(normal[r]))            (transh.py:48-51)
//   TransD  h' = l2n(h + (h.hp) rp)                 hp = ent_transfer[h], rp = rel_transfer[r]  (transd.py:56-57)
// TransD keeps its transfer vectors in the SAME tables, rows [base, base + n): the gradient scratch, the exchange
// and apply_rows treat them like any other row (same l2_norm flag, same optimiser -- transd.py:16-24).
// mode 0: score only; 1: score + gradient with the given dL/ds; 2: dL/ds from the per-triple loss (returns it in l).
template <int G, int IT>
__device__ float transd_triple(const float *__restrict__ ent, const float *__restrict__ rel, int ld, int lane,
                               int64_t item, int h, int r, int t, bool is_pos, const oea_step_cfg &cfg,
                               const StepWs &ws, int mode, float coef, float &l) {
    const int64_t eb = cfg.ent_transfer_base, rb = cfg.rel_transfer_base;
    Row<G, IT> yh, yt, yr, hp, tp, rp, H, T, g;
    load_row<G, IT>(ent + (int64_t)h * ld, ld, lane, yh);
    load_row<G, IT>(ent + (int64_t)t * ld, ld, lane, yt);
    load_row<G, IT>(rel + (int64_t)r * ld, ld, lane, yr);
    load_row<G, IT>(ent + (eb + h) * ld, ld, lane, hp);
    load_row<G, IT>(ent + (eb + t) * ld, ld, lane, tp);
    load_row<G, IT>(rel + (rb + r) * ld, ld, lane, rp);
    normalize<G, IT>(yh, cfg.ent_l2_norm);
    normalize<G, IT>(yt, cfg.ent_l2_norm);
    normalize<G, IT>(yr, cfg.rel_l2_norm);
    normalize<G, IT>(hp, cfg.ent_l2_norm);
    normalize<G, IT>(tp, cfg.ent_l2_norm);
    normalize<G, IT>(rp, cfg.rel_l2_norm);
    const float ah = dot<G, IT>(yh, hp), at = dot<G, IT>(yt, tp);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        H.v[it] = yh.v[it] + ah * rp.v[it];
        T.v[it] = yt.v[it] + at * rp.v[it];
    }
    const float ssh = sumsq<G, IT>(H), sst = sumsq<G, IT>(T);
    const float ih = rsqrtf(fmaxf(ssh, 1e-12f)), itl = rsqrtf(fmaxf(sst, 1e-12f));
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        H.v[it] *= ih;
        T.v[it] *= itl;
        const float d = H.v[it] + yr.v[it] - T.v[it];
        g.v[it] = d;
        s += cfg.l1 ? fabsf(d) : d * d;
    }
    s = group_sum<G>(s);
    l = 0.f;
    if (mode == 0) return s;
    if (mode == 2) triple_coef(cfg, is_pos, s, coef, l);
    if (coef == 0.f) return s;
    {
        Row<G, IT> d0 = g;
        dscore<G, IT>(d0, coef, cfg.l1, g);
    }
    // back through the two projections' normalisation: q = (g - (g.H) H) / |u|  (the clamp branch passes g / |u|)
    const float ch = ssh > 1e-12f ? dot<G, IT>(g, H) : 0.f, ct = sst > 1e-12f ? dot<G, IT>(g, T) : 0.f;
    Row<G, IT> qh, qt;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        qh.v[it] = (g.v[it] - ch * H.v[it]) * ih;
        qt.v[it] = (ct * T.v[it] - g.v[it]) * itl;
    }
    const float bh = dot<G, IT>(qh, rp), bt = dot<G, IT>(qt, rp);
    Row<G, IT> o;
#pragma unroll
    for (int it = 0; it < IT; ++it) o.v[it] = qh.v[it] + bh * hp.v[it];
    atomic_row<G, IT>(ws.ent_grad + (int64_t)h * ld, ld, lane, o, 1.f);
#pragma unroll
    for (int it = 0; it < IT; ++it) o.v[it] = bh * yh.v[it];
    atomic_row<G, IT>(ws.ent_grad + (eb + h) * ld, ld, lane, o, 1.f);
#pragma unroll
    for (int it = 0; it < IT; ++it) o.v[it] = qt.v[it] + bt * tp.v[it];
    atomic_row<G, IT>(ws.ent_grad + (int64_t)t * ld, ld, lane, o, 1.f);
#pragma unroll
    for (int it = 0; it < IT; ++it) o.v[it] = bt * yt.v[it];
    atomic_row<G, IT>(ws.ent_grad + (eb + t) * ld, ld, lane, o, 1.f);
#pragma unroll
    for (int it = 0; it < IT; ++it) o.v[it] = ah * qh.v[it] + at * qt.v[it];
    atomic_row<G, IT>(ws.rel_copy(item % kRelCopies) + (rb + r) * ld, ld, lane, o, 1.f);
    atomic_row<G, IT>(ws.rel_copy(item % kRelCopies) + (int64_t)r * ld, ld, lane, g, 1.f);
    if (lane == 0) {
        ws.ent_touched[h] = 1.f; ws.ent_touched[t] = 1.f; ws.ent_touched[eb + h] = 1.f; ws.ent_touched[eb + t] = 1.f;
        ws.rel_touched[r] = 1.f; ws.rel_touched[rb + r] = 1.f;
    }
    return s;
}

template <int G, int IT>
__device__ float transh_triple(const float *__restrict__ ent, const float *__restrict__ rel, int ld, int lane,
                               int64_t item, int h, int r, int t, bool is_pos, const oea_step_cfg &cfg,
                               const StepWs &ws, int mode, float coef, float &l) {
    Row<G, IT> yh, yr, yt, yn, g;
    load_row<G, IT>(ent + (int64_t)h * ld, ld, lane, yh);
    load_row<G, IT>(rel + (int64_t)r * ld, ld, lane, yr);
    load_row<G, IT>(ent + (int64_t)t * ld, ld, lane, yt);
    load_normal<G, IT>(cfg.normal, r, ld, lane, yn);
    normalize<G, IT>(yh, cfg.ent_l2_norm);
    normalize<G, IT>(yr, cfg.rel_l2_norm);
    normalize<G, IT>(yt, cfg.ent_l2_norm);
    const float ah = dot<G, IT>(yh, yn), at = dot<G, IT>(yt, yn);
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const float d = (yh.v[it] - ah * yn.v[it]) + yr.v[it] - (yt.v[it] - at * yn.v[it]);
        g.v[it] = d;
        s += cfg.l1 ? fabsf(d) : d * d;
    }
    s = group_sum<G>(s);
    l = 0.f;
    if (mode == 0) return s;
    if (mode == 2) triple_coef(cfg, is_pos, s, coef, l);
    if (coef == 0.f) return s;
    {
        Row<G, IT> d0 = g;
        dscore<G, IT>(d0, coef, cfg.l1, g);
    }
    const float gdn = dot<G, IT>(g, yn);
    Row<G, IT> pg, gn;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        pg.v[it] = g.v[it] - gdn * yn.v[it];
        gn.v[it] = (at - ah) * g.v[it] + gdn * (yt.v[it] - yh.v[it]);
    }
    atomic_row<G, IT>(ws.ent_grad + (int64_t)h * ld, ld, lane, pg, 1.f);
    atomic_row<G, IT>(ws.ent_grad + (int64_t)t * ld, ld, lane, pg, -1.f);
    atomic_row<G, IT>(ws.rel_copy(item % kRelCopies) + (int64_t)r * ld, ld, lane, g, 1.f);
    atomic_row<G, IT>(ws.nrm_copy(item % kRelCopies) + (int64_t)r * ld, ld, lane, gn, 1.f);
    if (lane == 0) { ws.ent_touched[h] = 1.f; ws.ent_touched[t] = 1.f; ws.rel_touched[r] = 1.f; ws.nrm_touched[r] = 1.f; }
    return s;
}

template <int G, int IT, int KIND>
__global__ __launch_bounds__(256) void triple_projected(
    const float *__restrict__ ent, const float *__restrict__ rel, int ld, const int32_t *__restrict__ pos,
    int64_t n_pos, const int32_t *__restrict__ neg, int64_t n_neg, oea_step_cfg cfg, StepWs ws) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    const bool margin = cfg.loss_kind == OEA_LOSS_MARGIN;
    const int64_t items = margin ? n_pos : n_pos + n_neg;
    double loss_local = 0.0;
    auto one = [&](int64_t item, const int32_t *tr, bool is_pos, int mode, float coef, float &l) {
        if (KIND == OEA_SCORE_TRANSD)
            return transd_triple<G, IT>(ent, rel, ld, lane, item, tr[0], tr[1], tr[2], is_pos, cfg, ws, mode, coef, l);
        return transh_triple<G, IT>(ent, rel, ld, lane, item, tr[0], tr[1], tr[2], is_pos, cfg, ws, mode, coef, l);
    };
    for (int64_t item = grp; item < items; item += ngrp) {
        float l;
        if (!margin) {
            const bool is_pos = item < n_pos;
            one(item, is_pos ? pos + 3 * item : neg + 3 * (item - n_pos), is_pos, 2, 0.f, l);
            if (lane == 0) loss_local += (double)l;
            continue;
        }
        // losses.py:15-27: sum relu(margin + s+ - s-), pos i paired with neg i; both are scored first, the active
        // pairs are evaluated again for their gradients (dL/ds+ = 1, dL/ds- = -1)
        float sc[2];
#pragma unroll 1
        for (int u = 0; u < 2; ++u) sc[u] = one(item, (u ? neg : pos) + 3 * item, u == 0, 0, 0.f, l);
        const float x = cfg.margin + sc[0] - sc[1];
        if (x <= 0.f) continue;
        if (lane == 0) loss_local += (double)x;
#pragma unroll 1
        for (int u = 0; u < 2; ++u) one(item, (u ? neg : pos) + 3 * item, u == 0, 1, u ? -1.f : 1.f, l);
    }
    block_loss_partial(loss_local, ws.partials);
}

// optimiser on the touched normal-vector rows: gradient back through BOTH normalisations
template <int G, int IT>
__global__ __launch_bounds__(256) void apply_normal_rows(int64_t n_rel, int ld, oea_step_cfg cfg, StepWs ws, int copies_folded) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t row = grp; row < n_rel; row += ngrp) {
        if (ws.nrm_touched[row] == 0) continue;
        float *v = cfg.normal + row * ld;
        Row<G, IT> rv, rg, y1, y2;
        grad_t q[IT];
        load_row<G, IT>(v, ld, lane, rv);
        load_grad_raw<G, IT>(ws.nrm_grad + row * ld, ld, lane, q);
        for (int cp = 1; cp < kRelCopies && !copies_folded; ++cp) {
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = it * G + lane;
                if (c < ld) {
                    const grad_t x = ws.nrm_copy(cp)[row * ld + c];
                    if (x != 0) { q[it] += x; ws.nrm_copy(cp)[row * ld + c] = 0; }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) rg.v[it] = oea::grad_val(q[it]);
        const float ss1 = sumsq<G, IT>(rv);
        const float inv1 = rsqrtf(fmaxf(ss1, 1e-12f));
#pragma unroll
        for (int it = 0; it < IT; ++it) y1.v[it] = rv.v[it] * inv1;
        const float ss2 = sumsq<G, IT>(y1);
        const float inv2 = rsqrtf(fmaxf(ss2, 1e-12f));
#pragma unroll
        for (int it = 0; it < IT; ++it) y2.v[it] = y1.v[it] * inv2;
        const float d2 = dot<G, IT>(y2, rg);
#pragma unroll
        for (int it = 0; it < IT; ++it) rg.v[it] = (rg.v[it] - y2.v[it] * d2) * inv2;
        const float d1 = ss1 > 1e-12f ? dot<G, IT>(y1, rg) : 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * G + lane;
            if (c < ld) {
                const float gv = (rg.v[it] - y1.v[it] * d1) * inv1;
                if (cfg.opt_kind == OEA_OPT_ADAGRAD) {
                    const float a = cfg.normal_acc[row * ld + c] + gv * gv;
                    cfg.normal_acc[row * ld + c] = a;
                    v[c] = rv.v[it] - cfg.lr * gv / sqrtf(a);
                } else {
                    v[c] = rv.v[it] - cfg.lr * gv;
                }
                ws.nrm_grad[row * ld + c] = 0;
            }
        }
        if (lane == 0) ws.nrm_touched[row] = 0;
    }
}

// ---- kernel 2: optimiser on touched rows (relation rows first, then entity rows) -------------------
// pull the summed gradient of one row back through the normalisation, apply Adagrad / SGD, clear the scratch row
// CLEAR = false: the gradient did not come from the scratch row (apply_rows_plan): nothing to zero
template <int G, int IT, bool CLEAR = true>
__device__ __forceinline__ void apply_one_row(float *__restrict__ v, float *__restrict__ acc, grad_t *__restrict__ g,
                                              flag_t *__restrict__ touched_flag, int ld, int lane, int on,
                                              const oea_step_cfg &cfg, const Row<G, IT> &rv, Row<G, IT> &rg,
                                              const Row<G, IT> &ra) {
    float inv = 1.f, ydg = 0.f;
    if (on) {
        const float ss = sumsq<G, IT>(rv);
        inv = rsqrtf(fmaxf(ss, 1e-12f));
        float dot = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) dot += rv.v[it] * rg.v[it];
        dot = group_sum<G>(dot) * inv;           // y . g
        ydg = ss > 1e-12f ? dot : 0.f;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = it * G + lane;
        if (c < ld) {
            const float gv = on ? (rg.v[it] - rv.v[it] * inv * ydg) * inv : rg.v[it];
            if (cfg.opt_kind == OEA_OPT_ADAGRAD) {
                const float a = ra.v[it] + gv * gv;
                acc[c] = a;
                v[c] = rv.v[it] - cfg.lr * gv / sqrtf(a);
            } else {
                v[c] = rv.v[it] - cfg.lr * gv;
            }
            if (CLEAR) g[c] = 0;
        }
    }
    if (CLEAR && lane == 0) *touched_flag = 0;
}

// one relation row: fetched together with its flag (most relations of a batch are touched); sums (fixed order) and clears the
// scratch copies
template <int G, int IT>
__device__ __forceinline__ void apply_relation_row(int64_t row, float *__restrict__ rel, float *__restrict__ rel_acc, int ld, int lane,
                                                   const oea_step_cfg &cfg, const StepWs &ws, int copies_folded) {
    float *v = rel + row * ld;
    float *acc = rel_acc + row * ld;
    grad_t *g = ws.rel_grad + row * ld;
    const float flag = (float)ws.rel_touched[row];
    Row<G, IT> rv, rg, ra;
    grad_t q[IT];                       // raw scratch elements: the copies are summed exactly in the fixed-point build
    load_row<G, IT>(v, ld, lane, rv);
    load_grad_raw<G, IT>(g, ld, lane, q);
    if (cfg.opt_kind == OEA_OPT_ADAGRAD) load_row<G, IT>(acc, ld, lane, ra);
    if (flag == 0.f) return;
    if (!copies_folded) {               // sum (fixed order) and clear the other copies
        // copies fetched together (all loads issued before use).  The fixed-point build keeps this loop rolled: unrolled, its
        // int64 elements (two registers each) took the kernel from 120 to 172 VGPRs = from 4 to 2 waves per SIMD, and the
        // latency-bound 15K shape paid double (11.3 -> 23.7 us, gpurun_out r04a)
        constexpr int CB = IT <= 4 ? 5 : 1;
#ifdef OEA_DET_SCRATCH
#pragma unroll 1
#endif
        for (int cp0 = 1; cp0 < kRelCopies; cp0 += CB) {
            grad_t tmp[CB][IT];
#pragma unroll
            for (int u = 0; u < CB; ++u)
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int c = it * G + lane;
                    tmp[u][it] = (cp0 + u < kRelCopies && c < ld) ? ws.rel_copy(cp0 + u)[row * ld + c] : (grad_t)0;
                }
#pragma unroll
            for (int u = 0; u < CB; ++u)
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int c = it * G + lane;
                    q[it] += tmp[u][it];
                    if (cp0 + u < kRelCopies && c < ld && tmp[u][it] != 0) ws.rel_copy(cp0 + u)[row * ld + c] = 0;
                }
        }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) rg.v[it] = oea::grad_val(q[it]);
    apply_one_row<G, IT>(v, acc, g, ws.rel_touched + row, ld, lane, cfg.rel_l2_norm, cfg, rv, rg, ra);
}

// Work item w of the grid: w < n_rel -> relation row w (sums its 16 scratch copies); else R consecutive entity rows.
// The kernel is a few thousand very short waves (three row loads, two reductions, two row stores): R > 1 keeps R rows'
// loads in flight per group and divides the wave count by R (latency-bound at the 15K shape).
template <int G, int IT, int R>
__global__ __launch_bounds__(256) void apply_rows(float *__restrict__ ent, float *__restrict__ ent_acc,
                                                  int64_t n_ent, float *__restrict__ rel,
                                                  float *__restrict__ rel_acc, int64_t n_rel, int ld,
                                                  oea_step_cfg cfg, StepWs ws, int n_partials,
                                                  double *__restrict__ loss_accum, int copies_folded, int flag_first) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    const int64_t n_items = n_rel + (n_ent + R - 1) / R;
    for (int64_t w = grp; w < n_items; w += ngrp) {
        if (w >= n_rel) {                                // ---- R entity rows ---------------------------------------
            const int64_t row0 = (w - n_rel) * R;
            float flag[R];
            Row<G, IT> rv[R], rg[R], ra[R];
            if (flag_first) {          // large tables: a batch leaves many rows untouched -- look before fetching 3 rows
                bool any = false;
#pragma unroll
                for (int r = 0; r < R; ++r) any |= row0 + r < n_ent && ws.ent_touched[row0 + r] != 0;
                if (!any) continue;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t row = row0 + r < n_ent ? row0 + r : n_ent - 1;
                flag[r] = row0 + r < n_ent ? (float)ws.ent_touched[row] : 0.f;
                load_row<G, IT>(ent + row * ld, ld, lane, rv[r]);
                load_grad_row<G, IT>(ws.ent_grad + row * ld, ld, lane, rg[r]);
                if (cfg.opt_kind == OEA_OPT_ADAGRAD) load_row<G, IT>(ent_acc + row * ld, ld, lane, ra[r]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (flag[r] == 0.f) continue;
                const int64_t row = row0 + r;
                apply_one_row<G, IT>(ent + row * ld, ent_acc + row * ld, ws.ent_grad + row * ld, ws.ent_touched + row, ld,
                                     lane, cfg.ent_l2_norm, cfg, rv[r], rg[r], ra[r]);
            }
            continue;
        }
        apply_relation_row<G, IT>(w, rel, rel_acc, ld, lane, cfg, ws, copies_folded);
    }
    // fixed-order reduction of the loss partials by one wave of block 0
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        double s = 0.0;
        for (int i = threadIdx.x; i < n_partials; i += 64) s += ws.partials[i];
        s = oea::wave_sum_d(s);
        if (threadIdx.x == 0) *loss_accum += s;
    }
}

// ---- kernel 2', round 6: the optimiser of a PLANNED step (step_plan.h), one launch, four kinds of blocks ----------------------------------
//   [0, rel_blocks)        relation rows, one lane group per row as in apply_rows (16 scratch copies summed in order);
//   (plan blocks, listed third in the grid)    one lane group per DISTINCT entity row the step's positives refer to as head or tail and that is no hub:
//                          gradient = the sum of its plan entries (sign, slot) over contrib[slot] in the plan's order -- the batch
//                          order, whatever the hardware does -- plus the row of the atomic scratch where its touched flag is up (an
//                          active negative corrupted INTO this row).  Chain of dependent loads: step_first -> record -> rows;
//   [.., + scan_blocks)    what the plan does not list: entity rows whose touched flag is up (corrupted rows of active negatives, hub
//                          rows, rows of positives outside the rule: a few thousand of 200,000).  A wave reads 64 flags with one
//                          load and its groups take the flagged rows of the ballot; chunk c = rows {c, c + n_chunk, ...}: entity ids
//                          are degree-ordered (read.py:64-79), 64 CONSECUTIVE rows at the head of the table are all hubs and one
//                          wave would work through them alone.  Rows the plan sums are left to their group (membership map);
//   last block             the loss partials, 256 lanes + a fixed tree (one wave walking 2,500 partials took 18 us).
// Same update everywhere: gradient back through the normalisation, Adagrad / SGD (apply_one_row).
template <int G, int IT>
__global__ __launch_bounds__(256) void apply_step_plan(float *__restrict__ ent, float *__restrict__ ent_acc, int64_t n_ent,
                                                       float *__restrict__ rel, float *__restrict__ rel_acc, int64_t n_rel, int ld,
                                                       oea_step_cfg cfg, StepWs ws, int n_partials, double *__restrict__ loss_accum,
                                                       int copies_folded, int rel_blocks, int scan_blocks,
                                                       const uint4 *__restrict__ recs, const uint32_t *__restrict__ vals,
                                                       const int32_t *__restrict__ step_first, int s, const uint8_t *__restrict__ inplan,
                                                       const float *__restrict__ contrib) {
    constexpr int GPB = 256 / G, GPW = 64 / G;
    const int lane = threadIdx.x % G;
    const int b = (int)blockIdx.x, nb = (int)gridDim.x;
    if (b < rel_blocks) {
        for (int64_t row = (int64_t)b * GPB + threadIdx.x / G; row < n_rel; row += (int64_t)rel_blocks * GPB)
            apply_relation_row<G, IT>(row, rel, rel_acc, ld, lane, cfg, ws, copies_folded);
    } else if (b >= rel_blocks + scan_blocks && b < nb - 1) {
        const int plan_blocks = nb - 1 - rel_blocks - scan_blocks;
        const int64_t grp = (int64_t)(b - rel_blocks - scan_blocks) * GPB + threadIdx.x / G, ngrp = (int64_t)plan_blocks * GPB;
        const int64_t i1 = step_first[s + 1];
        for (int64_t i = step_first[s] + grp; i < i1; i += ngrp) {
            const uint4 rec = recs[i];                                 // {row, first entry, entries, first entry's value}
            if (rec.z > oea::kPlanHubEntries) continue;                // a hub of this step: its gradient went through the atomic scratch
            const int64_t row = rec.x;
            const uint32_t e0 = rec.y, e1 = rec.y + rec.z, v0 = rec.w;
            const float flag = (float)ws.ent_touched[row];
            Row<G, IT> rv, ra, rg, rc;
            load_row<G, IT>(ent + row * ld, ld, lane, rv);
            if (cfg.opt_kind == OEA_OPT_ADAGRAD) load_row<G, IT>(ent_acc + row * ld, ld, lane, ra);
            load_row<G, IT>(contrib + (int64_t)(v0 & 0x7fffffffu) * ld, ld, lane, rc);
#pragma unroll
            for (int it = 0; it < IT; ++it) rg.v[it] = (v0 >> 31) ? -rc.v[it] : rc.v[it];
            for (uint32_t e = e0 + 1; e < e1; e += 3) {                // further references to the row, in batch order, three in flight
                uint32_t ve[3];
                Row<G, IT> r3[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) ve[u] = vals[min(e + u, e1 - 1)];
#pragma unroll
                for (int u = 0; u < 3; ++u) load_row<G, IT>(contrib + (int64_t)(ve[u] & 0x7fffffffu) * ld, ld, lane, r3[u]);
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (e + u < e1) {
#pragma unroll
                        for (int it = 0; it < IT; ++it) rg.v[it] += (ve[u] >> 31) ? -r3[u].v[it] : r3[u].v[it];
                    }
            }
            if (flag != 0.f) {                                         // + what the atomic scratch holds for this row
                grad_t *g = ws.ent_grad + row * ld;
                Row<G, IT> rs;
                load_grad_row<G, IT>(g, ld, lane, rs);
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    rg.v[it] += rs.v[it];
                    const int c = it * G + lane;
                    if (c < ld) g[c] = 0;
                }
                if (lane == 0) ws.ent_touched[row] = 0;
            }
            apply_one_row<G, IT, false>(ent + row * ld, ent_acc + row * ld, nullptr, nullptr, ld, lane, cfg.ent_l2_norm, cfg, rv, rg, ra);
        }
    } else if (b < nb - 1) {
        // (the scan's blocks come BEFORE the plan's in the grid: its waves work through their flagged rows one group-load at a time --
        //  the long pole late in training, when many negatives are active -- and should not wait for the plan's blocks to retire)
        const int first = rel_blocks;
        const int wl = threadIdx.x & 63, gw = wl / G;
        const int64_t wave = (int64_t)(b - first) * 4 + (threadIdx.x >> 6), nwave = (int64_t)scan_blocks * 4;
        const int64_t n_chunk = (n_ent + 63) / 64;
        const uint8_t *mine = inplan + (int64_t)s * n_ent;
        for (int64_t chunk = wave; chunk < n_chunk; chunk += nwave) {
            const int64_t r = (int64_t)wl * n_chunk + chunk;
            const bool todo = r < n_ent && (float)ws.ent_touched[r] != 0.f && mine[r] == 0;
            unsigned long long mask = __ballot(todo);
            while (mask) {
                int64_t row = -1;
#pragma unroll
                for (int q = 0; q < GPW; ++q)
                    if (mask) {
                        const int bit = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        if (gw == q) row = (int64_t)bit * n_chunk + chunk;
                    }
                if (row >= 0) {
                    Row<G, IT> rv, rg, ra;
                    load_row<G, IT>(ent + row * ld, ld, lane, rv);
                    load_grad_row<G, IT>(ws.ent_grad + row * ld, ld, lane, rg);
                    if (cfg.opt_kind == OEA_OPT_ADAGRAD) load_row<G, IT>(ent_acc + row * ld, ld, lane, ra);
                    apply_one_row<G, IT>(ent + row * ld, ent_acc + row * ld, ws.ent_grad + row * ld, ws.ent_touched + row, ld, lane,
                                         cfg.ent_l2_norm, cfg, rv, rg, ra);
                }
            }
        }
    } else {
        // the loss partials of the scoring kernel: every lane a strided share (all loads in flight), then a fixed tree
        __shared__ double sp[256];
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = u * 256 + (int)threadIdx.x; v[u] = i < n_partials ? ws.partials[i] : 0.0; }
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
        sp[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sp[threadIdx.x] += sp[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) *loss_accum += sp[0];
    }
}

#ifndef OEA_DET_SCRATCH
// ---- apply_step_plan with rows as float4 (round 6) ------------------------------------------------------------------------------------------
// The same launch and the same four kinds of blocks; the entity rows (plan rows and flag-scan rows) travel as 16 B per lane: a lane
// group of G lanes holds IT4 = ceil(ld / (4 G)) float4 per row (ld = 100, G = 16: two loads per row instead of seven dwords), a quarter
// of the memory instructions and of the registers per row stream -- the kernel is ONE residency of latency-bound chains
// (record -> rows -> stores), so what it issues per row is what it costs.  Relation rows and loss partials as in apply_step_plan.
template <int IT4>
struct Row4 {
    float4 v[IT4];
};
template <int G, int IT4>
__device__ __forceinline__ void load_row4(const float *__restrict__ base, int ld, int lane, Row4<IT4> &r) {
#pragma unroll
    for (int it = 0; it < IT4; ++it) {
        const int c = (it * G + lane) * 4;                          // ld % 4 == 0: a float4 is inside the row or past it
        r.v[it] = c < ld ? oea::ld4(base + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ float4 f4_axpy(float a, float4 x, float4 y) { return make_float4(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z), fmaf(a, x.w, y.w)); }
__device__ __forceinline__ float4 f4_add(float4 x, float4 y) { return make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); }
__device__ __forceinline__ float f4_dot(float4 x, float4 y) { return x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w; }

// apply_one_row on float4 fragments (fp32 scratch): back through the normalisation, Adagrad / SGD, optional clear of the scratch row
template <int G, int IT4, bool CLEAR>
__device__ __forceinline__ void apply_one_row4(float *__restrict__ v, float *__restrict__ acc, float *__restrict__ g, flag_t *__restrict__ touched_flag,
                                               int ld, int lane, int on, const oea_step_cfg &cfg, const Row4<IT4> &rv, const Row4<IT4> &rg,
                                               const Row4<IT4> &ra) {
    float inv = 1.f, ydg = 0.f;
    if (on) {
        float ss = 0.f, dot = 0.f;
#pragma unroll
        for (int it = 0; it < IT4; ++it) { ss += f4_dot(rv.v[it], rv.v[it]); dot += f4_dot(rv.v[it], rg.v[it]); }
        ss = group_sum<G>(ss);
        inv = rsqrtf(fmaxf(ss, 1e-12f));
        dot = group_sum<G>(dot) * inv;
        ydg = ss > 1e-12f ? dot : 0.f;
    }
#pragma unroll
    for (int it = 0; it < IT4; ++it) {
        const int c = (it * G + lane) * 4;
        if (c < ld) {
            float4 gv = rg.v[it];
            const float4 y = rv.v[it];
            if (on)                                                  // (g - y (y . g)) / |v|, y = v / |v|: apply_one_row's expression
                gv = make_float4((gv.x - y.x * inv * ydg) * inv, (gv.y - y.y * inv * ydg) * inv, (gv.z - y.z * inv * ydg) * inv,
                                 (gv.w - y.w * inv * ydg) * inv);
            float4 nv;
            if (cfg.opt_kind == OEA_OPT_ADAGRAD) {
                const float4 a = make_float4(ra.v[it].x + gv.x * gv.x, ra.v[it].y + gv.y * gv.y, ra.v[it].z + gv.z * gv.z,
                                             ra.v[it].w + gv.w * gv.w);
                oea::st4(acc + c, a);
                nv = make_float4(rv.v[it].x - cfg.lr * gv.x / sqrtf(a.x), rv.v[it].y - cfg.lr * gv.y / sqrtf(a.y),
                                 rv.v[it].z - cfg.lr * gv.z / sqrtf(a.z), rv.v[it].w - cfg.lr * gv.w / sqrtf(a.w));
            } else {
                nv = make_float4(y.x - cfg.lr * gv.x, y.y - cfg.lr * gv.y, y.z - cfg.lr * gv.z, y.w - cfg.lr * gv.w);
            }
            oea::st4(v + c, nv);
            if (CLEAR) oea::st4(g + c, make_float4(0.f, 0.f, 0.f, 0.f));
        }
    }
    if (CLEAR && lane == 0) *touched_flag = 0;
}

template <int G, int IT>
__global__ __launch_bounds__(256) void apply_step_plan_v4(float *__restrict__ ent, float *__restrict__ ent_acc, int64_t n_ent,
                                                          float *__restrict__ rel, float *__restrict__ rel_acc, int64_t n_rel, int ld,
                                                          oea_step_cfg cfg, StepWs ws, int n_partials, double *__restrict__ loss_accum,
                                                          int copies_folded, int rel_blocks, int scan_blocks,
                                                          const uint4 *__restrict__ recs, const uint32_t *__restrict__ vals,
                                                          const int32_t *__restrict__ step_first, int s, const uint8_t *__restrict__ inplan,
                                                          const float *__restrict__ contrib) {
    constexpr int GPB = 256 / G, GPW = 64 / G, IT4 = (IT + 3) / 4;
    const int lane = threadIdx.x % G;
    const int b = (int)blockIdx.x, nb = (int)gridDim.x;
    if (b < rel_blocks) {
        for (int64_t row = (int64_t)b * GPB + threadIdx.x / G; row < n_rel; row += (int64_t)rel_blocks * GPB)
            apply_relation_row<G, IT>(row, rel, rel_acc, ld, lane, cfg, ws, copies_folded);
    } else if (b >= rel_blocks + scan_blocks && b < nb - 1) {
        const int plan_blocks = nb - 1 - rel_blocks - scan_blocks;
        const int64_t grp = (int64_t)(b - rel_blocks - scan_blocks) * GPB + threadIdx.x / G, ngrp = (int64_t)plan_blocks * GPB;
        const int64_t i1 = step_first[s + 1];
        for (int64_t i = step_first[s] + grp; i < i1; i += ngrp) {
            const uint4 rec = recs[i];                                 // {row, first entry, entries, first entry's value}
            if (rec.z > oea::kPlanHubEntries) continue;
            const int64_t row = rec.x;
            const uint32_t e0 = rec.y, e1 = rec.y + rec.z, v0 = rec.w;
            const float flag = (float)ws.ent_touched[row];
            Row4<IT4> rv, ra, rg, rc;
            load_row4<G, IT4>(ent + row * ld, ld, lane, rv);
            if (cfg.opt_kind == OEA_OPT_ADAGRAD) load_row4<G, IT4>(ent_acc + row * ld, ld, lane, ra);
            load_row4<G, IT4>(contrib + (int64_t)(v0 & 0x7fffffffu) * ld, ld, lane, rc);
            const float s0 = (v0 >> 31) ? -1.f : 1.f;
#pragma unroll
            for (int it = 0; it < IT4; ++it) rg.v[it] = make_float4(s0 * rc.v[it].x, s0 * rc.v[it].y, s0 * rc.v[it].z, s0 * rc.v[it].w);
            for (uint32_t e = e0 + 1; e < e1; e += 3) {                // further references to the row, in batch order, three in flight
                uint32_t ve[3];
                Row4<IT4> r3[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) ve[u] = vals[min(e + u, e1 - 1)];
#pragma unroll
                for (int u = 0; u < 3; ++u) load_row4<G, IT4>(contrib + (int64_t)(ve[u] & 0x7fffffffu) * ld, ld, lane, r3[u]);
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (e + u < e1) {
                        const float su = (ve[u] >> 31) ? -1.f : 1.f;
#pragma unroll
                        for (int it = 0; it < IT4; ++it) rg.v[it] = f4_axpy(su, r3[u].v[it], rg.v[it]);
                    }
            }
            if (flag != 0.f) {                                         // + what the atomic scratch holds for this row
                float *g = ws.ent_grad + row * ld;
                Row4<IT4> rs;
                load_row4<G, IT4>(g, ld, lane, rs);
#pragma unroll
                for (int it = 0; it < IT4; ++it) {
                    rg.v[it] = f4_add(rg.v[it], rs.v[it]);
                    const int c = (it * G + lane) * 4;
                    if (c < ld) oea::st4(g + c, make_float4(0.f, 0.f, 0.f, 0.f));
                }
                if (lane == 0) ws.ent_touched[row] = 0;
            }
            apply_one_row4<G, IT4, false>(ent + row * ld, ent_acc + row * ld, nullptr, nullptr, ld, lane, cfg.ent_l2_norm, cfg, rv, rg, ra);
        }
    } else if (b < nb - 1) {
        const int first = rel_blocks;
        const int wl = threadIdx.x & 63, gw = wl / G;
        const int64_t wave = (int64_t)(b - first) * 4 + (threadIdx.x >> 6), nwave = (int64_t)scan_blocks * 4;
        const int64_t n_chunk = (n_ent + 63) / 64;
        const uint8_t *mine = inplan + (int64_t)s * n_ent;
        for (int64_t chunk = wave; chunk < n_chunk; chunk += nwave) {
            const int64_t r = (int64_t)wl * n_chunk + chunk;
            const bool todo = r < n_ent && (float)ws.ent_touched[r] != 0.f && mine[r] == 0;
            unsigned long long mask = __ballot(todo);
            while (mask) {
                int64_t row = -1;
#pragma unroll
                for (int q = 0; q < GPW; ++q)
                    if (mask) {
                        const int bit = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        if (gw == q) row = (int64_t)bit * n_chunk + chunk;
                    }
                if (row >= 0) {
                    Row4<IT4> rv, rg, ra;
                    load_row4<G, IT4>(ent + row * ld, ld, lane, rv);
                    load_row4<G, IT4>(ws.ent_grad + row * ld, ld, lane, rg);
                    if (cfg.opt_kind == OEA_OPT_ADAGRAD) load_row4<G, IT4>(ent_acc + row * ld, ld, lane, ra);
                    apply_one_row4<G, IT4, true>(ent + row * ld, ent_acc + row * ld, ws.ent_grad + row * ld, ws.ent_touched + row, ld, lane,
                                                 cfg.ent_l2_norm, cfg, rv, rg, ra);
                }
            }
        }
    } else {
        __shared__ double sp[256];
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = u * 256 + (int)threadIdx.x; v[u] = i < n_partials ? ws.partials[i] : 0.0; }
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
        sp[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sp[threadIdx.x] += sp[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) *loss_accum += sp[0];
    }
}
#endif  // !OEA_DET_SCRATCH

// ---- kernel 2b: optimisers whose update is DENSE (tf.train.AdamOptimizer / AdadeltaOptimizer, optimizers.py:13-16) -------
// The tables are l2_normalize(variable): the gradient of a gather comes back through the normalisation as a dense
// tensor, so TF's dense kernels run -- Adam moves every row every step (m and v decay where the gradient is zero),
// Adadelta's accumulators decay everywhere.  state: [2, rows, ld] = (m, v) for Adam, (accum, accum_update) for
// Adadelta.  The relation scratch copies must have been folded into copy 0 (fold_rel_copies_kernel).
template <int G, int IT>
__global__ __launch_bounds__(256) void apply_rows_dense(float *__restrict__ ent, float *__restrict__ ent_state,
                                                        int64_t n_ent, float *__restrict__ rel,
                                                        float *__restrict__ rel_state, int64_t n_rel, int ld,
                                                        oea_step_cfg cfg, StepWs ws, int n_partials,
                                                        double *__restrict__ loss_accum, float lr_t) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t row_all = grp; row_all < n_ent + n_rel; row_all += ngrp) {
        const bool is_rel = row_all < n_rel;
        const int64_t row = is_rel ? row_all : row_all - n_rel;
        const int64_t rows = is_rel ? n_rel : n_ent;
        flag_t *touched = is_rel ? ws.rel_touched : ws.ent_touched;
        float *v = (is_rel ? rel : ent) + row * ld;
        float *s0 = (is_rel ? rel_state : ent_state) + row * ld;
        float *s1 = s0 + rows * (int64_t)ld;
        grad_t *g = (is_rel ? ws.rel_grad : ws.ent_grad) + row * ld;
        const int on = is_rel ? cfg.rel_l2_norm : cfg.ent_l2_norm;
        const float flag = (float)touched[row];
        // Adadelta WITHOUT the l2_normalize in front of the lookup: TF's gradient is IndexedSlices and SparseApplyAdadelta
        // only visits the gathered rows -- the accumulators of untouched rows do not decay.  (With the normalisation the
        // gradient is dense; TF1's sparse Adam decays every row either way: optimizer._apply_sparse_shared.)
        if (cfg.opt_kind == OEA_OPT_ADADELTA && !on && flag == 0.f) continue;
        Row<G, IT> rv, rg, r0, r1;
        load_row<G, IT>(v, ld, lane, rv);
        load_grad_row<G, IT>(g, ld, lane, rg);
        load_row<G, IT>(s0, ld, lane, r0);
        load_row<G, IT>(s1, ld, lane, r1);
        float inv = 1.f, ydg = 0.f;
        if (on) {
            const float ss = sumsq<G, IT>(rv);
            inv = rsqrtf(fmaxf(ss, 1e-12f));
            float dot = 0.f;
#pragma unroll
            for (int it = 0; it < IT; ++it) dot += rv.v[it] * rg.v[it];
            dot = group_sum<G>(dot) * inv;
            ydg = ss > 1e-12f ? dot : 0.f;
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * G + lane;
            if (c < ld) {
                const float gv = on ? (rg.v[it] - rv.v[it] * inv * ydg) * inv : rg.v[it];
                if (cfg.opt_kind == OEA_OPT_ADAM) {          // training_ops ApplyAdam
                    const float m = r0.v[it] + (gv - r0.v[it]) * (1.f - cfg.beta1);
                    const float vv = r1.v[it] + (gv * gv - r1.v[it]) * (1.f - cfg.beta2);
                    s0[c] = m; s1[c] = vv;
                    v[c] = rv.v[it] - lr_t * m / (sqrtf(vv) + cfg.eps);
                } else {                                     // ApplyAdadelta (rho = beta1)
                    const float acc = r0.v[it] * cfg.beta1 + gv * gv * (1.f - cfg.beta1);
                    const float upd = sqrtf(r1.v[it] + cfg.eps) * rsqrtf(acc + cfg.eps) * gv;
                    s0[c] = acc;
                    s1[c] = r1.v[it] * cfg.beta1 + upd * upd * (1.f - cfg.beta1);
                    v[c] = rv.v[it] - cfg.lr * upd;
                }
                if (flag != 0.f) g[c] = 0;
            }
        }
        if (flag != 0.f && lane == 0) touched[row] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        double s = 0.0;
        for (int i = threadIdx.x; i < n_partials; i += 64) s += ws.partials[i];
        s = oea::wave_sum_d(s);
        if (threadIdx.x == 0) *loss_accum += s;
    }
}

// ---- entity-id partitioning of the step (one process per GPU; BASELINE.json north_star) -----------------------
// Owner of entity row id = id mod G (ids are degree-ordered with the two KGs interleaved, read.py:64-79: contiguous ranges
// would put every hub on rank 0).  Every rank keeps a full READ copy of the table for its gathers and the optimiser state
// of its own rows only.  Per step:
//   GRAD phase (unchanged kernels)        -> local gradient scratch
//   part_pack_kernel                      -> send buffer in OWNER-MAJOR order [G][rpr * (ld + 1)] (rows then touched flags),
//                                            scratch rows + flags cleared on the way; relation rows to a small buffer
//   reduce-scatter (RCCL)                 -> this rank's rows summed over the ranks [rpr * (ld + 1)]
//   all-reduce of the relation buffer     (a few hundred rows: replicated, every rank applies the same update)
//   part_apply_kernel                     -> optimiser on the owned rows (strided in the natural table), updated rows also
//                                            written to a contiguous block for the all-gather; relation rows
//   all-gather (RCCL)                     -> everyone's updated rows [G][rpr][ld]
//   part_unpack_kernel                    -> the other ranks' rows into the local read copy
// The collectives move (G-1)/G * E * (2 ld + 1) * 4 bytes per rank and step -- the volume of the all-reduce they replace --
// but the optimiser runs on 1/G of the rows and its state is sharded.
template <int G, int IT>
__global__ __launch_bounds__(256) void part_pack_kernel(StepWs ws, int64_t n_ent, int64_t n_rel, int ld, int world, int64_t rpr,
                                                        grad_t *__restrict__ send, grad_t *__restrict__ rel_x) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    const int64_t chunk = rpr * (ld + 1);
    const int64_t slots = (int64_t)world * rpr;
    for (int64_t w = grp; w < slots + n_rel; w += ngrp) {
        if (w >= slots) {                                          // relation row: grads (copies folded by the GRAD phase) + flag
            const int64_t r = w - slots;
            const flag_t f = ws.rel_touched[r];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = it * G + lane;
                if (c < ld) {
                    rel_x[r * ld + c] = f != 0 ? ws.rel_grad[r * ld + c] : (grad_t)0;
                    if (f != 0) ws.rel_grad[r * ld + c] = 0;
                }
            }
            if (lane == 0) { rel_x[n_rel * ld + r] = f; ws.rel_touched[r] = 0; }
            continue;
        }
        const int64_t o = w / rpr, j = w - o * rpr, id = j * world + o;        // slot (o, j) holds entity id
        const flag_t f = id < n_ent ? ws.ent_touched[id] : (flag_t)0;
        grad_t *dst = send + o * chunk + j * ld;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * G + lane;
            if (c < ld) {
                dst[c] = f != 0 ? ws.ent_grad[id * ld + c] : (grad_t)0;
                if (f != 0) ws.ent_grad[id * ld + c] = 0;
            }
        }
        if (lane == 0) {
            send[o * chunk + rpr * ld + j] = f;
            if (f != 0) ws.ent_touched[id] = 0;
        }
    }
}

template <int G, int IT>
__global__ __launch_bounds__(256) void part_apply_kernel(float *__restrict__ ent, float *__restrict__ acc_own, int64_t n_ent,
                                                         float *__restrict__ rel, float *__restrict__ rel_acc, int64_t n_rel,
                                                         int ld, int world, int rank, int64_t rpr, grad_t *__restrict__ own /* [rpr*(ld+1)] */,
                                                         grad_t *__restrict__ rel_x, float *__restrict__ upd /* [rpr, ld] */,
                                                         oea_step_cfg cfg, StepWs ws, int n_partials, double *__restrict__ loss_accum) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t w = grp; w < rpr + n_rel; w += ngrp) {
        Row<G, IT> rv, rg, ra;
        if (w >= rpr) {
            const int64_t r = w - rpr;
            if (rel_x[n_rel * ld + r] == 0) continue;
            load_row<G, IT>(rel + r * ld, ld, lane, rv);
            load_grad_row<G, IT>(rel_x + r * ld, ld, lane, rg);
            if (cfg.opt_kind == OEA_OPT_ADAGRAD) load_row<G, IT>(rel_acc + r * ld, ld, lane, ra);
            flag_t dummy;
            apply_one_row<G, IT>(rel + r * ld, rel_acc + r * ld, rel_x + r * ld, &dummy, ld, lane, cfg.rel_l2_norm, cfg, rv, rg, ra);
            continue;
        }
        const int64_t j = w, id = j * world + rank;
        if (id >= n_ent) {                                         // padding slot of the last stripe
#pragma unroll
            for (int it = 0; it < IT; ++it) { const int c = it * G + lane; if (c < ld) upd[j * ld + c] = 0.f; }
            continue;
        }
        float *v = ent + id * ld;
        load_row<G, IT>(v, ld, lane, rv);
        if (own[rpr * ld + j] != 0) {
            load_grad_row<G, IT>(own + j * ld, ld, lane, rg);
            if (cfg.opt_kind == OEA_OPT_ADAGRAD) load_row<G, IT>(acc_own + j * ld, ld, lane, ra);
            flag_t dummy;
            apply_one_row<G, IT>(v, acc_own + j * ld, own + j * ld, &dummy, ld, lane, cfg.ent_l2_norm, cfg, rv, rg, ra);
            load_row<G, IT>(v, ld, lane, rv);                      // the updated row (same lanes wrote it)
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) { const int c = it * G + lane; if (c < ld) upd[j * ld + c] = rv.v[it]; }
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {                      // fixed-order reduction of the loss partials
        double s = 0.0;
        for (int i = threadIdx.x; i < n_partials; i += 64) s += ws.partials[i];
        s = oea::wave_sum_d(s);
        if (threadIdx.x == 0) *loss_accum += s;
    }
}

__global__ void part_unpack_kernel(float *__restrict__ ent, int64_t n_ent, int ld, int world, int rank, int64_t rpr,
                                   const float *__restrict__ all /* [G][rpr][ld] */) {
    const int cpr = ld / 4;
    const int64_t total = n_ent * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t id = i / cpr;
        const int c = (int)(i - id * cpr);
        const int o = (int)(id % world);
        if (o == rank) continue;
        oea::st4(ent + id * ld + 4 * c, oea::ld4(all + ((int64_t)o * rpr + id / world) * ld + 4 * c));
    }
}

// data parallel: fold relation copies 1.. into copy 0 (and clear them) so that only copy 0 is exchanged
__global__ void fold_rel_copies_kernel(StepWs ws, int64_t n, int with_normal) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        grad_t sum = 0;
#pragma unroll
        for (int c = 0; c < kRelCopies - 1; ++c) {
            const grad_t v = ws.rel_extra[c * ws.rel_copy_stride + i];
            if (v != 0) { sum += v; ws.rel_extra[c * ws.rel_copy_stride + i] = 0; }
        }
        if (sum != 0) ws.rel_grad[i] += sum;
        if (with_normal) {
            grad_t sn = 0;
#pragma unroll
            for (int c = 0; c < kRelCopies - 1; ++c) {
                const grad_t v = ws.nrm_extra[c * ws.rel_copy_stride + i];
                if (v != 0) { sn += v; ws.nrm_extra[c * ws.rel_copy_stride + i] = 0; }
            }
            if (sn != 0) ws.nrm_grad[i] += sn;
        }
    }
}

// grad[ids[i], c] += src[i, c]  (gradients w.r.t. NORMALISED rows produced outside the fused step,
// e.g. MTransE's mapping loss, approaches/mtranse.py:84-96) + touched flags, so that apply_rows
// pulls them through the normalisation and the optimiser like any other row gradient.
__global__ void scatter_rows_kernel(grad_t *__restrict__ grad, flag_t *__restrict__ touched, int ld,
                                    const int32_t *__restrict__ ids, int64_t n, const float *__restrict__ src,
                                    int src_ld) {
    const int64_t total = n * ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / ld;
        const int c = (int)(i - row * ld);
        const float v = src[row * src_ld + c];
        const int32_t id = ids[row];
        if (v != 0.f) oea::grad_add(grad + (int64_t)id * ld + c, v);
        if (c == 0) touched[id] = 1;
    }
}

// ---- boundary-row ("halo") exchange of the partitioned step (round 5) --------------------------------------------------------------
// The dense protocol above moves every owned row every step: (G-1)/G * E * (2 ld + 1) * 4 B per rank (161 MB at the 100K shape)
// for a step that reads ~27,000 rows per rank.  BASELINE.json's north star asks for "all-gather of BOUNDARY embeddings": the rows
// a rank's share of the batch refers to.  Which rows those are is known before the step runs -- the epoch's positives and its
// negatives (drawn ahead for the whole epoch, the same Philox streams on every rank) name them -- so EVERY rank computes, for
// every step s and every rank r, the sorted list of entity rows r's share refers to, bucketed by owner (halo_mark / halo_compact;
// one copy of the [steps][G][G] counts to the host per call).  No index ever travels, the sizes of all exchanges of the call are
// known on the host, and a step is:
//   GRAD on the share -> PUSH: all-to-all of the gradient rows (+ flag) of list(s, me, o) to their owners o, added into the owner's
//   scratch (exact sums in the fixed-point build) -> relation rows all-reduced as before -> optimiser on the owned touched rows ->
//   PULL: all-to-all of the CURRENT values of list(s + 1, r, me) to the readers r of the next step.
// Rows a rank does not refer to stay stale in its local copy until the call's last step: one dense all-gather of the owned rows
// ends the range, so everything outside (evaluation, neighbour refresh, the other trainers) sees the single-GPU table.
// ~19 MB per rank and step at the 100K shape with 8 ranks (2,500 positives x (2 + 10) rows, 7/8 remote, ld + 1 floats out and ld
// back) instead of 161 MB.  TransE / TransH scores (TransD's stacked transfer rows are not in the lists: dense protocol).
struct HaloGeom {
    int world, rank;
    int64_t rpr, rpr32;            // owned rows per rank; the same rounded up to 32 (bitmap words per owner = rpr32 / 32)
    int64_t cap;                   // list entries per (step, rank)
    int nwords;                    // bitmap words per (step, rank) = world * rpr32 / 32
};

__device__ __forceinline__ void halo_mark_id(uint32_t *__restrict__ bm, const HaloGeom &g, int id) {
    const int64_t bit = (int64_t)(id % g.world) * g.rpr32 + id / g.world;
    atomicOr(bm + (bit >> 5), 1u << (bit & 31));
}

// one thread per (batch row, entry): entry 0 = the positive, 1..k its negatives; marks head and tail in the bitmap of the
// (step, rank) whose share holds the row
__global__ void halo_mark_kernel(const int32_t *__restrict__ pos_all, const int32_t *__restrict__ neg_all, int k,
                                 const int64_t *__restrict__ offsets, int s0, int s1, HaloGeom g, uint32_t *__restrict__ bitmaps) {
    const int64_t row_lo = offsets[s0], rows = offsets[s1] - row_lo;
    const int64_t total = rows * (k + 1);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = row_lo + i / (k + 1);
        const int e = (int)(i % (k + 1));
        int a = s0, b = s1;                                  // step of the row: offsets[s] <= row < offsets[s + 1]
        while (b - a > 1) { const int m = (a + b) >> 1; if (offsets[m] <= row) a = m; else b = m; }
        const int64_t b0 = offsets[a], nb = offsets[a + 1] - b0, p = row - b0;
        int r = (int)((p * g.world) / nb);
        while (r + 1 < g.world && nb * (r + 1) / g.world <= p) ++r;
        while (r > 0 && nb * r / g.world > p) --r;
        const int32_t *tr = e == 0 ? pos_all + 3 * row : neg_all + 3 * (row * k + (e - 1));
        uint32_t *bm = bitmaps + ((int64_t)(a - s0) * g.world + r) * g.nwords;
        halo_mark_id(bm, g, tr[0]);
        halo_mark_id(bm, g, tr[2]);
    }
}

// one workgroup per (step, rank): the set bits of owner o's words, ascending, as local row indices j (id = j * world + o);
// counts[(step, rank)][o]; the lists of the owners follow each other (prefix of the counts)
__global__ __launch_bounds__(256) void halo_compact_kernel(const uint32_t *__restrict__ bitmaps, HaloGeom g, int32_t *__restrict__ lists,
                                                           int32_t *__restrict__ counts, int32_t *__restrict__ err) {
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int sr = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t *bm = bitmaps + (int64_t)sr * g.nwords;
    int32_t *out = lists + (int64_t)sr * g.cap;
    const int wpo = (int)(g.rpr32 >> 5);                     // words per owner
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int o = 0; o < g.world; ++o) {
        const int start = s_base;
        for (int w0 = 0; w0 < wpo; w0 += 256) {
            const int w = w0 + tid;
            uint32_t bits = w < wpo ? bm[o * wpo + w] : 0u;
            const int c = __popc(bits);
            int incl = c;                                    // inclusive scan inside the wave
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
            if (lane == 63) s_wave[wave] = incl;
            __syncthreads();
            int before = 0;
            for (int u = 0; u < wave; ++u) before += s_wave[u];
            const int chunk_total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
            int at = s_base + before + incl - c;
            while (bits) {
                const int b = __ffs(bits) - 1;
                bits &= bits - 1;
                if (at < g.cap) out[at] = w * 32 + b;
                ++at;
            }
            __syncthreads();
            if (tid == 0) s_base += chunk_total;
            __syncthreads();
        }
        if (tid == 0) counts[(int64_t)sr * g.world + o] = s_base - start;
    }
    if (tid == 0 && s_base > g.cap) atomicMax(err, 1);
}

// entry q of a concatenation of per-peer lists -> (peer, position inside the peer's list); pfx [world + 1] ascending
__device__ __forceinline__ int halo_peer_of(const int32_t *__restrict__ pfx, int world, int64_t q) {
    int a = 0, b = world;
    while (b - a > 1) { const int m = (a + b) >> 1; if (pfx[m] <= q) a = m; else b = m; }
    return a;
}

// PUSH, reader side: gradient row + flag of every remote row of list(s, me, .) -> send[q * (ld + 1)], scratch row and flag cleared
template <int G, int IT>
__global__ __launch_bounds__(256) void halo_push_pack_kernel(StepWs ws, int ld, HaloGeom g, const int32_t *__restrict__ list /* (s, me) */,
                                                             const int32_t *__restrict__ pfx /* [world + 1] over owners */,
                                                             grad_t *__restrict__ send) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G, ngrp = (int64_t)gridDim.x * blockDim.x / G;
    const int64_t total = pfx[g.world];
    for (int64_t q = grp; q < total; q += ngrp) {
        const int o = halo_peer_of(pfx, g.world, q);
        if (o == g.rank) continue;                                   // own rows stay in the scratch
        const int64_t id = (int64_t)list[q] * g.world + o;
        const flag_t f = ws.ent_touched[id];
        grad_t *dst = send + q * (ld + 1);
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * G + lane;
            if (c < ld) {
                dst[c] = f != 0 ? ws.ent_grad[id * ld + c] : (grad_t)0;
                if (f != 0) ws.ent_grad[id * ld + c] = 0;
            }
        }
        if (lane == 0) { dst[ld] = f; if (f != 0) ws.ent_touched[id] = 0; }
    }
}

// PUSH, owner side: the rows received from reader r (= list(s, r, me), known here) added into the own scratch
template <int G, int IT>
__global__ __launch_bounds__(256) void halo_push_unpack_kernel(StepWs ws, int ld, HaloGeom g, const int32_t *__restrict__ lists_s /* (s, 0) */,
                                                               const int32_t *__restrict__ own_off /* [world]: where list(s, r, me) starts in (s, r) */,
                                                               const int32_t *__restrict__ rpfx /* [world + 1] over readers */,
                                                               const grad_t *__restrict__ recv) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G, ngrp = (int64_t)gridDim.x * blockDim.x / G;
    const int64_t total = rpfx[g.world];
    for (int64_t q = grp; q < total; q += ngrp) {
        const int r = halo_peer_of(rpfx, g.world, q);
        const int64_t j = lists_s[(int64_t)r * g.cap + own_off[r] + (q - rpfx[r])];
        const int64_t id = j * g.world + g.rank;
        const grad_t *src = recv + q * (ld + 1);
        if (src[ld] == 0) continue;                                  // the reader's share left the row alone
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * G + lane;
            if (c < ld) { const grad_t v = src[c]; if (v != 0) oea::grad_add_raw(ws.ent_grad + id * ld + c, v); }
        }
        if (lane == 0) ws.ent_touched[id] = 1;
    }
}

// optimiser on the OWNED touched rows (gradient sums in the own scratch) + the relation rows from rel_x; part_apply_kernel's
// arithmetic and group widths, nothing written for the other ranks
template <int G, int IT>
__global__ __launch_bounds__(256) void halo_apply_kernel(float *__restrict__ ent, float *__restrict__ acc_own, int64_t n_ent,
                                                         float *__restrict__ rel, float *__restrict__ rel_acc, int64_t n_rel,
                                                         int ld, int world, int rank, int64_t rpr, grad_t *__restrict__ rel_x,
                                                         oea_step_cfg cfg, StepWs ws, int n_partials, double *__restrict__ loss_accum) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t w = grp; w < rpr + n_rel; w += ngrp) {
        Row<G, IT> rv, rg, ra;
        if (w >= rpr) {
            const int64_t r = w - rpr;
            if (rel_x[n_rel * ld + r] == 0) continue;
            load_row<G, IT>(rel + r * ld, ld, lane, rv);
            load_grad_row<G, IT>(rel_x + r * ld, ld, lane, rg);
            if (cfg.opt_kind == OEA_OPT_ADAGRAD) load_row<G, IT>(rel_acc + r * ld, ld, lane, ra);
            flag_t dummy;
            apply_one_row<G, IT>(rel + r * ld, rel_acc + r * ld, rel_x + r * ld, &dummy, ld, lane, cfg.rel_l2_norm, cfg, rv, rg, ra);
            continue;
        }
        const int64_t j = w, id = j * world + rank;
        if (id >= n_ent || ws.ent_touched[id] == 0) continue;
        float *v = ent + id * ld;
        load_row<G, IT>(v, ld, lane, rv);
        load_grad_row<G, IT>(ws.ent_grad + id * ld, ld, lane, rg);
        if (cfg.opt_kind == OEA_OPT_ADAGRAD) load_row<G, IT>(acc_own + j * ld, ld, lane, ra);
        apply_one_row<G, IT>(v, acc_own + j * ld, ws.ent_grad + id * ld, ws.ent_touched + id, ld, lane, cfg.ent_l2_norm, cfg, rv, rg, ra);
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {                      // fixed-order reduction of the loss partials
        double s = 0.0;
        for (int i = threadIdx.x; i < n_partials; i += 64) s += ws.partials[i];
        s = oea::wave_sum_d(s);
        if (threadIdx.x == 0) *loss_accum += s;
    }
}

// relation rows of the scratch (copies folded by the GRAD phase) + flags -> rel_x, scratch cleared (part_pack_kernel's relation half)
__global__ void halo_rel_pack_kernel(StepWs ws, int64_t n_rel, int ld, grad_t *__restrict__ rel_x) {
    const int64_t total = n_rel * ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld;
        const flag_t f = ws.rel_touched[r];
        rel_x[i] = f != 0 ? ws.rel_grad[i] : (grad_t)0;
        if (f != 0) ws.rel_grad[i] = 0;
    }
}
__global__ void halo_rel_flags_kernel(StepWs ws, int64_t n_rel, int ld, grad_t *__restrict__ rel_x) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rel) { rel_x[n_rel * ld + r] = ws.rel_touched[r]; ws.rel_touched[r] = 0; }
}

// PULL: rows of the table <-> a packed buffer.  owner side (PACK): entry q of the readers' lists list(s, r, me) -> buf[q * ld];
// reader side (!PACK): buf[q * ld] -> row list(s, me, o) of owner o.  16-byte pieces.
template <bool PACK>
__global__ void halo_pull_kernel(float *__restrict__ ent, int ld, HaloGeom g, const int32_t *__restrict__ lists_s /* PACK: (s, 0); else (s, me) */,
                                 const int32_t *__restrict__ own_off, const int32_t *__restrict__ pfx, float *__restrict__ buf) {
    const int cpr = ld / 4;
    const int64_t total = (int64_t)pfx[g.world] * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = i / cpr;
        const int c = (int)(i - q * cpr);
        const int peer = halo_peer_of(pfx, g.world, q);
        if (peer == g.rank) continue;
        if (PACK) {
            const int64_t j = lists_s[(int64_t)peer * g.cap + own_off[peer] + (q - pfx[peer])];
            oea::st4(buf + q * ld + 4 * c, oea::ld4(ent + (j * g.world + g.rank) * ld + 4 * c));
        } else {
            const int64_t j = lists_s[q];
            oea::st4(ent + (j * g.world + peer) * ld + 4 * c, oea::ld4(buf + q * ld + 4 * c));
        }
    }
}

// the owned rows -> upd [rpr, ld] (source of the dense all-gather that ends a halo range)
__global__ void halo_owned_rows_kernel(const float *__restrict__ ent, int64_t n_ent, int ld, int world, int rank, int64_t rpr,
                                       float *__restrict__ upd) {
    const int cpr = ld / 4;
    const int64_t total = rpr * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = i / cpr, id = j * world + rank;
        const int c = (int)(i - j * cpr);
        oea::st4(upd + j * ld + 4 * c, id < n_ent ? oea::ld4(ent + id * ld + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

// triple_grouped specialised on (loss kind, norm) -- the per-triple losses that reach it (margin pairs go to
// triple_generic); OEA_STEP_RUNTIME_KIND=1 keeps the one kernel with the run-time switch (experiments).
// OEA_STEP_WAVE = 0: triple_grouped (two positives per wave) everywhere
static bool step_wave_enabled() {
    static const int env = [] { const char *e = getenv("OEA_STEP_WAVE"); return e ? atoi(e) : 1; }();
    return env != 0;
}

template <int IT64>
void launch_wave(int nb, int block, hipStream_t st, const float *ent, const float *rel, int ld, const int32_t *pos,
                 int64_t n_pos, const int32_t *neg, const oea_step_cfg &cfg, const StepWs &ws, float *contrib = nullptr,
                 const uint32_t *pflags = nullptr) {
    const int k = cfg.neg_group_k;
    static const int dbg = [] { const char *e = getenv("OEA_STEP_WAVE_DBG"); return e ? atoi(e) : 0; }();
#define OEA_WAVE(L1, KT)                                                                                                               \
    do {                                                                                                                               \
        if (contrib) oea::launch_timed(triple_wave<IT64, L1, KT, true>, nb, block, st, ent, rel, ld, pos, n_pos, neg, k, cfg, ws, dbg, contrib, pflags);   \
        else oea::launch_timed(triple_wave<IT64, L1, KT, false>, nb, block, st, ent, rel, ld, pos, n_pos, neg, k, cfg, ws, dbg, contrib, pflags);          \
    } while (0)
    if (k == 10) { if (cfg.l1) OEA_WAVE(1, 10); else OEA_WAVE(0, 10); }
    else { if (cfg.l1) OEA_WAVE(1, 0); else OEA_WAVE(0, 0); }
#undef OEA_WAVE
}

// the rule of triple_wave (and of the epoch plan that builds on it)
static bool wave_rule(const oea_step_cfg &cfg, int64_t n_ent, int64_t n_rel, int32_t ld) {
    return cfg.score_kind == OEA_SCORE_TRANSE && cfg.loss_kind == OEA_LOSS_LIMITED && cfg.ent_l2_norm && cfg.rel_l2_norm &&
           cfg.neg_group_k >= 1 && cfg.neg_group_k <= 10 && ld <= 256 &&
           std::max(n_ent, n_rel) * (int64_t)ld * (int64_t)sizeof(grad_t) < ((int64_t)1 << 30) && step_wave_enabled();
}

template <int G, int IT>
void launch_grouped(int nb, int block, hipStream_t st, const float *ent, const float *rel, int ld, const int32_t *pos,
                    int64_t n_pos, const int32_t *neg, const oea_step_cfg &cfg, const StepWs &ws, bool wave_fits, float *contrib = nullptr,
                    const uint32_t *pflags = nullptr) {
    static const int runtime_kind = [] { const char *e = getenv("OEA_STEP_RUNTIME_KIND"); return e ? atoi(e) : 0; }();
    const int k = cfg.neg_group_k;
    // one wave per positive (round 6): the per-triple losses with a compile-time kind, rows of at most 256 columns, tables whose
    // byte offsets fit the instruction's 32-bit scalar offset; same grid as triple_grouped with blocks of (256 / G) waves, so the
    // number of loss partials is launch_step's nb1 either way
    if constexpr ((G == 32 && IT <= 4) || (G == 64 && IT == 4)) {
        if (wave_fits && !runtime_kind && cfg.loss_kind == OEA_LOSS_LIMITED && cfg.ent_l2_norm && cfg.rel_l2_norm && k >= 0 && k <= 10 &&
            step_wave_enabled()) {
            constexpr int IT64 = G == 32 ? (IT + 1) / 2 : 4;
            // G == 32 (ld <= 128): the grid has one workgroup per 8 positives (launch_step's rule for the loss partials).  512 threads =
            // one positive per wave; 256 threads = two per wave (grid stride): half the waves, nearly all resident at once, and
            // 4-wave workgroups find their slots beside side-stream kernels where 8-wave workgroups starve (OEA_STEP_WAVE_BLOCK)
            static const int blk_env = [] { const char *e = getenv("OEA_STEP_WAVE_BLOCK"); return e ? atoi(e) : 0; }();
            const int wblock = G == 32 ? (blk_env == 256 || blk_env == 512 ? blk_env : 512) : (block / G) * 64;
            launch_wave<IT64>(nb, wblock, st, ent, rel, ld, pos, n_pos, neg, cfg, ws, contrib, pflags);
            return;
        }
    }
#define OEA_GROUPED(LOSS, L1) oea::launch_timed(triple_grouped<G, IT, LOSS, L1>, nb, block, st, ent, rel, ld, pos, n_pos, neg, k, cfg, ws)
    if (runtime_kind) OEA_GROUPED(-1, -1);
    else if (cfg.loss_kind == OEA_LOSS_LIMITED) { if (cfg.l1) OEA_GROUPED(OEA_LOSS_LIMITED, 1); else OEA_GROUPED(OEA_LOSS_LIMITED, 0); }
    else if (cfg.loss_kind == OEA_LOSS_LOGISTIC) { if (cfg.l1) OEA_GROUPED(OEA_LOSS_LOGISTIC, 1); else OEA_GROUPED(OEA_LOSS_LOGISTIC, 0); }
    else if (cfg.loss_kind == OEA_LOSS_POSITIVE) { if (cfg.l1) OEA_GROUPED(OEA_LOSS_POSITIVE, 1); else OEA_GROUPED(OEA_LOSS_POSITIVE, 0); }
    else OEA_GROUPED(-1, -1);
#undef OEA_GROUPED
}

// 16-lane groups for the optimiser kernels when the tables do not fit the caches (see launch_step).  ONE rule for apply_rows
// and part_apply_kernel, on the size of the WHOLE table: the group width decides the order of the row reductions, and the
// partitioned job must round like the single-GPU job (bit for bit in the fixed-point build).
static bool apply_g16(int64_t n_ent, int64_t n_rel, int32_t ld) {
    static const int env_g16 = [] { const char *e = getenv("OEA_APPLY_G16"); return e ? atoi(e) : -1; }();
    return env_g16 >= 0 ? env_g16 != 0 : (n_ent + n_rel) * (int64_t)ld * 12 > (int64_t)128 << 20;   // 3 arrays > 128 MB
}

// work items of the GRAD kernel of a step = how many loss partials it leaves (one per workgroup of ceil(items / groups per block)):
// ONE rule for launch_step and for everything that adds the partials up afterwards (oea_part_apply, the one-call epochs; ADVICE
// r04: the partitioned epoch took "grouped" from neg_group_k alone and under-counted TransD's partials -- the printed loss only)
static int64_t step_items(const oea_step_cfg &cfg, int64_t n_pos, int64_t n_neg) {
    const bool transh = cfg.score_kind == OEA_SCORE_TRANSH, transd = cfg.score_kind == OEA_SCORE_TRANSD;
    const bool transh_grouped = transh && cfg.loss_kind != OEA_LOSS_MARGIN && (n_neg == 0 || cfg.neg_group_k > 0);
    const bool projected = transd || (transh && !transh_grouped);
    const bool grouped = transh_grouped || (!projected && cfg.neg_group_k > 0 && cfg.loss_kind != OEA_LOSS_MARGIN);
    return (grouped || cfg.loss_kind == OEA_LOSS_MARGIN) ? n_pos : n_pos + n_neg;
}

template <int G, int IT>
int launch_step(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel,
                int32_t ld, const int32_t *pos, int64_t n_pos, const int32_t *neg, int64_t n_neg,
                const oea_step_cfg &cfg, const StepWs &ws, double *loss_accum, int phase, hipStream_t st,
                const oea::StepPlanView *plan = nullptr, int plan_step = 0, int64_t plan_first_pos = 0) {
    const int block = 256, gpb = block / G;
    const bool transh = cfg.score_kind == OEA_SCORE_TRANSH, transd = cfg.score_kind == OEA_SCORE_TRANSD;
    const bool dense_opt = cfg.opt_kind == OEA_OPT_ADAM || cfg.opt_kind == OEA_OPT_ADADELTA;
    // TransH with a per-triple loss on the sampler's grouped layout keeps its grouped kernel; margin pairs, free
    // negative lists and TransD take the one-item-per-group kernel
    const bool transh_grouped = transh && cfg.loss_kind != OEA_LOSS_MARGIN && (n_neg == 0 || cfg.neg_group_k > 0);
    const bool projected = transd || (transh && !transh_grouped);
    const bool grouped = transh_grouped || (!projected && cfg.neg_group_k > 0 && cfg.loss_kind != OEA_LOSS_MARGIN);
    const int64_t items = step_items(cfg, n_pos, n_neg);
    const int nb1 = (int)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(items, gpb), 1), kMaxBlocks);
    // profiling events, 4 per STEP: [m0 fwd_bwd m1] ... [m2 apply m3], each pair attached to its kernel's dispatch
    // (oea::launch_timed); a GRAD call takes the first pair and decides whether the step is sampled, the APPLY call
    // that follows takes the second pair
    if (phase != OEA_PHASE_APPLY) {
        oea::prof_call();
        if (transd)
            oea::launch_timed(triple_projected<G, IT, OEA_SCORE_TRANSD>, nb1, block, st, ent, rel, ld, pos, n_pos, neg, n_neg, cfg, ws);
        else if (projected)
            oea::launch_timed(triple_projected<G, IT, OEA_SCORE_TRANSH>, nb1, block, st, ent, rel, ld, pos, n_pos, neg, n_neg, cfg, ws);
        else if (transh)
            oea::launch_timed(triple_transh_grouped<G, IT>, nb1, block, st, ent, rel, ld, pos, n_pos, neg, n_neg ? cfg.neg_group_k : 0, cfg, ws);
        else if (grouped)
            launch_grouped<G, IT>(nb1, block, st, ent, rel, ld, pos, n_pos, neg, cfg, ws,
                                  std::max(n_ent, n_rel) * (int64_t)ld * (int64_t)sizeof(grad_t) < ((int64_t)1 << 30),
                                  plan ? plan->contrib : nullptr, plan ? plan->pflags + plan_first_pos : nullptr);
        else
            oea::launch_timed(triple_generic<G, IT>, nb1, block, st, ent, rel, ld, pos, n_pos, neg, n_neg, cfg, ws);
        if (phase == OEA_PHASE_GRAD || dense_opt)
            fold_rel_copies_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(n_rel * (int64_t)ld, 256), 1024), 256, 0, st>>>(
                ws, n_rel * (int64_t)ld, transh ? 1 : 0);
    }
    const int nb2 = (int)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(n_ent + n_rel, gpb), 1), 16384);
    if (phase != OEA_PHASE_GRAD) {
        // the APPLY phase's event pair: start on its first launch, stop on its last (the plan path has two)
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        (void)oea::prof_pair(&ev0, &ev1);
        if (dense_opt) {
            // lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)   (AdamOptimizer._apply_dense)
            const double t = (double)cfg.opt_t;
            const float lr_t = cfg.opt_kind == OEA_OPT_ADAM
                                   ? (float)((double)cfg.lr * std::sqrt(1.0 - std::pow((double)cfg.beta2, t)) / (1.0 - std::pow((double)cfg.beta1, t)))
                                   : cfg.lr;
            oea::launch_events(apply_rows_dense<G, IT>, nb2, block, st, ev0, ev1, ent, ent_acc, n_ent, rel, rel_acc, n_rel, ld, cfg, ws,
                              items > 0 ? nb1 : 0, loss_accum, lr_t);
        } else {
            // rows per lane group: 1.  Measured at the 15K shape (gpurun_out r02a, in-epoch HIP events): R = 1 17.4 us,
            // R = 2 20.1, R = 4 20.4 -- more (shorter) waves hide the row latency better than more loads per wave.
            // OEA_APPLY_ROWS overrides (1 / 2 / 4) for experiments.
            static const int env_r = [] { const char *e = getenv("OEA_APPLY_ROWS"); return e ? atoi(e) : 0; }();
            const int R = env_r ? env_r : 1;
            const int n_part = items > 0 ? nb1 : 0;   /* no triples scored: no loss partials to add */
            const int folded = phase == OEA_PHASE_APPLY;
            // OEA_APPLY_FLAG_FIRST=1: look at the touched flag before fetching the three rows.  Measured (gpurun_out r02d):
            // 15K shape 11.1 -> 12.8 us, 100K shape 64.0 -> 72.5 us -- a batch touches most of the table at both shapes
            // and the dependent round trip costs more than the rows it saves; off by default.
            static const int flag_first_env = [] { const char *e = getenv("OEA_APPLY_FLAG_FIRST"); return e ? atoi(e) : 0; }();
            const int flag_first = plan ? 1 : flag_first_env;     // behind the plan pass few entity rows are left: look at the flag first
            auto nb = [&](int r) { return (int)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(n_rel + oea::ceil_div(n_ent, r), gpb), 1), 16384); };
            // 16-lane groups (4 rows per wave, ceil(ld / 16) fragments per lane) when the tables do not fit the caches: the
            // kernel is then bound by HBM and the narrower groups waste fewer lanes on the row's tail -- 100K shape 64.5 ->
            // 54.9 us (step 0.139 -> 0.125 ms); at the 15K shape (latency-bound, cache-resident) they LOSE, 11.0 -> 16.3 us
            // (gpurun_out r02p).  OEA_APPLY_G16 = 0 / 1 overrides the size rule.
            const bool g16 = apply_g16(n_ent, n_rel, ld);
            if (plan) {
                // ONE launch: relation rows | the plan's rows (gathered sums) | flagged rows the plan does not list | loss partials
                static_assert(kMaxBlocks <= 16 * 256, "the partials block reads 16 per lane");
                const int64_t bound = 2 * n_pos;                       // at most two distinct rows per positive
                if (g16 && G == 32) {
                    const int it16 = (ld + 15) / 16;
                    const int relb = (int)std::max<int64_t>(oea::ceil_div(n_rel, 16), 1);
                    const int planb = (int)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(bound, 16), 1), 8192);
                    const int scanb = (int)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(n_ent, 256), 1), 4096);
#ifndef OEA_DET_SCRATCH
                    // rows as float4 (apply_step_plan_v4); OEA_APPLY_V4=0: dword fragments
                    static const bool v4 = [] { const char *e = getenv("OEA_APPLY_V4"); return !(e && e[0] == '0'); }();
#define OEA_APPLYP16(ITX) do { if (v4) oea::launch_events(apply_step_plan_v4<16, ITX>, relb + planb + scanb + 1, block, st, ev0, ev1, ent, ent_acc, n_ent, rel, rel_acc, n_rel, ld, cfg, ws, n_part, loss_accum, folded, relb, scanb, plan->recs, plan->vals_b, plan->step_first, plan_step, plan->inplan, plan->contrib); \
    else oea::launch_events(apply_step_plan<16, ITX>, relb + planb + scanb + 1, block, st, ev0, ev1, ent, ent_acc, n_ent, rel, rel_acc, n_rel, ld, cfg, ws, n_part, loss_accum, folded, relb, scanb, plan->recs, plan->vals_b, plan->step_first, plan_step, plan->inplan, plan->contrib); } while (0)
#else
#define OEA_APPLYP16(ITX) oea::launch_events(apply_step_plan<16, ITX>, relb + planb + scanb + 1, block, st, ev0, ev1, ent, ent_acc, n_ent, rel, rel_acc, n_rel, ld, cfg, ws, n_part, loss_accum, folded, relb, scanb, plan->recs, plan->vals_b, plan->step_first, plan_step, plan->inplan, plan->contrib)
#endif
                    if (it16 <= 2) OEA_APPLYP16(2);
                    else if (it16 <= 4) OEA_APPLYP16(4);
                    else if (it16 == 5) OEA_APPLYP16(5);
                    else if (it16 == 6) OEA_APPLYP16(6);
                    else if (it16 == 7) OEA_APPLYP16(7);
                    else OEA_APPLYP16(8);
#undef OEA_APPLYP16
                } else {
                    const int relb = (int)std::max<int64_t>(oea::ceil_div(n_rel, gpb), 1);
                    const int planb = (int)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(bound, gpb), 1), 8192);
                    const int scanb = (int)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(n_ent, 256), 1), 4096);
                    oea::launch_events(apply_step_plan<G, IT>, relb + planb + scanb + 1, block, st, ev0, ev1, ent, ent_acc, n_ent, rel, rel_acc,
                                       n_rel, ld, cfg, ws, n_part, loss_accum, folded, relb, scanb, plan->recs, plan->vals_b,
                                       plan->step_first, plan_step, plan->inplan, plan->contrib);
                }
            } else
            if (g16 && G == 32 && R == 1) {
                const int it16 = (ld + 15) / 16;
                const int nbg = (int)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(n_rel + n_ent, 16), 1), 16384);
#define OEA_APPLY16(ITX) oea::launch_events(apply_rows<16, ITX, 1>, nbg, block, st, ev0, ev1, ent, ent_acc, n_ent, rel, rel_acc, n_rel, ld, cfg, ws, n_part, loss_accum, folded, flag_first)
                if (it16 <= 2) OEA_APPLY16(2);
                else if (it16 <= 4) OEA_APPLY16(4);
                else if (it16 == 5) OEA_APPLY16(5);
                else if (it16 == 6) OEA_APPLY16(6);
                else if (it16 == 7) OEA_APPLY16(7);
                else OEA_APPLY16(8);
#undef OEA_APPLY16
            } else
            if (R >= 4 && IT <= 4)
                oea::launch_events(apply_rows<G, IT, 4>, nb(4), block, st, ev0, ev1, ent, ent_acc, n_ent, rel, rel_acc, n_rel, ld, cfg, ws, n_part, loss_accum, folded, flag_first);
            else if (R >= 2 && IT <= 8)
                oea::launch_events(apply_rows<G, IT, 2>, nb(2), block, st, ev0, ev1, ent, ent_acc, n_ent, rel, rel_acc, n_rel, ld, cfg, ws, n_part, loss_accum, folded, flag_first);
            else
                oea::launch_events(apply_rows<G, IT, 1>, nb(1), block, st, ev0, ev1, ent, ent_acc, n_ent, rel, rel_acc, n_rel, ld, cfg, ws, n_part, loss_accum, folded, flag_first);
        }
        if (transh)
            apply_normal_rows<G, IT><<<(unsigned)std::max<int64_t>(oea::ceil_div(n_rel, gpb), 1), block, 0, st>>>(
                n_rel, ld, cfg, ws, phase == OEA_PHASE_APPLY);
    }
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
    return (size_t)(reinterpret_cast<char *>(ws.rel_extra) - reinterpret_cast<char *>(256)) / sizeof(grad_t);
}

int32_t oea_step_scratch_elem_bytes(void) { return (int32_t)sizeof(grad_t); }

int oea_triple_step(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc,
                    int64_t n_rel, int32_t dim, int32_t ld, const int32_t *pos, int64_t n_pos,
                    const int32_t *neg, int64_t n_neg, const oea_step_cfg *cfg, void *workspace,
                    double *loss_accum, void *stream) {
    return oea_triple_step_phase(ent, ent_acc, n_ent, rel, rel_acc, n_rel, dim, ld, pos, n_pos, neg, n_neg, cfg,
                                 workspace, loss_accum, OEA_PHASE_BOTH, stream);
}

static int step_phase_impl(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc,
                           int64_t n_rel, int32_t dim, int32_t ld, const int32_t *pos, int64_t n_pos,
                           const int32_t *neg, int64_t n_neg, const oea_step_cfg *cfg, void *workspace,
                           double *loss_accum, int32_t phase, void *stream, const oea::StepPlanView *plan, int plan_step,
                           int64_t plan_first_pos);

int oea_triple_step_phase(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc,
                          int64_t n_rel, int32_t dim, int32_t ld, const int32_t *pos, int64_t n_pos,
                          const int32_t *neg, int64_t n_neg, const oea_step_cfg *cfg, void *workspace,
                          double *loss_accum, int32_t phase, void *stream) {
    return step_phase_impl(ent, ent_acc, n_ent, rel, rel_acc, n_rel, dim, ld, pos, n_pos, neg, n_neg, cfg, workspace, loss_accum, phase,
                           stream, nullptr, 0, 0);
}

static int step_phase_impl(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc,
                           int64_t n_rel, int32_t dim, int32_t ld, const int32_t *pos, int64_t n_pos,
                           const int32_t *neg, int64_t n_neg, const oea_step_cfg *cfg, void *workspace,
                           double *loss_accum, int32_t phase, void *stream, const oea::StepPlanView *plan, int plan_step,
                           int64_t plan_first_pos) {
    OEA_REQUIRE(phase >= OEA_PHASE_BOTH && phase <= OEA_PHASE_APPLY, "phase");
    OEA_REQUIRE(ent && rel && (pos || n_pos == 0) && cfg && workspace && loss_accum, "null pointer");
    OEA_REQUIRE(ld % 4 == 0 && dim <= ld && dim > 0, "ld % 4 == 0 and dim <= ld");
    OEA_REQUIRE(n_pos >= 0 && n_neg >= 0 && (neg || n_neg == 0), "neg == NULL needs n_neg == 0");
    OEA_REQUIRE(cfg->loss_kind >= OEA_LOSS_MARGIN && cfg->loss_kind <= OEA_LOSS_ALIGN, "loss_kind");
    OEA_REQUIRE(cfg->opt_kind >= OEA_OPT_SGD && cfg->opt_kind <= OEA_OPT_ADADELTA, "opt_kind");
    OEA_REQUIRE(cfg->opt_kind == OEA_OPT_SGD || phase == OEA_PHASE_GRAD || (ent_acc && rel_acc),
                "Adagrad / Adam / Adadelta need their state arrays (the GRAD phase alone does not)");
    if (cfg->opt_kind == OEA_OPT_ADAM || cfg->opt_kind == OEA_OPT_ADADELTA) {
        OEA_REQUIRE(cfg->score_kind != OEA_SCORE_TRANSH, "Adam / Adadelta are not built for the TransH normal-vector table");
        OEA_REQUIRE(cfg->opt_kind != OEA_OPT_ADAM || cfg->opt_t >= 1, "Adam: opt_t = 1-based step count");
        OEA_REQUIRE(cfg->beta1 > 0.f && cfg->beta1 < 1.f && cfg->eps > 0.f, "beta1 / eps");
    }
    if (cfg->loss_kind == OEA_LOSS_MARGIN) OEA_REQUIRE(n_neg == n_pos, "margin loss pairs pos i with neg i");
    if (cfg->loss_kind == OEA_LOSS_POSITIVE || cfg->loss_kind == OEA_LOSS_ALIGN)
        OEA_REQUIRE(n_neg == 0, "positive-only loss takes no negatives");
    OEA_REQUIRE(cfg->neg_group_k >= 0 && (cfg->neg_group_k == 0 || n_neg == n_pos * (int64_t)cfg->neg_group_k),
                "neg_group_k > 0 needs n_neg == n_pos * neg_group_k");
    OEA_REQUIRE(cfg->score_kind >= OEA_SCORE_TRANSE && cfg->score_kind <= OEA_SCORE_TRANSD, "score_kind");
    if (cfg->score_kind == OEA_SCORE_TRANSH)
        OEA_REQUIRE(cfg->normal && (cfg->opt_kind != OEA_OPT_ADAGRAD || cfg->normal_acc), "TransH needs the normal_vector table (+ accumulator)");
    if (cfg->score_kind == OEA_SCORE_TRANSD)
        OEA_REQUIRE(cfg->ent_transfer_base > 0 && cfg->rel_transfer_base > 0 && 2 * (int64_t)cfg->ent_transfer_base == n_ent &&
                        2 * (int64_t)cfg->rel_transfer_base == n_rel,
                    "TransD: the tables hold the embeddings in rows [0, base) and the transfer vectors in [base, 2 base)");
    if (n_pos + n_neg == 0 && phase != OEA_PHASE_APPLY) return OEA_OK;   // apply-only: externally scattered gradients
    StepWs ws;
    ws_layout(n_ent, n_rel, ld, workspace, &ws);
    hipStream_t st = oea::as_stream(stream);
#define OEA_STEP(G, IT) launch_step<G, IT>(ent, ent_acc, n_ent, rel, rel_acc, n_rel, ld, pos, n_pos, neg, n_neg, *cfg, ws, loss_accum, phase, st, plan, plan_step, plan_first_pos)
    // Measured and dropped (round 4, gpurun_out r04f): 64-lane groups for 64 < ld <= 128 (one positive per wave, two fragments
    // per lane): scoring kernel 17.8 -> 20.0 us at the 15K shape, 64.2 -> 60.4 us at the 100K shape, where the 64-lane optimiser
    // kernel loses 18 us against the 16-lane one -- 3 % at best for a second instantiation of every kernel.
    if (ld <= 32) OEA_STEP(32, 1);
    else if (ld <= 64) OEA_STEP(32, 2);
    else if (ld <= 96) OEA_STEP(32, 3);
    else if (ld <= 128) OEA_STEP(32, 4);
    else if (ld <= 256) OEA_STEP(64, 4);
    else if (ld <= 512) OEA_STEP(64, 8);
    else if (ld <= 1280) OEA_STEP(64, 20);
    else { oea::set_error("dim %d > 1280 unsupported", dim); return OEA_EUNSUPPORTED; }
#undef OEA_STEP
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_step_entity_scratch(void *workspace, int64_t n_ent, int64_t n_rel, int32_t ld, void **ent_grad, void **ent_touched) {
    OEA_REQUIRE(workspace && ent_grad && ent_touched, "null pointer");
    StepWs ws;
    ws_layout(n_ent, n_rel, ld, workspace, &ws);
    *ent_grad = ws.ent_grad;
    *ent_touched = ws.ent_touched;
    return OEA_OK;
}

int oea_step_scatter_ent_rows(void *workspace, int64_t n_ent, int64_t n_rel, int32_t ld, const int32_t *ids,
                              int64_t n, const float *src, int32_t src_ld, void *stream) {
    OEA_REQUIRE(workspace && ids && src && ld % 4 == 0 && src_ld >= ld, "shapes");
    if (n == 0) return OEA_OK;
    StepWs ws;
    ws_layout(n_ent, n_rel, ld, workspace, &ws);
    scatter_rows_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(n * ld, 256), 65535), 256, 0, oea::as_stream(stream)>>>(
        ws.ent_grad, ws.ent_touched, ld, ids, n, src, src_ld);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int64_t oea_part_rows_per_rank(int64_t n_ent, int32_t world) { return world > 0 ? (n_ent + world - 1) / world : 0; }

size_t oea_part_send_floats(int64_t n_ent, int32_t ld, int32_t world) {
    return (size_t)world * (size_t)oea_part_rows_per_rank(n_ent, world) * (size_t)(ld + 1);
}

#define OEA_PART_DISPATCH(CALL)                                         \
    if (ld <= 32) { CALL(32, 1); }                                      \
    else if (ld <= 64) { CALL(32, 2); }                                 \
    else if (ld <= 96) { CALL(32, 3); }                                 \
    else if (ld <= 128) { CALL(32, 4); }                                \
    else if (ld <= 256) { CALL(64, 4); }                                \
    else if (ld <= 512) { CALL(64, 8); }                                \
    else if (ld <= 1280) { CALL(64, 20); }                              \
    else { oea::set_error("ld %d > 1280 unsupported", ld); return OEA_EUNSUPPORTED; }

/* TransH under the entity-id partition: the normal-vector table is relation-sized and replicated, so its gradient scratch
 * (copy 0 after the GRAD phase folded the copies) and touched flags are summed over the ranks with two small all-reduces
 * and every rank applies the same update.  oea_step_normal_scratch: where the two regions sit in the workspace (byte
 * offsets + float counts); oea_step_apply_normals: the optimiser on the touched normal rows (cfg->normal / normal_acc). */
int oea_step_normal_scratch(int64_t n_ent, int64_t n_rel, int32_t ld, int64_t *grad_offset_bytes, int64_t *touched_offset_bytes) {
    OEA_REQUIRE(grad_offset_bytes && touched_offset_bytes, "null pointer");
    StepWs ws;
    ws_layout(n_ent, n_rel, ld, reinterpret_cast<void *>(256), &ws);           // fake base: only offsets matter
    *grad_offset_bytes = reinterpret_cast<char *>(ws.nrm_grad) - reinterpret_cast<char *>(256);
    *touched_offset_bytes = reinterpret_cast<char *>(ws.nrm_touched) - reinterpret_cast<char *>(256);
    return OEA_OK;
}

int oea_step_apply_normals(int64_t n_ent, int64_t n_rel, int32_t ld, const oea_step_cfg *cfg, void *workspace, void *stream) {
    OEA_REQUIRE(cfg && workspace && ld % 4 == 0, "arguments");
    OEA_REQUIRE(cfg->score_kind == OEA_SCORE_TRANSH && cfg->normal, "TransH score with its normal-vector table");
    OEA_REQUIRE(cfg->opt_kind == OEA_OPT_SGD || (cfg->opt_kind == OEA_OPT_ADAGRAD && cfg->normal_acc), "SGD or Adagrad (+ state)");
    if (n_rel == 0) return OEA_OK;
    StepWs ws;
    ws_layout(n_ent, n_rel, ld, workspace, &ws);
    hipStream_t st = oea::as_stream(stream);
#define OEA_CALL(G, IT) apply_normal_rows<G, IT><<<(unsigned)std::max<int64_t>(oea::ceil_div(n_rel, 256 / G), 1), 256, 0, st>>>(n_rel, ld, *cfg, ws, 1);
    OEA_PART_DISPATCH(OEA_CALL)
#undef OEA_CALL
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int64_t oea_step_items(const oea_step_cfg *cfg, int64_t n_pos, int64_t n_neg) { return cfg ? step_items(*cfg, n_pos, n_neg) : -1; }


int oea_part_pack(void *workspace, int64_t n_ent, int64_t n_rel, int32_t ld, int32_t world, void *send_, void *rel_x_,
                  void *stream) {
    grad_t *send = static_cast<grad_t *>(send_), *rel_x = static_cast<grad_t *>(rel_x_);
    OEA_REQUIRE(workspace && send && rel_x && world >= 1 && ld % 4 == 0, "arguments");
    StepWs ws;
    ws_layout(n_ent, n_rel, ld, workspace, &ws);
    const int64_t rpr = oea_part_rows_per_rank(n_ent, world);
    hipStream_t st = oea::as_stream(stream);
#define OEA_CALL(G, IT)                                                                                                   \
    part_pack_kernel<G, IT><<<(unsigned)std::min<int64_t>(oea::ceil_div(world * rpr + n_rel, 256 / G), 16384), 256, 0, st>>>( \
        ws, n_ent, n_rel, ld, world, rpr, send, rel_x)
    OEA_PART_DISPATCH(OEA_CALL)
#undef OEA_CALL
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_part_apply(float *ent, float *acc_own, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel, int32_t ld,
                   int32_t world, int32_t rank, void *own_, void *rel_x_, float *upd, const oea_step_cfg *cfg, void *workspace,
                   int64_t n_items, double *loss_accum, void *stream) {
    grad_t *own = static_cast<grad_t *>(own_), *rel_x = static_cast<grad_t *>(rel_x_);
    OEA_REQUIRE(ent && rel && own && rel_x && upd && cfg && workspace && loss_accum, "null pointer");
    OEA_REQUIRE(world >= 1 && rank >= 0 && rank < world && ld % 4 == 0, "world / rank / ld");
    OEA_REQUIRE(cfg->opt_kind == OEA_OPT_SGD || (cfg->opt_kind == OEA_OPT_ADAGRAD && acc_own && rel_acc), "SGD or Adagrad (+ state)");
    // TransD: both stacked tables are ordinary rows; TransH: the normal-vector table goes through oea_step_apply_normals
    OEA_REQUIRE(cfg->score_kind >= OEA_SCORE_TRANSE && cfg->score_kind <= OEA_SCORE_TRANSD, "score_kind");
    StepWs ws;
    ws_layout(n_ent, n_rel, ld, workspace, &ws);
    const int64_t rpr = oea_part_rows_per_rank(n_ent, world);
    hipStream_t st = oea::as_stream(stream);
    // loss partials: one per workgroup of the GRAD kernel that just ran on n_items work items (launch_step's nb1)
#define OEA_CALL(G, IT)                                                                                                       \
    {                                                                                                                         \
        const int gpb = 256 / G;                                                                                              \
        /* one partial per workgroup of the GRAD kernel, whose groups are 32 lanes wide up to ld = 128 and 64 beyond */     \
        const int grad_gpb = 256 / (ld <= 128 ? 32 : 64);                                                                     \
        const int n_part = n_items > 0 ? (int)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(n_items, grad_gpb), 1), kMaxBlocks) : 0; \
        /* the second event pair of a sampled step (the GRAD call took the first): bench.py's apply timing under partitioning */ \
        oea::launch_timed(part_apply_kernel<G, IT>, (unsigned)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(rpr + n_rel, gpb), 1), 16384), 256, st, \
            ent, acc_own, n_ent, rel, rel_acc, n_rel, ld, world, rank, rpr, own, rel_x, upd, *cfg, ws, n_part, loss_accum);   \
    }
    if (ld <= 128 && apply_g16(n_ent, n_rel, ld)) {                 // the single-GPU job's group width at this table size
        const int it16 = (ld + 15) / 16;
        if (it16 <= 2) OEA_CALL(16, 2)
        else if (it16 <= 4) OEA_CALL(16, 4)
        else if (it16 == 5) OEA_CALL(16, 5)
        else if (it16 == 6) OEA_CALL(16, 6)
        else if (it16 == 7) OEA_CALL(16, 7)
        else OEA_CALL(16, 8)
    } else {
        OEA_PART_DISPATCH(OEA_CALL)
    }
#undef OEA_CALL
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_part_unpack(float *ent, int64_t n_ent, int32_t ld, int32_t world, int32_t rank, const float *all, void *stream) {
    OEA_REQUIRE(ent && all && world >= 1 && rank >= 0 && rank < world && ld % 4 == 0, "arguments");
    if (world == 1) return OEA_OK;
    const int64_t rpr = oea_part_rows_per_rank(n_ent, world);
    part_unpack_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(n_ent * (ld / 4), 256), 16384), 256, 0, oea::as_stream(stream)>>>(
        ent, n_ent, ld, world, rank, rpr, all);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_triple_epoch(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel,
                     int32_t dim, int32_t ld, const int32_t *pos_all, const int64_t *offsets_host,
                     const int64_t *splits_host, int32_t steps, int32_t k, const oea_sampler_side *side0,
                     const oea_sampler_side *side1, uint64_t seed, uint32_t step_base, int32_t *neg_buf,
                     int32_t *err_flag, const oea_step_cfg *cfg, void *workspace, double *loss_accum,
                     const int64_t *offsets_dev, const int64_t *splits_dev, void *stream) {
    return oea_triple_epoch_range(ent, ent_acc, n_ent, rel, rel_acc, n_rel, dim, ld, pos_all, offsets_host, splits_host,
                                  steps, 0, steps, k, side0, side1, seed, step_base, neg_buf, err_flag, cfg, workspace,
                                  loss_accum, offsets_dev, splits_dev, stream);
}

int oea_triple_epoch_range(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel,
                           int32_t dim, int32_t ld, const int32_t *pos_all, const int64_t *offsets_host,
                           const int64_t *splits_host, int32_t steps, int32_t step_begin, int32_t step_end, int32_t k,
                           const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed,
                           uint32_t step_base, int32_t *neg_buf, int32_t *err_flag, const oea_step_cfg *cfg,
                           void *workspace, double *loss_accum, const int64_t *offsets_dev, const int64_t *splits_dev,
                           void *stream) {
    return oea_triple_epoch_range_shard(ent, ent_acc, n_ent, rel, rel_acc, n_rel, dim, ld, pos_all, offsets_host, splits_host,
                                        steps, step_begin, step_end, k, side0, side1, seed, step_base, neg_buf, err_flag, cfg,
                                        workspace, loss_accum, offsets_dev, splits_dev, 0, 1, stream);
}

static int epoch_range_impl(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel,
                            int32_t dim, int32_t ld, const int32_t *pos_all, const int64_t *offsets_host,
                            const int64_t *splits_host, int32_t steps, int32_t step_begin, int32_t step_end, int32_t k,
                            const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed,
                            uint32_t step_base, int32_t *neg_buf, int32_t *err_flag, const oea_step_cfg *cfg,
                            void *workspace, double *loss_accum, const int64_t *offsets_dev, const int64_t *splits_dev,
                            int32_t rank, int32_t world, void *plan, size_t plan_bytes, int32_t plan_built, void *stream);

int oea_triple_epoch_range_shard(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel,
                                 int32_t dim, int32_t ld, const int32_t *pos_all, const int64_t *offsets_host,
                                 const int64_t *splits_host, int32_t steps, int32_t step_begin, int32_t step_end, int32_t k,
                                 const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed,
                                 uint32_t step_base, int32_t *neg_buf, int32_t *err_flag, const oea_step_cfg *cfg,
                                 void *workspace, double *loss_accum, const int64_t *offsets_dev, const int64_t *splits_dev,
                                 int32_t rank, int32_t world, void *stream) {
    return epoch_range_impl(ent, ent_acc, n_ent, rel, rel_acc, n_rel, dim, ld, pos_all, offsets_host, splits_host, steps, step_begin,
                            step_end, k, side0, side1, seed, step_base, neg_buf, err_flag, cfg, workspace, loss_accum, offsets_dev,
                            splits_dev, rank, world, nullptr, 0, 0, stream);
}

// 1 when an epoch of this configuration would run on the gathered-sum plan (step_plan.h): triple_wave's rule, SGD / Adagrad, the
// fp32 scratch (the fixed-point build keeps ONE summation path for one GPU and for G ranks), OEA_STEP_PLAN != 0
int32_t oea_step_plan_supported(const oea_step_cfg *cfg, int64_t n_ent, int64_t n_rel, int32_t ld, int32_t k) {
    static const int env = [] { const char *e = getenv("OEA_STEP_PLAN"); return e ? atoi(e) : 1; }();
    if (!cfg || env == 0 || oea::kDetScratch) return 0;
    // tables that fit the caches (the 15K shape: 9 MB) keep the flag-driven optimiser: its streaming pass over the whole table
    // costs 11 us there and the plan's chains of dependent loads 20 (measured, tools/r06/d.sh); OEA_STEP_PLAN=2 forces the plan
    if (env != 2 && !apply_g16(n_ent, n_rel, ld)) return 0;
    return wave_rule(*cfg, n_ent, n_rel, ld) && cfg->neg_group_k == k && (cfg->opt_kind == OEA_OPT_SGD || cfg->opt_kind == OEA_OPT_ADAGRAD);
}

int oea_triple_epoch_range_plan(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel,
                                int32_t dim, int32_t ld, const int32_t *pos_all, const int64_t *offsets_host,
                                const int64_t *splits_host, int32_t steps, int32_t step_begin, int32_t step_end, int32_t k,
                                const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed,
                                uint32_t step_base, int32_t *neg_buf, int32_t *err_flag, const oea_step_cfg *cfg,
                                void *workspace, double *loss_accum, const int64_t *offsets_dev, const int64_t *splits_dev,
                                void *plan, size_t plan_bytes, int32_t plan_built, void *stream) {
    return epoch_range_impl(ent, ent_acc, n_ent, rel, rel_acc, n_rel, dim, ld, pos_all, offsets_host, splits_host, steps, step_begin,
                            step_end, k, side0, side1, seed, step_base, neg_buf, err_flag, cfg, workspace, loss_accum, offsets_dev,
                            splits_dev, 0, 1, plan, plan_bytes, plan_built, stream);
}

static int epoch_range_impl(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel,
                            int32_t dim, int32_t ld, const int32_t *pos_all, const int64_t *offsets_host,
                            const int64_t *splits_host, int32_t steps, int32_t step_begin, int32_t step_end, int32_t k,
                            const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed,
                            uint32_t step_base, int32_t *neg_buf, int32_t *err_flag, const oea_step_cfg *cfg,
                            void *workspace, double *loss_accum, const int64_t *offsets_dev, const int64_t *splits_dev,
                            int32_t rank, int32_t world, void *plan, size_t plan_bytes, int32_t plan_built, void *stream) {
    OEA_REQUIRE(pos_all && offsets_host && splits_host && cfg, "null pointer");
    OEA_REQUIRE(steps >= 0 && k >= 0, "steps, k >= 0");
    OEA_REQUIRE(0 <= step_begin && step_begin <= step_end && step_end <= steps, "0 <= step_begin <= step_end <= steps");
    OEA_REQUIRE((offsets_dev == nullptr) == (splits_dev == nullptr), "offsets_dev and splits_dev go together");
    OEA_REQUIRE(world >= 1 && rank >= 0 && rank < world, "0 <= rank < world");
    // side0 == NULL with the device layout given: neg_buf already holds the epoch's negatives (the caller drew them
    // with oea_sample_negatives_epoch, e.g. on another stream while the previous epoch was running)
    const bool presampled = k > 0 && side0 == nullptr && side1 == nullptr && offsets_dev != nullptr;
    OEA_REQUIRE(k == 0 || (neg_buf && (presampled || (err_flag && side0 && side1))), "sampling needs neg_buf, err_flag and both sides");
    const bool ahead = k > 0 && offsets_dev != nullptr && steps > 0;      // neg_buf covers the whole epoch
    // the sampler does not read the tables: a range that starts the epoch draws ALL its negatives in one launch
    // (later ranges of the same epoch find them in neg_buf: pass side0 = side1 = NULL, or let them be drawn again
    // step by step -- the Philox streams are the same either way).  A rank of a sharded job draws the WHOLE epoch as
    // well (same streams on every rank: the union of the ranks' slices is the single-process draw) and uses its rows.
    const bool sample_all = ahead && !presampled && step_begin == 0;
    if (sample_all) {
        const int rc = oea_sample_negatives_epoch(pos_all, offsets_host[steps], offsets_dev, splits_dev, steps, k, side0,
                                                  side1, seed, step_base, 10, neg_buf, err_flag, stream);
        if (rc != OEA_OK) return rc;
    }
    // the gathered-sum plan (step_plan.h): needs the whole epoch's negatives to exist before its first step
    oea::StepPlanView pv;
    const bool use_plan = plan && world == 1 && ahead && (presampled || sample_all) && oea_step_plan_supported(cfg, n_ent, n_rel, ld, k);
    if (use_plan) {
        int64_t max_batch = 0;
        for (int32_t s = 0; s < steps; ++s) max_batch = std::max(max_batch, offsets_host[s + 1] - offsets_host[s]);
        const size_t need = oea::step_plan_layout(offsets_host[steps], steps, max_batch, n_ent, ld, plan, &pv);
        OEA_REQUIRE(plan_bytes >= need, "plan workspace smaller than oea_step_plan_bytes");
        if (!plan_built) {
            const int rc = oea_step_plan_build(pos_all, neg_buf, k, offsets_dev, offsets_host[steps], steps, max_batch, n_ent, ld, plan,
                                               plan_bytes, stream);
            if (rc != OEA_OK) return rc;
        }
    }
    oea_step_cfg step_cfg = *cfg;             // Adam: opt_t counts the steps actually run (cfg->opt_t = count of the first)
    for (int32_t s = step_begin; s < step_end; ++s) {
        const int64_t b0 = offsets_host[s], nb = offsets_host[s + 1] - b0;
        if (nb <= 0) continue;
        // this rank's contiguous share of the batch rows (models/dist.py:shard_batch; the whole batch when world == 1)
        const int64_t r_lo = nb * rank / world, r_hi = nb * (rank + 1) / world;
        const int64_t lo = b0 + r_lo, n = r_hi - r_lo;
        int64_t split = splits_host[s] - r_lo;
        split = split < 0 ? 0 : (split > n ? n : split);
        const int32_t *pos = pos_all + 3 * lo;
        int32_t *negs = ahead ? neg_buf + 3 * lo * (int64_t)k : neg_buf;
        if (n > 0 && k > 0 && !presampled && !sample_all) {
            const int rc = oea_sample_negatives_pair(pos, n, split, k, side0, side1, seed, step_base + (uint32_t)s,
                                                     (uint32_t)r_lo, 10, negs, err_flag, stream);
            if (rc != OEA_OK) return rc;
        }
        if (n > 0) {
            const int rc = step_phase_impl(ent, ent_acc, n_ent, rel, rel_acc, n_rel, dim, ld, pos, n,
                                           k > 0 ? negs : nullptr, n * (int64_t)k, &step_cfg, workspace, loss_accum,
                                           OEA_PHASE_BOTH, stream, use_plan ? &pv : nullptr, s, lo);
            if (rc != OEA_OK) return rc;
        }
        ++step_cfg.opt_t;
    }
    return OEA_OK;
}

// The steps [step_begin, step_end) of a data-parallel epoch under the entity-id partition, enqueued by ONE call: per step
// GRAD on this rank's share of the batch -> pack -> reduce-scatter (RCCL) + all-reduce of the relation rows -> optimiser on
// the owned rows -> all-gather -> unpack; everything on `stream`, no host work between the steps (driven from Python the
// exchange costs six library calls and three torch.distributed calls per step).  The communicator is the C ABI's own
// (oea_comm_*: RCCL through dlopen), so a non-Python host runs the same job.  Buffers as in the step-by-step protocol
// (include/openea_hip.h): send [world * rpr * (ld + 1)], own [rpr * (ld + 1)], rel_x [n_rel * (ld + 1)], upd [rpr, ld],
// all [world, rpr, ld]; acc_own = the optimiser state of the owned rows.  TransH: the normal vectors' scratch is all-reduced
// and applied on every rank.  Same Philox streams as the single-GPU epoch; result = the step-by-step partitioned job.
int oea_triple_epoch_range_comm(oea_comm_t comm, float *ent, float *acc_own, int64_t n_ent, float *rel, float *rel_acc,
                                int64_t n_rel, int32_t dim, int32_t ld, const int32_t *pos_all, const int64_t *offsets_host,
                                const int64_t *splits_host, int32_t steps, int32_t step_begin, int32_t step_end, int32_t k,
                                const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed, uint32_t step_base,
                                int32_t *neg_buf, int32_t *err_flag, const oea_step_cfg *cfg, void *workspace, double *loss_accum,
                                const int64_t *offsets_dev, const int64_t *splits_dev, void *send, void *own, void *rel_x,
                                float *upd, float *all, void *stream) {
    OEA_REQUIRE(comm && pos_all && offsets_host && splits_host && cfg && send && own && rel_x && upd && all, "null pointer");
    OEA_REQUIRE(steps >= 0 && k >= 0 && 0 <= step_begin && step_begin <= step_end && step_end <= steps, "step range");
    OEA_REQUIRE((offsets_dev == nullptr) == (splits_dev == nullptr), "offsets_dev and splits_dev go together");
    OEA_REQUIRE(cfg->opt_kind == OEA_OPT_SGD || cfg->opt_kind == OEA_OPT_ADAGRAD, "the partition runs SGD / Adagrad");
    const int32_t world = oea_comm_size(comm), rank = oea_comm_rank(comm);
    const bool presampled = k > 0 && side0 == nullptr && side1 == nullptr && offsets_dev != nullptr;
    OEA_REQUIRE(k == 0 || (neg_buf && (presampled || (err_flag && side0 && side1))), "sampling needs neg_buf, err_flag and both sides");
    const bool ahead = k > 0 && offsets_dev != nullptr && steps > 0;
    const bool sample_all = ahead && !presampled && step_begin == 0;
    if (sample_all) {
        const int rc = oea_sample_negatives_epoch(pos_all, offsets_host[steps], offsets_dev, splits_dev, steps, k, side0, side1, seed,
                                                  step_base, 10, neg_buf, err_flag, stream);
        if (rc != OEA_OK) return rc;
    }
    const int64_t rpr = oea_part_rows_per_rank(n_ent, world);
    const int64_t chunk = rpr * (ld + 1);
    const bool transh = cfg->score_kind == OEA_SCORE_TRANSH;
    void *nrm_grad = nullptr, *nrm_touched = nullptr;
    if (transh) {
        int64_t g_off = 0, t_off = 0;
        int rc = oea_step_normal_scratch(n_ent, n_rel, ld, &g_off, &t_off);
        if (rc != OEA_OK) return rc;
        nrm_grad = static_cast<char *>(workspace) + g_off;
        nrm_touched = static_cast<char *>(workspace) + t_off;
    }
    // the gradients travel in the scratch's own type: fp32, or int64 fixed point in the deterministic build (exact sums:
    // the G-rank job then equals the single-GPU job bit for bit)
    const int32_t gdt = oea::kDetScratch ? OEA_COMM_I64 : OEA_COMM_F32;
    hipStream_t st = oea::as_stream(stream);
    oea_step_cfg step_cfg = *cfg;
#define OEA_TRY_RC(call) do { const int _rc = (call); if (_rc != OEA_OK) return _rc; } while (0)
    for (int32_t s = step_begin; s < step_end; ++s) {
        const int64_t b0 = offsets_host[s], nb = offsets_host[s + 1] - b0;
        if (nb <= 0) continue;
        const int64_t r_lo = nb * rank / world, r_hi = nb * (rank + 1) / world;
        const int64_t lo = b0 + r_lo, n = r_hi - r_lo;
        int64_t split = splits_host[s] - r_lo;
        split = split < 0 ? 0 : (split > n ? n : split);
        const int32_t *pos = pos_all + 3 * lo;
        int32_t *negs = ahead ? neg_buf + 3 * lo * (int64_t)k : neg_buf;
        if (n > 0 && k > 0 && !presampled && !sample_all)
            OEA_TRY_RC(oea_sample_negatives_pair(pos, n, split, k, side0, side1, seed, step_base + (uint32_t)s, (uint32_t)r_lo, 10, negs,
                                                 err_flag, stream));
        // every rank takes part in every exchange, also with an empty share of the batch (n == 0: nothing scored)
        OEA_TRY_RC(oea::comm_phase_mark(comm, st));
        OEA_TRY_RC(oea_triple_step_phase(ent, nullptr, n_ent, rel, rel_acc, n_rel, dim, ld, pos, n, k > 0 ? negs : nullptr, n * (int64_t)k,
                                         &step_cfg, workspace, loss_accum, OEA_PHASE_GRAD, stream));
        OEA_TRY_RC(oea::comm_phase_mark(comm, st));
        OEA_TRY_RC(oea_part_pack(workspace, n_ent, n_rel, ld, world, send, rel_x, stream));
        OEA_TRY_RC(oea::comm_phase_mark(comm, st));
        OEA_TRY_RC(oea_comm_reduce_scatter(comm, send, own, chunk, gdt, stream));
        OEA_TRY_RC(oea_comm_allreduce(comm, rel_x, n_rel * (int64_t)(ld + 1), gdt, stream));
        if (transh) {
            OEA_TRY_RC(oea_comm_allreduce(comm, nrm_grad, n_rel * (int64_t)ld, gdt, stream));
            OEA_TRY_RC(oea_comm_allreduce(comm, nrm_touched, n_rel, gdt, stream));
        }
        OEA_TRY_RC(oea::comm_phase_mark(comm, st));
        const int64_t n_items = step_items(step_cfg, n, n * (int64_t)k);
        OEA_TRY_RC(oea_part_apply(ent, acc_own, n_ent, rel, rel_acc, n_rel, ld, world, rank, own, rel_x, upd, &step_cfg, workspace, n_items,
                                  loss_accum, stream));
        if (transh) OEA_TRY_RC(oea_step_apply_normals(n_ent, n_rel, ld, &step_cfg, workspace, stream));
        OEA_TRY_RC(oea::comm_phase_mark(comm, st));
        OEA_TRY_RC(oea_allgather_rows(comm, upd, all, rpr, ld, stream));
        OEA_TRY_RC(oea::comm_phase_mark(comm, st));
        OEA_TRY_RC(oea_part_unpack(ent, n_ent, ld, world, rank, all, stream));
        OEA_TRY_RC(oea::comm_phase_mark(comm, st));
        ++step_cfg.opt_t;
    }
#undef OEA_TRY_RC
    return OEA_OK;
}

}  // extern "C"

// ---- the halo form of the one-call partitioned epoch ---------------------------------------------------------------------------------
static HaloGeom halo_geom(int64_t n_ent, int32_t world, int32_t rank, int64_t max_batch, int32_t k) {
    HaloGeom g;
    g.world = world; g.rank = rank;
    g.rpr = oea_part_rows_per_rank(n_ent, world);
    g.rpr32 = (g.rpr + 31) / 32 * 32;
    g.nwords = (int)((int64_t)world * g.rpr32 / 32);
    const int64_t share = oea::ceil_div(max_batch, world) + 1;
    g.cap = std::min<int64_t>(2 * share * (1 + (int64_t)k), (int64_t)world * g.rpr);
    g.cap = (g.cap + 3) / 4 * 4;
    return g;
}
struct HaloWs { uint32_t *bitmaps; int32_t *lists, *counts, *tables, *err; };
// tables: per step [4][world + 1] int32: 0 = prefix over owners of count(s, me, .), 1 = prefix over readers of count(s, ., me) (me
// excluded), 2 = where list(s, r, me) starts inside (s, r) (index r), 3 = spare
static size_t halo_ws_layout(const HaloGeom &g, int32_t steps, void *base, HaloWs *ws) {
    size_t off = 0;
    char *b = static_cast<char *>(base);
    auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return b ? b + o : nullptr; };
    uint32_t *bm = (uint32_t *)take(4 * (size_t)steps * g.world * g.nwords);
    int32_t *li = (int32_t *)take(4 * (size_t)steps * g.world * g.cap);
    int32_t *co = (int32_t *)take(4 * (size_t)steps * g.world * g.world);
    int32_t *ta = (int32_t *)take(4 * (size_t)steps * 4 * (g.world + 1));
    int32_t *er = (int32_t *)take(256);
    if (ws) { ws->bitmaps = bm; ws->lists = li; ws->counts = co; ws->tables = ta; ws->err = er; }
    return off;
}

extern "C" {

size_t oea_halo_workspace_bytes(int64_t n_ent, int32_t world, int32_t steps, int64_t max_batch, int32_t k) {
    if (world < 1 || steps < 0) return 0;
    return halo_ws_layout(halo_geom(n_ent, world, 0, max_batch, k), steps, nullptr, nullptr);
}

// bytes of EACH of the two exchange buffers: the larger of what a rank can send (its lists: <= cap rows) and receive (the other
// ranks' lists of its rows: <= (world - 1) * min(cap, rpr) rows), ld + 1 scratch elements per row
size_t oea_halo_buffer_bytes(int64_t n_ent, int32_t world, int64_t max_batch, int32_t k, int32_t ld) {
    if (world < 1) return 0;
    const HaloGeom g = halo_geom(n_ent, world, 0, max_batch, k);
    const int64_t rows = std::max<int64_t>(g.cap, (int64_t)(world - 1) * std::min<int64_t>(g.cap, g.rpr));
    return align256((size_t)rows * (ld + 1) * sizeof(grad_t));
}

// the plan alone: counts_host [step_end - step_begin][world][world] = distinct rows rank r's share of step s refers to that rank o owns
// (what sizes every message of oea_triple_epoch_range_halo), lists_host (may be NULL) [steps][world][cap] the row lists themselves
// (local indices j, id = j * world + owner, owner-major, ascending); *cap_out = entries per (step, rank).  Parity: tests hold both to
// a numpy restatement (oracle/np_oracle.py:halo_plan).
int oea_halo_plan(const int32_t *pos_all, const int32_t *neg_all, int32_t k, const int64_t *offsets_host, const int64_t *offsets_dev,
                  int32_t steps, int32_t step_begin, int32_t step_end, int64_t n_ent, int32_t world, void *halo_ws, size_t halo_ws_bytes,
                  int32_t *counts_host, int32_t *lists_host, int64_t *cap_out, void *stream) {
    OEA_REQUIRE(pos_all && offsets_host && offsets_dev && halo_ws && counts_host, "null pointer");
    OEA_REQUIRE(steps >= 0 && k >= 0 && (k == 0 || neg_all) && 0 <= step_begin && step_begin <= step_end && step_end <= steps && world >= 1, "arguments");
    hipStream_t st = oea::as_stream(stream);
    const int32_t nS = step_end - step_begin;
    int64_t max_batch = 0;
    for (int32_t s = 0; s < steps; ++s) max_batch = std::max(max_batch, offsets_host[s + 1] - offsets_host[s]);
    const HaloGeom g = halo_geom(n_ent, world, 0, max_batch, k);
    if (cap_out) *cap_out = g.cap;
    if (nS == 0) return OEA_OK;
    HaloWs hw;
    OEA_REQUIRE(halo_ws_bytes >= halo_ws_layout(g, nS, halo_ws, &hw), "halo workspace smaller than oea_halo_workspace_bytes");
    OEA_CHECK_HIP(hipMemsetAsync(hw.bitmaps, 0, 4 * (size_t)nS * world * g.nwords, st));
    OEA_CHECK_HIP(hipMemsetAsync(hw.err, 0, 4, st));
    const int64_t rows = offsets_host[step_end] - offsets_host[step_begin];
    if (rows > 0)
        halo_mark_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(rows * (k + 1), 256), 16384), 256, 0, st>>>(
            pos_all, neg_all, k, offsets_dev, step_begin, step_end, g, hw.bitmaps);
    halo_compact_kernel<<<(unsigned)(nS * world), 256, 0, st>>>(hw.bitmaps, g, hw.lists, hw.counts, hw.err);
    int32_t err = 0;
    OEA_CHECK_HIP(hipMemcpyAsync(counts_host, hw.counts, 4 * (size_t)nS * world * world, hipMemcpyDeviceToHost, st));
    if (lists_host) OEA_CHECK_HIP(hipMemcpyAsync(lists_host, hw.lists, 4 * (size_t)nS * world * g.cap, hipMemcpyDeviceToHost, st));
    OEA_CHECK_HIP(hipMemcpyAsync(&err, hw.err, 4, hipMemcpyDeviceToHost, st));
    OEA_CHECK_HIP(hipStreamSynchronize(st));
    OEA_REQUIRE(err == 0, "halo lists overflowed their capacity");
    return OEA_OK;
}

int oea_triple_epoch_range_halo(oea_comm_t comm, float *ent, float *acc_own, int64_t n_ent, float *rel, float *rel_acc,
                                int64_t n_rel, int32_t dim, int32_t ld, const int32_t *pos_all, const int64_t *offsets_host,
                                const int64_t *splits_host, int32_t steps, int32_t step_begin, int32_t step_end, int32_t k,
                                const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed, uint32_t step_base,
                                int32_t *neg_buf, int32_t *err_flag, const oea_step_cfg *cfg, void *workspace, double *loss_accum,
                                const int64_t *offsets_dev, const int64_t *splits_dev, void *halo_ws, size_t halo_ws_bytes,
                                void *buf_a, void *buf_b, size_t buf_bytes, void *rel_x, float *upd, float *all,
                                int64_t *stats_host, void *stream) {
    OEA_REQUIRE(comm && ent && rel && pos_all && offsets_host && splits_host && cfg && workspace && loss_accum && halo_ws && buf_a && buf_b &&
                rel_x && upd && all, "null pointer");
    OEA_REQUIRE(steps >= 0 && k >= 0 && 0 <= step_begin && step_begin <= step_end && step_end <= steps, "step range");
    OEA_REQUIRE(offsets_dev && splits_dev, "the halo exchange plans from the epoch's batches on the device: offsets_dev / splits_dev");
    OEA_REQUIRE(cfg->opt_kind == OEA_OPT_SGD || cfg->opt_kind == OEA_OPT_ADAGRAD, "the partition runs SGD / Adagrad");
    OEA_REQUIRE(cfg->score_kind == OEA_SCORE_TRANSE || cfg->score_kind == OEA_SCORE_TRANSH, "halo exchange: TransE / TransH scores");
    OEA_REQUIRE(ld % 4 == 0, "ld % 4 == 0");
    const int32_t world = oea_comm_size(comm), rank = oea_comm_rank(comm);
    const bool presampled = k > 0 && side0 == nullptr && side1 == nullptr;
    OEA_REQUIRE(k == 0 || (neg_buf && (presampled || (err_flag && side0 && side1))), "sampling needs neg_buf, err_flag and both sides");
    OEA_REQUIRE(k == 0 || presampled || step_begin == 0, "the whole epoch's negatives are drawn by the range that starts it");
    hipStream_t st = oea::as_stream(stream);
    if (k > 0 && !presampled) {
        const int rc = oea_sample_negatives_epoch(pos_all, offsets_host[steps], offsets_dev, splits_dev, steps, k, side0, side1, seed,
                                                  step_base, 10, neg_buf, err_flag, stream);
        if (rc != OEA_OK) return rc;
    }
    const int32_t nS = step_end - step_begin;
    if (stats_host) { stats_host[0] = stats_host[1] = stats_host[2] = stats_host[3] = 0; }
    if (nS == 0) return OEA_OK;
    int64_t max_batch = 0;
    for (int32_t s = 0; s < steps; ++s) max_batch = std::max(max_batch, offsets_host[s + 1] - offsets_host[s]);
    const HaloGeom g = halo_geom(n_ent, world, rank, max_batch, k);
    HaloWs hw;
    OEA_REQUIRE(halo_ws_bytes >= halo_ws_layout(g, nS, halo_ws, &hw), "halo workspace smaller than oea_halo_workspace_bytes(n_ent, world, steps, max_batch, k)");
    OEA_REQUIRE(buf_bytes >= oea_halo_buffer_bytes(n_ent, world, max_batch, k, ld), "exchange buffers smaller than oea_halo_buffer_bytes");
    // ---- plan: who refers to which rows in which step (every rank computes the same tables) -----------------------------------
    OEA_CHECK_HIP(hipMemsetAsync(hw.bitmaps, 0, 4 * (size_t)nS * world * g.nwords, st));
    OEA_CHECK_HIP(hipMemsetAsync(hw.err, 0, 4, st));
    const int64_t rows = offsets_host[step_end] - offsets_host[step_begin];
    if (rows > 0)
        halo_mark_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(rows * (k + 1), 256), 16384), 256, 0, st>>>(
            pos_all, neg_buf, k, offsets_dev, step_begin, step_end, g, hw.bitmaps);
    halo_compact_kernel<<<(unsigned)(nS * world), 256, 0, st>>>(hw.bitmaps, g, hw.lists, hw.counts, hw.err);
    std::vector<int32_t> counts((size_t)nS * world * world + 1);
    OEA_CHECK_HIP(hipMemcpyAsync(counts.data(), hw.counts, 4 * (size_t)nS * world * world, hipMemcpyDeviceToHost, st));
    OEA_CHECK_HIP(hipMemcpyAsync(counts.data() + (size_t)nS * world * world, hw.err, 4, hipMemcpyDeviceToHost, st));
    OEA_CHECK_HIP(hipStreamSynchronize(st));                          // the ONE host read of the call
    OEA_REQUIRE(counts[(size_t)nS * world * world] == 0, "halo lists overflowed their capacity (entries of a batch that are not corruptions?)");
    auto cnt = [&](int s, int r, int o) { return (int64_t)counts[((size_t)s * world + r) * world + o]; };
    std::vector<int32_t> tables((size_t)nS * 4 * (world + 1), 0);
    for (int s = 0; s < nS; ++s) {
        int32_t *t0 = tables.data() + (size_t)s * 4 * (world + 1), *t1 = t0 + (world + 1), *t2 = t1 + (world + 1);
        for (int o = 0; o < world; ++o) t0[o + 1] = t0[o] + (int32_t)cnt(s, rank, o);
        for (int r = 0; r < world; ++r) {
            t1[r + 1] = t1[r] + (r == rank ? 0 : (int32_t)cnt(s, r, rank));
            int32_t off = 0;
            for (int o = 0; o < rank; ++o) off += (int32_t)cnt(s, r, o);
            t2[r] = off;
        }
    }
    OEA_CHECK_HIP(hipMemcpyAsync(hw.tables, tables.data(), 4 * tables.size(), hipMemcpyHostToDevice, st));
    OEA_CHECK_HIP(hipStreamSynchronize(st));                          // (tables lives on this stack frame)
    const int64_t rpr = g.rpr;
    const bool transh = cfg->score_kind == OEA_SCORE_TRANSH;
    void *nrm_grad = nullptr, *nrm_touched = nullptr;
    if (transh) {
        int64_t g_off = 0, t_off = 0;
        const int rc = oea_step_normal_scratch(n_ent, n_rel, ld, &g_off, &t_off);
        if (rc != OEA_OK) return rc;
        nrm_grad = static_cast<char *>(workspace) + g_off;
        nrm_touched = static_cast<char *>(workspace) + t_off;
    }
    const int32_t gdt = oea::kDetScratch ? OEA_COMM_I64 : OEA_COMM_F32;
    StepWs ws;
    ws_layout(n_ent, n_rel, ld, workspace, &ws);
    grad_t *xa = static_cast<grad_t *>(buf_a), *xb = static_cast<grad_t *>(buf_b), *relx = static_cast<grad_t *>(rel_x);
    std::vector<int64_t> sc(world), sd(world), rc_(world), rd(world);
    oea_step_cfg step_cfg = *cfg;
#define OEA_TRY_RC(call) do { const int _rc = (call); if (_rc != OEA_OK) return _rc; } while (0)
    for (int32_t s = step_begin; s < step_end; ++s) {
        const int sl = s - step_begin;
        const int64_t b0 = offsets_host[s], nb = offsets_host[s + 1] - b0;
        if (nb <= 0) continue;
        const int64_t r_lo = nb * rank / world, r_hi = nb * (rank + 1) / world;
        const int64_t lo = b0 + r_lo, n = r_hi - r_lo;
        const int32_t *pos = pos_all + 3 * lo;
        int32_t *negs = k > 0 ? neg_buf + 3 * lo * (int64_t)k : nullptr;
        const int32_t *tab = hw.tables + (size_t)sl * 4 * (world + 1);
        const int32_t *t0h = tables.data() + (size_t)sl * 4 * (world + 1), *t1h = t0h + (world + 1);
        OEA_TRY_RC(oea::comm_phase_mark(comm, st));
        OEA_TRY_RC(oea_triple_step_phase(ent, nullptr, n_ent, rel, rel_acc, n_rel, dim, ld, pos, n, negs, n * (int64_t)k, &step_cfg, workspace,
                                         loss_accum, OEA_PHASE_GRAD, stream));
        OEA_TRY_RC(oea::comm_phase_mark(comm, st));
        // ---- PUSH -------------------------------------------------------------------------------------------------------------
        const int64_t n_send = t0h[world], n_recv = t1h[world];
        const int32_t *list_me = hw.lists + ((size_t)sl * world + rank) * g.cap, *lists_s = hw.lists + (size_t)sl * world * g.cap;
#define OEA_CALL(G, IT)                                                                                                             \
        if (n_send > 0) halo_push_pack_kernel<G, IT><<<(unsigned)std::min<int64_t>(oea::ceil_div(n_send, 256 / G), 16384), 256, 0, st>>>( \
            ws, ld, g, list_me, tab, xa);
        OEA_PART_DISPATCH(OEA_CALL)
#undef OEA_CALL
        halo_rel_pack_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(n_rel * (int64_t)ld, 256), 4096), 256, 0, st>>>(ws, n_rel, ld, relx);
        halo_rel_flags_kernel<<<(unsigned)oea::ceil_div(n_rel, 256), 256, 0, st>>>(ws, n_rel, ld, relx);
        OEA_TRY_RC(oea::comm_phase_mark(comm, st));
        for (int p = 0; p < world; ++p) {
            sc[p] = p == rank ? 0 : cnt(sl, rank, p) * (ld + 1);
            sd[p] = (int64_t)t0h[p] * (ld + 1);
            rc_[p] = p == rank ? 0 : cnt(sl, p, rank) * (ld + 1);
            rd[p] = (int64_t)t1h[p] * (ld + 1);
        }
        OEA_TRY_RC(oea_comm_alltoallv(comm, xa, sc.data(), sd.data(), xb, rc_.data(), rd.data(), gdt, stream));
        OEA_TRY_RC(oea_comm_allreduce(comm, relx, n_rel * (int64_t)(ld + 1), gdt, stream));
        if (transh) {
            OEA_TRY_RC(oea_comm_allreduce(comm, nrm_grad, n_rel * (int64_t)ld, gdt, stream));
            OEA_TRY_RC(oea_comm_allreduce(comm, nrm_touched, n_rel, gdt, stream));
        }
        if (stats_host) {
            int64_t out_rows = 0;
            for (int p = 0; p < world; ++p) out_rows += p == rank ? 0 : cnt(sl, rank, p);
            stats_host[0] += out_rows * (ld + 1) * (int64_t)sizeof(grad_t);
            stats_host[2] = std::max(stats_host[2], out_rows);
            stats_host[3] += 1;
        }
        OEA_TRY_RC(oea::comm_phase_mark(comm, st));
#define OEA_CALL(G, IT)                                                                                                             \
        if (n_recv > 0) halo_push_unpack_kernel<G, IT><<<(unsigned)std::min<int64_t>(oea::ceil_div(n_recv, 256 / G), 16384), 256, 0, st>>>( \
            ws, ld, g, lists_s, tab + 2 * (world + 1), tab + (world + 1), xb);
        OEA_PART_DISPATCH(OEA_CALL)
#undef OEA_CALL
        // ---- optimiser on the owned rows (the single-GPU job's group width at this table size) ---------------------------------
        const int64_t n_items = step_items(step_cfg, n, n * (int64_t)k);
        {
            const int grad_gpb = 256 / (ld <= 128 ? 32 : 64);
            const int n_part = n_items > 0 ? (int)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(n_items, grad_gpb), 1), kMaxBlocks) : 0;
#define OEA_CALL(G, IT)                                                                                                             \
            oea::launch_timed(halo_apply_kernel<G, IT>, (unsigned)std::min<int64_t>(std::max<int64_t>(oea::ceil_div(rpr + n_rel, 256 / G), 1), 16384), \
                              256, st, ent, acc_own, n_ent, rel, rel_acc, n_rel, ld, world, rank, rpr, relx, step_cfg, ws, n_part, loss_accum);
            if (ld <= 128 && apply_g16(n_ent, n_rel, ld)) {
                const int it16 = (ld + 15) / 16;
                if (it16 <= 2) { OEA_CALL(16, 2) }
                else if (it16 <= 4) { OEA_CALL(16, 4) }
                else if (it16 == 5) { OEA_CALL(16, 5) }
                else if (it16 == 6) { OEA_CALL(16, 6) }
                else if (it16 == 7) { OEA_CALL(16, 7) }
                else { OEA_CALL(16, 8) }
            } else {
                OEA_PART_DISPATCH(OEA_CALL)
            }
#undef OEA_CALL
        }
        if (transh) OEA_TRY_RC(oea_step_apply_normals(n_ent, n_rel, ld, &step_cfg, workspace, stream));
        OEA_TRY_RC(oea::comm_phase_mark(comm, st));
        // ---- PULL for the next step of the range: the current values of the rows its readers refer to ---------------------------
        int32_t s2 = s + 1;
        while (s2 < step_end && offsets_host[s2 + 1] - offsets_host[s2] <= 0) ++s2;
        if (s2 < step_end) {
            const int sl2 = s2 - step_begin;
            const int32_t *tab2 = hw.tables + (size_t)sl2 * 4 * (world + 1);
            const int32_t *u0 = tables.data() + (size_t)sl2 * 4 * (world + 1), *u1 = u0 + (world + 1);
            const int64_t n_out = u1[world], n_in = u0[world];
            float *fa = reinterpret_cast<float *>(xa), *fb = reinterpret_cast<float *>(xb);
            if (n_out > 0)
                halo_pull_kernel<true><<<(unsigned)std::min<int64_t>(oea::ceil_div(n_out * (ld / 4), 256), 16384), 256, 0, st>>>(
                    ent, ld, g, hw.lists + (size_t)sl2 * world * g.cap, tab2 + 2 * (world + 1), tab2 + (world + 1), fb);
            for (int p = 0; p < world; ++p) {
                sc[p] = p == rank ? 0 : cnt(sl2, p, rank) * ld;
                sd[p] = (int64_t)u1[p] * ld;
                rc_[p] = p == rank ? 0 : cnt(sl2, rank, p) * ld;
                rd[p] = (int64_t)u0[p] * ld;
            }
            OEA_TRY_RC(oea_comm_alltoallv(comm, fb, sc.data(), sd.data(), fa, rc_.data(), rd.data(), OEA_COMM_F32, stream));
            OEA_TRY_RC(oea::comm_phase_mark(comm, st));
            if (n_in > 0)
                halo_pull_kernel<false><<<(unsigned)std::min<int64_t>(oea::ceil_div(n_in * (ld / 4), 256), 16384), 256, 0, st>>>(
                    ent, ld, g, hw.lists + ((size_t)sl2 * world + rank) * g.cap, nullptr, tab2, fa);
            if (stats_host) {
                int64_t in_rows = 0;
                for (int p = 0; p < world; ++p) in_rows += p == rank ? 0 : cnt(sl2, rank, p);
                stats_host[1] += in_rows * ld * 4;
            }
        } else {
            // the range ends: one dense all-gather of the owned rows makes every copy the single-GPU table
            halo_owned_rows_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(rpr * (ld / 4), 256), 16384), 256, 0, st>>>(ent, n_ent, ld, world, rank, rpr, upd);
            OEA_TRY_RC(oea_allgather_rows(comm, upd, all, rpr, ld, stream));
            OEA_TRY_RC(oea::comm_phase_mark(comm, st));
            OEA_TRY_RC(oea_part_unpack(ent, n_ent, ld, world, rank, all, stream));
        }
        OEA_TRY_RC(oea::comm_phase_mark(comm, st));
        ++step_cfg.opt_t;
    }
#undef OEA_TRY_RC
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
