// topk.hip -- k-nearest neighbour search for truncated negative sampling.
//
// Replaces find_neighbours (modules/train/batch.py:157-165):
//     sim_mat = np.matmul(sub_embed, embed.T); np.argpartition(-sim_mat[i], k)[:k]
// k is large here (int((1-eps)*N): 1,499 of 15,000; 2,000 of 100,000), so instead of keeping k
// candidates per row on chip the search is a per-row SELECT on the fp32 keys:
//   1. a strip of rows of S is produced by the MFMA tile kernel (sim_rank.hip) into an HBM
//      workspace (strips of a few thousand rows: a full wave of workgroups per launch matters more than
//      keeping the strip in the 256 MB Infinity Cache -- see ops.topk_inner),
//   2. one workgroup per row, three coalesced reads of the row (16 B per lane):
//        a. histogram over 2048 LINEAR buckets between a sampled [lo, hi] of the row (monotone in
//           the value, ends clamped) -> the bucket b* holding the k-th largest value,
//        b. the few entries of bucket b* go to LDS and are ranked exactly by (value desc, column asc);
//           every wave counts the entries above b* in its quarter of the row,
//        c. ordered compaction by wave ballots: key > T, or key == T and column <= T's column.
//      Rows whose threshold bucket holds more than kCandCap entries (constant / heavily tied rows)
//      take the radix path instead: three 11+11+10-bit histogram passes + scan compaction.
//      Rows of >= 16,384 columns first try a ONE-read variant (row_select_sampled_kernel, below): bucket threshold
//      from a sample of the row, candidates kept in LDS, exact selection on the LDS copy.
//   All paths give the (value desc, column asc) selection in ascending column order, bit-identical with
//   oracle_topk_inner.
//
// Which path oea_topk_inner takes (all give the same result):
//   queries == candidates, 12,288 <= N <= 131,072   stream form (round 4): sampled thresholds; upper-triangle tile sweep on the bf16
//                                                   hi / lo split writing per-wave record streams (sim_rank.hip:
//                                                   topk_stream_sym_kernel, redo list + overflow pool for crowded waves);
//                                                   topk_bucket_kernel -> compact per-row lists; list_select_kernel (exact
//                                                   chains around the k-th approximate value); strip fallback for failed rows
//   queries == candidates, N >= 32,768, else        per-row segment lists from the symmetric sweep (round 3; bf16 or fp32 sweep)
//   nc >= 32,768, nq >= 4,096                       per-query segment lists from the full sweep (fp32)
//   otherwise                                       N x N strips + per-row select (above)
#include <math.h>
#include <stdlib.h>

#include <cmath>

#include <vector>

#include "common.h"

namespace {

__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f + 0.0f);   // -0 -> +0 so that key order == float order
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int SEL_THREADS = 256;

// block-wide exclusive scan of one int per thread (256 threads = 4 waves); returns the
// exclusive prefix, *total gets the block sum.
__device__ __forceinline__ int block_excl_scan(int v, int *s_wave /*[4]*/, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    *total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return base + incl - v;
}

// radix path (tie-heavy rows): hist = 2048 ints of LDS
__device__ void radix_select_row(const float *__restrict__ src, int64_t nc, int k, const int32_t *__restrict__ id_map,
                                 int32_t *__restrict__ o, int *hist) {
    __shared__ int s_wave[4];
    __shared__ uint32_t s_prefix;
    __shared__ int s_need;
    const int tid = threadIdx.x;

    uint32_t prefix = 0;        // key bits decided so far
    uint32_t mask = 0;          // which bits of the key are decided
    int need = k;               // how many still to take among keys matching the prefix
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = shifts[pass], bins = 1 << widths[pass];
        for (int b = tid; b < bins; b += SEL_THREADS) hist[b] = 0;
        __syncthreads();
        for (int64_t j = tid; j < nc; j += SEL_THREADS) {
            const uint32_t key = f2ord(src[j]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & (bins - 1)], 1);
        }
        __syncthreads();
        // find the bin (from the top) where the cumulative count reaches `need`
        if (tid < 64) {
            // each lane owns a contiguous range of bins, scanned from high to low
            const int per = bins / 64;
            const int hi = bins - 1 - tid * per;          // lane 0 owns the highest bins
            int sum = 0;
            for (int b = 0; b < per; ++b) sum += hist[hi - b];
            // exclusive prefix over lanes (lane 0 first)
            int incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (tid >= off) incl += t;
            }
            const int before = incl - sum;
            if (before < need && incl >= need) {
                int acc = before;
                for (int b = 0; b < per; ++b) {
                    const int c = hist[hi - b];
                    if (acc + c >= need) {
                        s_prefix = prefix | ((uint32_t)(hi - b) << shift);
                        s_need = need - acc;
                        break;
                    }
                    acc += c;
                }
            }
        }
        __syncthreads();
        prefix = s_prefix;
        need = s_need;
        mask |= (uint32_t)(bins - 1) << shift;
        __syncthreads();
    }
    // prefix is now the exact key T of the k-th largest value; `need` of the == T entries are taken.
    const uint32_t T = prefix;
    int taken_gt_eq = 0;   // running output position
    int taken_eq = 0;      // running count of == T entries seen
    for (int64_t base = 0; base < nc; base += SEL_THREADS) {
        const int64_t j = base + tid;
        uint32_t key = 0;
        const bool valid = j < nc;
        if (valid) key = f2ord(src[j]);
        const int is_eq = valid && key == T;
        const int is_gt = valid && key > T;
        int tot_eq, tot_sel;
        const int eq_before = block_excl_scan(is_eq, s_wave, &tot_eq);
        const int sel = is_gt || (is_eq && (taken_eq + eq_before) < need);
        const int pos = block_excl_scan(sel, s_wave, &tot_sel);
        if (sel) o[taken_gt_eq + pos] = id_map ? id_map[j] : (int32_t)j;
        taken_gt_eq += tot_sel;
        taken_eq += tot_eq;
    }
}

constexpr int kBins = 2048;
constexpr int kCandCap = 1024;

__device__ __forceinline__ int lin_bin(float v, float lo, float scale) {
    // monotone non-decreasing in v; bins 0 and kBins-1 collect what falls outside the sampled range
    const float b = fminf(fmaxf((v - lo) * scale + 1.0f, 0.0f), (float)(kBins - 1));
    return (int)b;
}

// the three-read select of one row (every thread of the workgroup calls it)
__device__ void select_row_3pass(const float *__restrict__ src, int64_t nc, int k, const int32_t *__restrict__ id_map,
                                 int32_t *__restrict__ o, int *hist /*[kBins]*/, uint32_t *c_key /*[kCandCap]*/,
                                 int *c_col /*[kCandCap]*/, int stop_after = 0 /* experiments: leave after phase N */) {
    __shared__ float s_red[8];
    __shared__ int s_bstar, s_need, s_ncand, s_gt[4], s_cbefore[4], s_tcol;
    __shared__ uint32_t s_tkey;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- sampled range ------------------------------------------------------------------------
    float mn = INFINITY, mx = -INFINITY;
    {
        const int64_t ns = nc < 1024 ? nc : 1024;
        const int64_t stride = nc / ns;
        for (int64_t i = tid; i < ns; i += SEL_THREADS) {
            const float v = src[i * stride];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, off, 64));
            mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        }
        if (lane == 0) { s_red[wave] = mn; s_red[4 + wave] = mx; }
    }
    for (int b = tid; b < kBins; b += SEL_THREADS) hist[b] = 0;
    if (tid < 4) { s_gt[tid] = 0; s_cbefore[tid] = 0; }
    if (tid == 0) s_ncand = 0;
    __syncthreads();
    const float lo = fminf(fminf(s_red[0], s_red[1]), fminf(s_red[2], s_red[3]));
    const float hi = fmaxf(fmaxf(s_red[4], s_red[5]), fmaxf(s_red[6], s_red[7]));
    const float scale = hi > lo ? (float)(kBins - 2) / (hi - lo) : 0.0f;

    // every wave owns a contiguous, tile-aligned quarter of the row (tiles of 256 columns)
    const int64_t tiles = (nc + 255) / 256;
    const int64_t tiles_per_wave = (tiles + 3) / 4;
    const int64_t seg0 = wave * tiles_per_wave * 256;
    const int64_t seg1 = seg0 + tiles_per_wave * 256 < nc ? seg0 + tiles_per_wave * 256 : nc;

    // ---- read 1: histogram --------------------------------------------------------------------
    constexpr int U = 4;                                   // tiles in flight per wave (latency hiding)
    for (int64_t t0 = seg0 + lane * 4; t0 - lane * 4 < seg1; t0 += 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                      // ld % 4 == 0: a 16 B load at c0 < nc stays inside the row
            const int64_t c0 = t0 + u * 256;
            v[u] = c0 < seg1 ? *reinterpret_cast<const float4 *>(src + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c0 = t0 + u * 256;
            if (c0 < seg1) atomicAdd(&hist[lin_bin(v[u].x, lo, scale)], 1);
            if (c0 + 1 < seg1) atomicAdd(&hist[lin_bin(v[u].y, lo, scale)], 1);
            if (c0 + 2 < seg1) atomicAdd(&hist[lin_bin(v[u].z, lo, scale)], 1);
            if (c0 + 3 < seg1) atomicAdd(&hist[lin_bin(v[u].w, lo, scale)], 1);
        }
    }
    __syncthreads();
    if (tid < 64) {          // the bucket (from the top) where the cumulative count reaches k
        constexpr int per = kBins / 64;
        const int top = kBins - 1 - tid * per;
        int sum = 0;
        for (int b = 0; b < per; ++b) sum += hist[top - b];
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (tid >= off) incl += t;
        }
        const int before = incl - sum;
        if (before < k && incl >= k) {
            int acc = before;
            for (int b = 0; b < per; ++b) {
                const int c = hist[top - b];
                if (acc + c >= k) { s_bstar = top - b; s_need = k - acc; break; }
                acc += c;
            }
        }
    }
    __syncthreads();
    const int bstar = s_bstar, need = s_need;
    if (stop_after == 1) return;
    if (hist[bstar] > kCandCap) {                       // block-uniform
        __syncthreads();
        radix_select_row(src, nc, k, id_map, o, hist);
        return;
    }

    // ---- read 2: candidates of bucket b*, per-wave count above it ---------------------------------
    int gt = 0;
    for (int64_t t0 = seg0 + lane * 4; t0 - lane * 4 < seg1; t0 += 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c0 = t0 + u * 256;
            v[u] = c0 < seg1 ? *reinterpret_cast<const float4 *>(src + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c0 = t0 + u * 256;
            const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (c0 + i < seg1) {
                    const int b = lin_bin(vv[i], lo, scale);
                    gt += b > bstar;
                    if (b == bstar) {
                        const int p = atomicAdd(&s_ncand, 1);
                        c_key[p] = f2ord(vv[i]);
                        c_col[p] = (int)(c0 + i);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) gt += __shfl_xor(gt, off, 64);
    if (lane == 0) s_gt[wave] = gt;
    __syncthreads();
    if (stop_after == 2) return;
    // exact rank of every candidate by (value desc, column asc)
    const int ncand = s_ncand;
    for (int i = tid; i < ncand; i += SEL_THREADS) {
        const uint32_t ki = c_key[i];
        const int ci = c_col[i];
        int rank = 0;
        for (int j = 0; j < ncand; ++j) {
            const uint32_t kj = c_key[j];
            rank += (kj > ki) || (kj == ki && c_col[j] < ci);
        }
        if (rank == need - 1) { s_tkey = ki; s_tcol = ci; }
        if (rank < need) {
            for (int w = 1; w < 4; ++w)
                if (ci < w * tiles_per_wave * 256) atomicAdd(&s_cbefore[w], 1);
        }
    }
    __syncthreads();
    const uint32_t tkey = s_tkey;
    const int tcol = s_tcol;
    if (stop_after == 3) return;
    int running = s_cbefore[wave];
    for (int w = 0; w < wave; ++w) running += s_gt[w];

    // ---- read 3: ordered compaction ---------------------------------------------------------------
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int64_t t0 = seg0 + lane * 4; t0 - lane * 4 < seg1; t0 += 256 * U) {       // wave-uniform trip count
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c0 = t0 + u * 256;
            v[u] = c0 < seg1 ? *reinterpret_cast<const float4 *>(src + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {                       // tiles in column order
            const int64_t c0 = t0 + u * 256;
            const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            bool sel[4];
            uint64_t bal[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t key = f2ord(vv[i]);
                sel[i] = (c0 + i < seg1) && (key > tkey || (key == tkey && (int)(c0 + i) <= tcol));
                bal[i] = __ballot(sel[i]);
            }
            if ((bal[0] | bal[1] | bal[2] | bal[3]) == 0ull) continue;
            int p = running + __popcll(bal[0] & lt) + __popcll(bal[1] & lt) + __popcll(bal[2] & lt) + __popcll(bal[3] & lt);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (sel[i]) o[p++] = id_map ? id_map[c0 + i] : (int32_t)(c0 + i);
            running += __popcll(bal[0]) + __popcll(bal[1]) + __popcll(bal[2]) + __popcll(bal[3]);
        }
    }
}

// The same select for rows of <= 1024 * TPW columns with the row held in REGISTERS: every wave loads its contiguous quarter
// of the row once (TPW float4 per lane, all requests in flight together) and the three passes of select_row_3pass run on
// the registers.  The three-read kernel exposed the L2 latency twelve times per row (3 passes x 4 trips of 4 tiles); at
// 15,000 columns the select was 0.85 ms of the 1.36 ms search (read 1 + histogram 0.32, read 2 0.12, read 3 0.38).
// The bucket range is the row's exact [min, max] here (it was a 1,024-entry sample); the selection is exact either way.
template <int TPW>
__device__ void select_row_regs(const float *__restrict__ src, int64_t nc, int k, const int32_t *__restrict__ id_map,
                                int32_t *__restrict__ o, int *hist /*[kBins]*/, uint32_t *c_key /*[kCandCap]*/,
                                int *c_col /*[kCandCap]*/) {
    __shared__ float s_red[8];
    __shared__ int s_bstar, s_need, s_ncand, s_gt[4], s_cbefore[4], s_tcol;
    __shared__ uint32_t s_tkey;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tiles = (nc + 255) / 256;
    const int64_t tiles_per_wave = (tiles + 3) / 4;                // <= TPW (checked by the launcher)
    const int64_t seg0 = wave * tiles_per_wave * 256;
    const int64_t seg1 = seg0 + tiles_per_wave * 256 < nc ? seg0 + tiles_per_wave * 256 : nc;
    float4 v[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {                                // ld % 4 == 0: a 16 B load at c0 < nc stays inside the row
        const int64_t c0 = seg0 + u * 256 + lane * 4;
        v[u] = c0 < seg1 ? *reinterpret_cast<const float4 *>(src + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int b = tid; b < kBins; b += SEL_THREADS) hist[b] = 0;
    if (tid < 4) { s_gt[tid] = 0; s_cbefore[tid] = 0; }
    if (tid == 0) s_ncand = 0;
    float mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int64_t c0 = seg0 + u * 256 + lane * 4;
        const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (c0 + i < seg1) { mn = fminf(mn, vv[i]); mx = fmaxf(mx, vv[i]); }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off, 64));
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    }
    if (lane == 0) { s_red[wave] = mn; s_red[4 + wave] = mx; }
    __syncthreads();
    const float lo = fminf(fminf(s_red[0], s_red[1]), fminf(s_red[2], s_red[3]));
    const float hi = fmaxf(fmaxf(s_red[4], s_red[5]), fmaxf(s_red[6], s_red[7]));
    const float scale = hi > lo ? (float)(kBins - 2) / (hi - lo) : 0.0f;
    // ---- pass 1: histogram ----------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int64_t c0 = seg0 + u * 256 + lane * 4;
        const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (c0 + i < seg1) atomicAdd(&hist[lin_bin(vv[i], lo, scale)], 1);
    }
    __syncthreads();
    if (tid < 64) {          // the bucket (from the top) where the cumulative count reaches k
        constexpr int per = kBins / 64;
        const int top = kBins - 1 - tid * per;
        int sum = 0;
        for (int b = 0; b < per; ++b) sum += hist[top - b];
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (tid >= off) incl += t;
        }
        const int before = incl - sum;
        if (before < k && incl >= k) {
            int acc = before;
            for (int b = 0; b < per; ++b) {
                const int c = hist[top - b];
                if (acc + c >= k) { s_bstar = top - b; s_need = k - acc; break; }
                acc += c;
            }
        }
    }
    __syncthreads();
    const int bstar = s_bstar, need = s_need;
    if (hist[bstar] > kCandCap) {                       // block-uniform: tie-heavy row, the radix path re-reads it
        __syncthreads();
        radix_select_row(src, nc, k, id_map, o, hist);
        return;
    }
    // ---- pass 2: candidates of bucket b*, per-wave count above it ----------------------------------------
    int gt = 0;
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int64_t c0 = seg0 + u * 256 + lane * 4;
        const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (c0 + i < seg1) {
                const int b = lin_bin(vv[i], lo, scale);
                gt += b > bstar;
                if (b == bstar) {
                    const int p = atomicAdd(&s_ncand, 1);
                    c_key[p] = f2ord(vv[i]);
                    c_col[p] = (int)(c0 + i);
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) gt += __shfl_xor(gt, off, 64);
    if (lane == 0) s_gt[wave] = gt;
    __syncthreads();
    const int ncand = s_ncand;
    for (int i = tid; i < ncand; i += SEL_THREADS) {     // exact rank of every candidate by (value desc, column asc)
        const uint32_t ki = c_key[i];
        const int ci = c_col[i];
        int rank = 0;
        for (int j = 0; j < ncand; ++j) {
            const uint32_t kj = c_key[j];
            rank += (kj > ki) || (kj == ki && c_col[j] < ci);
        }
        if (rank == need - 1) { s_tkey = ki; s_tcol = ci; }
        if (rank < need) {
            for (int w = 1; w < 4; ++w)
                if (ci < w * tiles_per_wave * 256) atomicAdd(&s_cbefore[w], 1);
        }
    }
    __syncthreads();
    const uint32_t tkey = s_tkey;
    const int tcol = s_tcol;
    int running = s_cbefore[wave];
    for (int w = 0; w < wave; ++w) running += s_gt[w];
    // ---- pass 3: ordered compaction (tiles in column order) ------------------------------------------------
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int64_t c0 = seg0 + u * 256 + lane * 4;
        const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        bool sel[4];
        uint64_t bal[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t key = f2ord(vv[i]);
            sel[i] = (c0 + i < seg1) && (key > tkey || (key == tkey && (int)(c0 + i) <= tcol));
            bal[i] = __ballot(sel[i]);
        }
        if ((bal[0] | bal[1] | bal[2] | bal[3]) == 0ull) continue;
        int p = running + __popcll(bal[0] & lt) + __popcll(bal[1] & lt) + __popcll(bal[2] & lt) + __popcll(bal[3] & lt);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (sel[i]) o[p++] = id_map ? id_map[c0 + i] : (int32_t)(c0 + i);
        running += __popcll(bal[0]) + __popcll(bal[1]) + __popcll(bal[2]) + __popcll(bal[3]);
    }
}

template <int TPW>
__global__ __launch_bounds__(SEL_THREADS) void row_select_regs_kernel(const float *__restrict__ s, int64_t n_rows, int64_t nc,
                                                                      int64_t ld, int k, const int32_t *__restrict__ id_map,
                                                                      int32_t *__restrict__ out /* [n_rows, k] */) {
    __shared__ int hist[kBins];
    __shared__ uint32_t c_key[kCandCap];
    __shared__ int c_col[kCandCap];
    select_row_regs<TPW>(s + (int64_t)blockIdx.x * ld, nc, k, id_map, out + (int64_t)blockIdx.x * k, hist, c_key, c_col);
}

__global__ __launch_bounds__(SEL_THREADS) void row_select_kernel(const float *__restrict__ s, int64_t n_rows, int64_t nc,
                                                                 int64_t ld, int k, const int32_t *__restrict__ id_map,
                                                                 int32_t *__restrict__ out /* [n_rows, k] */, int stop_after) {
    __shared__ int hist[kBins];
    __shared__ uint32_t c_key[kCandCap];
    __shared__ int c_col[kCandCap];
    select_row_3pass(s + (int64_t)blockIdx.x * ld, nc, k, id_map, out + (int64_t)blockIdx.x * k, hist, c_key, c_col, stop_after);
}

// The same select with the row held in LDS: ONE read of the strip instead of three (rows of up to ~32,000 columns fit the
// 160 KB of a CU; 15,000 columns = 60 KB + 16 KB of tables -> two workgroups per CU).  An experiment (see launch_select):
// the 15,000^2, k = 1,499 search spends 0.85 of its 1.36 ms in the three-read select, but NOT on the re-reads.
__global__ __launch_bounds__(SEL_THREADS) void row_select_cached_kernel(const float *__restrict__ s, int64_t n_rows, int64_t nc,
                                                                        int64_t ld, int k, const int32_t *__restrict__ id_map,
                                                                        int32_t *__restrict__ out /* [n_rows, k] */) {
    extern __shared__ __attribute__((aligned(16))) float row_lds[];          // nc rounded up to 4 floats
    __shared__ int hist[kBins];
    __shared__ uint32_t c_key[kCandCap];
    __shared__ int c_col[kCandCap];
    const float *src = s + (int64_t)blockIdx.x * ld;
    for (int64_t c = (int64_t)threadIdx.x * 4; c < nc; c += SEL_THREADS * 4)     // ld % 4 == 0: the last 16 B stay inside the row
        *reinterpret_cast<float4 *>(row_lds + c) = *reinterpret_cast<const float4 *>(src + c);
    __syncthreads();
    select_row_3pass(row_lds, nc, k, id_map, out + (int64_t)blockIdx.x * k, hist, c_key, c_col);
}

// ---- one-read select for long rows -------------------------------------------------------------------------------
// A SAMPLE of the row (64 contiguous runs of 128 entries spread over it) gives a bucket threshold that the true
// k-th largest value clears with ~3 sigma; ONE pass over the row then keeps, per wave and in column order, the
// ~1.25 k entries at or above that bucket in LDS; the exact (value desc, column asc) selection and the ordered
// compaction run on the LDS copy.  Rows whose candidates overflow the LDS lists or fall short of k (bad sample,
// heavy ties) take the three-read path -- the result is the same either way.
constexpr int kWaveCap = 1024;          // candidates per wave (4 waves): 32 KB of LDS, 48 KB in all -> 3 workgroups per CU
constexpr int kSampleRuns = 64, kSampleRun = 128;

__global__ __launch_bounds__(SEL_THREADS) void row_select_sampled_kernel(const float *__restrict__ s, int64_t n_rows, int64_t nc,
                                                                         int64_t ld, int k, const int32_t *__restrict__ id_map,
                                                                         int32_t *__restrict__ out) {
    __shared__ int hist[kBins];
    __shared__ float w_val[4][kWaveCap];
    __shared__ int w_col[4][kWaveCap];
    __shared__ uint32_t c_key[kCandCap];
    __shared__ int c_col[kCandCap];
    __shared__ float s_red[8];
    __shared__ int s_wcnt[4], s_wsel[4], s_bs, s_bstar, s_need, s_ncand, s_tcol;
    __shared__ uint32_t s_tkey;
    const int64_t row = blockIdx.x;
    const float *src = s + row * ld;
    int32_t *o = out + row * (int64_t)k;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- sample: run b starts at (b * (nc - 128) / 63) rounded down to a multiple of 4 ------------------------------
    float4 sv[8];
    {
        const int b = tid >> 2, part = tid & 3;
        const int64_t start = ((int64_t)b * (nc - kSampleRun) / (kSampleRuns - 1)) & ~(int64_t)3;
#pragma unroll
        for (int i = 0; i < 8; ++i) sv[i] = *reinterpret_cast<const float4 *>(src + start + part * 32 + i * 4);
    }
    float mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mn = fminf(fminf(mn, fminf(sv[i].x, sv[i].y)), fminf(sv[i].z, sv[i].w));
        mx = fmaxf(fmaxf(mx, fmaxf(sv[i].x, sv[i].y)), fmaxf(sv[i].z, sv[i].w));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off, 64));
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    }
    if (lane == 0) { s_red[wave] = mn; s_red[4 + wave] = mx; }
    for (int b = tid; b < kBins; b += SEL_THREADS) hist[b] = 0;
    if (tid == 0) { s_ncand = 0; s_bs = 0; }
    __syncthreads();
    const float lo = fminf(fminf(s_red[0], s_red[1]), fminf(s_red[2], s_red[3]));
    const float hi = fmaxf(fmaxf(s_red[4], s_red[5]), fmaxf(s_red[6], s_red[7]));
    const float scale = hi > lo ? (float)(kBins - 2) / (hi - lo) : 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        atomicAdd(&hist[lin_bin(sv[i].x, lo, scale)], 1);
        atomicAdd(&hist[lin_bin(sv[i].y, lo, scale)], 1);
        atomicAdd(&hist[lin_bin(sv[i].z, lo, scale)], 1);
        atomicAdd(&hist[lin_bin(sv[i].w, lo, scale)], 1);
    }
    __syncthreads();
    {   // bucket where the sample's cumulative count from the top reaches the expectation + 3 sigma + slack
        const float e = (float)k * (float)(kSampleRuns * kSampleRun) / (float)nc;
        const int want = (int)(e + 3.0f * sqrtf(e) + 8.0f);
        if (tid < 64) {
            constexpr int per = kBins / 64;
            const int top = kBins - 1 - tid * per;
            int sum = 0;
            for (int b = 0; b < per; ++b) sum += hist[top - b];
            int incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (tid >= off) incl += t;
            }
            const int before = incl - sum;
            if (before < want && incl >= want) {
                int acc = before;
                for (int b = 0; b < per; ++b) {
                    acc += hist[top - b];
                    if (acc >= want) { s_bs = top - b; break; }
                }
            }
        }
    }
    __syncthreads();
    const int bs = s_bs;                 // 0 when the sample never reaches `want`: every entry is a candidate -> overflow -> 3-pass

    // ---- the one pass: per-wave candidate lists in column order -----------------------------------------------------
    const int64_t tiles = (nc + 255) / 256;
    const int64_t tiles_per_wave = (tiles + 3) / 4;
    const int64_t seg0 = wave * tiles_per_wave * 256;
    const int64_t seg1 = seg0 + tiles_per_wave * 256 < nc ? seg0 + tiles_per_wave * 256 : nc;
    const uint64_t lt = (1ull << lane) - 1ull;
    constexpr int U = 4;
    int running = 0;
    for (int64_t t0 = seg0 + lane * 4; t0 - lane * 4 < seg1; t0 += 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c0 = t0 + u * 256;
            v[u] = c0 < seg1 ? *reinterpret_cast<const float4 *>(src + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c0 = t0 + u * 256;
            const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            bool sel[4];
            uint64_t bal[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sel[i] = (c0 + i < seg1) && lin_bin(vv[i], lo, scale) >= bs;
                bal[i] = __ballot(sel[i]);
            }
            if ((bal[0] | bal[1] | bal[2] | bal[3]) == 0ull) continue;
            int p = running + __popcll(bal[0] & lt) + __popcll(bal[1] & lt) + __popcll(bal[2] & lt) + __popcll(bal[3] & lt);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (sel[i]) {
                    if (p < kWaveCap) { w_val[wave][p] = vv[i]; w_col[wave][p] = (int)(c0 + i); }
                    ++p;
                }
            running += __popcll(bal[0]) + __popcll(bal[1]) + __popcll(bal[2]) + __popcll(bal[3]);
        }
    }
    if (lane == 0) s_wcnt[wave] = running;
    for (int b = tid; b < kBins; b += SEL_THREADS) hist[b] = 0;
    __syncthreads();
    const int n0 = s_wcnt[0], n1 = s_wcnt[1], n2 = s_wcnt[2], n3 = s_wcnt[3];
    bool fail = n0 > kWaveCap || n1 > kWaveCap || n2 > kWaveCap || n3 > kWaveCap || n0 + n1 + n2 + n3 < k;
    if (!fail) {
        // ---- exact k-th among the candidates: bucket histogram, then the threshold bucket ranked pairwise ----------------
        const int mine = s_wcnt[wave];
        for (int i = lane; i < mine; i += 64) atomicAdd(&hist[lin_bin(w_val[wave][i], lo, scale)], 1);
        __syncthreads();
        if (tid < 64) {
            constexpr int per = kBins / 64;
            const int top = kBins - 1 - tid * per;
            int sum = 0;
            for (int b = 0; b < per; ++b) sum += hist[top - b];
            int incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (tid >= off) incl += t;
            }
            const int before = incl - sum;
            if (before < k && incl >= k) {
                int acc = before;
                for (int b = 0; b < per; ++b) {
                    const int c = hist[top - b];
                    if (acc + c >= k) { s_bstar = top - b; s_need = k - acc; break; }
                    acc += c;
                }
            }
        }
        __syncthreads();
        const int bstar = s_bstar, need = s_need;
        fail = hist[bstar] > kCandCap;                      // block-uniform
        if (!fail) {
            for (int i = lane; i < mine; i += 64) {
                const float v = w_val[wave][i];
                if (lin_bin(v, lo, scale) == bstar) {
                    const int p = atomicAdd(&s_ncand, 1);
                    c_key[p] = f2ord(v);
                    c_col[p] = w_col[wave][i];
                }
            }
            __syncthreads();
            const int ncand = s_ncand;
            for (int i = tid; i < ncand; i += SEL_THREADS) {
                const uint32_t ki = c_key[i];
                const int ci = c_col[i];
                int rank = 0;
                for (int j = 0; j < ncand; ++j) {
                    const uint32_t kj = c_key[j];
                    rank += (kj > ki) || (kj == ki && c_col[j] < ci);
                }
                if (rank == need - 1) { s_tkey = ki; s_tcol = ci; }
            }
            __syncthreads();
            const uint32_t tkey = s_tkey;
            const int tcol = s_tcol;
            // ---- ordered compaction of the LDS lists: count per wave, then write ----------------------------------------
            int cnt = 0;
            for (int i = lane; i < mine; i += 64) {
                const uint32_t key = f2ord(w_val[wave][i]);
                cnt += key > tkey || (key == tkey && w_col[wave][i] <= tcol);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
            if (lane == 0) s_wsel[wave] = cnt;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wave; ++w) base += s_wsel[w];
            for (int i0 = 0; i0 < mine; i0 += 64) {
                const int i = i0 + lane;
                bool sel = false;
                int col = 0;
                if (i < mine) {
                    const uint32_t key = f2ord(w_val[wave][i]);
                    col = w_col[wave][i];
                    sel = key > tkey || (key == tkey && col <= tcol);
                }
                const uint64_t bal = __ballot(sel);
                if (sel) o[base + __popcll(bal & lt)] = id_map ? id_map[col] : col;
                base += __popcll(bal);
            }
            return;
        }
    }
    __syncthreads();
    select_row_3pass(src, nc, k, id_map, o, hist, c_key, c_col);
}


// ---- strip-free search for long candidate lists ---------------------------------------------------------------------
// (1) thr[q] = the r-th largest similarity of query q to a strided SAMPLE of kSample candidates: with r a little above
//     k * kSample / nc (3 sigma + slack) the whole row holds >= k values at or above it, and only ~1.35 k of them;
// (2) the tile sweep appends exactly those survivors to per-query list segments (sim_rank.hip: topk_append_kernel) --
//     the nq x nc strip is never written (it was 2 x 40 GB of traffic at 100,000 x 100,000);
// (3) list_select_kernel picks the exact top-k by (value desc, column asc) among a query's survivors in LDS and writes
//     them in ascending column order (bitonic sort of the k selected columns);
// (4) an append that finds its segment full goes to the row's spill list (kSpillCap entries, one global atomic each);
// (5) queries the select still gives up on (spill list full, fewer than k survivors, tie-heavy) are redone through the strip
//     path in batches inside the dead list storage (redo_failed_rows: one host read of their count).
// The result equals the strip path's bit for bit.
constexpr int kSample = 2048;          // 2,048 sampled candidates: survivors ~ r N / S +- 1/sqrt(r) (r ~ 71 at k/N = 2 %)
constexpr int kMaxSeg = 256;              // query-side segments per row
constexpr int kSpillCap = 512;             // entries of a row's spill list (appends that found their segment full)
constexpr int kMaxSegAll = 2304;          // + two candidate-side segments per query tile (symmetric search: T <= 1,100 tiles)

__device__ __forceinline__ float ord2f(uint32_t key) {
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key);
}

// one wave per row: the r-th largest of S = 64 * PER values held in registers, r << S (the sampled thresholds: r = 11..70 of
// 2,048 / 4,096).  Round 3: (1) radix select of the r-th largest of the 64 LANE MAXIMA -- a lower bound L of the answer (r
// distinct values are >= L); (2) the few hundred values >= L go to a per-wave LDS list (ballot + prefix count per register);
// (3) exact radix select of the r-th largest of that list.  32 + 32 * ceil(m / 64) ballots instead of 32 * PER (2,048 for 4,096
// samples: 0.81 ms per 70,000 rows, a tenth of a CSLS evaluation); a row whose list would overflow (heavy ties) takes the
// full register select.  Same value as before, bit for bit.
constexpr int kKthCap = 512;

template <int PER>
__global__ __launch_bounds__(256) void kth_value_kernel(const float *__restrict__ s, int64_t n_rows, int64_t ld, int r,
                                                        float *__restrict__ thr) {
    __shared__ uint32_t s_list[4][kKthCap];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= n_rows) return;                        // whole waves exit together
    uint32_t key[PER];
    const float *src = s + row * ld;
    uint32_t lmax = 0;
#pragma unroll
    for (int i = 0; i < PER / 4; ++i) {
        const float4 v = *reinterpret_cast<const float4 *>(src + (i * 64 + lane) * 4);
        key[4 * i] = f2ord(v.x); key[4 * i + 1] = f2ord(v.y); key[4 * i + 2] = f2ord(v.z); key[4 * i + 3] = f2ord(v.w);
        lmax = max(max(lmax, key[4 * i]), max(key[4 * i + 1], max(key[4 * i + 2], key[4 * i + 3])));
    }
    uint32_t prefix = 0;
    int need = r;
    bool full = r > 64;                               // fewer lanes than r: no bound from the lane maxima
    if (!full) {
        for (int bit = 31; bit >= 0; --bit) {         // r-th largest of the 64 lane maxima
            const uint32_t cand = prefix | (1u << bit), mask = ~((1u << bit) - 1u);
            const int cnt = __popcll(__ballot((lmax & mask) == cand));
            if (cnt >= need) prefix = cand; else need -= cnt;
        }
        const uint32_t bound = prefix;
        uint32_t *list = s_list[wave];
        int m = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const bool in = key[i] >= bound;
            const unsigned long long bal = __ballot(in);
            const int at = m + __popcll(bal & ((1ull << lane) - 1ull));
            if (in && at < kKthCap) list[at] = key[i];
            m += __popcll(bal);
        }
        full = m > kKthCap;                           // wave-uniform
        if (!full) {
            constexpr int LPER = kKthCap / 64;
            uint32_t lk[LPER];
#pragma unroll
            for (int i = 0; i < LPER; ++i) lk[i] = (i * 64 + lane) < m ? list[i * 64 + lane] : 0u;     // same wave wrote it: no barrier
            const int used = (m + 63) / 64;
            prefix = 0;
            need = r;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t cand = prefix | (1u << bit), mask = ~((1u << bit) - 1u);
                int cnt = 0;
#pragma unroll
                for (int i = 0; i < LPER; ++i)
                    if (i < used) cnt += __popcll(__ballot((i * 64 + lane) < m && (lk[i] & mask) == cand));
                if (cnt >= need) prefix = cand; else need -= cnt;
            }
        }
    }
    if (full) {
        prefix = 0;
        need = r;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = prefix | (1u << bit), mask = ~((1u << bit) - 1u);
            int cnt = 0;                              // wave total straight from the compare masks (scalar popcounts, no shuffles)
#pragma unroll
            for (int i = 0; i < PER; ++i) cnt += __popcll(__ballot((key[i] & mask) == cand));
            if (cnt >= need) prefix = cand; else need -= cnt;
        }
    }
    if (lane == 0) thr[row] = ord2f(prefix);
}

constexpr int kStreamMaxT = 1024;             // target tiles: <= 131,072 rows (the slice table is (64 + 4 T) * 8 B of LDS)
constexpr int kOvfChunkRecs = 512;            // == kOvfChunk (sim_rank.hip)
constexpr int kPerThread = 24;            // list entries a thread keeps in registers: lists of up to 6,144 survivors
constexpr int kBitWords = 8192;           // bitmap of selected columns in (dynamic) LDS: nc <= 262,144

// One workgroup per query: the survivors (a few thousand (value, column) pairs in <= kMaxSeg segments) are read ONCE into
// registers; the exact k-th by (value desc, column asc) comes from a 2048-bucket histogram over [thr, row max] plus a
// pairwise ranking inside the threshold bucket (as in the strip select); the selected columns are set in an LDS bitmap
// and enumerated in ascending order -- no sort.
__global__ __launch_bounds__(SEL_THREADS, 5) void list_select_kernel(const float *__restrict__ list_vals, const int32_t *__restrict__ list_cols,
                                                                   const int32_t *__restrict__ counts,
                                                                   const float *__restrict__ thr, int nseg, int cap, int64_t nc,
                                                                   int k, const int32_t *__restrict__ id_map, int32_t *__restrict__ out,
                                                                   int32_t *__restrict__ fail_rows, int32_t *__restrict__ n_fail,
                                                                   const uint2 *__restrict__ clists, const uint8_t *__restrict__ ccounts,
                                                                   int T, int ccap, const int32_t *__restrict__ spill_cnt,
                                                                   const uint2 *__restrict__ spill, int stop_after,
                                                                   const float *__restrict__ exact_src, int ld_src, int dim,
                                                                   const float *__restrict__ tol_ptr,
                                                                   const uint2 *__restrict__ compact, const int32_t *__restrict__ compact_cnt,
                                                                   int compact_cap, const uint8_t *__restrict__ row_fail,
                                                                   const float *__restrict__ exact_q = nullptr, int ld_q = 0) {
    // compact != NULL: the row's survivors are ONE list of compact_cnt[row] (value, column) pairs (topk_bucket_kernel); rows a
    // full stream or a full list may have lost entries of (row_fail) go to the strip fallback
    // tol_ptr != NULL: the list values are APPROXIMATE (v~ of the bf16 sweep, |v~ - v| <= tol = *tol_ptr) and the lists hold every
    // pair with v~ >= thr - tol.  With t~ = the k-th largest v~ (found exactly as before) the exact k-th value t is within tol
    // of t~, so  v~ > t~ + 2 tol  =>  v > t: selected;   v~ < t~ - 2 tol  =>  v < t: not selected;  the band in between (a dozen
    // entries) is decided by the EXACT k-ordered chains against rows of exact_src, (value desc, column asc) as everywhere.
    // The band must lie inside the lists (t~ - 2 tol >= thr - tol), else the row goes to the strip fallback.
    // exact_q != NULL: the query rows of this launch come from another table than the candidates (row r of the launch = exact_q + r ld_q)
    // stop_after (experiments, OEA_TOPK_SELECT_STOP): leave after phase 1 (lengths + scan), 2 (gather), 3 (histogram + bucket),
    // 4 (threshold bucket ranked), 5 (bitmap set); 0 = run to the end.  Results are garbage when it is set.
    // symmetric search (T > 0): T more segments per row, one per query tile, of (value, column) pairs (topk_append_sym_kernel)
    __shared__ int hist[kBins];
    // the segment offsets (lengths + scan, gather: phases 1-2) and the candidate tables (threshold bucket, band: phases 4-5) never live
    // at the same time (barriers in between): one array -- 8 KB less static LDS = FIVE workgroups per CU instead of four at nc = 100,000
    __shared__ uint32_t s_raw[(kMaxSegAll + 1 > 2 * kCandCap) ? kMaxSegAll + 1 : 2 * kCandCap];
    uint32_t *c_key = s_raw;
    int *c_col = reinterpret_cast<int *>(s_raw + kCandCap);
    int *s_off = reinterpret_cast<int *>(s_raw);
    extern __shared__ uint32_t bitmap[];      // ceil(nc / 32) words (dynamic: 12.5 KB at nc = 100,000)
    __shared__ float s_red[4];
    __shared__ int s_wave[4];
    __shared__ int s_bad, s_bstar, s_need, s_ncand, s_tcol, s_nband;
    __shared__ uint32_t s_tkey;
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_bad = 0; s_ncand = 0; s_nband = 0; }
    __syncthreads();
    const float tol = tol_ptr ? *tol_ptr : 0.f;
    // segment lengths -> exclusive offsets (block scan: the symmetric search has hundreds of segments per row)
    // + the row's spill list as the last segment (entries whose own segment was full)
    const int nst = nseg + T + 1;
    constexpr int kSegPer = (kMaxSegAll + SEL_THREADS - 1) / SEL_THREADS;
    int seg_c[kSegPer];
    int mine = 0;
    if (compact) {
        const int c = compact_cnt[row];
        if (tid == 0 && row_fail[row]) s_bad = 1;
        mine = tid == 0 ? c : 0;
    }
#pragma unroll
    for (int u = 0; u < kSegPer; ++u) {
        if (compact) { seg_c[u] = 0; continue; }
        const int sg = tid * kSegPer + u;
        int c = 0;
        if (sg < nseg) {
            c = min(counts[row * nseg + sg], cap - 1);             // the last slot of a segment is scratch (topk_append_kernel)
        } else if (sg < nseg + T) {
            c = min((int)ccounts[row * T + (sg - nseg)], ccap - 1);
        } else if (sg < nst) {
            c = spill_cnt[row];
            if (c > kSpillCap) s_bad = 1;                        // the spill list overflowed as well: strip path
        }
        seg_c[u] = c;
        mine += c;
    }
    int total_all;
    int run = block_excl_scan(mine, s_wave, &total_all);
#pragma unroll
    for (int u = 0; u < kSegPer; ++u) {
        const int sg = tid * kSegPer + u;
        if (sg <= nst) s_off[sg] = run;
        run += seg_c[u];
    }
    if (tid == 0 && (total_all < k || total_all > kPerThread * SEL_THREADS)) s_bad = 1;
    const int words = (int)((nc + 31) / 32);
    for (int b = tid; b < kBins; b += SEL_THREADS) hist[b] = 0;
    // owner[i] = the segment of list position i, written by the thread that counted the segment: the gather below finds an
    // entry with two LDS reads instead of a binary search over ~1,600 segment offsets (11 dependent reads per entry: the
    // gather was 2.3 ms of the select's 4.7 at 100,000 rows).  The table lives in the bitmap's storage, which is zeroed after
    // the gather.
    uint16_t *owner = reinterpret_cast<uint16_t *>(bitmap);
    if (!compact && total_all <= kPerThread * SEL_THREADS) {
        int at = run;                                            // == s_off[tid * kSegPer + kSegPer] after the loop above
#pragma unroll
        for (int u = kSegPer - 1; u >= 0; --u) {
            at -= seg_c[u];
            const int sg = tid * kSegPer + u;
            for (int e = 0; e < seg_c[u]; ++e) owner[at + e] = (uint16_t)sg;
        }
    }
    __syncthreads();
    const int total = compact ? total_all : s_off[nst];
    bool fail = s_bad != 0;                                      // block-uniform from here on
    if (stop_after == 1) return;
    if (!fail) {
        float val[kPerThread];
        int col[kPerThread];
        float mx = -INFINITY;
        const float *vb = list_vals + row * nseg * (int64_t)cap;
        const int32_t *cb = list_cols + row * nseg * (int64_t)cap;
#pragma unroll
        for (int e = 0; e < kPerThread; ++e) {
            const int i = tid + e * SEL_THREADS;
            val[e] = -INFINITY;
            col[e] = -1;
            if (i < total && compact) {
                const uint2 pr = compact[row * compact_cap + i];
                val[e] = __uint_as_float(pr.x);
                col[e] = (int)pr.y;
                mx = fmaxf(mx, val[e]);
            } else if (i < total) {
                // (round 3, measured and dropped: consecutive positions per thread with one binary search + a forward walk
                //  instead of a binary search per position -- 4.41 -> 5.02 ms: the strided assignment keeps a wave's loads in
                //  neighbouring entries of the same segments)
                const int lo_s = owner[i];                       // s_off[lo_s] <= i < s_off[lo_s + 1]
                if (lo_s < nseg) {
                    const int64_t at = (int64_t)lo_s * cap + (i - s_off[lo_s]);
                    val[e] = vb[at];
                    col[e] = cb[at];
                } else if (lo_s == nst - 1) {
                    const uint2 pr = spill[row * kSpillCap + (i - s_off[lo_s])];
                    val[e] = __uint_as_float(pr.x);
                    col[e] = (int)pr.y;
                } else {
                    const uint2 pr = clists[((int64_t)row * T + (lo_s - nseg)) * ccap + (i - s_off[lo_s])];
                    val[e] = __uint_as_float(pr.x);
                    col[e] = (int)pr.y;
                }
                mx = fmaxf(mx, val[e]);
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        if (lane == 0) s_red[wave] = mx;
        __syncthreads();
        for (int w = tid; w < words; w += SEL_THREADS) bitmap[w] = 0u;       // the owner table is dead: its storage becomes the bitmap
        if (stop_after == 2) { if (mx == 12345.f) out[row] = col[0] + col[kPerThread - 1]; return; }
        const float lo = thr[row] - tol;                         // every survivor is >= the sweep's cut (the same expression)
        const float hi = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        const float scale = hi > lo ? (float)(kBins - 2) / (hi - lo) : 0.0f;
#pragma unroll
        for (int e = 0; e < kPerThread; ++e)
            if (col[e] >= 0) atomicAdd(&hist[lin_bin(val[e], lo, scale)], 1);
        __syncthreads();
        if (tid < 64) {          // the bucket (from the top) where the cumulative count reaches k
            constexpr int per = kBins / 64;
            const int top = kBins - 1 - tid * per;
            int sum = 0;
            for (int b = 0; b < per; ++b) sum += hist[top - b];
            int incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (tid >= off) incl += t;
            }
            const int before = incl - sum;
            if (before < k && incl >= k) {
                int acc = before;
                for (int b = 0; b < per; ++b) {
                    const int c = hist[top - b];
                    if (acc + c >= k) { s_bstar = top - b; s_need = k - acc; break; }
                    acc += c;
                }
            }
        }
        __syncthreads();
        const int bstar = s_bstar, need = s_need;
        if (stop_after == 3) return;
        fail = hist[bstar] > kCandCap;                          // tie-heavy row: the fallback's radix path handles it
        if (!fail) {
#pragma unroll
            for (int e = 0; e < kPerThread; ++e)
                if (col[e] >= 0 && lin_bin(val[e], lo, scale) == bstar) {
                    const int p = atomicAdd(&s_ncand, 1);
                    c_key[p] = f2ord(val[e]);
                    c_col[p] = col[e];
                }
            __syncthreads();
            const int ncand = s_ncand;
            for (int i = tid; i < ncand; i += SEL_THREADS) {
                const uint32_t ki = c_key[i];
                const int ci = c_col[i];
                int rank = 0;
                for (int j = 0; j < ncand; ++j) {
                    const uint32_t kj = c_key[j];
                    rank += (kj > ki) || (kj == ki && c_col[j] < ci);
                }
                if (rank == need - 1) { s_tkey = ki; s_tcol = ci; }
            }
            __syncthreads();
            const uint32_t tkey = s_tkey;
            const int tcol = s_tcol;
            if (stop_after == 4) return;
            if (tol_ptr) {
                const float tv = ord2f(tkey);
                const float hi2 = tv + 2.0f * tol, lo2 = tv - 2.0f * tol;      // complementary classes: > hi2 | [lo2, hi2] | < lo2
                int above = 0;
#pragma unroll
                for (int e = 0; e < kPerThread; ++e) {
                    if (col[e] < 0) continue;
                    if (val[e] > hi2) {
                        atomicOr(&bitmap[col[e] >> 5], 1u << (col[e] & 31));
                        ++above;
                    } else if (val[e] >= lo2) {
                        const int p = atomicAdd(&s_nband, 1);
                        if (p < kCandCap) c_col[p] = col[e];
                    }
                }
                int n_above;
                block_excl_scan(above, s_wave, &n_above);                     // (two barriers: s_nband and c_col are complete)
                const int nband = s_nband, need2 = k - n_above;
                fail = lo2 < lo || nband > kCandCap || need2 < 0 || need2 > nband;      // block-uniform
                if (!fail) {
                    const float *__restrict__ a = exact_q ? exact_q + row * (int64_t)ld_q : exact_src + row * (int64_t)ld_src;
                    for (int i = tid; i < nband; i += SEL_THREADS) {
                        const float *__restrict__ b = exact_src + (int64_t)c_col[i] * ld_src;
                        float acc = 0.f;
                        int kk = 0;
                        for (; kk + 32 <= dim; kk += 32) {                    // 16 loads in flight, then the chain in k order
                            float4 x[8], y[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) { x[u] = oea::ld4(a + kk + 4 * u); y[u] = oea::ld4(b + kk + 4 * u); }
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                acc = fmaf(x[u].x, y[u].x, acc); acc = fmaf(x[u].y, y[u].y, acc);
                                acc = fmaf(x[u].z, y[u].z, acc); acc = fmaf(x[u].w, y[u].w, acc);
                            }
                        }
                        for (; kk + 4 <= dim; kk += 4) {
                            const float4 x = oea::ld4(a + kk), y = oea::ld4(b + kk);
                            acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
                        }
                        for (; kk < dim; ++kk) acc = fmaf(a[kk], b[kk], acc);
                        c_key[i] = f2ord(acc);
                    }
                    __syncthreads();
                    for (int i = tid; i < nband; i += SEL_THREADS) {
                        const uint32_t ki = c_key[i];
                        const int ci = c_col[i];
                        int rank = 0;
                        for (int j = 0; j < nband; ++j) {
                            const uint32_t kj = c_key[j];
                            rank += (kj > ki) || (kj == ki && c_col[j] < ci);
                        }
                        if (rank < need2) atomicOr(&bitmap[ci >> 5], 1u << (ci & 31));
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < kPerThread; ++e) {
                    if (col[e] < 0) continue;
                    const uint32_t key = f2ord(val[e]);
                    if (key > tkey || (key == tkey && col[e] <= tcol)) atomicOr(&bitmap[col[e] >> 5], 1u << (col[e] & 31));   // exactly k bits
                }
            }
            __syncthreads();
            if (!fail) {
            if (stop_after == 5) return;
            // enumerate the set bits in ascending column order: contiguous word ranges per thread + block scan
            const int wpt = (words + SEL_THREADS - 1) / SEL_THREADS;
            const int w0 = tid * wpt, w1 = min(words, w0 + wpt);
            int cnt = 0;
            for (int w = w0; w < w1; ++w) cnt += __popc(bitmap[w]);
            int tot;
            int pos = block_excl_scan(cnt, s_wave, &tot);
            int32_t *o = out + row * (int64_t)k;
            if (k <= kBins) {
                // the k columns are staged in LDS (the histogram is dead) and leave coalesced, their id_map look-ups
                // independent of each other: a thread writing its own ~8 columns one by one issued 64 partial-line stores
                // per instruction and waited for every look-up (1.0 ms of the select at 100,000 rows without id_map)
                for (int w = w0; w < w1; ++w) {
                    uint32_t bits = bitmap[w];
                    while (bits) {
                        const int b = __ffs((int)bits) - 1;
                        bits &= bits - 1;
                        hist[pos++] = 32 * w + b;
                    }
                }
                __syncthreads();
                for (int i = tid; i < k; i += SEL_THREADS) {
                    const int c = hist[i];
                    o[i] = id_map ? id_map[c] : c;
                }
                return;
            }
            for (int w = w0; w < w1; ++w) {
                uint32_t bits = bitmap[w];
                while (bits) {
                    const int b = __ffs((int)bits) - 1;
                    bits &= bits - 1;
                    const int c = 32 * w + b;
                    o[pos++] = id_map ? id_map[c] : c;
                }
            }
            return;
            }
        }
    }
    if (tid == 0) fail_rows[atomicAdd(n_fail, 1)] = (int32_t)row;
}

// ---- gather of <= kFbRows failed rows (the CSLS means' bulk fallback, sim_rank.hip) ---------------------------------------
constexpr int kFbRows = 128;

__global__ void gather_fail_rows_kernel(const float *__restrict__ qp, int kp, const int32_t *__restrict__ fail_rows,
                                        const int32_t *__restrict__ n_fail, float *__restrict__ dst) {
    const int nf = min(*n_fail, kFbRows);
    const int cpr = kp / 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kFbRows * cpr; i += gridDim.x * blockDim.x) {
        const int f = i / cpr, c = i - f * cpr;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < nf) v = oea::ld4(qp + (int64_t)fail_rows[f] * kp + 4 * c);
        oea::st4(dst + (int64_t)f * kp + 4 * c, v);
    }
}


struct ListPlan {
    bool ok = false;
    int r = 0, cap = 0, chunks = 0, nseg = 0;
    int64_t rows_per = 0, stride = 0, ld = 0;
    size_t off_thr = 0, off_counts = 0, off_fail = 0, off_nfail = 0, off_lists = 0, off_strip = 0, cols_off = 0, off_spcnt = 0, off_spill = 0;
};

// sample rank of the threshold = e + sigma * sqrt(e) + slack (e = k S / N).  A row with fewer than k survivors costs 0.4 us in
// the batched strip fallback, so the margin is modest: 2.5 sigma + 4 (r = 61 at k / N = 2 %, ~3.0 % of a row survives, ~0.6 % of
// the rows come up short) measured 20.0 / 23.3 ms (random / trained tables, 100,000^2, k = 2,000) against 20.6 / 24.1 ms at
// 3.5 sigma + 8 and 20.7 / 25.1 ms at 1.5 sigma + 2.  OEA_TOPK_SIGMA / OEA_TOPK_SLACK override.
static int threshold_rank(double e) {
    static const double sigma = [] { const char *v = getenv("OEA_TOPK_SIGMA"); return v ? atof(v) : 2.5; }();
    static const double slack = [] { const char *v = getenv("OEA_TOPK_SLACK"); return v ? atof(v) : 4.0; }();
    return (int)(e + sigma * std::sqrt(e) + slack);
}

// dynamic LDS of list_select_kernel: the bitmap of selected columns, or the owner table of the gather that precedes it
static size_t select_lds_bytes(int64_t nc) {
    return std::max(sizeof(uint32_t) * (size_t)((nc + 31) / 32), sizeof(uint16_t) * (size_t)kPerThread * SEL_THREADS);
}

static int select_stop() {
    static const int v = [] { const char *e = getenv("OEA_TOPK_SELECT_STOP"); return e ? atoi(e) : 0; }();
    return v;
}

// workspace layout of one pass of `rows` queries; ok = false when the strip path should run instead
static ListPlan plan_lists(int64_t nq, int64_t nc, int k, size_t ws_bytes) {
    ListPlan p;
    // OEA_TOPK_LISTS_MIN = 8192 / 16384 measured on random rows (gpurun_out r03q): 30,000^2 x 100, k = 600 2.84 vs 3.39 ms through
    // strips, 15,000^2 even; not enabled: trained tables overflow the lists at these sizes (see plan_sym)
    static const int64_t min_nc = [] { const char *e = getenv("OEA_TOPK_LISTS_MIN"); return e ? (int64_t)atoll(e) : (int64_t)32768; }();
    if (nq < 4096 || nc < min_nc) return p;
    const double e = (double)k * kSample / (double)nc;
    // sample rank of the threshold: the row then holds N * Beta(r, S - r + 1) survivors, i.e. about r N / S +- a relative
    // 1 / sqrt(r); rows left with fewer than k survivors go through the batched strip fallback (threshold_rank)
    p.r = threshold_rank(e);
    const double m_total = (double)p.r * (double)nc / kSample;
    if (p.r >= kSample / 2 || m_total * 1.4 > kPerThread * SEL_THREADS || nc > (int64_t)kBitWords * 32) return p;
    p.stride = nc / kSample;
    p.ld = (nc + 31) / 32 * 32;
    auto a256 = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t fixed = 4096;
    if (ws_bytes <= fixed) return p;
    int64_t rows = nq;
    for (int iter = 0; iter < 8; ++iter) {
        const int chunks = oea::topk_append_chunks(rows, nc);
        const int nseg = 4 * chunks;
        if (nseg > kMaxSeg) return p;
        const double m = m_total / nseg;
        const int cap = ((int)(m + 8.0 * std::sqrt(m) + 32.0) + 7) / 8 * 8;
        const size_t per_row = sizeof(float) * kSample + (size_t)nseg * cap * 8 + (size_t)nseg * 4 + 4 + 4 + 16 + 4 + 8 * (size_t)kSpillCap;
        if ((size_t)128 * nseg * cap * 4 >= ((size_t)1 << 31)) return p;          // 32-bit byte offsets inside a query tile
        int64_t fit = (int64_t)((ws_bytes - fixed) / per_row) / 128 * 128;
        if (fit >= nq) fit = nq;
        if (fit < 128) return p;
        p.chunks = chunks; p.nseg = nseg; p.cap = cap;
        if (fit >= rows) { p.rows_per = rows; p.ok = true; break; }
        rows = fit;
    }
    if (!p.ok) return p;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += a256(bytes); return o; };
    p.off_thr = take(sizeof(float) * (size_t)p.rows_per);
    p.off_counts = take(sizeof(int32_t) * (size_t)p.rows_per * p.nseg);
    p.off_fail = take(sizeof(int32_t) * (size_t)p.rows_per);
    p.off_nfail = take(256);
    p.off_spcnt = take(sizeof(int32_t) * (size_t)p.rows_per);
    p.off_spill = take(8 * (size_t)p.rows_per * kSpillCap);
    p.cols_off = a256((size_t)p.rows_per * p.nseg * p.cap * 4);
    p.off_lists = take(2 * p.cols_off);
    p.off_strip = take(sizeof(float) * (size_t)p.rows_per * kSample);
    // the batched strip fallback lives in the list storage: at least one 128-row batch must fit there
    p.ok = off <= ws_bytes && 2 * p.cols_off >= 128 * (sizeof(float) * ((size_t)4096 + (size_t)p.ld) + sizeof(int32_t) * (size_t)k) + 256;
    return p;
}

// ---- symmetric search (queries == candidates): the tiles on and above the diagonal serve rows AND columns ---------------
struct SymPlan {
    bool ok = false;
    int r = 0, T = 0, L = 0, groups = 0, nseg = 0, cap = 0, ccap = 0, n_items = 0;
    int64_t stride = 0, ld = 0;
    size_t off_thr = 0, off_counts = 0, off_ccounts = 0, off_fail = 0, off_nfail = 0, off_items = 0, off_vals = 0, off_cols = 0,
           off_clists = 0, off_strip = 0, off_spcnt = 0, off_spill = 0, total = 0;
};

static SymPlan plan_sym(int64_t n, int k, size_t ws_bytes) {
    SymPlan p;
    // Measured in round 3 (OEA_TOPK_SYM_MIN = 8192): on RANDOM unit rows the upper-triangle sweep + list select beat the N x N
    // strip + three-read row select below 32,768 rows too (15,000 rows, k = 1,499: 1.10 vs 1.39 ms; 30,000 rows, k = 600: 2.29 vs
    // 3.39 ms, gpurun_out r03p) -- but on TRAINED tables, which is what a refresh sees, the neighbours crowd into few candidate
    // ranges, list segments overflow and the failed rows are redone through the strip: 1.74 vs 1.41 ms at 15,000 rows in the
    // bench line (gpurun_out r03r).  The limit stays at 32,768.
    static const int64_t min_n = [] { const char *e = getenv("OEA_TOPK_SYM_MIN"); return e ? (int64_t)atoll(e) : (int64_t)32768; }();
    if (n < min_n) return p;
    const double e = (double)k * kSample / (double)n;
    p.r = threshold_rank(e);
    const double frac = (double)p.r / kSample;                  // expected survivor fraction of a row
    const double m_total = frac * (double)n;
    if (p.r >= kSample / 2 || m_total * 1.4 > kPerThread * SEL_THREADS || n > (int64_t)kBitWords * 32) return p;
    p.T = (int)oea::ceil_div(n, 128);
    p.groups = 16;                                              // work items per (full) query tile row: L tiles each
    p.L = std::max(8, (int)oea::ceil_div(p.T, p.groups));
    p.groups = (int)oea::ceil_div(p.T, p.L);
    p.nseg = 4 * p.groups;
    if (p.nseg > kMaxSeg || p.nseg + 2 * p.T + 1 >= kMaxSegAll) return p;
    const double m = frac * p.L * 128.0 / 4.0;                  // per query-side segment (2 wave rows x 2 half-waves share an item)
    p.cap = ((int)(m * (1.0 + 4.0 / std::sqrt((double)p.r)) + 8.0 * std::sqrt(m) + 32.0) + 7) / 8 * 8;
    const double mc = frac * 64.0;                              // per candidate-side segment (the 64 queries of one wave column)
    p.ccap = ((int)(mc * (1.0 + 4.0 / std::sqrt((double)p.r)) + 6.0 * std::sqrt(mc) + 4.0) + 3) / 4 * 4;
    static const int env_ccap = [] { const char *e = getenv("OEA_TOPK_CCAP"); return e ? atoi(e) : 0; }();       // experiments
    if (env_ccap >= 4) p.ccap = env_ccap;
    if (p.ccap > 248 || (size_t)128 * p.nseg * p.cap * 4 >= ((size_t)1 << 31)) return p;
    int64_t items = 0;
    for (int c = 0; c < p.groups; ++c) items += std::min(p.T, (c + 1) * p.L);
    p.n_items = (int)items;
    p.stride = n / kSample;
    p.ld = (n + 31) / 32 * 32;
    auto a256 = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += a256(bytes); return o; };
    p.off_thr = take(sizeof(float) * (size_t)n);
    p.off_counts = take(sizeof(int32_t) * (size_t)n * p.nseg);
    p.off_ccounts = take((size_t)n * p.T * 2);
    p.off_fail = take(sizeof(int32_t) * (size_t)n);
    p.off_nfail = take(256);
    p.off_items = take(sizeof(int32_t) * 4 * (size_t)p.n_items);
    p.off_spcnt = take(sizeof(int32_t) * (size_t)n);
    p.off_spill = take(8 * (size_t)n * kSpillCap);
    p.off_vals = take(sizeof(float) * (size_t)n * p.nseg * p.cap);
    p.off_cols = take(sizeof(int32_t) * (size_t)n * p.nseg * p.cap);
    p.off_clists = take(8 * (size_t)n * p.T * 2 * p.ccap);
    p.off_strip = take(sizeof(float) * (size_t)n * kSample);
    p.total = off;
    p.ok = off <= ws_bytes;
    return p;
}

struct StreamPlan {
    bool ok = false;
    int r = 0, T = 0, L = 0, groups = 0, n_items = 0, rcap = 0, ccap = 0, row_cap = 0, ovf_chunks = 0;
    int64_t stride = 0, ld = 0;
    size_t off_thr = 0, off_fail = 0, off_nfail = 0, off_items = 0, off_rcnt = 0, off_coff = 0, off_lcnt = 0, off_rowfail = 0, off_strip = 0,
           off_rstream = 0, off_cstream = 0, off_lists = 0, off_ovf = 0, off_ovflen = 0, off_redo = 0, stream_bytes = 0, total = 0;
    int redo_cap = 0;
};

// the stream form of the symmetric search (bf16 sweep only): same thresholds and work items as plan_sym
static StreamPlan plan_stream(int64_t n, int k, size_t ws_bytes) {
    StreamPlan p;
    static const bool on = [] {
        const char *e = getenv("OEA_TOPK_STREAM"), *b = getenv("OEA_TOPK_BF16");      // the streams hold the bf16 sweep's records
        return !(e && e[0] == '0') && !(b && b[0] == '0');
    }();
    // from 12,288 rows on (the 15K datasets: 15,000 rows, k = 1,499: 0.95 ms on random rows / 1.26 ms on the trained table against
    // 1.21 / 1.31 ms of the N x N strip path; with one bucketing workgroup per tile it lost there)
    static const int64_t min_n = [] { const char *e = getenv("OEA_TOPK_SYM_MIN"); return e ? (int64_t)atoll(e) : (int64_t)12288; }();
    if (!on || n < min_n) return p;
    const double e = (double)k * kSample / (double)n;
    p.r = threshold_rank(e);
    const double frac = (double)p.r / kSample;
    const double m_total = frac * (double)n;
    if (p.r >= kSample / 2 || m_total * 1.4 > kPerThread * SEL_THREADS || n > (int64_t)kBitWords * 32) return p;
    p.T = (int)oea::ceil_div(n, 128);
    if (p.T > kStreamMaxT) return p;
    p.groups = 16;
    p.L = std::max(8, (int)oea::ceil_div(p.T, p.groups));
    p.groups = (int)oea::ceil_div(p.T, p.L);
    // records per wave and side: L tiles of 64 x 64 pairs; the thresholds' common noise averages over the wave's 64 rows
    const double ew = frac * p.L * 4096.0;
    p.rcap = p.ccap = ((int)(ew * 1.25 + 8.0 * std::sqrt(ew) + 256.0) + 63) / 64 * 64;
    // tests: OEA_TOPK_STREAM_CAP shrinks the streams (every wave then overflows into the pool), OEA_TOPK_OVF_CHUNKS the pool (it
    // runs dry: rows go to the strip fallback) -- both read per call
    if (const char *ec = getenv("OEA_TOPK_STREAM_CAP")) p.rcap = p.ccap = std::max(64, atoi(ec) / 64 * 64);
    p.row_cap = std::min(kPerThread * SEL_THREADS,
                         ((int)(m_total * (1.0 + 4.0 / std::sqrt((double)p.r)) + 8.0 * std::sqrt(m_total) + 64.0) + 7) / 8 * 8);
    int64_t items = 0;
    for (int c = 0; c < p.groups; ++c) items += std::min(p.T, (c + 1) * p.L);
    p.n_items = (int)items;
    if ((size_t)items * 4 * (size_t)p.rcap >= ((size_t)1 << 32)) return p;            // record indices are 32-bit
    p.stride = n / kSample;
    p.ld = (n + 31) / 32 * 32;
    auto a256 = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += a256(bytes); return o; };
    p.off_thr = take(sizeof(float) * (size_t)n);
    p.off_fail = take(sizeof(int32_t) * (size_t)n);
    p.off_nfail = take(256);
    p.off_items = take(sizeof(int32_t) * 4 * (size_t)p.n_items);
    p.off_rcnt = take(sizeof(int32_t) * 4 * (size_t)p.n_items);
    p.off_coff = take(sizeof(int32_t) * 4 * (size_t)p.n_items * (p.L + 1));
    p.off_lcnt = take(sizeof(int32_t) * (size_t)p.T * 128);
    p.off_rowfail = take((size_t)p.T * 128);
    p.off_strip = take(sizeof(float) * (size_t)n * kSample);
    p.stream_bytes = 8 * (size_t)p.n_items * 4 * (size_t)p.rcap;
    p.off_rstream = take(p.stream_bytes);
    p.off_cstream = take(p.stream_bytes);
    p.off_lists = take(8 * (size_t)p.T * 128 * (size_t)p.row_cap);
    // overflow pool: 40 % of the expected records (both sides) + a chunk per wave
    p.ovf_chunks = (int)std::min<double>(0.4 * 2.0 * ew * 4.0 * (double)items / kOvfChunkRecs + 4.0 * (double)items, 4.0e6);
    if (getenv("OEA_TOPK_STREAM_CAP")) p.ovf_chunks = (int)std::min<double>(2.0 * 2.0 * ew * 4.0 * (double)items / kOvfChunkRecs + 8.0 * (double)items, 4.0e6);
    if (const char *eo = getenv("OEA_TOPK_OVF_CHUNKS")) p.ovf_chunks = std::max(1, atoi(eo));
    p.off_ovf = take(16 * (size_t)kOvfChunkRecs * (size_t)p.ovf_chunks);
    p.off_ovflen = take(sizeof(int32_t) * (size_t)p.ovf_chunks);
    p.redo_cap = (int)std::min<int64_t>(items * 4 * p.L, 1 << 22);     // (work item, tile, wave) triples: all of them, up to 4 M
    p.off_redo = take(32 * (size_t)p.redo_cap);
    p.total = off;
    p.ok = off <= ws_bytes;
    return p;
}

// long rows with k well inside the LDS candidate lists take the one-read kernel
static void launch_select(const float *s, int64_t n_rows, int64_t nc, int64_t ld, int k, const int32_t *id_map, int32_t *out,
                          hipStream_t st) {
    // rows that fit the LDS beside the select's tables (<= 128 KB of row: 32,768 columns) CAN be selected from one read
    // instead of three (OEA_TOPK_SELECT_CACHED=1).  Measured (round 3, gpurun_out r03ag3): 15,000^2, k = 1,499 1.39 -> 1.66 ms --
    // the re-reads come out of L2 and were not the bound; the histogram / candidate passes are LDS work either way and two
    // workgroups per CU (76 KB each) hide less of it than the eight of the three-read kernel.  Off by default.
    static const bool cached_ok = [] {
        const char *e = getenv("OEA_TOPK_SELECT_CACHED");
        if (!(e && e[0] == '1')) return false;
        return hipFuncSetAttribute(reinterpret_cast<const void *>(row_select_cached_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   128 * 1024) == hipSuccess;
    }();
    const size_t row_bytes = sizeof(float) * (size_t)((nc + 3) / 4 * 4);
    // rows of <= 16,384 columns: the row in registers, one read (OEA_TOPK_SELECT_REGS=0: the three-read kernel)
    static const bool regs_on = [] { const char *e = getenv("OEA_TOPK_SELECT_REGS"); return !(e && e[0] == '0'); }();
    const bool aligned16 = (((uintptr_t)s | (uintptr_t)(ld * 4)) & 15) == 0;
    if (nc >= 16384 && (int64_t)k * 5 <= (int64_t)kWaveCap * 4 * 3)        // expected candidates ~1.3 k <= 3/4 of the lists
        row_select_sampled_kernel<<<(unsigned)n_rows, SEL_THREADS, 0, st>>>(s, n_rows, nc, ld, k, id_map, out);
    else if (cached_ok && nc >= 2048 && row_bytes <= 128 * 1024 && (((uintptr_t)s | (uintptr_t)(ld * 4)) & 15) == 0)
        row_select_cached_kernel<<<(unsigned)n_rows, SEL_THREADS, row_bytes, st>>>(s, n_rows, nc, ld, k, id_map, out);
    else if (regs_on && nc <= 4096 && aligned16)
        row_select_regs_kernel<4><<<(unsigned)n_rows, SEL_THREADS, 0, st>>>(s, n_rows, nc, ld, k, id_map, out);
    else if (regs_on && nc <= 8192 && aligned16)
        row_select_regs_kernel<8><<<(unsigned)n_rows, SEL_THREADS, 0, st>>>(s, n_rows, nc, ld, k, id_map, out);
    else if (regs_on && nc <= 16384 && aligned16)
        row_select_regs_kernel<16><<<(unsigned)n_rows, SEL_THREADS, 0, st>>>(s, n_rows, nc, ld, k, id_map, out);
    else
        row_select_kernel<<<(unsigned)n_rows, SEL_THREADS, 0, st>>>(s, n_rows, nc, ld, k, id_map, out, select_stop());
}

// Work items of the symmetric sweep on an ABSOLUTE grid of candidate chunks: item (c, qt) = candidate tiles
// [max(qt, c L), min(T, (c + 1) L)) of query tile qt, for every chunk c that reaches past the diagonal (qt < (c + 1) L).
// Items are numbered chunk-major, so the workgroups in flight at one time walk the SAME candidate tiles in step and share
// them in their XCD's L2 (windows that start at each query tile's own diagonal put 64 different panels per XCD in flight:
// 21 % L2 hits against 68 % in the plain rank sweep).  Segment group of an item = c - qt / L.
__global__ void sym_items_kernel(int T, int L, int chunks, int4 *__restrict__ items) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= chunks * T) return;
    const int c = i / T, qt = i - c * T;
    if (qt >= min(T, (c + 1) * L)) return;
    int base = 0;
    for (int h = 0; h < c; ++h) base += min(T, (h + 1) * L);
    items[base + qt] = make_int4(qt, max(qt, c * L), min(T, (c + 1) * L), c - qt / L);
}

// ---- stream form of the symmetric search (bf16 sweep): the records of one target tile -> its 128 rows' compact lists -----------
// topk_stream_sym_kernel (sim_rank.hip) leaves, per wave of every work item, a stream of query-side records (targets: the 128
// rows of the item's query tile) and a stream of candidate-side records whose position at every candidate-tile boundary is
// known (col_off).  One workgroup per target tile tau collects its records: the whole query-side streams of the items (tau, c),
// c >= tau / L, and from every item (qt <= tau, c = tau / L) the slice of the candidate-side streams written while it was on
// candidate tile tau.  The slices (a few thousand, ~100 records each) are laid end to end through a prefix table in LDS, every
// thread walks the concatenation with a stride of the workgroup (its slice index only moves forward), and a record goes to slot
// atomicAdd(LDS counter of its target row) of that row's compact list -- (value, other index) pairs in no particular order: the
// select (histogram + bitmap) needs none.

__device__ __forceinline__ int sym_item_base(int T, int L, int c) {
    int b = 0;
    for (int h = 0; h < c; ++h) b += min(T, (h + 1) * L);
    return b;
}

constexpr int kBucketThreads = 512, kBucketFlight = 8;         // 4,096 records in flight per workgroup, four workgroups per CU: the pass is
                                                               // HBM latency; every target tile's workgroup resident at once up to 1,024 tiles

__global__ __launch_bounds__(kBucketThreads) void topk_bucket_kernel(int T, int L, int groups, int64_t n, const uint2 *__restrict__ row_streams, int rcap,
                                                          const int32_t *__restrict__ row_cnt, const uint2 *__restrict__ col_streams,
                                                          int ccap, const int32_t *__restrict__ col_off, int lp1,
                                                          uint2 *__restrict__ lists, int row_cap, int32_t *__restrict__ counts,
                                                          uint8_t *__restrict__ row_fail, int parts) {
    // `parts` workgroups share a target tile (each an equal share of the concatenated records): a row's slots then come from ONE
    // returning atomic on its (zeroed) global count per round and workgroup -- 128 per 4,096 records -- instead of a counter in LDS.
    // With one workgroup per tile a 15,000-row table has 118 workgroups for 256 CUs, a 100,000-row one 782 for 512 slots.
    extern __shared__ uint32_t seg_tab[];                    // prefix[nseg + 1] | src[nseg]
    __shared__ int hist[128], loff[129], gbase[128];
    __shared__ int s_part[kBucketThreads / 64];
    const int tau = blockIdx.x / parts, part = blockIdx.x - tau * parts, tid = threadIdx.x;
    const int c0 = tau / L;
    const int nrow = (groups - c0) * 4, ncol = (tau + 1) * 4, nseg = nrow + ncol;
    uint32_t *prefix = seg_tab, *src = seg_tab + (64 + 4 * T + 1);
    if (tid < 128) hist[tid] = 0;
    const int base_c0 = sym_item_base(T, L, c0);
    const int per = (nseg + kBucketThreads - 1) / kBucketThreads;
    int mine = 0;
    for (int u = 0; u < per; ++u) {
        const int sg = tid * per + u;
        if (sg >= nseg) break;
        uint32_t start;
        int len;
        if (sg < nrow) {                                     // query-side stream of item (tau, c0 + sg / 4), wave sg % 4: all of it
            const int c = c0 + (sg >> 2);
            const size_t wid = (size_t)(sym_item_base(T, L, c) + tau) * 4 + (sg & 3);
            start = (uint32_t)(wid * rcap);
            len = row_cnt[wid];
        } else {                                             // candidate-side stream of item (qt, c0), wave: the slice of tile tau
            const int qt = (sg - nrow) >> 2;
            const size_t wid = (size_t)(base_c0 + qt) * 4 + ((sg - nrow) & 3);
            const int ti = tau - max(qt, c0 * L);
            const int o0 = min(col_off[wid * lp1 + ti], ccap), o1 = min(col_off[wid * lp1 + ti + 1], ccap);
            start = (uint32_t)(wid * ccap) + (uint32_t)o0;
            len = o1 - o0;
        }
        src[sg] = start;
        prefix[sg + 1] = (uint32_t)len;
        mine += len;
    }
    int total, run;
    {   // exclusive scan over the workgroup's 16 waves
        const int lane = tid & 63, wv = tid >> 6;
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_part[wv] = incl;
        __syncthreads();
        int base = 0;
        total = 0;
        for (int w = 0; w < kBucketThreads / 64; ++w) {
            if (w < wv) base += s_part[w];
            total += s_part[w];
        }
        run = base + incl - mine;
    }
    for (int u = 0; u < per; ++u) {
        const int sg = tid * per + u;
        if (sg >= nseg) break;
        run += (int)prefix[sg + 1];
        prefix[sg + 1] = (uint32_t)run;
    }
    if (tid == 0) prefix[0] = 0u;
    __syncthreads();
    uint2 *__restrict__ out = lists + (size_t)tau * 128 * row_cap;
    // every wave takes a contiguous sixteenth of the concatenation, its lanes consecutive records (512 B per load instruction);
    // the slice index of a lane then moves forward by less than one slice per step
    const uint32_t per_part = ((uint32_t)total + parts - 1) / parts;
    const uint32_t p_begin = min((uint32_t)part * per_part, (uint32_t)total), p_end = min(p_begin + per_part, (uint32_t)total);
    const uint32_t per_wave = (p_end - p_begin + kBucketThreads / 64 - 1) / (kBucketThreads / 64);
    const uint32_t w_begin = min(p_begin + (uint32_t)(tid >> 6) * per_wave, p_end), w_end = min(w_begin + per_wave, p_end);
    int sg = 0;
    {
        int lo = 0, hi = nseg;                               // prefix[lo] <= w_begin < prefix[hi] (when the range is not empty)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (prefix[mid] <= w_begin) lo = mid; else hi = mid;
        }
        sg = lo;
    }
    // Rounds of kBucketFlight * kBucketThreads records: a returning LDS atomic ranks every record inside its target row's bucket
    // of the round, the records are laid out bucket by bucket in LDS and leave in that order -- a row's records of the round
    // (~30) go to consecutive slots of its list: runs of ~256 B instead of 8-byte stores to 128 different lines (which the L2
    // evicted half-written: 2.4 ms for this pass against 0.9 now at 100,000 rows).
    uint2 *sorted = reinterpret_cast<uint2 *>(seg_tab + 2 * (64 + 4 * T) + 2);
    const uint32_t n_rounds = (per_wave + kBucketFlight * 64 - 1) / (kBucketFlight * 64);
    for (uint32_t round = 0; round < n_rounds; ++round) {
        const uint32_t g0 = w_begin + (tid & 63) + round * (kBucketFlight * 64);
        uint2 rec[kBucketFlight];
        int rk[kBucketFlight];
#pragma unroll
        for (int u = 0; u < kBucketFlight; ++u) {
            const uint32_t g = g0 + u * 64;
            rec[u] = make_uint2(0u, 0xFFFFFFFFu);
            if (g < w_end) {
                while (prefix[sg + 1] <= g) ++sg;
                const uint32_t idx = src[sg] + (g - prefix[sg]);
                rec[u] = sg < nrow ? row_streams[idx] : col_streams[idx];
            }
        }
#pragma unroll
        for (int u = 0; u < kBucketFlight; ++u)
            if (g0 + u * 64 < w_end) rk[u] = atomicAdd(&hist[rec[u].y >> 24], 1);
        __syncthreads();
        if (tid < 64) {                                      // one wave: offsets of the 128 buckets inside the round, their list slots
            const int h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
            int incl = h0 + h1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (tid >= o) incl += t;
            }
            const int excl = incl - h0 - h1;
            loff[2 * tid] = excl;
            loff[2 * tid + 1] = excl + h0;
            if (tid == 63) loff[128] = incl;
            const int64_t row0 = (int64_t)tau * 128 + 2 * tid;
            gbase[2 * tid] = h0 ? atomicAdd(counts + row0, h0) : 0;
            gbase[2 * tid + 1] = h1 ? atomicAdd(counts + row0 + 1, h1) : 0;
            if (h0 && gbase[2 * tid] + h0 > row_cap) row_fail[row0] = 1;
            if (h1 && gbase[2 * tid + 1] + h1 > row_cap) row_fail[row0 + 1] = 1;
            hist[2 * tid] = 0;
            hist[2 * tid + 1] = 0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kBucketFlight; ++u)
            if (g0 + u * 64 < w_end) sorted[loff[rec[u].y >> 24] + rk[u]] = rec[u];
        __syncthreads();
        const int n_round = loff[128];
        for (int i = tid; i < n_round; i += kBucketThreads) {
            const uint2 r = sorted[i];
            const int t = (int)(r.y >> 24);
            const int slot = gbase[t] + (i - loff[t]);
            if (slot < row_cap) out[(size_t)t * row_cap + slot] = make_uint2(r.x, r.y & 0xFFFFFFu);
        }
        __syncthreads();
    }
}

// records that did not fit their wave's stream (16 B: value, target row, other index; chunks of kOvfChunk from the shared pool):
// appended to the compact lists behind the bucketed records, one returning atomic on the row's count each
__global__ __launch_bounds__(256) void topk_overflow_kernel(const uint4 *__restrict__ pool, const int32_t *__restrict__ alloc,
                                                            const int32_t *__restrict__ len, int cap_chunks, uint2 *__restrict__ lists,
                                                            int row_cap, int32_t *__restrict__ counts, uint8_t *__restrict__ row_fail) {
    const int n_chunks = min(*alloc, cap_chunks);
    const int lane = threadIdx.x & 63;
    for (int ch = blockIdx.x * 4 + (threadIdx.x >> 6); ch < n_chunks; ch += gridDim.x * 4) {
        const int l = min(len[ch], kOvfChunkRecs);
        for (int i = lane; i < l; i += 64) {
            const uint4 rec = pool[(size_t)ch * kOvfChunkRecs + i];
            const int slot = atomicAdd(counts + rec.y, 1);
            if (slot < row_cap) lists[(size_t)rec.y * row_cap + slot] = make_uint2(rec.x, rec.z);
            else row_fail[rec.y] = 1;
        }
    }
}

// ---- rows the list select gave up on: redone through the strip path, in batches -------------------------------------------
// (overflowed segments -- trained embeddings cluster: a row's neighbours crowd into a few candidate ranges -- fewer than k
// survivors, tie-heavy rows.)  The survivor lists are dead once list_select_kernel has run, so their storage holds the
// gathered packed query rows, the similarity strip of a batch and its selected columns.  One host read of the failure
// count (the only synchronisation of the search); the work is proportional to the failed rows: 0.4 us per row at 100,000
// candidates, against ~80 us per row of the per-row kernel this replaces.
__global__ void gather_rows_list_kernel(const float *__restrict__ qp, int kp, const int32_t *__restrict__ rows, int n, int n_pad,
                                        float *__restrict__ dst) {
    const int cpr = kp / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)n_pad * cpr; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i / cpr), c = (int)(i - (int64_t)f * cpr);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < n) v = oea::ld4(qp + (int64_t)rows[f] * kp + 4 * c);
        oea::st4(dst + (int64_t)f * kp + 4 * c, v);
    }
}
__global__ void scatter_selected_kernel(const int32_t *__restrict__ sel, int k, const int32_t *__restrict__ rows, int n,
                                        int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)n * k; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i / k);
        out[(int64_t)rows[f] * k + (i - (int64_t)f * k)] = sel[i];
    }
}

static int redo_failed_rows(const float *qp, int kp, const float *cp, int64_t nc, int dim, int k, const int32_t *id_map, int32_t *out,
                            const int32_t *fail_rows, const int32_t *n_fail_dev, void *region, size_t region_bytes, int64_t ld,
                            hipStream_t st) {
    int32_t nf = 0;
    OEA_CHECK_HIP(hipMemcpyAsync(&nf, n_fail_dev, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    OEA_CHECK_HIP(hipStreamSynchronize(st));
    static const bool dbg = getenv("OEA_TOPK_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[oea_topk_inner] rows redone through the strip path: %d\n", nf);
    if (nf <= 0) return OEA_OK;
    const size_t per_row = sizeof(float) * ((size_t)kp + (size_t)ld) + sizeof(int32_t) * (size_t)k;
    int64_t batch = (int64_t)(region_bytes / per_row) / 128 * 128;
    batch = std::min<int64_t>(batch, 16384);
    OEA_REQUIRE(batch >= 128, "list storage smaller than one 128-row fallback batch");
    char *w = static_cast<char *>(region);
    float *gq = reinterpret_cast<float *>(w);
    int32_t *sel = reinterpret_cast<int32_t *>(w + sizeof(float) * (size_t)batch * kp);
    float *strip = reinterpret_cast<float *>(w + sizeof(float) * (size_t)batch * kp + (sizeof(int32_t) * (size_t)batch * k + 255) / 256 * 256);
    // (the three areas fit: batch * per_row <= region_bytes, the 256-byte round-up is inside the slack of the /128*128)
    if (sizeof(float) * (size_t)batch * kp + (sizeof(int32_t) * (size_t)batch * k + 255) / 256 * 256 + sizeof(float) * (size_t)batch * ld > region_bytes)
        batch -= 128;
    OEA_REQUIRE(batch >= 128, "list storage smaller than one 128-row fallback batch");
    for (int64_t b0 = 0; b0 < nf; b0 += batch) {
        const int cnt = (int)std::min<int64_t>(batch, nf - b0);
        const int cnt_pad = (cnt + 127) / 128 * 128;
        gather_rows_list_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div((int64_t)cnt_pad * (kp / 4), 256), 4096), 256, 0, st>>>(
            qp, kp, fail_rows + b0, cnt, cnt_pad, gq);
        oea::sim_inner_store_packed(gq, cnt, cp, nc, kp, dim, strip, ld, st);
        launch_select(strip, cnt, nc, ld, k, id_map, sel, st);
        scatter_selected_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div((int64_t)cnt * k, 256), 8192), 256, 0, st>>>(
            sel, k, fail_rows + b0, cnt, out);
    }
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // namespace

namespace oea {
// shared with the one-sweep CSLS means (sim_rank.hip)
int kth_value(const float *strip, int64_t rows, int sample, int r, float *thr, hipStream_t st) {
    const unsigned grid = (unsigned)ceil_div(rows, 4);
    if (sample == 1024) kth_value_kernel<16><<<grid, 256, 0, st>>>(strip, rows, sample, r, thr);
    else if (sample == 2048) kth_value_kernel<32><<<grid, 256, 0, st>>>(strip, rows, sample, r, thr);
    else if (sample == 4096) kth_value_kernel<64><<<grid, 256, 0, st>>>(strip, rows, sample, r, thr);
    else return OEA_EINVAL;
    return OEA_OK;
}
void gather_packed_rows(const float *qp, int kp, const int32_t *rows, const int32_t *n_rows, float *dst, hipStream_t st) {
    gather_fail_rows_kernel<<<32, 256, 0, st>>>(qp, kp, rows, n_rows, dst);
}
}  // namespace oea

// ---- short candidate lists: the k best of a row of at most 1,024 values by RANKING (no sort) ---------------------------------------------
// One wave per row; every lane owns the columns lane, lane + 64, ...; rank of a value = how many values of the row beat it (better
// value, or the same value in an earlier column); rank < k selects.  The selected columns leave in ascending column order (a wave
// ballot prefix), optionally mapped through the row's id list; the k-th best value (rank k - 1) is written beside them.  Replaces
// torch.argsort / sort / topk on [rows, k + margin] matrices (approaches/rdgcn.py:get_neg, ops.l1_grid_topk_means).
template <typename T, bool LARGEST>
__global__ __launch_bounds__(256) void row_rank_select_kernel(const T *__restrict__ vals, int64_t n_rows, int nc, int64_t ld, int k,
                                                              const int32_t *__restrict__ ids, int64_t ld_ids,
                                                              int32_t *__restrict__ out_sel, T *__restrict__ out_kth) {
    constexpr int PER = 16;                                   // 64 x 16 = 1,024 columns
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= n_rows) return;
    __shared__ T sh[4][1024];
    T *mine = sh[threadIdx.x >> 6];
    const T *v = vals + row * ld;
    for (int c = lane; c < nc; c += 64) mine[c] = v[c];
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    int rank[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) rank[u] = 0;
    for (int j = 0; j < nc; ++j) {
        const T o = mine[j];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = lane + 64 * u;
            if (c < nc) {
                const T x = mine[c];
                const bool beats = LARGEST ? (o > x) : (o < x);
                rank[u] += (beats || (o == x && j < c)) ? 1 : 0;
            }
        }
    }
    int base = 0;                                             // selected columns before this group of 64
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int c = lane + 64 * u;
        if (64 * u >= nc) break;                              // (uniform)
        const bool sel = c < nc && rank[u] < k;
        const unsigned long long m = __ballot(sel);
        if (sel) {
            const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
            if (out_sel) out_sel[row * k + slot] = ids ? ids[row * ld_ids + c] : c;
            if (out_kth && rank[u] == k - 1) out_kth[row] = mine[c];
        }
        base += __popcll(m);
    }
}

template <typename T>
static int rank_select(const T *vals, int64_t n_rows, int32_t nc, int64_t ld, int32_t k, int32_t largest, const int32_t *ids, int64_t ld_ids,
                       int32_t *out_sel, T *out_kth, void *stream) {
    OEA_REQUIRE(vals && (out_sel || out_kth), "null pointer");
    OEA_REQUIRE(nc >= 1 && nc <= 1024 && k >= 1 && k <= nc && ld >= nc, "1 <= k <= nc <= 1024, ld >= nc");
    if (n_rows == 0) return OEA_OK;
    const unsigned nb = (unsigned)oea::ceil_div(n_rows, 4);
    hipStream_t st = oea::as_stream(stream);
    if (largest) row_rank_select_kernel<T, true><<<nb, 256, 0, st>>>(vals, n_rows, nc, ld, k, ids, ld_ids, out_sel, out_kth);
    else row_rank_select_kernel<T, false><<<nb, 256, 0, st>>>(vals, n_rows, nc, ld, k, ids, ld_ids, out_sel, out_kth);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

extern "C" {

size_t oea_topk_workspace_bytes(int64_t nq, int64_t nc) {
    const int64_t ld = (nc + 31) / 32 * 32;
    return (size_t)nq * (size_t)ld * sizeof(float);
}

size_t oea_topk_sym_workspace_bytes(int64_t n, int32_t k) {
    const StreamPlan q = plan_stream(n, k, ~(size_t)0);         // the form oea_topk_inner takes when it is covered
    if (q.ok) return q.total;
    const SymPlan p = plan_sym(n, k, ~(size_t)0);
    return p.ok ? p.total : 0;
}

int oea_row_rank_select_f32(const float *vals, int64_t n_rows, int32_t nc, int64_t ld, int32_t k, int32_t largest, const int32_t *ids,
                            int64_t ld_ids, int32_t *out_sel, float *out_kth, void *stream) {
    return rank_select<float>(vals, n_rows, nc, ld, k, largest, ids, ld_ids, out_sel, out_kth, stream);
}

int oea_row_rank_select_f64(const double *vals, int64_t n_rows, int32_t nc, int64_t ld, int32_t k, int32_t largest, const int32_t *ids,
                            int64_t ld_ids, int32_t *out_sel, double *out_kth, void *stream) {
    return rank_select<double>(vals, n_rows, nc, ld, k, largest, ids, ld_ids, out_sel, out_kth, stream);
}

int oea_topk_rows(const float *s, int64_t n_rows, int64_t nc, int64_t ld, int32_t k, const int32_t *id_map,
                  int32_t *out_idx, void *stream) {
    OEA_REQUIRE(s && out_idx, "null pointer");
    OEA_REQUIRE(k >= 1 && k <= nc && ld >= nc && ld % 4 == 0, "1 <= k <= nc <= ld, ld % 4 == 0 (rows are read 16 B at a time)");
    if (n_rows == 0) return OEA_OK;
    launch_select(s, n_rows, nc, ld, k, id_map, out_idx, oea::as_stream(stream));
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_topk_inner(const float *q, int64_t nq, int32_t ldq, const float *c, int64_t nc, int32_t ldc,
                   int32_t dim, int32_t k, const int32_t *id_map, int32_t *out_idx, void *workspace,
                   size_t ws_bytes, void *stream) {
    OEA_REQUIRE(q && c && out_idx && workspace, "null pointer");
    OEA_REQUIRE(k >= 1 && k <= nc, "1 <= k <= nc");
    OEA_REQUIRE(nc < (1 << 24) - 32, "nc < 2^24");
    OEA_REQUIRE(ldq % 4 == 0 && ldc % 4 == 0 && dim > 0 && dim <= ldq && dim <= ldc, "ld % 4 == 0, dim <= ld");
    if (nq == 0) return OEA_OK;
    const int64_t ld = (nc + 31) / 32 * 32;
    int64_t rows_per = (int64_t)(ws_bytes / (sizeof(float) * (size_t)ld));
    rows_per = rows_per / 128 * 128;
    OEA_REQUIRE(rows_per >= 128 || rows_per >= nq || (int64_t)(ws_bytes / (sizeof(float) * (size_t)ld)) >= nq,
                "workspace smaller than one 128-row strip");
    if (rows_per < 128) rows_per = nq;
    float *strip = static_cast<float *>(workspace);
    hipStream_t st = oea::as_stream(stream);
    const bool packed = oea::tile_glds_enabled();
    float *qp = nullptr, *cp = nullptr;
    int kp = 0;
    if (packed) {                                   // both operands packed once for all strips (sim_rank.hip)
        int rc = oea::pack_rows(0, q, nq, ldq, dim, st, &qp, &kp);
        if (rc == OEA_OK) rc = oea::pack_rows(1, c, nc, ldc, dim, st, &cp, &kp);
        if (rc != OEA_OK) return rc;
    }
    static const bool lists_on = [] { const char *e = getenv("OEA_TOPK_LISTS"); return !(e && e[0] == '0'); }();
    static const bool sym_on = [] { const char *e = getenv("OEA_TOPK_SYM"); return !(e && e[0] == '0'); }();
    const bool same = q == c && nq == nc && ldq == ldc;
    static const bool bf16_sweep = [] { const char *e = getenv("OEA_TOPK_BF16"); return !(e && e[0] == '0'); }();
    const StreamPlan sp = (same && packed && lists_on && sym_on && bf16_sweep && dim <= 2048) ? plan_stream(nc, k, ws_bytes) : StreamPlan();
    if (sp.ok) {                                    // stream form: wave-private record streams -> per-tile bucketing -> compact lists
        char *w = static_cast<char *>(workspace);
        float *thr = reinterpret_cast<float *>(w + sp.off_thr);
        int32_t *fail_rows = reinterpret_cast<int32_t *>(w + sp.off_fail);
        int32_t *n_fail = reinterpret_cast<int32_t *>(w + sp.off_nfail);
        float *tol_dev = reinterpret_cast<float *>(w + sp.off_nfail + 64);
        int32_t *items_dev = reinterpret_cast<int32_t *>(w + sp.off_items);
        int32_t *row_cnt = reinterpret_cast<int32_t *>(w + sp.off_rcnt);
        int32_t *col_off = reinterpret_cast<int32_t *>(w + sp.off_coff);
        int32_t *list_cnt = reinterpret_cast<int32_t *>(w + sp.off_lcnt);
        uint8_t *row_fail = reinterpret_cast<uint8_t *>(w + sp.off_rowfail);
        float *sstrip = reinterpret_cast<float *>(w + sp.off_strip);
        uint2 *lists = reinterpret_cast<uint2 *>(w + sp.off_lists);
        OEA_REQUIRE(kp <= 4096, "dim <= 4096 on the list path");
        sym_items_kernel<<<(unsigned)oea::ceil_div((int64_t)sp.groups * sp.T, 256), 256, 0, st>>>(sp.T, sp.L, sp.groups,
                                                                                              reinterpret_cast<int4 *>(items_dev));
        // the sample strip on the bf16 split as well (OEA_TOPK_BF16_STRIP=0: fp32): thresholds are estimates, see sample_strip_bf16_pack
        static const bool bf16_strip = [] { const char *e = getenv("OEA_TOPK_BF16_STRIP"); return !(e && e[0] == '0'); }();
        int rc = OEA_OK;
        if (bf16_strip) {
            const float *qs = nullptr, *ss = nullptr;
            int kps2 = 0;
            rc = oea::sample_strip_bf16_pack(c, nc, ldc, c, kSample, ldc * (int)sp.stride, dim, st, &qs, &ss, &kps2);
            if (rc != OEA_OK) return rc;
            oea::sample_strip_bf16_launch(qs, nq, ss, kSample, kps2, dim, sstrip, st);
        } else {
            float *smp = nullptr;
            int kps = 0;
            rc = oea::pack_rows(2, c, kSample, ldc * (int)sp.stride, dim, st, &smp, &kps);
            if (rc != OEA_OK) return rc;
            oea::sim_inner_store_packed(qp, nq, smp, kSample, kp, dim, sstrip, kSample, st);
        }
        kth_value_kernel<kSample / 64><<<(unsigned)oea::ceil_div(nq, 4), 256, 0, st>>>(sstrip, nq, kSample, sp.r, thr);
        int32_t *ovf_alloc = reinterpret_cast<int32_t *>(w + sp.off_nfail + 128);
        int32_t *ovf_len = reinterpret_cast<int32_t *>(w + sp.off_ovflen);
        OEA_CHECK_HIP(hipMemsetAsync(n_fail, 0, 256, st));           // failure count, tolerance block, overflow chunk count
        OEA_CHECK_HIP(hipMemsetAsync(row_fail, 0, (size_t)nq, st));
        rc = oea::topk_stream_sym_bf16(c, nc, ldc, dim, thr, items_dev, sp.n_items, w + sp.off_rstream, sp.rcap, w + sp.off_cstream, sp.ccap,
                                       row_cnt, col_off, sp.L + 1, row_fail, tol_dev, w + sp.off_ovf, ovf_alloc, ovf_len, sp.ovf_chunks,
                                       ovf_alloc + 1, w + sp.off_redo, sp.redo_cap, st, bf16_strip);
        if (rc != OEA_OK) return rc;
        const size_t bucket_lds = sizeof(uint32_t) * (2 * (64 + 4 * (size_t)sp.T) + 2) + 8 * (size_t)kBucketFlight * kBucketThreads;
        static const hipError_t bucket_attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&topk_bucket_kernel),
                                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 1024 * (4 + kBucketFlight * kBucketThreads / 1024) + 1024);
        OEA_CHECK_HIP(bucket_attr);
        // (rows past n of the last tile never get records; the counts array covers T * 128 rows)
        OEA_CHECK_HIP(hipMemsetAsync(list_cnt, 0, sizeof(int32_t) * (size_t)sp.T * 128, st));
        const int parts = std::max(1, std::min(16, (int)oea::ceil_div(1024, sp.T)));
        topk_bucket_kernel<<<(unsigned)(sp.T * parts), kBucketThreads, bucket_lds, st>>>(
            sp.T, sp.L, sp.groups, nq, reinterpret_cast<const uint2 *>(w + sp.off_rstream), sp.rcap, row_cnt,
            reinterpret_cast<const uint2 *>(w + sp.off_cstream), sp.ccap, col_off, sp.L + 1, lists, sp.row_cap, list_cnt, row_fail, parts);
        topk_overflow_kernel<<<1024, 256, 0, st>>>(reinterpret_cast<const uint4 *>(w + sp.off_ovf), ovf_alloc, ovf_len, sp.ovf_chunks, lists,
                                                   sp.row_cap, list_cnt, row_fail);
        static const bool dbg_ovf = getenv("OEA_TOPK_DEBUG") != nullptr;
        if (dbg_ovf) {
            int32_t na = 0;
            OEA_CHECK_HIP(hipMemcpyAsync(&na, ovf_alloc, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            OEA_CHECK_HIP(hipStreamSynchronize(st));
            int32_t nr = 0;
            OEA_CHECK_HIP(hipMemcpy(&nr, ovf_alloc + 1, sizeof(int32_t), hipMemcpyDeviceToHost));
            fprintf(stderr, "[oea_topk_inner] overflow chunks: %d of %d, redone wave tiles: %d of %d\n", na, sp.ovf_chunks, nr, sp.redo_cap);
        }
        list_select_kernel<<<(unsigned)nq, SEL_THREADS, select_lds_bytes(nc), st>>>(
            nullptr, nullptr, nullptr, thr, 0, 0, nc, k, id_map, out_idx, fail_rows, n_fail, nullptr, nullptr, 0, 0, nullptr, nullptr,
            select_stop(), c, ldc, dim, tol_dev, lists, list_cnt, sp.row_cap, row_fail);
        rc = redo_failed_rows(qp, kp, cp, nc, dim, k, id_map, out_idx, fail_rows, n_fail, w + sp.off_rstream, 2 * sp.stream_bytes, sp.ld, st);
        if (rc != OEA_OK) return rc;
        rc = oea::release_packed_rows(st);
        if (rc != OEA_OK) return rc;
        OEA_CHECK_HIP(hipGetLastError());
        return OEA_OK;
    }
    const SymPlan sy = (same && packed && lists_on && sym_on && dim <= 2048) ? plan_sym(nc, k, ws_bytes) : SymPlan();
    if (sy.ok) {                                    // queries == candidates: the upper triangle's tiles feed rows and columns
        char *w = static_cast<char *>(workspace);
        float *thr = reinterpret_cast<float *>(w + sy.off_thr);
        int32_t *counts = reinterpret_cast<int32_t *>(w + sy.off_counts);
        uint8_t *ccounts = reinterpret_cast<uint8_t *>(w + sy.off_ccounts);
        int32_t *fail_rows = reinterpret_cast<int32_t *>(w + sy.off_fail);
        int32_t *n_fail = reinterpret_cast<int32_t *>(w + sy.off_nfail);
        int32_t *items_dev = reinterpret_cast<int32_t *>(w + sy.off_items);
        float *list_vals = reinterpret_cast<float *>(w + sy.off_vals);
        int32_t *list_cols = reinterpret_cast<int32_t *>(w + sy.off_cols);
        void *clists = w + sy.off_clists;
        int32_t *spill_cnt = reinterpret_cast<int32_t *>(w + sy.off_spcnt);
        void *spill = w + sy.off_spill;
        float *sstrip = reinterpret_cast<float *>(w + sy.off_strip);
        OEA_REQUIRE(kp <= 4096, "dim <= 4096 on the list path");
        // work items, chunk-major: (query tile, first candidate tile, one past the last, segment group)
        sym_items_kernel<<<(unsigned)oea::ceil_div((int64_t)sy.groups * sy.T, 256), 256, 0, st>>>(sy.T, sy.L, sy.groups,
                                                                                              reinterpret_cast<int4 *>(items_dev));
        float *sp = nullptr;
        int kps = 0;
        int rc = oea::pack_rows(2, c, kSample, ldc * (int)sy.stride, dim, st, &sp, &kps);
        if (rc != OEA_OK) return rc;
        oea::sim_inner_store_packed(qp, nq, sp, kSample, kp, dim, sstrip, kSample, st);
        kth_value_kernel<kSample / 64><<<(unsigned)oea::ceil_div(nq, 4), 256, 0, st>>>(sstrip, nq, kSample, sy.r, thr);
        OEA_CHECK_HIP(hipMemsetAsync(n_fail, 0, sizeof(int32_t), st));
        // segments no work item writes (the lower triangle's) must read as empty
        OEA_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)nq * sy.nseg, st));
        OEA_CHECK_HIP(hipMemsetAsync(ccounts, 0, (size_t)nq * sy.T * 2, st));
        OEA_CHECK_HIP(hipMemsetAsync(spill_cnt, 0, sizeof(int32_t) * (size_t)nq, st));
        // the sweep on the bf16 hi / lo split of the rows (3/16 of the fp32 matrix time; OEA_TOPK_BF16=0: the exact fp32 sweep):
        // approximate list values, the select decides the neighbourhood of the k-th value with exact chains -- same result
        static const bool bf16_on = [] { const char *e = getenv("OEA_TOPK_BF16"); return !(e && e[0] == '0'); }();
        float *tol_dev = reinterpret_cast<float *>(w + sy.off_nfail + 64);
        if (bf16_on) {
            rc = oea::topk_append_sym_bf16(c, nc, ldc, dim, thr, items_dev, sy.n_items, sy.nseg, sy.cap, list_vals, list_cols, counts, sy.T,
                                           sy.ccap, clists, ccounts, spill_cnt, spill, kSpillCap, tol_dev, st);
            if (rc != OEA_OK) return rc;
        } else {
            oea::topk_append_sym_packed(qp, nq, kp, dim, thr, items_dev, sy.n_items, sy.nseg, sy.cap, list_vals, list_cols, counts, sy.T,
                                        sy.ccap, clists, ccounts, spill_cnt, spill, kSpillCap, st);
        }
        list_select_kernel<<<(unsigned)nq, SEL_THREADS, select_lds_bytes(nc), st>>>(
            list_vals, list_cols, counts, thr, sy.nseg, sy.cap, nc, k, id_map, out_idx, fail_rows, n_fail,
            static_cast<const uint2 *>(clists), ccounts, 2 * sy.T, sy.ccap, spill_cnt, static_cast<const uint2 *>(spill), select_stop(),
            c, ldc, dim, bf16_on ? tol_dev : nullptr, nullptr, nullptr, 0, nullptr);
        rc = redo_failed_rows(qp, kp, cp, nc, dim, k, id_map, out_idx, fail_rows, n_fail, clists,
                              8 * (size_t)nq * sy.T * 2 * sy.ccap, sy.ld, st);
        if (rc != OEA_OK) return rc;
        rc = oea::release_packed_rows(st);
        if (rc != OEA_OK) return rc;
        OEA_CHECK_HIP(hipGetLastError());
        return OEA_OK;
    }
    const ListPlan lp = (packed && lists_on && dim <= 2048) ? plan_lists(nq, nc, k, ws_bytes) : ListPlan();
    if (lp.ok) {                                    // strip-free path (see above)
        char *w = static_cast<char *>(workspace);
        float *thr = reinterpret_cast<float *>(w + lp.off_thr);
        int32_t *counts = reinterpret_cast<int32_t *>(w + lp.off_counts);
        int32_t *fail_rows = reinterpret_cast<int32_t *>(w + lp.off_fail);
        int32_t *n_fail = reinterpret_cast<int32_t *>(w + lp.off_nfail);
        float *list_vals = reinterpret_cast<float *>(w + lp.off_lists);
        int32_t *list_cols = reinterpret_cast<int32_t *>(w + lp.off_lists + lp.cols_off);
        float *sstrip = reinterpret_cast<float *>(w + lp.off_strip);
        int32_t *spill_cnt = reinterpret_cast<int32_t *>(w + lp.off_spcnt);
        void *spill = w + lp.off_spill;
        OEA_REQUIRE(kp <= 4096, "dim <= 4096 on the list path");
        // the sweep on the bf16 hi / lo split of both tables (3 / 16 of the fp32 matrix time; OEA_TOPK_BF16=0: the exact fp32 sweep):
        // approximate list values, the select decides the neighbourhood of the k-th value with exact chains -- same sets; the sample
        // strip of the thresholds on the split as well (estimates)
        static const bool bf16_strip = [] { const char *e = getenv("OEA_TOPK_BF16_STRIP"); return !(e && e[0] == '0'); }();
        const bool strip16 = bf16_sweep && bf16_strip;
        float *tol_dev = reinterpret_cast<float *>(w + lp.off_nfail + 64);
        const float *qs = nullptr, *cs = nullptr, *ss = nullptr;
        int kps2 = 0, rc = OEA_OK;
        float *sp = nullptr;
        int kps = 0;
        if (strip16) rc = oea::sample_strip_bf16_pack(q, nq, ldq, c, kSample, ldc * (int)lp.stride, dim, st, &qs, &ss, &kps2);
        else rc = oea::pack_rows(2, c, kSample, ldc * (int)lp.stride, dim, st, &sp, &kps);   // every stride-th candidate row
        if (rc != OEA_OK) return rc;
        if (bf16_sweep) {
            rc = oea::topk_append_bf16_prepare(q, nq, ldq, c, nc, ldc, dim, tol_dev, st, &qs, &cs, &kps2, strip16);
            if (rc != OEA_OK) return rc;
        }
        for (int64_t r0 = 0; r0 < nq; r0 += lp.rows_per) {
            const int64_t rows = std::min<int64_t>(lp.rows_per, nq - r0);
            if (strip16) oea::sample_strip_bf16_launch(qs + r0 * kps2, rows, ss, kSample, kps2, dim, sstrip, st);
            else oea::sim_inner_store_packed(qp + r0 * kp, rows, sp, kSample, kp, dim, sstrip, kSample, st);
            kth_value_kernel<kSample / 64><<<(unsigned)oea::ceil_div(rows, 4), 256, 0, st>>>(sstrip, rows, kSample, lp.r, thr);
            OEA_CHECK_HIP(hipMemsetAsync(n_fail, 0, sizeof(int32_t), st));
            // the chunk count (hence the segment layout) is the one planned for a full pass: a short last pass reuses it
            OEA_CHECK_HIP(hipMemsetAsync(spill_cnt, 0, sizeof(int32_t) * (size_t)rows, st));
            if (bf16_sweep)
                oea::topk_append_bf16_launch(qs + r0 * kps2, rows, cs, nc, kps2, dim, thr, lp.cap, lp.chunks, list_vals, list_cols, counts,
                                             spill_cnt, spill, kSpillCap, tol_dev, st);
            else
                oea::topk_append_packed(qp + r0 * kp, rows, cp, nc, kp, dim, thr, lp.cap, lp.chunks, list_vals, list_cols, counts, spill_cnt,
                                        spill, kSpillCap, st);
            list_select_kernel<<<(unsigned)rows, SEL_THREADS, select_lds_bytes(nc), st>>>(
                list_vals, list_cols, counts, thr, lp.nseg, lp.cap, nc,
                                                                      k, id_map, out_idx + r0 * (int64_t)k, fail_rows, n_fail, nullptr, nullptr, 0, 0,
                                                                      spill_cnt, static_cast<const uint2 *>(spill), select_stop(),
                                                                      bf16_sweep ? c : nullptr, ldc, dim, bf16_sweep ? tol_dev : nullptr,
                                                                      nullptr, nullptr, 0, nullptr, bf16_sweep ? q + r0 * (int64_t)ldq : nullptr, ldq);
            // rows the select gave up on: through the strip path, in batches, inside the (now dead) list storage
            rc = redo_failed_rows(qp + r0 * kp, kp, cp, nc, dim, k, id_map, out_idx + r0 * (int64_t)k, fail_rows, n_fail, list_vals,
                                  2 * lp.cols_off, lp.ld, st);
            if (rc != OEA_OK) return rc;
        }
        rc = oea::release_packed_rows(st);
        if (rc != OEA_OK) return rc;
        OEA_CHECK_HIP(hipGetLastError());
        return OEA_OK;
    }
    for (int64_t r0 = 0; r0 < nq; r0 += rows_per) {              // r0 is a multiple of 128: strips start on tile rows
        const int64_t rows = std::min<int64_t>(rows_per, nq - r0);
        if (packed) {
            oea::sim_inner_store_packed(qp + r0 * kp, rows, cp, nc, kp, dim, strip, ld, st);
        } else {
            int rc = oea_sim_matrix(q + r0 * ldq, rows, ldq, c, nc, ldc, dim, OEA_METRIC_INNER, strip, ld, stream);
            if (rc != OEA_OK) return rc;
        }
        launch_select(strip, rows, nc, ld, k, id_map, out_idx + r0 * (int64_t)k, st);
    }
    if (packed) {
        const int rc = oea::release_packed_rows(st);
        if (rc != OEA_OK) return rc;
    }
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
