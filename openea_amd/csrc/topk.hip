// topk.hip -- k-nearest neighbour search for truncated negative sampling.
//
// Replaces find_neighbours (modules/train/batch.py:157-165):
//     sim_mat = np.matmul(sub_embed, embed.T); np.argpartition(-sim_mat[i], k)[:k]
// k is large here (int((1-eps)*N): 1,499 of 15,000; 2,000 of 100,000), so instead of keeping k
// candidates per row on chip the search is a per-row radix SELECT on the fp32 keys:
//   1. a strip of rows of S is produced by the MFMA tile kernel (sim_rank.hip) into an HBM
//      workspace (the strip is L2 / Infinity-Cache resident when read back),
//   2. one workgroup per row finds the k-th largest key with three histogram passes
//      (11 + 11 + 10 bits, histograms in LDS),
//   3. a fourth pass compacts the selected columns in ascending column order:
//      key > T, plus the first `need` columns with key == T  -> (value desc, column asc)
//      selection, bit-identical with oracle_topk_inner.
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

__global__ __launch_bounds__(SEL_THREADS) void row_select_kernel(const float *__restrict__ s, int64_t n_rows, int64_t nc,
                                                                 int64_t ld, int k, const int32_t *__restrict__ id_map,
                                                                 int32_t *__restrict__ out /* [n_rows, k] */) {
    __shared__ int hist[2048];
    __shared__ int s_wave[4];
    __shared__ uint32_t s_prefix;
    __shared__ int s_need;
    const int64_t row = blockIdx.x;
    const float *src = s + row * ld;
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
    int32_t *o = out + row * (int64_t)k;
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

}  // namespace

extern "C" {

size_t oea_topk_workspace_bytes(int64_t nq, int64_t nc) {
    const int64_t ld = (nc + 31) / 32 * 32;
    return (size_t)nq * (size_t)ld * sizeof(float);
}

int oea_topk_rows(const float *s, int64_t n_rows, int64_t nc, int64_t ld, int32_t k, const int32_t *id_map,
                  int32_t *out_idx, void *stream) {
    OEA_REQUIRE(s && out_idx, "null pointer");
    OEA_REQUIRE(k >= 1 && k <= nc && ld >= nc, "1 <= k <= nc <= ld");
    if (n_rows == 0) return OEA_OK;
    row_select_kernel<<<(unsigned)n_rows, SEL_THREADS, 0, oea::as_stream(stream)>>>(s, n_rows, nc, ld, k, id_map, out_idx);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_topk_inner(const float *q, int64_t nq, int32_t ldq, const float *c, int64_t nc, int32_t ldc,
                   int32_t dim, int32_t k, const int32_t *id_map, int32_t *out_idx, void *workspace,
                   size_t ws_bytes, void *stream) {
    OEA_REQUIRE(q && c && out_idx && workspace, "null pointer");
    OEA_REQUIRE(k >= 1 && k <= nc, "1 <= k <= nc");
    OEA_REQUIRE(ldq % 4 == 0 && ldc % 4 == 0 && dim > 0 && dim <= ldq && dim <= ldc, "ld % 4 == 0, dim <= ld");
    if (nq == 0) return OEA_OK;
    const int64_t ld = (nc + 31) / 32 * 32;
    int64_t rows_per = (int64_t)(ws_bytes / (sizeof(float) * (size_t)ld));
    rows_per = rows_per / 128 * 128;
    OEA_REQUIRE(rows_per >= 128 || rows_per >= nq || (int64_t)(ws_bytes / (sizeof(float) * (size_t)ld)) >= nq,
                "workspace smaller than one 128-row strip");
    if (rows_per < 128) rows_per = nq;
    float *strip = static_cast<float *>(workspace);
    for (int64_t r0 = 0; r0 < nq; r0 += rows_per) {
        const int64_t rows = std::min<int64_t>(rows_per, nq - r0);
        int rc = oea_sim_matrix(q + r0 * ldq, rows, ldq, c, nc, ldc, dim, OEA_METRIC_INNER, strip, ld, stream);
        if (rc != OEA_OK) return rc;
        row_select_kernel<<<(unsigned)rows, SEL_THREADS, 0, oea::as_stream(stream)>>>(strip, rows, nc, ld, k, id_map,
                                                                                    out_idx + r0 * (int64_t)k);
    }
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
