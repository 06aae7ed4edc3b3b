// common.h -- shared helpers for libopenea_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/openea_hip.h"

namespace oea {

void set_error(const char *fmt, ...);

// optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg)
bool prof_enabled();
void prof_call();                  // start of a profiled call: is it one of the sampled ones?
void prof_mark(hipStream_t st);   // records the next event of the current profile session (sampled calls only)
// sampled call: the next TWO events of the session, to be attached to one kernel dispatch (hipExtLaunchKernelGGL start /
// stop events: the dispatch's own begin / end timestamps -- what rocprofv3's kernel trace reports -- instead of two
// hipEventRecord barrier packets around it, which add ~5 us of packet processing to a 17 us kernel)
bool prof_pair(hipEvent_t *start, hipEvent_t *stop);

#define OEA_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            oea::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return OEA_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

#define OEA_REQUIRE(cond, msg)                                              \
    do {                                                                    \
        if (!(cond)) {                                                      \
            oea::set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, msg); \
            return OEA_EINVAL;                                              \
        }                                                                   \
    } while (0)

// packed-operand path of the similarity tiles (sim_rank.hip), for the kNN strips of topk.hip
bool tile_glds_enabled();
int pack_rows(int slot, const float *src, int64_t n, int ld, int dim, hipStream_t st, float **packed, int *kp);
int release_packed_rows(hipStream_t st);
void sim_inner_store_packed(const float *e1p, int64_t n1, const float *e2p, int64_t n2, int kp, int dim, float *out,
                            int64_t ld_out, hipStream_t st);
void sim_inner_store_packed_gated(const float *e1p, int64_t n1, const float *e2p, int64_t n2, int kp, int dim, float *out,
                                  int64_t ld_out, const int32_t *gate, hipStream_t st);
int topk_append_chunks(int64_t nq, int64_t nc);
int kth_value(const float *strip, int64_t rows, int sample, int r, float *thr, hipStream_t st);          // topk.hip
void gather_packed_rows(const float *qp, int kp, const int32_t *rows, const int32_t *n_rows, float *dst, hipStream_t st);   // <= 128 rows
void topk_append_packed(const float *qp, int64_t nq, const float *cp, int64_t nc, int kp, int dim, const float *thr, int cap,
                        int chunks, float *list_vals, int32_t *list_cols, int32_t *counts, int32_t *spill_cnt, void *spill, int sp_cap,
                        hipStream_t st);

void topk_append_sym_packed(const float *ep, int64_t n, int kp, int dim, const float *thr, const void *items, int n_items, int nseg,
                            int cap, float *list_vals, int32_t *list_cols, int32_t *counts, int T, int ccap, void *clists,
                            uint8_t *ccounts, int32_t *spill_cnt, void *spill, int sp_cap, hipStream_t st);

int topk_append_bf16_prepare(const float *q, int64_t nq, int ldq, const float *c, int64_t nc, int ldc, int dim, float *tol_dev,
                             hipStream_t st, const float **qs, const float **cs, int *kp, bool q_packed);
int sample_strip_bf16_pack(const float *q, int64_t nq, int ldq, const float *c, int64_t n_sample, int ld_sample, int dim, hipStream_t st,
                           const float **qs, const float **ss, int *kp);
void sample_strip_bf16_launch(const float *qs, int64_t rows, const float *ss, int64_t n_sample, int kp, int dim, float *strip, hipStream_t st);
void topk_append_bf16_launch(const float *qs, int64_t nq, const float *cs, int64_t nc, int kp, int dim, const float *thr, int cap,
                             int chunks, float *list_vals, int32_t *list_cols, int32_t *counts, int32_t *spill_cnt, void *spill,
                             int sp_cap, const float *tol_dev, hipStream_t st);
int topk_append_sym_bf16(const float *src, int64_t n, int ld, int dim, const float *thr, const void *items, int n_items, int nseg,
                         int cap, float *list_vals, int32_t *list_cols, int32_t *counts, int T, int ccap, void *clists,
                         uint8_t *ccounts, int32_t *spill_cnt, void *spill, int sp_cap, float *tol_dev, hipStream_t st);

int topk_stream_sym_bf16(const float *src, int64_t n, int ld, int dim, const float *thr, const void *items, int n_items, void *row_streams,
                         int rcap, void *col_streams, int ccap, int32_t *row_cnt, int32_t *col_off, int lp1, uint8_t *row_fail,
                         float *tol_dev, void *ovf_pool, int32_t *ovf_alloc, int32_t *ovf_len, int ovf_chunks, int32_t *redo_cnt,
                         void *redo, int redo_cap, hipStream_t st, bool packed);
int comm_phase_mark(struct ::oea_comm *c, hipStream_t st);      // comm.hip: phase boundary of the one-call partitioned epoch

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// kernel launch that carries the profile session's events when this call is a sampled one
template <class K, class... A>
static inline void launch_timed(K kernel, dim3 grid, dim3 block, hipStream_t st, A... args) {
    hipEvent_t e0, e1;
    if (prof_pair(&e0, &e1)) hipExtLaunchKernelGGL(kernel, grid, block, 0, st, e0, e1, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, 0, st, args...);
}

// the same with the events given by the caller (a phase of SEVERAL launches timed as one: start event on the first, stop on the last)
template <class K, class... A>
static inline void launch_events(K kernel, dim3 grid, dim3 block, hipStream_t st, hipEvent_t e0, hipEvent_t e1, A... args) {
    if (e0 || e1) hipExtLaunchKernelGGL(kernel, grid, block, 0, st, e0, e1, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, 0, st, args...);
}

// ---- wave64 helpers ---------------------------------------------------------------------
// Sum over the G-lane group (G power of two <= 64) containing this lane; every lane of the
// group gets the result.  Butterfly with __shfl_xor: fixed order -> deterministic.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// fp32 butterfly on the VALU cross-lane paths instead of ds_bpermute (each LDS-crossbar hop costs
// ~100 cycles of dependent latency; the fused step does ~25 reductions per wave): DPP quad_perm /
// row_half_mirror / row_mirror inside a row of 16, v_permlane16_swap / v_permlane32_swap (gfx950)
// across rows.  Same pairing as the xor butterfly, so the sums are bit-identical with it.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "group width");
    if (G >= 2) v += dpp_f32<0xB1>(v);        // quad_perm [1,0,3,2]
    if (G >= 4) v += dpp_f32<0x4E>(v);        // quad_perm [2,3,0,1]
    if (G >= 8) v += dpp_f32<0x141>(v);       // row_half_mirror
    if (G >= 16) v += dpp_f32<0x140>(v);      // row_mirror
    if (G >= 32) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);    // [even rows | odd rows] of the pair
        v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    if (G >= 64) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    return v;
}
// Sum over the wave's 64 lanes as a SCALAR (wave-uniform) value: the four butterfly steps inside the rows of 16, then the gfx9 row
// broadcasts -- row_bcast15 (every row adds lane 15 of the row on its left), row_bcast31 (rows 2 and 3 add lane 31) -- leave the
// total in lane 63, read with v_readlane: 6 DPP adds instead of 4 + 2 x (copy, permlane swap, add), and the result can steer scalar
// branches.  Fixed order -> deterministic (not the order of group_sum<64>).
__device__ __forceinline__ float wave_sum_uniform(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    v += dpp_f32<0x142>(v);        // row_bcast15, all rows: row i += row i-1 (old values); only row 3 = r2 + r3 and row 1 = r0 + r1 matter
    v += dpp_f32<0x143>(v);        // row_bcast31: rows 2, 3 += lane 31 = r0 + r1
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
template <int G>
__device__ __forceinline__ double group_sum_d(double v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) { return group_sum_d<64>(v); }

// hardware fp32 atomic add (global_atomic_add_f32, no CAS loop); device scope.
__device__ __forceinline__ void atomic_add_f32(float *p, float v) { unsafeAtomicAdd(p, v); }

// ---- gradient scratch of the translational step --------------------------------------------------------------------
// Two builds of the library from the same sources (csrc/Makefile):
//   libopenea_hip.so      grad_t = float: hardware fp32 atomics.  The sum of a row's contributions depends on the order the
//                         atomics arrive in, so two runs of one job differ in the last bits.
//   libopenea_hip_det.so  (-DOEA_DET_SCRATCH) grad_t = int64 FIXED POINT with 32 fractional bits, 64-bit integer atomics
//                         (global_atomic_add_x2).  Integer addition is associative: the sum is the same whatever the order --
//                         run to run, and whether one GPU or G ranks computed the contributions (the partition exchanges
//                         the int64 sums).  Every contribution is rounded ONCE to the 2^-32 grid (|v| < 2^19; larger values
//                         saturate), the row sum is exact and rounded once to fp32 when the optimiser reads it: closer to
//                         the fp64 oracle than any fp32 summation order.  TF sums the duplicate rows of a gather's gradient
//                         before the optimiser (optimizers.py:4-7): this is that sum, without an order.
#ifdef OEA_DET_SCRATCH
typedef long long grad_t;
typedef long long flag_t;
constexpr int kDetScratch = 1;
// float -> round(v * 2^32) as int64: v + 1.5 * 2^20 in fp64 has ulp 2^-32 and keeps its exponent for |v| < 2^19, so the
// difference of the two bit patterns IS the fixed-point value (one cvt, one fp64 add, one 32-bit subtract)
__device__ __forceinline__ long long to_fixed(float v) {
    const float c = __builtin_amdgcn_fmed3f(v, -524287.f, 524287.f);
    const double d = (double)c + 1572864.0;                       // 1.5 * 2^20 = 0x4138000000000000
    return __double_as_longlong(d) - 0x4138000000000000LL;
}
__device__ __forceinline__ float grad_val(long long q) { return (float)((double)q * 2.3283064365386963e-10); }   // 2^-32
__device__ __forceinline__ void grad_add(long long *p, float v) {
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(p), (unsigned long long)to_fixed(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
// a value already in the scratch's own type (another rank's partial sum): exact integer add
__device__ __forceinline__ void grad_add_raw(long long *p, long long q) {
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(p), (unsigned long long)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#else
typedef float grad_t;
typedef float flag_t;
constexpr int kDetScratch = 0;
__device__ __forceinline__ float grad_val(float q) { return q; }
__device__ __forceinline__ void grad_add(float *p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void grad_add_raw(float *p, float q) { unsafeAtomicAdd(p, q); }
#endif

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// ---- Philox4x32-10 (bit-identical with oracle/c/oracle.c) ------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return make_uint4(c0, c1, c2, c3);
}

// ---- triple membership set (layout shared with oracle/c/oracle.c) ---------------------------
#define OEA_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
__host__ __device__ __forceinline__ uint64_t pack_triple(uint32_t h, uint32_t r, uint32_t t) {
    return ((uint64_t)h << 40) | ((uint64_t)(r & 0xFFFFu) << 24) | (uint64_t)(t & 0xFFFFFFu);
}
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

}  // namespace oea
