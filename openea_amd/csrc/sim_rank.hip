// sim_rank.hip -- similarity tiles on the fp32 matrix cores + fused alignment-rank epilogue.
//
// Replaces sim() / csls_sim() / calculate_rank() of the reference's greedy_alignment
// (modules/finding/similarity.py:11-83, modules/finding/alignment.py:13-84,146-168):
//   np.matmul(e1, e2.T)             -> 128x128 tiles of v_mfma_f32_32x32x2_f32 (exact fp32:
//                                      a k-ordered fmaf chain, k = 0,1,2,... -- identical to
//                                      oracle/c/oracle.c:dot_chain, hence bit-exact ranks)
//   argsort(-row) + np.where(==gold) -> rank_i = #{j : S_ij > S_ii} (+ stable tie rule),
//                                      counted in the epilogue; the N1 x N2 matrix is never
//                                      written (19.6 GB at 70,000^2 in the reference)
//   scipy cdist('cityblock')         -> fp64 VALU tiles, sequential k (bit-exact vs scipy)
//
// Operand layout: both operands are row-major [n, ld] fp32 (K contiguous).  A K-chunk of 32
// is staged global -> registers -> LDS with the k index permuted inside groups of 8
// (positions 0..3 hold k = 0,2,4,6; 4..7 hold k = 1,3,5,7) so that one ds_read_b128 gives a
// lane its operand for four consecutive MFMA k-steps in natural k order.  Row stride 36
// floats (144 B) makes the b128 reads bank-conflict free.
#include "common.h"
#include <stdlib.h>
#include <cmath>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TILE = 128;      // block tile edge (both operands)
constexpr int BK = 32;         // K chunk
constexpr int LDS_LD = BK + 4; // padded row stride (floats)

// ---- XCD-aware order of the (query tile, candidate chunk) work items (round 5) --------------------------------------------------
// Block b of a launch runs on XCD b % 8 (observed dispatch rule, MI355X_MICROARCH.md: "for speed only"), and every XCD has its own
// 4 MB L2.  In launch order (query tile fastest) the 64 workgroups resident on one XCD hold 64 DIFFERENT query tiles: from K = 300
// on their operand panels (128 rows x Kp x 4 B each, re-read once per candidate tile) no longer fit the L2 and every chunk comes
// from the Infinity Cache -- 185 GB per 70,000^2 x 1,200 sweep.  Here XCD c takes the contiguous share [c * per, (c + 1) * per) of
// the items in CHUNK-FASTEST order: its resident workgroups are ~64 / ny query tiles x all ny candidate chunks, i.e. 64 / ny + ny
// operand streams instead of 65, each k chunk fetched from the fabric once and hit in L2 by the other workgroups that walk the
// same panel in step.  per = 0: the plain order (OEA_XCD_MAP=0, ablation).  Correctness never depends on the placement.
struct TileGrid { unsigned nx, ny, per; };

__device__ __forceinline__ bool tile_grid_item(const TileGrid &g, unsigned &bx, unsigned &by) {
    const unsigned L = blockIdx.x;
    if (g.per == 0u) { bx = L % g.nx; by = L / g.nx; return L < g.nx * g.ny; }
    const unsigned slot = L >> 3, w = (L & 7u) * g.per + slot;
    if (slot >= g.per || w >= g.nx * g.ny) return false;
    bx = w / g.ny; by = w % g.ny;
    return true;
}

__device__ __forceinline__ uint32_t f2ord(float f) {   // order-preserving float -> uint
    uint32_t u = __float_as_uint(f + 0.0f);   // -0 -> +0 so that key order == float order
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// A K-chunk of one 128-row operand tile in flight between HBM/L2 and LDS: 4 float4 per thread.
struct Frag { float4 v[4]; };

// rows [row0, row0+128) x k [k0, k0+32) of `src` -> registers.  Rows past the matrix are clamped to
// its last row (their products are discarded by the epilogues), so that interior chunks are four
// plain 16-byte loads from a tile base + 32-bit offsets; only the chunk that crosses `dim` masks.
__device__ __forceinline__ void load_frag(const float *__restrict__ src, int64_t n, int ld, int dim, int64_t row0,
                                          int k0, int tid, Frag &f) {
    const float *base = src + row0 * ld;                                  // wave-uniform
    const int last = (int)(n - 1 - row0 < TILE - 1 ? n - 1 - row0 : TILE - 1);
    const int c = k0 + (tid & 7) * 4;
    if (k0 + BK <= dim) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = min((tid >> 3) + pass * 32, last);
            f.v[pass] = oea::ld4(base + r * ld + c);
        }
    } else {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = min((tid >> 3) + pass * 32, last);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < ld) {
                v = oea::ld4(base + r * ld + c);
                if (c + 0 >= dim) v.x = 0.f;
                if (c + 1 >= dim) v.y = 0.f;
                if (c + 2 >= dim) v.z = 0.f;
                if (c + 3 >= dim) v.w = 0.f;
            }
            f.v[pass] = v;
        }
    }
}

// registers -> LDS with the k permutation described above
__device__ __forceinline__ void store_frag(const Frag &f, float *__restrict__ dst, int tid) {
    const int q = tid & 7, g = q >> 1, half = q & 1;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        float *p = dst + ((tid >> 3) + pass * 32) * LDS_LD + g * 8;
        *reinterpret_cast<float2 *>(p + 2 * half) = make_float2(f.v[pass].x, f.v[pass].z);
        *reinterpret_cast<float2 *>(p + 4 + 2 * half) = make_float2(f.v[pass].y, f.v[pass].w);
    }
}

// acc += A-chunk (M rows) x B-chunk (N rows) over `groups` groups of 8 k; per wave a 64x64 sub-tile
__device__ __forceinline__ void mma_chunk(const float *__restrict__ As, const float *__restrict__ Bs, int groups,
                                          f32x16 (&acc)[2][2]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const float *ap = As + (wm * 64 + (lane & 31)) * LDS_LD + 4 * (lane >> 5);
    const float *bp = Bs + (wn * 64 + (lane & 31)) * LDS_LD + 4 * (lane >> 5);
    for (int g = 0; g < groups; ++g) {
        const float4 a0 = *reinterpret_cast<const float4 *>(ap + g * 8);
        const float4 a1 = *reinterpret_cast<const float4 *>(ap + 32 * LDS_LD + g * 8);
        const float4 b0 = *reinterpret_cast<const float4 *>(bp + g * 8);
        const float4 b1 = *reinterpret_cast<const float4 *>(bp + 32 * LDS_LD + g * 8);
        const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
        const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][s], bv[0][s], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][s], bv[1][s], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][s], bv[0][s], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][s], bv[1][s], acc[1][1], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
}

// Software pipeline over a sequence of `n_tiles` M-tiles against one fixed N-tile: chunk i+1 travels
// HBM/L2 -> registers while chunk i is on the matrix cores, and lands in the other LDS buffer
// (one barrier per chunk).  `epilogue(t, acc)` runs after the last chunk of M-tile t with the
// first chunk of tile t+1 already in flight.  LDS: As/Bs = 2 buffers of TILE*LDS_LD floats each.
template <class MTile, class Epilogue>
__device__ __forceinline__ void tile_pipeline(const float *__restrict__ am, int64_t m_rows, int lda,
                                              const float *__restrict__ bn, int64_t n_rows, int ldb, int dim,
                                              int64_t n0, int64_t n_tiles, MTile m_tile, float *As, float *Bs,
                                              Epilogue epilogue) {
    const int tid = threadIdx.x;
    const int kend = (dim + 7) / 8 * 8;
    const int nchunk = (kend + BK - 1) / BK;
    const int64_t total = n_tiles * nchunk;
    if (total == 0) return;
    constexpr int BUF = TILE * LDS_LD;
    Frag fa, fb;
    load_frag(am, m_rows, lda, dim, m_tile(0), 0, tid, fa);
    load_frag(bn, n_rows, ldb, dim, n0, 0, tid, fb);
    store_frag(fa, As, tid);
    store_frag(fb, Bs, tid);
    __syncthreads();
    if (total > 1) {
        const int64_t t1 = 1 / nchunk;
        const int k1 = (1 % nchunk) * BK;
        load_frag(am, m_rows, lda, dim, m_tile(t1), k1, tid, fa);
        load_frag(bn, n_rows, ldb, dim, n0, k1, tid, fb);
    }
    f32x16 acc[2][2];
    zero_acc(acc);
    int64_t t = 0;
    int kc = 0;
    for (int64_t it = 0; it < total; ++it) {
        const int cur = (int)(it & 1);
        mma_chunk(As + cur * BUF, Bs + cur * BUF, min(BK, kend - kc * BK) / 8, acc);
        if (it + 1 < total) {
            store_frag(fa, As + (cur ^ 1) * BUF, tid);
            store_frag(fb, Bs + (cur ^ 1) * BUF, tid);
        }
        __syncthreads();
        if (it + 2 < total) {
            int kc2 = kc + 2;
            int64_t t2 = t;
            while (kc2 >= nchunk) { kc2 -= nchunk; ++t2; }
            load_frag(am, m_rows, lda, dim, m_tile(t2), kc2 * BK, tid, fa);
            load_frag(bn, n_rows, ldb, dim, n0, kc2 * BK, tid, fb);
        }
        if (++kc == nchunk) {
            epilogue(t, acc);
            zero_acc(acc);
            kc = 0;
            ++t;
        }
    }
}

// ---- packed operands + LDS-DMA staging --------------------------------------------------------------------------
// The same pipeline fed by `global_load_lds_dwordx4` (gfx950): a chunk goes HBM/L2 -> LDS without passing through
// registers and without ds_write instructions.  The LDS-DMA destination of one wave instruction is lane-linear
// (base + 16 B * lane), so the k permutation of the reg-staged path cannot be applied on the way: the operands are
// packed once per call by pack_rows_kernel ([n_pad, Kp] with n_pad % 128 == 0 and Kp % 32 == 0, zero filled, every
// group of 8 k stored as k = 0,2,4,6,1,3,5,7 -- O(n d) work in front of an O(n^2 d) sweep).  The LDS image of a chunk
// is 128 rows x 32 floats WITHOUT padding; bank conflicts are avoided by an XOR swizzle of the eight 16-byte columns
// of a row with ((row >> 1) & 7), applied to the per-lane SOURCE address of the DMA and to the ds_read address (the same
// involution on both sides).  A 128-byte row covers half of the 64 banks, so 16 consecutive rows of one column must
// land on all 16 (row parity, position) combinations: with (row & 7) rows r and r + 8 collided -- SQ_LDS_BANK_CONFLICT
// was half of SQ_LDS_IDX_ACTIVE -- with ((row >> 1) & 7) they do not.
constexpr int PLD = BK;                 // unpadded LDS row stride (floats)

__global__ void pack_rows_kernel(const float *__restrict__ src, int64_t n, int ld, int dim, float *__restrict__ dst,
                                 int64_t n_pad, int kp) {
    const int cpr = kp / 4;                                            // 16-byte columns per packed row
    const int64_t total = n_pad * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const int c = (int)(i - row * cpr);
        const int k = 8 * (c >> 1) + (c & 1);                          // k, k+2, k+4, k+6
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < n) {
            const float *r = src + row * ld;
            if (k < dim) v.x = r[k];
            if (k + 2 < dim) v.y = r[k + 2];
            if (k + 4 < dim) v.z = r[k + 4];
            if (k + 6 < dim) v.w = r[k + 6];
        }
        oea::st4(dst + row * kp + 4 * c, v);
    }
}

// rows [row0, row0 + 128) x packed k [k0, k0 + 32) -> LDS buffer `dst` (128 x 32 floats), 4 DMA instructions per wave
__device__ __forceinline__ void stage_packed(const float *__restrict__ packed, int kp, int64_t row0, int k0,
                                             float *__restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 3;                                         // row inside the instruction's 8-row block
    const float *src = packed + (row0 + wave * 32 + sub) * kp + k0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float *base = dst + (wave * 4 + j) * 8 * PLD;                  // wave-uniform: 1 KB per instruction
        const int swz = (4 * j + (sub >> 1)) & 7;                      // ((row >> 1) & 7) of row = 32 wave + 8 j + sub
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (int64_t)j * 8 * kp + 4 * ((lane & 7) ^ swz)),
                                         reinterpret_cast<__attribute__((address_space(3))) void *>(reinterpret_cast<uintptr_t>(base)),
                                         16, 0, 0);
    }
}

__device__ __forceinline__ void mma_chunk_packed(const float *__restrict__ As, const float *__restrict__ Bs, int groups,
                                                 f32x16 (&acc)[2][2]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int x = (lane >> 1) & 7, half = lane >> 5;                   // ((row >> 1) & 7): tile offsets are multiples of 16
    const float *ap = As + (wm * 64 + (lane & 31)) * PLD;
    const float *bp = Bs + (wn * 64 + (lane & 31)) * PLD;
    for (int g = 0; g < groups; ++g) {
        const int off = 4 * ((2 * g + half) ^ x);
        const float4 a0 = *reinterpret_cast<const float4 *>(ap + off);
        const float4 a1 = *reinterpret_cast<const float4 *>(ap + 32 * PLD + off);
        const float4 b0 = *reinterpret_cast<const float4 *>(bp + off);
        const float4 b1 = *reinterpret_cast<const float4 *>(bp + 32 * PLD + off);
        const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
        const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][s], bv[0][s], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][s], bv[1][s], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][s], bv[0][s], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][s], bv[1][s], acc[1][1], 0, 0, 0);
        }
    }
}

// tile_pipeline on packed operands (am / bn: packed arrays, lda = ldb = Kp; rows are padded, so no clamping): the DMA of
// chunk i+1 is issued into the other LDS buffer before chunk i goes to the matrix cores; the barrier at the end of the
// iteration (with the vmcnt(0) the compiler puts in front of it) makes it visible and frees the buffer just read.
template <class MTile, class Epilogue>
__device__ __forceinline__ void tile_pipeline_packed(const float *__restrict__ am, int kp, const float *__restrict__ bn,
                                                     int dim, int64_t n0, int64_t n_tiles, MTile m_tile, float *As,
                                                     float *Bs, Epilogue epilogue) {
    const int kend = (dim + 7) / 8 * 8;
    const int nchunk = (kend + BK - 1) / BK;
    const int64_t total = n_tiles * nchunk;
    if (total == 0) return;
    constexpr int BUF = TILE * LDS_LD;                                // the buffers keep the reg-staged path's size
    stage_packed(am, kp, m_tile(0), 0, As);
    stage_packed(bn, kp, n0, 0, Bs);
    __syncthreads();
    f32x16 acc[2][2];
    zero_acc(acc);
    int64_t t = 0;
    int kc = 0;
    for (int64_t it = 0; it < total; ++it) {
        const int cur = (int)(it & 1);
        if (it + 1 < total) {
            int kc1 = kc + 1;
            int64_t t1 = t;
            if (kc1 == nchunk) { kc1 = 0; ++t1; }
            stage_packed(am, kp, m_tile(t1), kc1 * BK, As + (cur ^ 1) * BUF);
            stage_packed(bn, kp, n0, kc1 * BK, Bs + (cur ^ 1) * BUF);
        }
        mma_chunk_packed(As + cur * BUF, Bs + cur * BUF, min(BK, kend - kc * BK) / 8, acc);
        __syncthreads();
        if (++kc == nchunk) {
            epilogue(t, acc);
            zero_acc(acc);
            kc = 0;
            ++t;
        }
    }
}

// ---- bf16 hi / lo split operands (certified prefilter: see the section before CslsPlan) ----------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ uint32_t bf16_rne(float x) {                   // round to nearest even (finite inputs)
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// packed row = kp "float slots" (kp = dim rounded up to 32): every 32-k chunk is 128 B = 8 granules of 8 bf16; granule
// (s * 2 + h) * 2 + p holds k = 32 chunk + 16 s + 8 h + [0, 8) of the hi (p = 0) or lo (p = 1) part -- one MFMA operand of lane
// half h in k-step s.  Same bytes per row as the fp32 packed layout, so stage_packed moves it unchanged.
__global__ void pack_rows_bf16_kernel(const float *__restrict__ src, int64_t n, int ld, int dim, uint4 *__restrict__ dst,
                                      int64_t n_pad, int kp) {
    const int cpr = kp / 4;
    const int64_t total = n_pad * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const int c = (int)(i - row * cpr);
        const int g = c & 7, k0 = 32 * (c >> 3) + 16 * (g >> 2) + 8 * ((g >> 1) & 1), lo = g & 1;
        uint32_t h[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float x = (row < n && k0 + t < dim) ? src[row * ld + k0 + t] : 0.f;
            const uint32_t hi = bf16_rne(x);
            h[t] = lo ? bf16_rne(x - __uint_as_float(hi << 16)) : hi;
        }
        dst[i] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    }
}

// max row norm of a table as the bits of a non-negative float (integer order == float order)
__global__ void row_norm_max_kernel(const float *__restrict__ src, int64_t n, int ld, int dim, unsigned *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float nr = 0.f;
    if (i < n) {
        float ss = 0.f;
        for (int k = 0; k < dim; ++k) ss = fmaf(src[i * ld + k], src[i * ld + k], ss);
        nr = sqrtf(ss) * 1.0000005f;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nr = fmaxf(nr, __shfl_xor(nr, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(nr));
}

// acc += hi.hi + hi.lo + lo.hi of one chunk (ksteps = 1 or 2 k-steps of 16)
// EARLY (the 256 x 128 kernels): all eight fragment reads of a k-step are issued in front of its MFMAs (a scheduling fence keeps
// them there) in the order the products consume them, so the waits are counted lgkmcnt(6 / 4 / 2 / 0) behind running MFMAs; left
// to itself hipcc re-uses the hi fragments' registers for the lo ones and waits for the LDS with lgkmcnt(0) three times per k-step
template <bool EARLY = false>
__device__ __forceinline__ void mma_chunk_bf16(const float *__restrict__ As, const float *__restrict__ Bs, int ksteps,
                                               f32x16 (&acc)[2][2]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int x = (lane >> 1) & 7, half = lane >> 5;
    const float *ap = As + (wm * 64 + (lane & 31)) * PLD;
    const float *bp = Bs + (wn * 64 + (lane & 31)) * PLD;
    for (int s = 0; s < ksteps; ++s) {
        const int g = 4 * s + 2 * half;
        const int oh = 4 * (g ^ x), ol = 4 * ((g + 1) ^ x);
        const bf16x8 a0h = *reinterpret_cast<const bf16x8 *>(ap + oh), b0h = *reinterpret_cast<const bf16x8 *>(bp + oh);
        const bf16x8 b1h = *reinterpret_cast<const bf16x8 *>(bp + 32 * PLD + oh), a1h = *reinterpret_cast<const bf16x8 *>(ap + 32 * PLD + oh);
        const bf16x8 b0l = *reinterpret_cast<const bf16x8 *>(bp + ol), b1l = *reinterpret_cast<const bf16x8 *>(bp + 32 * PLD + ol);
        const bf16x8 a0l = *reinterpret_cast<const bf16x8 *>(ap + ol), a1l = *reinterpret_cast<const bf16x8 *>(ap + 32 * PLD + ol);
        if constexpr (EARLY) __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b0h, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b1h, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b0h, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b1h, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b0l, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b1l, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b0l, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b1l, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, b0h, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, b1h, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, b0h, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, b1h, acc[1][1], 0, 0, 0);
    }
}

// tile_pipeline_packed for the bf16 layout: chunks of two k-steps (32 k = 128 B per row), LDS-DMA staging unchanged.
// K BLOCKS (round 5; KBLK: the evaluation / CSLS sweeps -- the neighbour sweeps' epilogues have no 64 registers to spare and run at
// K = 100): the MFMA accumulators restart from zero every kBf16BlockChunks chunks (128 k) and
// the block results are added in block order on the VALU (64 v_add per 96 MFMAs).  The accumulation term of the certificate
// (bf16_eps_rel) then counts the products of ONE block: the error of n additions in any order is <= n u (sum of |terms|), and the
// blocks' sums of |terms| add up to <= |q||c| (Cauchy-Schwarz per block, then over the blocks) -- 3 * 128 + (blocks - 1)
// roundings instead of 3 * Kp: at K = 1,200 the bound drops from 5.9e-4 to 1.3e-4 and with it the records per row.
constexpr int kBf16BlockChunks = 4;

__device__ __forceinline__ void add_acc(f32x16 (&tot)[2][2], const f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[a][b][r] += acc[a][b][r];
}

template <bool KBLK, class MTile, class Epilogue>
__device__ __forceinline__ void tile_pipeline_bf16(const float *__restrict__ am, int kp, const float *__restrict__ bn, int dim,
                                                   int64_t n0, int64_t n_tiles, MTile m_tile, float *As, float *Bs,
                                                   Epilogue epilogue) {
    const int S = (dim + 15) / 16;
    const int nchunk = (S + 1) / 2;
    const int64_t total = n_tiles * nchunk;
    if (total == 0) return;
    constexpr int BUF = TILE * LDS_LD;
    stage_packed(am, kp, m_tile(0), 0, As);
    stage_packed(bn, kp, n0, 0, Bs);
    __syncthreads();
    f32x16 acc[2][2], tot[2][2];
    zero_acc(acc);
    zero_acc(tot);
    int64_t t = 0;
    int kc = 0;
    for (int64_t it = 0; it < total; ++it) {
        const int cur = (int)(it & 1);
        if (it + 1 < total) {
            int kc1 = kc + 1;
            int64_t t1 = t;
            if (kc1 == nchunk) { kc1 = 0; ++t1; }
            stage_packed(am, kp, m_tile(t1), kc1 * BK, As + (cur ^ 1) * BUF);
            stage_packed(bn, kp, n0, kc1 * BK, Bs + (cur ^ 1) * BUF);
        }
        mma_chunk_bf16(As + cur * BUF, Bs + cur * BUF, min(2, S - 2 * kc), acc);
        ++kc;
        if constexpr (KBLK) {
            if ((kc & (kBf16BlockChunks - 1)) == 0 || kc == nchunk) {   // end of a 128-k block: block sums in block order (one
                add_acc(tot, acc);                                      // block when K <= 128: tot = 0 + acc, the same values)
                zero_acc(acc);
            }
        }
        // the epilogue works on registers only: it runs BEFORE the barrier (and the vmcnt(0) in front of it), so the DMA of the
        // next tile's first chunk lands behind it instead of being waited for first (the bf16 chunks are too short to hide it)
        if (kc == nchunk) {
            if constexpr (KBLK) { epilogue(t, tot); zero_acc(tot); }
            else { epilogue(t, acc); zero_acc(acc); }
            kc = 0;
            ++t;
        }
        __syncthreads();
    }
}

// ---- 256 x 128 tiles, three LDS stages (round 5: K > 128, i.e. AliNet's 1,200-d / RDGCN's 300-d evaluation rows) ---------------
// tile_pipeline_bf16 keeps ONE 32 KB chunk per workgroup in flight (64 KB per CU): with ~1.9 us between the issue of a chunk's
// LDS-DMA and its arrival under load that is 34 GB/s per CU -- the 70,000^2 x 1,200 sweep sat at 34 % of the bf16 pipe whatever
// the epilogue did.  Here 512 threads (8 waves as 4 (M) x 2 (N), each still a 64 x 64 sub-tile: every epilogue is unchanged) own
// 256 candidate rows x 128 query rows -- 3/4 of the operand bytes per flop -- and a ring of THREE 48 KB stages keeps TWO chunks in
// flight (96 KB per CU): a chunk is waited for with a COUNTED s_waitcnt vmcnt(6) (6 DMA instructions per wave and stage: the
// newer stage stays in flight) in front of a RAW s_barrier -- __syncthreads() would drain the DMA queue (vmcnt(0)) on every chunk.
// K blocks as in tile_pipeline_bf16<true>.  LDS: 3 x (256 + 128) x 128 B = 144 KB (dynamic), one workgroup per CU.
constexpr int BIG_MT = 256;                       // candidate rows of a tile
constexpr int BIG_A = BIG_MT * PLD;               // floats of an A stage
constexpr int BIG_STAGE = BIG_A + TILE * PLD;     // + the B stage: 48 KB
constexpr int BIG_STAGES = 3;
constexpr int BIG_LDS_BYTES = BIG_STAGES * BIG_STAGE * 4;

// per-lane part of the DMA source addresses of a stage (elements): the swizzled 16-byte column of the lane's row for the even /
// odd 8-row pieces; everything else of an address is wave-uniform (tile row, k chunk, piece) and stays on the scalar unit
struct BigLane {
    unsigned a_even, a_odd, b_even, b_odd;
    float *a_dst, *b_dst;                 // LDS offsets of the wave's pieces inside a stage
};

__device__ __forceinline__ BigLane big_lane(int kp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;          // 8 waves: 32 A rows + 16 B rows each
    const int sub = lane >> 3;
    BigLane l;
    const unsigned c_even = 4u * ((lane & 7) ^ ((sub >> 1) & 7)), c_odd = 4u * ((lane & 7) ^ ((4 + (sub >> 1)) & 7));   // ((row >> 1) & 7), row = 8 j + sub (+ 16 or 32 wave)
    l.a_even = (unsigned)(wave * 32 + sub) * (unsigned)kp + c_even;
    l.a_odd = (unsigned)(wave * 32 + sub) * (unsigned)kp + c_odd;
    l.b_even = (unsigned)(wave * 16 + sub) * (unsigned)kp + c_even;
    l.b_odd = (unsigned)(wave * 16 + sub) * (unsigned)kp + c_odd;
    l.a_dst = nullptr; l.b_dst = nullptr;
    return l;
}

__device__ __forceinline__ void stage_packed_big(const float *__restrict__ a_tile /* am + row0 * kp + k0: wave-uniform */,
                                                 const float *__restrict__ b_tile, int kp, const BigLane &l, float *__restrict__ slot) {
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float *base = slot + (wave * 4 + j) * 8 * PLD;
        const float *src = a_tile + (size_t)((j & 1 ? l.a_odd : l.a_even) + (unsigned)(j * 8 * kp));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         reinterpret_cast<__attribute__((address_space(3))) void *>(reinterpret_cast<uintptr_t>(base)),
                                         16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float *base = slot + BIG_A + (wave * 2 + j) * 8 * PLD;
        const float *src = b_tile + (size_t)((j & 1 ? l.b_odd : l.b_even) + (unsigned)(j * 8 * kp));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         reinterpret_cast<__attribute__((address_space(3))) void *>(reinterpret_cast<uintptr_t>(base)),
                                         16, 0, 0);
    }
}

// the eight fragments of one k-step of a wave's 64 x 64 sub-tile
struct BfFrag { bf16x8 a0h, a0l, a1h, a1l, b0h, b0l, b1h, b1l; };

__device__ __forceinline__ void load_bf_frag(const float *__restrict__ ap, const float *__restrict__ bp, int s, int half, int x, BfFrag &f) {
    const int g = 4 * s + 2 * half;
    const int oh = 4 * (g ^ x), ol = 4 * ((g + 1) ^ x);
    f.a0h = *reinterpret_cast<const bf16x8 *>(ap + oh); f.b0h = *reinterpret_cast<const bf16x8 *>(bp + oh);
    f.b1h = *reinterpret_cast<const bf16x8 *>(bp + 32 * PLD + oh); f.a1h = *reinterpret_cast<const bf16x8 *>(ap + 32 * PLD + oh);
    f.b0l = *reinterpret_cast<const bf16x8 *>(bp + ol); f.b1l = *reinterpret_cast<const bf16x8 *>(bp + 32 * PLD + ol);
    f.a0l = *reinterpret_cast<const bf16x8 *>(ap + ol); f.a1l = *reinterpret_cast<const bf16x8 *>(ap + 32 * PLD + ol);
}

__device__ __forceinline__ void mma_bf_frag(const BfFrag &f, f32x16 (&acc)[2][2]) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a0h, f.b0h, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a0h, f.b1h, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a1h, f.b0h, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a1h, f.b1h, acc[1][1], 0, 0, 0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a0h, f.b0l, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a0h, f.b1l, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a1h, f.b0l, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a1h, f.b1l, acc[1][1], 0, 0, 0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a0l, f.b0h, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a0l, f.b1h, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a1l, f.b0h, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a1l, f.b1h, acc[1][1], 0, 0, 0);
}

// PF (OEA_BF16_BIG_MODE=1, the default; 0 = !PF): the fragment reads never wait in front of idle matrix cores.  A chunk is two k-steps; the loop
// body is  [reads of k-step 1 of chunk c] [MFMAs of k-step 0] [chunk c + 1 landed? counted vmcnt, raw barrier] [DMA of chunk c + 3 into
// the slot chunk c just left] [reads of k-step 0 of chunk c + 1] [MFMAs of k-step 1 of chunk c]: every read is issued one MFMA group
// (12 x 32 cycles) before its use, the barrier sits between the two groups, and the DMA runs three chunks ahead.  The second k-step of
// a tile's last chunk may be zero padding of the packed rows (Kp is a multiple of 32): it adds +0.  !PF = the first form of this
// pipeline (reads in front of each group, barrier at the top), kept for the ablation.
template <bool PF, class MTile, class Epilogue>
__device__ __forceinline__ void tile_pipeline_bf16_big(const float *__restrict__ am, int kp, const float *__restrict__ bn, int dim,
                                                       int64_t n0, int64_t n_tiles, MTile m_tile, float *lds, Epilogue epilogue) {
    const int S = (dim + 15) / 16;
    const int nchunk = (S + 1) / 2;
    const int total = (int)n_tiles * nchunk;          // (a work item walks at most a few thousand chunks)
    if (total <= 0) return;
    const BigLane lane_off = big_lane(kp);
    const float *b_base = bn + n0 * kp;
    int ti = 0, ki = 0, si = 0;                       // (tile, chunk, LDS stage) of the next stage to issue
    const float *a_base = am + m_tile(0) * kp;
    auto issue = [&]() {
        stage_packed_big(a_base + ki * BK, b_base + ki * BK, kp, lane_off, lds + si * BIG_STAGE);
        si = si == BIG_STAGES - 1 ? 0 : si + 1;
        if (++ki == nchunk) { ki = 0; ++ti; a_base = am + m_tile(ti) * kp; }
    };
    f32x16 acc[2][2], tot[2][2];
    zero_acc(acc);
    zero_acc(tot);
    int t = 0, kc = 0, sc = 0;                        // (tile, chunk, stage) being multiplied
    if constexpr (!PF) {
        issue();
        if (total > 1) issue();
        for (int it = 0; it < total; ++it) {
            // stage `it` has landed when at most the newer stage's 6 DMA instructions of this wave are outstanding (loads complete in
            // order; whatever the epilogue issued since only makes the count stricter); the barrier then covers the other waves' parts
            // and tells everybody that the slot read in iteration it - 1 is free
            if (it + 1 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (it + 2 < total) issue();
            const float *slot = lds + sc * BIG_STAGE;
            sc = sc == BIG_STAGES - 1 ? 0 : sc + 1;
            mma_chunk_bf16<true>(slot, slot + BIG_A, min(2, S - 2 * kc), acc);
            ++kc;
            if ((kc & (kBf16BlockChunks - 1)) == 0 || kc == nchunk) {
                add_acc(tot, acc);
                zero_acc(acc);
            }
            if (kc == nchunk) {
                epilogue((int64_t)t, tot);
                zero_acc(tot);
                kc = 0;
                ++t;
            }
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int x = (lane >> 1) & 7, half = lane >> 5;
    const int a_off = (wm * 64 + (lane & 31)) * PLD, b_off = BIG_A + (wn * 64 + (lane & 31)) * PLD;
    // prologue: three chunks on their way, chunk 0 landed, its first fragments requested
    issue();
    if (total > 1) issue();
    if (total > 2) issue();
    if (total > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (total > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    BfFrag f0, f1;
    load_bf_frag(lds + a_off, lds + b_off, 0, half, x, f0);
    for (int it = 0; it < total; ++it) {
        const float *slot = lds + sc * BIG_STAGE;
        load_bf_frag(slot + a_off, slot + b_off, 1, half, x, f1);            // k-step 1 of this chunk, behind the MFMAs of k-step 0
        __builtin_amdgcn_sched_barrier(0);
        mma_bf_frag(f0, acc);
        __builtin_amdgcn_sched_barrier(0);
        sc = sc == BIG_STAGES - 1 ? 0 : sc + 1;
        if (it + 1 < total) {
            // chunk it + 1 must have landed: of this wave's DMA only chunk it + 2's (6 instructions) may still be out; f1 has arrived
            // too (lgkmcnt(0)): after the barrier nobody reads chunk it's slot any more
            if (it + 2 < total) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (it + 3 < total) issue();                                     // into the slot of chunk it
            const float *nxt = lds + sc * BIG_STAGE;
            load_bf_frag(nxt + a_off, nxt + b_off, 0, half, x, f0);          // k-step 0 of the next chunk, behind the MFMAs of k-step 1
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_bf_frag(f1, acc);
        __builtin_amdgcn_sched_barrier(0);
        ++kc;
        if ((kc & (kBf16BlockChunks - 1)) == 0 || kc == nchunk) {
            add_acc(tot, acc);
            zero_acc(acc);
        }
        if (kc == nchunk) {
            epilogue((int64_t)t, tot);
            zero_acc(tot);
            kc = 0;
            ++t;
        }
    }
}

// ---- producer / consumer waves (round 5, OEA_BF16_BIG_MODE=2; not the default: measured equal to mode 1) -------------------------
// Hypothesis: what the 256 x 128 pipeline above spends its time on is the ISSUE of its LDS-DMA: a `global_load_lds_dwordx4` costs the
// issuing wave ~60-180 cycles when it sits between MFMAs and fragment reads (MI355X_MICROARCH.md), 6 per wave and chunk = a third
// of an iteration on every SIMD; a wave that does nothing else issues one in ~25 cycles.  So of the 8 waves of the workgroup (two per
// SIMD, 256 VGPRs each: more waves would halve the register budget of the multiplying ones) waves 6 and 7 become LOADERS -- each
// issues half of a stage's 40 instructions, waits for the stage the consumers need next (counted vmcnt) and meets them at the
// barrier -- and waves 0-5 are CONSUMERS as 3 (M) x 2 (N) on 192 candidate rows x 128 query rows: fragment reads one MFMA group
// ahead, MFMAs, K-block sums, epilogue, and nothing else.  LDS: 3 stages x (192 + 128) x 128 B = 120 KB.
// Measured at 70,000^2 x 1,200: 35.3 ms against 34.7 ms for mode 1 -- six multiplying waves each run 1.3x faster than eight did, which
// cancels; identical results.  (hipcc still answers the loop-carried fragment reads with lgkmcnt(0) in front of the first MFMA
// group and counted waits on the NEXT chunk's reads in the second: the prefetch of both modes is only half effective.)
constexpr int SPEC_MT = 192, SPEC_NC = 6;
constexpr int SPEC_A = SPEC_MT * PLD;
constexpr int SPEC_STAGE = SPEC_A + TILE * PLD;            // 40 KB
constexpr int SPEC_BLOCKS = (SPEC_MT + TILE) / 8;          // 8-row pieces of a stage: 24 of A, 16 of B

template <class MTile, class Epilogue>
__device__ __forceinline__ void tile_pipeline_bf16_spec(const float *__restrict__ am, int kp, const float *__restrict__ bn, int dim,
                                                        int64_t n0, int64_t n_tiles, MTile m_tile, float *lds, Epilogue epilogue) {
    const int S = (dim + 15) / 16;
    const int nchunk = (S + 1) / 2;
    const int total = (int)n_tiles * nchunk;
    if (total <= 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= SPEC_NC) {
        // ---- loader ------------------------------------------------------------------------------------------------------------
        const int lw = wave - SPEC_NC, sub = lane >> 3;
        const unsigned c_even = 4u * ((lane & 7) ^ ((sub >> 1) & 7)), c_odd = 4u * ((lane & 7) ^ ((4 + (sub >> 1)) & 7));
        const float *b_base = bn + n0 * kp;
        int ti = 0, ki = 0, si = 0;
        const float *a_base = am + m_tile(0) * kp;
        auto issue = [&]() {
            float *slot = lds + si * SPEC_STAGE;
            const float *a_tile = a_base + ki * BK, *b_tile = b_base + ki * BK;
#pragma unroll
            for (int u = 0; u < SPEC_BLOCKS / 2; ++u) {
                const int b = lw * (SPEC_BLOCKS / 2) + u;                    // loader 0: A pieces 0-19; loader 1: A 20-23, B 0-15
                const bool is_a = b < SPEC_MT / 8;
                const int piece = is_a ? b : b - SPEC_MT / 8;
                const float *src = (is_a ? a_tile : b_tile) + (size_t)((unsigned)(piece * 8 + sub) * (unsigned)kp + (piece & 1 ? c_odd : c_even));
                float *base = slot + (is_a ? 0 : SPEC_A) + piece * 8 * PLD;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 reinterpret_cast<__attribute__((address_space(3))) void *>(reinterpret_cast<uintptr_t>(base)),
                                                 16, 0, 0);
            }
            si = si == BIG_STAGES - 1 ? 0 : si + 1;
            if (++ki == nchunk) { ki = 0; ++ti; a_base = am + m_tile(ti) * kp; }
        };
        issue();
        if (total > 1) issue();
        if (total > 2) issue();
        if (total > 2) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
        else if (total > 1) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int it = 0; it + 1 < total; ++it) {
            // chunk it + 1 landed (of this wave's DMA only chunk it + 2's 20 instructions may still be out)
            if (it + 2 < total) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (it + 3 < total) issue();                                     // into the slot the consumers have just left
        }
        return;
    }
    // ---- consumer ----------------------------------------------------------------------------------------------------------------
    const int wm = wave >> 1, wn = wave & 1;
    const int x = (lane >> 1) & 7, half = lane >> 5;
    const int a_off = (wm * 64 + (lane & 31)) * PLD, b_off = SPEC_A + (wn * 64 + (lane & 31)) * PLD;
    f32x16 acc[2][2], tot[2][2];
    zero_acc(acc);
    zero_acc(tot);
    int t = 0, kc = 0, sc = 0;
    __builtin_amdgcn_s_barrier();                                            // chunk 0 landed
    asm volatile("" ::: "memory");
    BfFrag f0, f1;
    load_bf_frag(lds + a_off, lds + b_off, 0, half, x, f0);
    for (int it = 0; it < total; ++it) {
        const float *slot = lds + sc * SPEC_STAGE;
        load_bf_frag(slot + a_off, slot + b_off, 1, half, x, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma_bf_frag(f0, acc);
        __builtin_amdgcn_sched_barrier(0);
        sc = sc == BIG_STAGES - 1 ? 0 : sc + 1;
        if (it + 1 < total) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // f1 is here: nobody reads chunk it's slot after the barrier
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const float *nxt = lds + sc * SPEC_STAGE;
            load_bf_frag(nxt + a_off, nxt + b_off, 0, half, x, f0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_bf_frag(f1, acc);
        __builtin_amdgcn_sched_barrier(0);
        ++kc;
        if ((kc & (kBf16BlockChunks - 1)) == 0 || kc == nchunk) {
            add_acc(tot, acc);
            zero_acc(acc);
        }
        if (kc == nchunk) {
            epilogue((int64_t)t, tot);
            zero_acc(tot);
            kc = 0;
            ++t;
        }
    }
}

// B IN REGISTERS (round 4, Kp = 32 NCH <= 128).  tile_pipeline_bf16 stages a 16 KB chunk of BOTH operands per iteration and waits for
// it at the next barrier: with the matrix part of a chunk down to ~700 cycles the L2 / fabric latency of the chunk (~2 us) is what
// an iteration takes -- the bf16 pipeline alone ran at 15 % of the matrix pipe.  The B tile of a work item never changes: here every
// lane loads its B fragments of ALL k-steps once, straight from the packed rows into registers (32 NCH VGPRs), the LDS holds only A
// stages, and a stage is TWO chunks (32 KB): half the barriers and exposed latencies per tile, half the LDS reads, no B re-reads.
// As: 2 slots x 2 chunks x (128 x 32) floats = 64 KB.
template <int NCH>
struct BRegs {
    bf16x8 h[2 * NCH][2], l[2 * NCH][2];             // [k-step][row block tn]: hi and lo fragments of this lane
};

template <int NCH>
__device__ __forceinline__ void load_bregs(const float *__restrict__ bn, int kp, int64_t n0, BRegs<NCH> &b) {
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const int wn = wave & 1, half = lane >> 5;
    const float *row = bn + (n0 + wn * 64 + (lane & 31)) * kp + half * 8;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const float *p = row + (int64_t)tn * 32 * kp + c * 32 + s2 * 16;     // granules ((s2 * 2 + half) * 2 + {0, 1}) of chunk c
                b.h[2 * c + s2][tn] = *reinterpret_cast<const bf16x8 *>(p);
                b.l[2 * c + s2][tn] = *reinterpret_cast<const bf16x8 *>(p + 4);
            }
}

// acc += the three split products of chunk c (two k-steps): A fragments from the staged chunk, B fragments from registers
// one_step: only the chunk's first k-step (the second one is the zero padding of a row whose 16-k steps are odd in number: dim = 100 is
// seven steps -- an eighth of the tile's MFMAs and fragment reads)
template <int NCH>
__device__ __forceinline__ void mma_chunk_breg(const float *__restrict__ As, const BRegs<NCH> &b, int c, f32x16 (&acc)[2][2], bool one_step = false) {
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const int wm = wave >> 1;
    const int x = (lane >> 1) & 7, half = lane >> 5;
    const float *ap = As + (wm * 64 + (lane & 31)) * PLD;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        if (s2 == 1 && one_step) break;                               // wave-uniform
        const int g = 4 * s2 + 2 * half;
        const int oh = 4 * (g ^ x), ol = 4 * ((g + 1) ^ x);
        const bf16x8 a0h = *reinterpret_cast<const bf16x8 *>(ap + oh), a0l = *reinterpret_cast<const bf16x8 *>(ap + ol);
        const bf16x8 a1h = *reinterpret_cast<const bf16x8 *>(ap + 32 * PLD + oh), a1l = *reinterpret_cast<const bf16x8 *>(ap + 32 * PLD + ol);
        const bf16x8 b0h = b.h[2 * c + s2][0], b0l = b.l[2 * c + s2][0], b1h = b.h[2 * c + s2][1], b1l = b.l[2 * c + s2][1];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b0h, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b1h, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b0h, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b1h, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b0l, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, b1l, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b0l, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, b1l, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, b0h, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, b1h, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, b0h, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, b1h, acc[1][1], 0, 0, 0);
    }
}

template <int NCH, class MTile, class Epilogue>
__device__ __forceinline__ void tile_pipeline_bf16_breg(const float *__restrict__ am, int kp, const float *__restrict__ bn, int64_t n0,
                                                        int64_t n_tiles, MTile m_tile, float *As, Epilogue epilogue, int dim = 0) {
    constexpr int BUF = TILE * PLD;                                   // one chunk
    constexpr int NS = (NCH + 1) / 2;                                 // stages per tile: chunks [2 s, min(2 s + 2, NCH))
    const int64_t total = n_tiles * NS;
    if (total == 0) return;
    const bool odd_steps = dim > 0 && (dim + 15) / 16 == 2 * NCH - 1; // the last chunk's second k-step is padding
    BRegs<NCH> b;
    load_bregs<NCH>(bn, kp, n0, b);
    auto stage = [&](int64_t t, int s, float *slot) {
        stage_packed(am, kp, m_tile(t), (2 * s) * BK, slot);
        if (2 * s + 1 < NCH) stage_packed(am, kp, m_tile(t), (2 * s + 1) * BK, slot + BUF);
    };
    stage(0, 0, As);
    __syncthreads();
    f32x16 acc[2][2];
    zero_acc(acc);
    int64_t t = 0;
    int sidx = 0;
    for (int64_t it = 0; it < total; ++it) {
        float *cur = As + (it & 1) * 2 * BUF, *nxt = As + ((it & 1) ^ 1) * 2 * BUF;
        if (it + 1 < total) {
            int s1 = sidx + 1;
            int64_t t1 = t;
            if (s1 == NS) { s1 = 0; ++t1; }
            stage(t1, s1, nxt);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s)                                 // (compile-time chunk indices for the register file)
            if (s == sidx) {
                mma_chunk_breg<NCH>(cur, b, 2 * s, acc, odd_steps && 2 * s == NCH - 1);
                if (2 * s + 1 < NCH) mma_chunk_breg<NCH>(cur + BUF, b, 2 * s + 1, acc, odd_steps && 2 * s + 1 == NCH - 1);
            }
        if (++sidx == NS) {                                          // before the barrier: see tile_pipeline_bf16
            epilogue(t, acc);
            zero_acc(acc);
            sidx = 0;
            ++t;
        }
        __syncthreads();
    }
}

// one call site for both stagings: PACKED kernels receive packed operands and their Kp in the ld arguments
template <bool PACKED, class MTile, class Epilogue>
__device__ __forceinline__ void run_tiles(const float *__restrict__ am, int64_t m_rows, int lda, const float *__restrict__ bn,
                                          int64_t n_rows, int ldb, int dim, int64_t n0, int64_t n_tiles, MTile m_tile,
                                          float *As, float *Bs, Epilogue epilogue) {
    if constexpr (PACKED) tile_pipeline_packed(am, lda, bn, dim, n0, n_tiles, m_tile, As, Bs, epilogue);
    else tile_pipeline(am, m_rows, lda, bn, n_rows, ldb, dim, n0, n_tiles, m_tile, As, Bs, epilogue);
}

// ---- gold similarity: S_ii as the same k-ordered fmaf chain (one lane per query) ---------------
__global__ void gold_inner_kernel(const float *__restrict__ e1, int64_t n1, int ld1, const float *__restrict__ e2,
                                  int ld2, int dim, const float *__restrict__ csls_r,
                                  const float *__restrict__ csls_c, float *__restrict__ gold) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    const float *a = e1 + i * ld1, *b = e2 + i * ld2;
    float acc = 0.f;
    int k = 0;
    for (; k + 4 <= dim; k += 4) {                    // 16-byte loads (ld % 4 == 0), the chain itself stays k-ordered
        const float4 x = oea::ld4(a + k), y = oea::ld4(b + k);
        acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
    }
    for (; k < dim; ++k) acc = fmaf(a[k], b[k], acc);
    if (csls_r) acc = (2.0f * acc - csls_r[i]) - csls_c[i];
    gold[i] = acc;
}

// ---- fused evaluation: ONE prologue launch + the sweep (oea_rank_eval_metrics) -------------------------------------------------
// The 10,500^2 evaluation of BASELINE configs[1] is a 0.20 ms tile sweep; with its operands packed by two launches, the gold
// similarities by a third, two memsets, the argmax extraction and the metric reduction by two more, and a zero-fill of the
// result buffer, the call took 0.28 ms.  Prologue: both packs + golds + zeroing in one grid-stride
// kernel.  Epilogue: the sweep's workgroups take a ticket after their integer atomics; the last one reads the merged ranks /
// keys back (device-coherent loads), writes argmax and reduces Hits@k / sum(rank + 1) / sum 1 / (rank + 1) in a fixed order.
struct EvalTail {
    unsigned *done;            // NULL: plain oea_rank_eval (argmax / metrics by separate launches); else [1 + query tiles] tickets
    int32_t *argmax;
    long long *hits;           // [nk], then rank_sum at hits[nk]
    double *rr_sum;
    long long *tile_part;      // [query tiles][10]: Hits@k x 8, sum(rank + 1), bits of the tile's sum 1 / (rank + 1)
    int tk[8];
    int nk;
};

// Two levels, so that nobody walks all n1 rows alone (a first version let the globally last workgroup read every rank with
// device-coherent loads: 41 dependent round trips per thread at n1 = 10,500 -- slower than the launches it replaced):
//  (1) the last of the gridDim.y workgroups of a QUERY TILE finalises the tile's 128 rows -- one coherent load per thread --
//      and leaves the tile's partial sums (integers; the reciprocal ranks added in a fixed order);
//  (2) the last tile to do so adds the partials in tile order.
__device__ __forceinline__ void eval_tail(const EvalTail &t, const int32_t *rank, const unsigned long long *best_key, int64_t n1,
                                          long long *s_i, double *s_d, int *s_flag) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Everything the tail reads was written by device-scope ATOMICS (rank: atomicAdd, keys: atomicMax, tile partials:
    // atomic stores), i.e. at the point of coherence already -- ordering them before the ticket only needs this wave's
    // outstanding memory operations to have completed (s_waitcnt), NOT a cache action.  A __threadfence() here costs every
    // workgroup an L2 write-back + invalidate: the operand panels its XCD neighbours share in L2 are thrown away 2,000 times
    // per sweep (measured: 0.283 -> 0.365 ms per 10,500^2 evaluation, gpurun_out r03e / r03f).
    // The wait is spelled out: a workgroup-scope release need not lower to s_waitcnt vmcnt(0) when the
    // workgroup's waves share one L2 anyway (no tgsplit), and this wave's no-return atomics go to other L2 channels than the
    // ticket -- they must have been acknowledged before thread 0 takes it.  s_waitcnt 0 = vmcnt(0) expcnt(0) lgkmcnt(0).
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) *s_flag = atomicAdd(t.done + 1 + blockIdx.x, 1u) == gridDim.y - 1u;
    __syncthreads();
    if (!*s_flag) return;
    long long h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double rr = 0.0;
    const int64_t i = (int64_t)blockIdx.x * TILE + tid;
    if (tid < TILE && i < n1) {
        const int r = __hip_atomic_load(rank + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // other XCDs' atomics
        const unsigned long long key = __hip_atomic_load(best_key + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t.argmax[i] = (int32_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
#pragma unroll
        for (int k = 0; k < 8; ++k) h[k] = (k < t.nk && r < t.tk[k]);
        h[8] = r + 1;
        rr = 1.0 / (double)(r + 1);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) h[k] += __shfl_xor(h[k], off, 64);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rr += __shfl_xor(rr, off, 64);
    if (lane == 0) {                 // the tile buffers are free: every wave is past its last MFMA chunk (barriers above)
#pragma unroll
        for (int k = 0; k < 9; ++k) s_i[wave * 9 + k] = h[k];
        s_d[wave] = rr;
    }
    __syncthreads();
    if (tid < 10) {
        long long v = 0;
        if (tid < 9) { for (int w = 0; w < 4; ++w) v += s_i[w * 9 + tid]; }
        else { double d = 0.0; for (int w = 0; w < 4; ++w) d += s_d[w]; v = __double_as_longlong(d); }
        __hip_atomic_store(t.tile_part + (int64_t)blockIdx.x * 10 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) *s_flag = atomicAdd(t.done, 1u) == gridDim.x - 1u;
    __syncthreads();
    if (!*s_flag) return;
    long long g[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double grr = 0.0;
    for (int q = tid; q < (int)gridDim.x; q += blockDim.x) {
#pragma unroll
        for (int k = 0; k < 9; ++k) g[k] += __hip_atomic_load(t.tile_part + (int64_t)q * 10 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        grr += __longlong_as_double(__hip_atomic_load(t.tile_part + (int64_t)q * 10 + 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) g[k] += __shfl_xor(g[k], off, 64);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) grr += __shfl_xor(grr, off, 64);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) s_i[wave * 9 + k] = g[k];
        s_d[wave] = grr;
    }
    __syncthreads();
    if (tid < 9) {
        long long v = 0;
        for (int w = 0; w < 4; ++w) v += s_i[w * 9 + tid];
        if (tid < t.nk) t.hits[tid] = v;
        else if (tid == 8) t.hits[t.nk] = v;
    } else if (tid == 9) {
        double v = 0.0;
        for (int w = 0; w < 4; ++w) v += s_d[w];
        *t.rr_sum = v;
    }
}

__global__ __launch_bounds__(256) void eval_prologue_kernel(const float *__restrict__ e1, int64_t n1, int ld1,
                                                            const float *__restrict__ e2, int64_t n2, int ld2, int dim,
                                                            float *__restrict__ p1, int64_t n1_pad, float *__restrict__ p2,
                                                            int64_t n2_pad, int kp, const float *__restrict__ csls_r,
                                                            const float *__restrict__ csls_c, int64_t gold_off,
                                                            float *__restrict__ gold, unsigned long long *__restrict__ keys,
                                                            int32_t *__restrict__ rank, unsigned *__restrict__ done, int n_done) {
    const int cpr = kp / 4;
    const int64_t t1 = n1_pad * cpr, t2 = n2_pad * cpr;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = gid; i < t1 + t2; i += stride) {                 // pack_rows_kernel for both operands
        const bool second = i >= t1;
        const int64_t ii = second ? i - t1 : i;
        const int64_t row = ii / cpr;
        const int c = (int)(ii - row * cpr);
        const int k = 8 * (c >> 1) + (c & 1);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < (second ? n2 : n1)) {
            const float *r = second ? e2 + row * ld2 : e1 + row * ld1;
            if (k < dim) v.x = r[k];
            if (k + 2 < dim) v.y = r[k + 2];
            if (k + 4 < dim) v.z = r[k + 4];
            if (k + 6 < dim) v.w = r[k + 6];
        }
        oea::st4((second ? p2 : p1) + row * kp + 4 * c, v);
    }
    for (int64_t i = gid; i < n1; i += stride) {                      // gold_inner_kernel + the two memsets
        const float *a = e1 + i * ld1, *b = e2 + (gold_off + i) * ld2;
        float acc = 0.f;
        int k = 0;
        for (; k + 4 <= dim; k += 4) {
            const float4 x = oea::ld4(a + k), y = oea::ld4(b + k);
            acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
        }
        for (; k < dim; ++k) acc = fmaf(a[k], b[k], acc);
        if (csls_r) acc = (2.0f * acc - csls_r[i]) - csls_c[gold_off + i];
        gold[i] = acc;
        keys[i] = 0ull;
        rank[i] = 0;
    }
    for (int64_t i = gid; i < n_done; i += stride) done[i] = 0u;
}

// ---- fused rank epilogue: M = candidates (e2 rows), N = queries (e1 rows) -----------------------
// grid.x = query tiles, grid.y = candidate chunks.  Per lane: one query (MFMA column) and 16
// candidates per MFMA tile, so the per-query reductions stay in registers across the whole
// candidate sweep; partial results are merged with integer atomics (order independent).
template <bool CSLS, bool PACKED>
__global__ __launch_bounds__(256, 2) void rank_inner_kernel(
    const float *__restrict__ e1, int64_t n1, int ld1, const float *__restrict__ e2, int64_t n2, int ld2,
    int dim, const float *__restrict__ gold, const float *__restrict__ csls_r, const float *__restrict__ csls_c,
    int tiles_per_chunk, int64_t gold_off, int32_t *__restrict__ rank, unsigned long long *__restrict__ best_key,
    EvalTail tail) {
    __shared__ __attribute__((aligned(16))) float As[2 * TILE * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2 * TILE * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t q0 = (int64_t)blockIdx.x * TILE;
    const int64_t nct = (n2 + TILE - 1) / TILE;
    const int64_t ct_begin = (int64_t)blockIdx.y * tiles_per_chunk;
    const int64_t ct_end = (ct_begin + tiles_per_chunk < nct) ? ct_begin + tiles_per_chunk : nct;

    int64_t qi[2];
    float g[2], rq[2];
    int cnt[2] = {0, 0};
    float best[2] = {-INFINITY, -INFINITY};
    int bidx[2] = {0x7fffffff, 0x7fffffff};
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        qi[tn] = q0 + wn * 64 + tn * 32 + (lane & 31);
        const bool ok = qi[tn] < n1;
        g[tn] = ok ? gold[qi[tn]] : 0.f;
        rq[tn] = (ok && CSLS) ? csls_r[qi[tn]] : 0.f;
    }
    run_tiles<PACKED>(
        e2, n2, ld2, e1, n1, ld1, dim, q0, ct_end > ct_begin ? ct_end - ct_begin : 0,
        [=](int64_t t) { return (ct_begin + t) * TILE; }, As, Bs,
        [&](int64_t t, f32x16 (&acc)[2][2]) {
            const int64_t c0 = (ct_begin + t) * TILE;
            const int64_t g_lo = q0 + gold_off;              // the golds of this query tile: columns [g_lo, g_lo + 128)
            const int jb = (int)c0 + wm * 64 + 4 * (lane >> 5);
            const bool below = c0 + TILE <= g_lo, above = c0 >= g_lo + TILE;
            if (c0 + TILE <= n2 && (below || above)) {
                // interior tile away from the golds: every j is on one side of every gold, so the tie
                // rule is a plain >= (j < gold) or > (j > gold); a lane meets its candidates in
                // ascending j, so a strict > keeps the smallest argmax
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = jb + tm * 32 + (r & 3) + 8 * (r >> 2);
                        const float cj = CSLS ? csls_c[j] : 0.f;
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn) {
                            float v = acc[tm][tn][r];
                            if (CSLS) v = fmaf(2.0f, v, -rq[tn]) - cj;       // 2v is exact: == (2v - r) - c
                            cnt[tn] += below ? (v >= g[tn]) : (v > g[tn]);
                            if (v > best[tn]) { best[tn] = v; bidx[tn] = j; }
                        }
                    }
                }
                return;
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t j = jb + tm * 32 + (r & 3) + 8 * (r >> 2);
                    if (j < n2) {
                        const float cj = CSLS ? csls_c[j] : 0.f;
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn) {
                            float v = acc[tm][tn][r];
                            if (CSLS) v = fmaf(2.0f, v, -rq[tn]) - cj;
                            const int64_t i = qi[tn];
                            cnt[tn] += (j != i + gold_off) && (v > g[tn] || (v == g[tn] && j < i + gold_off));
                            if (v > best[tn] || (v == best[tn] && (int)j < bidx[tn])) { best[tn] = v; bidx[tn] = (int)j; }
                        }
                    }
                }
            }
        });
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        // merge the two half-waves (same query, disjoint candidates)
        cnt[tn] += __shfl_xor(cnt[tn], 32, 64);
        const float ob = __shfl_xor(best[tn], 32, 64);
        const int oi = __shfl_xor(bidx[tn], 32, 64);
        if (ob > best[tn] || (ob == best[tn] && oi < bidx[tn])) { best[tn] = ob; bidx[tn] = oi; }
        if (lane < 32 && qi[tn] < n1 && ct_end > ct_begin) {
            if (cnt[tn]) atomicAdd(rank + qi[tn], cnt[tn]);
            const unsigned long long key = ((unsigned long long)f2ord(best[tn]) << 32) | (0xFFFFFFFFu - (uint32_t)bidx[tn]);
            atomicMax(best_key + qi[tn], key);
        }
    }
    if (tail.done) {                 // fused evaluation (oea_rank_eval_metrics): the last workgroups to arrive finish the job
        __shared__ int s_flag;
        eval_tail(tail, rank, best_key, n1, reinterpret_cast<long long *>(As), reinterpret_cast<double *>(Bs), &s_flag);
    }
}

// ---- threshold-append epilogue (strip-free neighbour search, topk.hip): M = candidates, N = queries -------------------
// Same sweep as rank_inner_kernel; the epilogue keeps only the similarities at or above the query's threshold thr[q]
// (estimated from a column sample, so that a few thousand of the n2 candidates survive) and appends (value, column) to a
// list segment PRIVATE to the lane: a query's survivors of chunk y come from the two wave rows (wm) and the two half-waves
// that share it, hence 4 * gridDim.y segments of `cap` entries per query, each in ascending column order.  No atomics, no
// N x N strip in HBM.  Values and columns go to two arrays (one dword store each from the register that holds them) at a
// 32-bit byte offset from a wave-uniform base.  counts[q * nseg + seg] may exceed cap: the entries past cap went to the
// query's spill list (one global atomic each; rare unless the neighbours of a row crowd into one candidate range).
// BF16 (round 6): q / c are the hi / lo split packed rows (ldq = ldc = Kp), the sweep runs on the bf16 matrix pipe and the lists hold
// every pair with v~ >= thr - tol (tol = *tol_ptr bounds |v~ - v|): list_select_kernel decides the neighbourhood of the k-th value
// with exact chains -- the same sets as the fp32 sweep
// NCH > 0 (BF16 only): the query operand in registers (tile_pipeline_bf16_breg, Kp = 32 NCH)
template <bool PACKED, bool BF16 = false, int NCH = 0>
__global__ __launch_bounds__(256, 2) void topk_append_kernel(
    const float *__restrict__ q, int64_t nq, int ldq, const float *__restrict__ c, int64_t nc, int ldc, int dim,
    const float *__restrict__ thr, int tiles_per_chunk, int cap, float *__restrict__ list_vals, int32_t *__restrict__ list_cols,
    int32_t *__restrict__ counts, int32_t *__restrict__ spill_cnt, uint2 *__restrict__ spill, int sp_cap,
    const float *__restrict__ tol_ptr = nullptr) {
    __shared__ __attribute__((aligned(16))) float lds[4 * TILE * LDS_LD];          // one array: the register form uses it as 2 x 2 chunks of A
    float *As = lds, *Bs = lds + 2 * TILE * LDS_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t q0 = (int64_t)blockIdx.x * TILE;
    const int64_t nct = (nc + TILE - 1) / TILE;
    const int64_t ct_begin = (int64_t)blockIdx.y * tiles_per_chunk;
    const int64_t ct_end = (ct_begin + tiles_per_chunk < nct) ? ct_begin + tiles_per_chunk : nct;
    const int nseg = 4 * (int)gridDim.y;
    const int sidx = ((int)blockIdx.y * 2 + wm) * 2 + (lane >> 5);
    float th[2];
    uint32_t boff[2], bbeg[2], blast[2];                          // byte offsets in the lane's segment from the workgroup's base
    int64_t qi[2];
    char *__restrict__ vbase = reinterpret_cast<char *>(list_vals + q0 * nseg * (int64_t)cap);      // wave-uniform
    char *__restrict__ cbase = reinterpret_cast<char *>(list_cols + q0 * nseg * (int64_t)cap);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int ql = wn * 64 + tn * 32 + (lane & 31);
        qi[tn] = q0 + ql;
        th[tn] = qi[tn] < nq ? thr[qi[tn]] - (BF16 ? *tol_ptr : 0.f) : INFINITY;            // padding rows never append
        bbeg[tn] = boff[tn] = 4u * (uint32_t)((ql * nseg + sidx) * cap);
        blast[tn] = boff[tn] + 4u * (uint32_t)(cap - 1);          // survivors past cap overwrite the last slot; boff keeps counting
    }
    auto sweep = [&](f32x16 (&acc)[2][2], int jb, int j_end) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jb + tm * 32 + (r & 3) + 8 * (r >> 2);       // ascending in (tm, r >> 2, r & 3)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const float v = acc[tm][tn][r];
                    if (v >= th[tn] && j < j_end) {
                        // the store path stays branch-free: entries from index cap - 1 on land in the segment's LAST slot
                        // (scratch: the select reads min(count, cap - 1) entries) and go to the row's spill list, one global
                        // atomic each -- rare unless the neighbours of a row crowd into one candidate range
                        const uint32_t at = min(boff[tn], blast[tn]);
                        *reinterpret_cast<float *>(vbase + at) = v;
                        *reinterpret_cast<int32_t *>(cbase + at) = j;
                        if (boff[tn] >= blast[tn]) {
                            const int pos = atomicAdd(spill_cnt + qi[tn], 1);
                            if (pos < sp_cap) spill[qi[tn] * sp_cap + pos] = make_uint2(__float_as_uint(v), (uint32_t)j);
                        }
                        boff[tn] += 4u;
                    }
                }
            }
        }
    };
    auto epilogue = [&](int64_t t, f32x16 (&acc)[2][2]) {
        const int64_t c0 = (ct_begin + t) * TILE;
        const int jb = (int)c0 + wm * 64 + 4 * (lane >> 5);
        if (c0 + TILE <= nc) sweep(acc, jb, 0x7fffffff);          // interior tile: the bound folds away
        else sweep(acc, jb, (int)nc);
    };
    const int64_t n_tiles = ct_end > ct_begin ? ct_end - ct_begin : 0;
    auto m_tile = [=](int64_t t) { return (ct_begin + t) * TILE; };
    if constexpr (BF16 && NCH > 0) tile_pipeline_bf16_breg<NCH>(c, ldc, q, q0, n_tiles, m_tile, As, epilogue, dim);
    else if constexpr (BF16) tile_pipeline_bf16<false>(c, ldc, q, dim, q0, n_tiles, m_tile, As, Bs, epilogue);
    else run_tiles<PACKED>(c, nc, ldc, q, nq, ldq, dim, q0, n_tiles, m_tile, As, Bs, epilogue);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
        if (qi[tn] < nq) counts[qi[tn] * nseg + sidx] = (int32_t)((boff[tn] - bbeg[tn]) >> 2);
}

// ---- symmetric neighbour search: queries == candidates, S = E E^T ---------------------------------------------------------
// Only the tiles on and above the diagonal are computed (bitwise S_ij == S_ji: the k-ordered fmaf chain multiplies the same
// pairs in the same order).  A work item = (query tile qt, candidate tiles [ct_begin, ct_end) with ct_begin >= qt); its tiles
// feed TWO sets of lists: the query side as in topk_append_kernel (lane-private segments of the rows of tile qt), and -- off
// the diagonal -- the CANDIDATE side: row j of tile ct receives (S_ji, i) for the rows i of tile qt at or above thr[j].
// Candidate-side entries go to one small segment per (row j, query tile qt, wave column wn): the 32 lanes of a half-wave
// hold the SAME candidate j for 32 different queries, so the slots of an MFMA register's survivors are the prefix counts of
// a wave ballot -- no atomics, no LDS, no barrier; the segment is written by this wave only, its length goes to
// ccounts[(j * T + qt) * 2 + wn] (uint8).  (First version: one segment per (j, qt), slots from an LDS counter per candidate,
// a returning LDS atomic per survivor between two barriers: 17.0 ms for the 100,000^2 sweep.)
// BF16: e = the hi / lo split rows (pack_rows_bf16_kernel), the values appended are v~ with |v~ - v| <= *tol_ptr, and the cut is
// thr - tol: every pair whose EXACT value reaches thr is in the lists (the select resolves the neighbourhood of the k-th
// value with exact chains, list_select_kernel).
template <bool PACKED, bool BF16>
__global__ __launch_bounds__(256, 2) void topk_append_sym_kernel(
    const float *__restrict__ e, int64_t n, int ld, int dim, const float *__restrict__ thr, const int4 *__restrict__ items,
    int nseg, int cap, float *__restrict__ list_vals, int32_t *__restrict__ list_cols, int32_t *__restrict__ counts, int T, int ccap,
    uint2 *__restrict__ clists, uint8_t *__restrict__ ccounts, int32_t *__restrict__ spill_cnt, uint2 *__restrict__ spill, int sp_cap,
    const float *__restrict__ tol_ptr) {
    __shared__ __attribute__((aligned(16))) float As[2 * TILE * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2 * TILE * LDS_LD];
    const int4 item = items[blockIdx.x];                           // (qt, ct_begin, ct_end, segment group)
    const int qt = item.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l32 = lane & 31;
    const int64_t q0 = (int64_t)qt * TILE;
    const int sidx = (item.w * 2 + wm) * 2 + half;
    const float tol = BF16 ? *tol_ptr : 0.f;
    float th[2];
    uint32_t boff[2], bbeg[2], blast[2];
    int64_t qi[2];
    char *__restrict__ vbase = reinterpret_cast<char *>(list_vals + q0 * nseg * (int64_t)cap);
    char *__restrict__ cbase = reinterpret_cast<char *>(list_cols + q0 * nseg * (int64_t)cap);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int ql = wn * 64 + tn * 32 + l32;
        qi[tn] = q0 + ql;
        th[tn] = qi[tn] < n ? thr[qi[tn]] - tol : INFINITY;
        bbeg[tn] = boff[tn] = 4u * (uint32_t)((ql * nseg + sidx) * cap);
        blast[tn] = boff[tn] + 4u * (uint32_t)(cap - 1);
    }
    // the candidate of accumulator register (tm, r) in this half-wave: local row jl0 + tm * 32 + (r & 3) + 8 * (r >> 2);
    // lane l32 = tm * 16 + r looks after that candidate's threshold and segment length
    const int jl0 = wm * 64 + 4 * half;
    const int my_jl = jl0 + (l32 >> 4) * 32 + (l32 & 3) + 8 * ((l32 & 15) >> 2);
    const uint32_t below = (1u << l32) - 1u;
    auto epilogue = [&](int64_t t, f32x16 (&acc)[2][2]) {
            const int ct = item.y + (int)t;
            const int64_t c0 = (int64_t)ct * TILE;
            const bool offdiag = ct != qt;                           // workgroup-uniform
            const int64_t my_j = c0 + my_jl;
            const float my_tc = (offdiag && my_j < n) ? thr[my_j] - tol : INFINITY;
            int my_cnt = 0;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jl = jl0 + tm * 32 + (r & 3) + 8 * (r >> 2);
                    const int j = (int)c0 + jl;
                    const bool jin = j < n;
                    const float tc = __shfl(my_tc, (lane & 32) + tm * 16 + r, 64);
                    uint2 *__restrict__ seg = clists + (((int64_t)j * T + qt) * 2 + wn) * ccap;
                    int cnt = 0;
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        const float v = acc[tm][tn][r];
                        if (v >= th[tn] && jin) {
                            const uint32_t at = min(boff[tn], blast[tn]);           // last slot = scratch, see topk_append_kernel
                            *reinterpret_cast<float *>(vbase + at) = v;
                            *reinterpret_cast<int32_t *>(cbase + at) = j;
                            if (boff[tn] >= blast[tn]) {
                                const int pos = atomicAdd(spill_cnt + qi[tn], 1);
                                if (pos < sp_cap) spill[qi[tn] * sp_cap + pos] = make_uint2(__float_as_uint(v), (uint32_t)j);
                            }
                            boff[tn] += 4u;
                        }
                        const bool pc = v >= tc && qi[tn] < n;       // tc = +inf on the diagonal and past the last row
                        const unsigned long long bal = __ballot(pc);
                        const uint32_t bh = half ? (uint32_t)(bal >> 32) : (uint32_t)bal;
                        const int slot = cnt + __popc(bh & below);
                        if (pc) seg[min(slot, ccap - 1)] = make_uint2(__float_as_uint(v), (uint32_t)qi[tn]);
                        if (pc && slot >= ccap - 1) {                 // last slot = scratch; row j's spill list
                            const int pos = atomicAdd(spill_cnt + j, 1);
                            if (pos < sp_cap) spill[(int64_t)j * sp_cap + pos] = make_uint2(__float_as_uint(v), (uint32_t)qi[tn]);
                        }
                        cnt += __popc(bh);
                    }
                    if (l32 == tm * 16 + r) my_cnt = cnt;
                }
            }
            // lengths saturate at 255; the select reads min(length, ccap - 1) entries, the rest sits in the row's spill list
            if (offdiag && my_j < n) ccounts[(my_j * T + qt) * 2 + wn] = (uint8_t)min(my_cnt, 255);
        };
    auto m_tile = [=](int64_t t) { return (int64_t)(item.y + t) * TILE; };
    if constexpr (BF16) tile_pipeline_bf16<false>(e, ld, e, dim, q0, (int64_t)(item.z - item.y), m_tile, As, Bs, epilogue);
    else run_tiles<PACKED>(e, n, ld, e, n, ld, dim, q0, (int64_t)(item.z - item.y), m_tile, As, Bs, epilogue);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
        if (qi[tn] < n) counts[qi[tn] * nseg + sidx] = (int32_t)((boff[tn] - bbeg[tn]) >> 2);
}

// ---- stream form of the symmetric neighbour sweep (round 4) ------------------------------------------------------------------
// The append kernel above gives every (query, segment) and every (candidate, query tile) its own little list: 4-byte stores to
// ~64 different lines per instruction and ~28 vector instructions per accumulator element.  Here every WAVE owns two
// append-only streams of 8-byte records (value bits, tag): survivors of the query side (v~ >= cut of the query; tag = local
// query row << 24 | candidate) and of the candidate side (v~ >= cut of the candidate; tag = local candidate row << 24 |
// query), slots = stream position + prefix of a ballot -- full-line stores, no per-lane bookkeeping.  The candidate stream's
// position at every tile boundary is kept (col_off), so that the records aimed at one candidate tile can be found again:
// topk_bucket_kernel (topk.hip) deals the records of one target tile to its 128 rows' compact lists, list_select_kernel selects
// from those.  Trained tables crowd a row's neighbours into few candidate ranges, so some waves see several times the average
// number of records.  A wave whose stream is full only notes (work item, tile, wave, positions at the tile's start) in a redo
// list; topk_stream_redo_kernel computes those tiles again and puts the records that did not fit (stream_ovf_tile: same ballots,
// same positions), as self-describing 16-byte records (value, target row, other index), into 512-record chunks taken from a
// shared pool with one atomic per chunk; topk_overflow_kernel (topk.hip) appends them to the compact lists after the bucketing.
// (Handling the overflow inside the sweep -- a second pass over the accumulators -- cost the sweep 1.2 ms of spills and code size
// on tables that never overflow.)  Only a pool or redo list that runs dry marks rows as lost (row_fail): strip fallback.
constexpr int kOvfChunk = 512;                       // records per overflow chunk

struct OvfState {                                    // per wave (wave-uniform)
    uint4 *__restrict__ pool;
    int32_t *__restrict__ alloc, *__restrict__ len;
    uint8_t *__restrict__ row_fail;
    uint32_t cap_chunks, chunk, used;
};

__device__ __forceinline__ void ovf_append(OvfState &o, bool over, float v, uint32_t target, uint32_t other, int lane) {
    const unsigned long long om = __ballot(over);
    if (om == 0ull) return;                          // wave-uniform
    const uint32_t cnt = (uint32_t)__popcll(om);
    if (o.used + cnt > (uint32_t)kOvfChunk) {
        if (o.chunk < o.cap_chunks && lane == 0) o.len[o.chunk] = (int32_t)o.used;
        uint32_t idx = 0;
        if (lane == 0) idx = (uint32_t)atomicAdd(o.alloc, 1);
        o.chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
        o.used = 0;
    }
    if (over) {
        const uint32_t slot = o.used + __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u));
        if (o.chunk < o.cap_chunks) o.pool[(size_t)o.chunk * kOvfChunk + slot] = make_uint4(__float_as_uint(v), target, other, 0u);
        else o.row_fail[target] = 1;                 // the pool ran dry: this row goes to the strip fallback
    }
    o.used += cnt;
}

// second pass over a tile whose records did not all fit: identical predicates and positions, only the overflow is written
template <bool INTERIOR>
__device__ __forceinline__ void stream_ovf_tile(const f32x16 (&acc)[2][2], int c0, int jl0, int n, const float (&th)[2], const uint32_t (&qidx)[2],
                                                float my_tc, int lane, uint32_t rbytes, uint32_t cbytes, uint32_t rpos, uint32_t cpos,
                                                OvfState &o) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = c0 + jl0 + tm * 32 + (r & 3) + 8 * (r >> 2);
            const float tc = __shfl(my_tc, (lane & 32) + tm * 16 + r, 64);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const float v = acc[tm][tn][r];
                const bool pr = INTERIOR ? v >= th[tn] : (v >= th[tn] && j < n);
                unsigned long long m = __ballot(pr);
                uint32_t at = rpos + 8u * __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                ovf_append(o, pr && at >= rbytes, v, qidx[tn], (uint32_t)j, lane);
                rpos += 8u * (uint32_t)__popcll(m);
                const bool pc = v >= tc;
                m = __ballot(pc);
                at = cpos + 8u * __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                ovf_append(o, pc && at >= cbytes, v, (uint32_t)j, qidx[tn], lane);
                cpos += 8u * (uint32_t)__popcll(m);
            }
        }
    }
}

template <bool INTERIOR>
__device__ __forceinline__ void stream_tile(const f32x16 (&acc)[2][2], int c0, int jl0, int n, const float (&th)[2], const uint32_t (&rtag)[2],
                                            const uint32_t (&qidx)[2], float my_tc, int lane, char *__restrict__ rs, uint32_t rbytes,
                                            char *__restrict__ cs, uint32_t cbytes, uint32_t &rpos, uint32_t &cpos) {
    // positions and capacities in BYTES (32-bit offsets from wave-uniform bases: one shift per record, no 64-bit address math)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jl = jl0 + tm * 32 + (r & 3) + 8 * (r >> 2);
            const int j = c0 + jl;
            const float tc = __shfl(my_tc, (lane & 32) + tm * 16 + r, 64);
            const uint32_t ctag = (uint32_t)jl << 24;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const float v = acc[tm][tn][r];
                const bool pr = INTERIOR ? v >= th[tn] : (v >= th[tn] && j < n);
                unsigned long long m = __ballot(pr);
                if (pr) {
                    const uint32_t at = rpos + 8u * __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (at < rbytes) *reinterpret_cast<uint2 *>(rs + at) = make_uint2(__float_as_uint(v), rtag[tn] | (uint32_t)j);
                }
                rpos += 8u * (uint32_t)__popcll(m);
                const bool pc = v >= tc;                          // tc = +inf on the diagonal, past the last row; th-side rows past n never match
                m = __ballot(pc);
                if (pc) {
                    const uint32_t at = cpos + 8u * __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (at < cbytes) *reinterpret_cast<uint2 *>(cs + at) = make_uint2(__float_as_uint(v), ctag | qidx[tn]);
                }
                cpos += 8u * (uint32_t)__popcll(m);
            }
        }
    }
}

// ---- the same records without a branch per accumulator (round 6) --------------------------------------------------------------------
// stream_tile's `if (pr)` is taken for 4 of 5 accumulators (k / n = 2 %: some lane of the 64 passes), each time through
// s_and_saveexec / s_cbranch / 64-bit address arithmetic / a second compare against the capacity: ~190 cycles per accumulator and
// side when nothing overlaps it (measured with one wave per SIMD: 24 K cycles of epilogue beside 3 K of MFMA per tile).  Here one
// accumulator and side is ONE straight block: v_cmpx puts the predicate into EXEC, the slot is mbcnt(EXEC), the record leaves through
// a raw buffer whose num_records IS the stream's capacity (the hardware drops what does not fit: positions, and so the redo pass,
// are unchanged), the position advances on the scalar unit, EXEC is restored.  5 VALU + 2 stores + 3 SALU, no branch.
template <uint32_t TAG_ADD>
__device__ __forceinline__ void stream_append(float v, float cut, uint32_t tag_base, __amdgpu_buffer_rsrc_t srd, uint32_t &pos,
                                              unsigned long long full) {
    // (the record as ONE 8-byte store from a register pair was slower: 12.4 against 11.6 ms per 100,000^2 search -- the pair costs a
    //  copy of the accumulator for all lanes and registers the kernel does not have)
    uint32_t c, t, cnt;
    asm volatile(
        "v_cmpx_ge_f32 vcc, %[v], %[cut]\n\t"
        "s_nop 1\n\t"                                      // a VALU result in a scalar register (EXEC) read as DATA by the next VALU: 2 wait states
        "v_mbcnt_lo_u32_b32 %[c], exec_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[c], exec_hi, %[c]\n\t"
        "v_lshl_add_u32 %[c], %[c], 3, %[pos]\n\t"
        "v_add_u32 %[t], %[ta], %[tb]\n\t"
        "buffer_store_dword %[v], %[c], %[srd], 0 offen\n\t"
        "buffer_store_dword %[t], %[c], %[srd], 0 offen offset:4\n\t"
        "s_bcnt1_i32_b64 %[cnt], exec\n\t"
        "s_lshl3_add_u32 %[pos], %[cnt], %[pos]\n\t"
        "s_mov_b64 exec, %[full]\n\t"
        : [c] "=&v"(c), [t] "=&v"(t), [cnt] "=&s"(cnt), [pos] "+s"(pos)
        : [v] "v"(v), [cut] "v"(cut), [tb] "v"(tag_base), [ta] "n"(TAG_ADD), [srd] "s"(srd), [full] "s"(full)
        : "vcc", "scc", "memory");
}

// interior tiles (every candidate row exists); rj[tn] = rtag[tn] + c0 + jl0 and cq[tn] = (jl0 << 24) | qidx[tn] carry the lane's part of
// the second record word, the (tm, r) part is a constant of the step I = 16 tm + r
template <int I>
__device__ __forceinline__ void stream_fast_steps(const f32x16 (&acc)[2][2], const float (&th)[2], const uint32_t (&rj)[2], const uint32_t (&cq)[2],
                                                  float my_tc, bool upper, __amdgpu_buffer_rsrc_t srd_r, __amdgpu_buffer_rsrc_t srd_c,
                                                  uint32_t &rpos, uint32_t &cpos, unsigned long long full) {
    if constexpr (I < 32) {
        constexpr int tm = I >> 4, r = I & 15;
        constexpr uint32_t jc = (uint32_t)(tm * 32 + (r & 3) + 8 * (r >> 2));
        const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_tc), tm * 16 + r));
        const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_tc), 32 + tm * 16 + r));
        const float tc = upper ? t1 : t0;
        stream_append<jc>(acc[tm][0][r], th[0], rj[0], srd_r, rpos, full);
        stream_append<(jc << 24)>(acc[tm][0][r], tc, cq[0], srd_c, cpos, full);
        stream_append<jc>(acc[tm][1][r], th[1], rj[1], srd_r, rpos, full);
        stream_append<(jc << 24)>(acc[tm][1][r], tc, cq[1], srd_c, cpos, full);
        stream_fast_steps<I + 1>(acc, th, rj, cq, my_tc, upper, srd_r, srd_c, rpos, cpos, full);
    }
}

__device__ __forceinline__ void stream_tile_fast(const f32x16 (&acc)[2][2], const float (&th)[2], const uint32_t (&rj)[2], const uint32_t (&cq)[2],
                                                 float my_tc, int lane, __amdgpu_buffer_rsrc_t srd_r, __amdgpu_buffer_rsrc_t srd_c,
                                                 uint32_t &rpos, uint32_t &cpos) {
    const unsigned long long full = __builtin_amdgcn_ballot_w64(true);
    stream_fast_steps<0>(acc, th, rj, cq, my_tc, lane >= 32, srd_r, srd_c, rpos, cpos, full);
}

// NCH = Kp / 32 in {1..4}: the query tile's operand in registers (tile_pipeline_bf16_breg); 0: both operands through LDS
// FAST: every tile through stream_tile_fast (compile-time: the branching epilogue and its registers are not in the kernel)
template <int NCH, bool FAST = false>
__global__ __launch_bounds__(256, 2) void topk_stream_sym_kernel(
    const float *__restrict__ e, int64_t n, int kp, int dim, const float *__restrict__ thr, const int4 *__restrict__ items,
    uint2 *__restrict__ row_streams, int rcap, uint2 *__restrict__ col_streams, int ccap, int32_t *__restrict__ row_cnt,
    int32_t *__restrict__ col_off, int lp1, uint8_t *__restrict__ row_fail, const float *__restrict__ tol_ptr,
    int32_t *__restrict__ redo_cnt, int4 *__restrict__ redo, int redo_cap, int fast) {
    __shared__ __attribute__((aligned(16))) float lds[NCH > 0 ? 4 * TILE * PLD : 4 * TILE * LDS_LD];
    float *As = lds, *Bs = lds + 2 * TILE * LDS_LD;
    const int4 item = items[blockIdx.x];                           // (qt, ct_begin, ct_end, segment group)
    const int qt = item.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l32 = lane & 31;
    const int64_t q0 = (int64_t)qt * TILE;
    const size_t wid = (size_t)blockIdx.x * 4 + wave;
    char *__restrict__ rs = reinterpret_cast<char *>(row_streams + wid * rcap);
    char *__restrict__ cs = reinterpret_cast<char *>(col_streams + wid * ccap);
    int32_t *__restrict__ coff = col_off + wid * lp1;
    const uint32_t rbytes = 8u * (uint32_t)rcap, cbytes = 8u * (uint32_t)ccap;
    const float tol = *tol_ptr;
    float th[2];
    uint32_t rtag[2], qidx[2];
    int64_t qi[2];
    bool q_in = true;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int ql = wn * 64 + tn * 32 + l32;
        qi[tn] = q0 + ql;
        q_in = q_in && qi[tn] < n;
        th[tn] = qi[tn] < n ? thr[qi[tn]] - tol : INFINITY;
        rtag[tn] = (uint32_t)ql << 24;
        qidx[tn] = (uint32_t)qi[tn];
    }
    const bool q_all = __ballot(!q_in) == 0ull;                    // every query row of this wave exists (else: candidate side per lane)
    const int jl0 = wm * 64 + 4 * half;
    const int my_jl = jl0 + (l32 >> 4) * 32 + (l32 & 3) + 8 * ((l32 & 15) >> 2);
    uint32_t rpos = 0, cpos = 0;                                   // wave-uniform, bytes
    auto uniform_ptr = [](char *p) {                              // the wave's stream base, on the scalar unit
        const uintptr_t u = reinterpret_cast<uintptr_t>(p);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
        return reinterpret_cast<char *>(((uintptr_t)hi << 32) | lo);
    };
    const __amdgpu_buffer_rsrc_t srd_r = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(rs), 0, (int)rbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(cs), 0, (int)cbytes, 0x00020000);
    const uint32_t cq[2] = {((uint32_t)jl0 << 24) | qidx[0], ((uint32_t)jl0 << 24) | qidx[1]};
    const bool fast_on = fast != 0;
    float thr_pre = INFINITY;                                        // thr of this lane's candidate row in the tile about to be finished
    {
        const int64_t j0 = (int64_t)item.y * TILE + my_jl;
        if (item.y < item.z && j0 < n) thr_pre = thr[j0];
    }
    auto epilogue = [&](int64_t t, f32x16 (&acc)[2][2]) {
        const int ct = item.y + (int)t;
        const int64_t c0 = (int64_t)ct * TILE;
        const bool offdiag = ct != qt;                               // workgroup-uniform
        const int64_t my_j = c0 + my_jl;
        // the candidate-side cut is +inf where nothing may be recorded: on the diagonal and past the last candidate row
        const float my_tc = (offdiag && my_j < n) ? thr_pre - tol : INFINITY;
        {                                                            // the next tile's threshold, one tile ahead (an exposed L2 round trip
            const int64_t nj = c0 + TILE + my_jl;                    // per tile and wave when asked for here)
            thr_pre = (ct + 1 < item.z && nj < n) ? thr[nj] : INFINITY;
        }
        if (lane == 0) coff[t] = (int32_t)(cpos >> 3);
        if (!q_all) {                                                // ragged last query tile: the (zero) rows past n must not reach
                                                                     // the candidate lists: their accumulators become -inf
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        if (qi[tn] >= n) acc[tm][tn][r] = -INFINITY;
        }
        const uint32_t r0 = rpos, p0 = cpos;
        if (FAST || (c0 + TILE <= n && fast_on)) {
            if (FAST && c0 + TILE > n) {                             // ragged last candidate tile: rows past n never pass a cut
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((int)c0 + jl0 + tm * 32 + (r & 3) + 8 * (r >> 2) >= (int)n) { acc[tm][0][r] = -INFINITY; acc[tm][1][r] = -INFINITY; }
            }
            const uint32_t rj[2] = {rtag[0] + (uint32_t)((int)c0 + jl0), rtag[1] + (uint32_t)((int)c0 + jl0)};
            stream_tile_fast(acc, th, rj, cq, my_tc, lane, srd_r, srd_c, rpos, cpos);
        } else if (c0 + TILE <= n) stream_tile<true>(acc, (int)c0, jl0, (int)n, th, rtag, qidx, my_tc, lane, rs, rbytes, cs, cbytes, rpos, cpos);
        else stream_tile<false>(acc, (int)c0, jl0, (int)n, th, rtag, qidx, my_tc, lane, rs, rbytes, cs, cbytes, rpos, cpos);
        if (rpos > rbytes || cpos > cbytes) {                        // wave-uniform, rare: records of this tile did not fit
            int slot = 0;
            if (lane == 0) {
                slot = atomicAdd(redo_cnt, 1);
                if (slot < redo_cap) {
                    redo[2 * (size_t)slot] = make_int4((int)blockIdx.x, (int)t, wave, 0);
                    redo[2 * (size_t)slot + 1] = make_int4((int)r0, (int)p0, 0, 0);
                }
            }
            slot = __builtin_amdgcn_readfirstlane(slot);
            if (slot >= redo_cap) {                                  // (never, in practice) the rows this wave may have lost
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    if (qi[tn] < n) row_fail[qi[tn]] = 1;
                if (offdiag && my_j < n) row_fail[my_j] = 1;
            }
        }
    };
    const int64_t n_tiles = (int64_t)(item.z - item.y);
    auto m_tile = [=](int64_t t) { return (int64_t)(item.y + t) * TILE; };
    if constexpr (NCH > 0) tile_pipeline_bf16_breg<NCH>(e, kp, e, q0, n_tiles, m_tile, As, epilogue, dim);
    else tile_pipeline_bf16<false>(e, kp, e, dim, q0, n_tiles, m_tile, As, Bs, epilogue);
    if (lane == 0) {
        coff[n_tiles] = (int32_t)(cpos >> 3);
        row_cnt[wid] = (int32_t)(min(rpos, rbytes) >> 3);
    }
}

// the tiles of the redo list, again: only the recorded wave writes, and only what did not fit its streams
__global__ __launch_bounds__(256, 2) void topk_stream_redo_kernel(
    const float *__restrict__ e, int64_t n, int kp, int dim, const float *__restrict__ thr, const int4 *__restrict__ items, int rcap, int ccap,
    uint8_t *__restrict__ row_fail, const float *__restrict__ tol_ptr, const int32_t *__restrict__ redo_cnt, const int4 *__restrict__ redo,
    int redo_cap, uint4 *__restrict__ ovf_pool, int32_t *__restrict__ ovf_alloc, int32_t *__restrict__ ovf_len, int ovf_chunks) {
    __shared__ __attribute__((aligned(16))) float As[2 * TILE * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2 * TILE * LDS_LD];
    const int n_redo = min(*redo_cnt, redo_cap);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l32 = lane & 31;
    const uint32_t rbytes = 8u * (uint32_t)rcap, cbytes = 8u * (uint32_t)ccap;
    const float tol = *tol_ptr;
    OvfState ovf{ovf_pool, ovf_alloc, ovf_len, row_fail, (uint32_t)ovf_chunks, 0xFFFFFFFFu, (uint32_t)kOvfChunk};
    const int jl0 = wm * 64 + 4 * half;
    const int my_jl = jl0 + (l32 >> 4) * 32 + (l32 & 3) + 8 * ((l32 & 15) >> 2);
    for (int idx = blockIdx.x; idx < n_redo; idx += gridDim.x) {
        const int4 ent = redo[2 * (size_t)idx], pos = redo[2 * (size_t)idx + 1];
        const int4 item = items[ent.x];
        const int qt = item.x, ct = item.y + ent.y;
        const int64_t q0 = (int64_t)qt * TILE, c0 = (int64_t)ct * TILE;
        float th[2];
        uint32_t qidx[2];
        int64_t qi[2];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            qi[tn] = q0 + wn * 64 + tn * 32 + l32;
            th[tn] = qi[tn] < n ? thr[qi[tn]] - tol : INFINITY;
            qidx[tn] = (uint32_t)qi[tn];
        }
        const int64_t my_j = c0 + my_jl;
        const float my_tc = (ct != qt && my_j < n) ? thr[my_j] - tol : INFINITY;
        tile_pipeline_bf16<false>(e, kp, e, dim, q0, 1, [=](int64_t) { return c0; }, As, Bs, [&](int64_t, f32x16 (&acc)[2][2]) {
            if (wave != ent.z) return;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        if (qi[tn] >= n) acc[tm][tn][r] = -INFINITY;
            if (c0 + TILE <= n) stream_ovf_tile<true>(acc, (int)c0, jl0, (int)n, th, qidx, my_tc, lane, rbytes, cbytes, (uint32_t)pos.x, (uint32_t)pos.y, ovf);
            else stream_ovf_tile<false>(acc, (int)c0, jl0, (int)n, th, qidx, my_tc, lane, rbytes, cbytes, (uint32_t)pos.x, (uint32_t)pos.y, ovf);
        });
        __syncthreads();                                             // the LDS buffers are staged again by the next entry
    }
    if (lane == 0 && ovf.chunk < ovf.cap_chunks) ovf_len[ovf.chunk] = (int32_t)ovf.used;
}

__global__ void rank_finalize_kernel(const unsigned long long *__restrict__ best_key, int64_t n1,
                                     int32_t *__restrict__ argmax) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n1) argmax[i] = (int32_t)(0xFFFFFFFFu - (uint32_t)(best_key[i] & 0xFFFFFFFFull));
}

// ---- store epilogue: M = e1 rows (output rows), N = e2 rows (output columns) ----------------------
template <bool PACKED>
__global__ __launch_bounds__(256, 2) void sim_inner_store_kernel(const float *__restrict__ e1, int64_t n1, int ld1,
                                                              const float *__restrict__ e2, int64_t n2, int ld2,
                                                              int dim, float *__restrict__ out, int64_t ld_out,
                                                              const int32_t *__restrict__ gate) {
    __shared__ __attribute__((aligned(16))) float As[2 * TILE * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2 * TILE * LDS_LD];
    if (gate && *gate == 0) return;          // device-side gate (the neighbour search's fallback sweep: nothing failed)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * TILE, c0 = (int64_t)blockIdx.x * TILE;
    run_tiles<PACKED>(
        e1, n1, ld1, e2, n2, ld2, dim, c0, 1, [=](int64_t) { return m0; }, As, Bs,
        [&](int64_t, f32x16 (&acc)[2][2]) {
            float *tile = out + m0 * ld_out + c0;                                    // wave-uniform
            const int rows_left = (int)(n1 - m0 < TILE ? n1 - m0 : TILE), cols_left = (int)(n2 - c0 < TILE ? n2 - c0 : TILE);
            const int col = wn * 64 + (lane & 31), rbase = wm * 64 + 4 * (lane >> 5);
            const bool ok0 = col < cols_left, ok1 = col + 32 < cols_left;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + tm * 32 + (r & 3) + 8 * (r >> 2);
                    if (row < rows_left) {
                        float *p = tile + (row * (int)ld_out + col);               // 128 rows * ld_out < 2^31
                        if (ok0) p[0] = acc[tm][0][r];
                        if (ok1) p[32] = acc[tm][1][r];
                    }
                    asm volatile("" ::: "memory");      // keep the 32 row addresses from all being live at once
                }
        });
}

// ---- fp64 VALU tiles: manhattan / euclidean --------------------------------------------------------
// 64 queries x 64 candidates per block, 4x4 per thread, K chunk 32 staged as doubles.
// sim = float(1 - sum_k |a_k - b_k|)  (scipy cdist cityblock, similarity.py:46-48), sequential k.
constexpr int VT = 64, VK = 32;

__device__ __forceinline__ void stage_f64(const float *__restrict__ src, int64_t n, int ld, int dim, int64_t row0,
                                          int k0, double *__restrict__ dst /* [VK][VT+1] */, int tid) {
    // thread -> (row = tid / 4, 8 k values starting at (tid % 4) * 8)
    const int r = tid >> 2, kq = (tid & 3) * 8;
    const int64_t row = row0 + r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + kq + e;
        float v = 0.f;
        if (row < n && k < dim) v = src[row * ld + k];
        dst[(kq + e) * (VT + 1) + r] = (double)v;
    }
}

template <int METRIC>
__device__ __forceinline__ void valu_tile(const float *__restrict__ e1, int64_t n1, int ld1,
                                          const float *__restrict__ e2, int64_t n2, int ld2, int dim, int64_t q0,
                                          int64_t c0, double *Qs, double *Cs, float (&simv)[4][4]) {
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int k0 = 0; k0 < dim; k0 += VK) {
        __syncthreads();
        stage_f64(e1, n1, ld1, dim, q0, k0, Qs, tid);
        stage_f64(e2, n2, ld2, dim, c0, k0, Cs, tid);
        __syncthreads();
        const int kk = min(VK, dim - k0);
        for (int k = 0; k < kk; ++k) {
            double qv[4], cv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) qv[a] = Qs[k * (VT + 1) + ty * 4 + a];
#pragma unroll
            for (int b = 0; b < 4; ++b) cv[b] = Cs[k * (VT + 1) + tx + 16 * b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double t = qv[a] - cv[b];
                    if (METRIC == OEA_METRIC_MANHATTAN) acc[a][b] += fabs(t);
                    else acc[a][b] += t * t;
                }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            simv[a][b] = METRIC == OEA_METRIC_MANHATTAN ? (float)(1.0 - acc[a][b]) : (float)(1.0 - sqrt(acc[a][b]));
}

template <int METRIC>
__global__ void gold_valu_kernel(const float *__restrict__ e1, int64_t n1, int ld1, const float *__restrict__ e2,
                                 int ld2, int dim, const float *__restrict__ csls_r,
                                 const float *__restrict__ csls_c, float *__restrict__ gold) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    const float *a = e1 + i * ld1, *b = e2 + i * ld2;
    double s = 0.0;
    for (int k = 0; k < dim; ++k) {
        const double t = (double)a[k] - (double)b[k];
        if (METRIC == OEA_METRIC_MANHATTAN) s += fabs(t);
        else s += t * t;
    }
    float v = METRIC == OEA_METRIC_MANHATTAN ? (float)(1.0 - s) : (float)(1.0 - sqrt(s));
    if (csls_r) v = (2.0f * v - csls_r[i]) - csls_c[i];
    gold[i] = v;
}

template <int METRIC>
__global__ __launch_bounds__(256) void rank_valu_kernel(
    const float *__restrict__ e1, int64_t n1, int ld1, const float *__restrict__ e2, int64_t n2, int ld2, int dim,
    const float *__restrict__ gold, const float *__restrict__ csls_r, const float *__restrict__ csls_c,
    int tiles_per_chunk, int64_t gold_off, int32_t *__restrict__ rank, unsigned long long *__restrict__ best_key) {
    __shared__ double Qs[VK * (VT + 1)];
    __shared__ double Cs[VK * (VT + 1)];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t q0 = (int64_t)blockIdx.x * VT;
    const int64_t nct = (n2 + VT - 1) / VT;
    const int64_t ct_begin = (int64_t)blockIdx.y * tiles_per_chunk;
    const int64_t ct_end = (ct_begin + tiles_per_chunk < nct) ? ct_begin + tiles_per_chunk : nct;
    int cnt[4] = {0, 0, 0, 0};
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bidx[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    float g[4], rq[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int64_t i = q0 + ty * 4 + a;
        g[a] = i < n1 ? gold[i] : 0.f;
        rq[a] = (i < n1 && csls_r) ? csls_r[i] : 0.f;
    }
    for (int64_t ct = ct_begin; ct < ct_end; ++ct) {
        const int64_t c0 = ct * VT;
        float simv[4][4];
        valu_tile<METRIC>(e1, n1, ld1, e2, n2, ld2, dim, q0, c0, Qs, Cs, simv);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int64_t j = c0 + tx + 16 * b;
            if (j < n2) {
                const float cj = csls_c ? csls_c[j] : 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int64_t i = q0 + ty * 4 + a;
                    float v = simv[a][b];
                    if (csls_r) v = (2.0f * v - rq[a]) - cj;
                    cnt[a] += (j != i + gold_off) && (v > g[a] || (v == g[a] && j < i + gold_off));
                    if (v > best[a] || (v == best[a] && (int)j < bidx[a])) { best[a] = v; bidx[a] = (int)j; }
                }
            }
        }
    }
    if (ct_end > ct_begin) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int64_t i = q0 + ty * 4 + a;
            if (i < n1) {
                if (cnt[a]) atomicAdd(rank + i, cnt[a]);
                const unsigned long long key = ((unsigned long long)f2ord(best[a]) << 32) | (0xFFFFFFFFu - (uint32_t)bidx[a]);
                atomicMax(best_key + i, key);
            }
        }
    }
}

// ---- L1 similarity in fp32 (pre-filter of RDGCN's hard-negative mining, rdgcn.py:75-87) ---------------------------------------
// 1 - sum |a - b| with fp32 accumulation: NOT the bits of scipy's cdist (that is sim_valu_store_kernel's fp64 chain, 92 ms for
// 20,000 x 200,000 x 300 at 0.68 of the fp64 vector peak); it only has to rank the candidates well enough that the true k
// nearest are among the k + margin it keeps -- those are then re-ranked with exact fp64 distances (oea_pair_l1_f64).
// 128 x 128 tile per workgroup, 8 x 8 outputs per thread: per k a thread reads its 8 + 8 operands with FOUR ds_read_b128
// (rows / columns t*4..+4 and 64 + t*4..+4: a 16-lane group reads 256 contiguous bytes, no bank conflict) for 128 vector
// operations -- the first version (64 x 64 tile, 4 x 4 outputs, 8 ds_read_b32 per 32 operations) spent as many LDS cycles as
// VALU cycles and was no faster than the fp64 kernel (88 vs 93 ms).  |.| is a free source modifier of the add.
constexpr int LT = 128, LK = 32;
__global__ __launch_bounds__(256) void sim_l1_f32_store_kernel(const float *__restrict__ e1, int64_t n1, int ld1,
                                                               const float *__restrict__ e2, int64_t n2, int ld2, int dim,
                                                               float *__restrict__ out, int64_t ld_out) {
    __shared__ __attribute__((aligned(16))) float Qs[LK * LT];
    __shared__ __attribute__((aligned(16))) float Cs[LK * LT];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t q0 = (int64_t)blockIdx.y * LT, c0 = (int64_t)blockIdx.x * LT;
    float acc[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = 0.f;
    // staging: thread -> (row = tid / 2, 16 k values starting at (tid % 2) * 16), stored k-major: [k][row]
    const int r = tid >> 1, kq = (tid & 1) * 16;
    for (int k0 = 0; k0 < dim; k0 += LK) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
            const int k = k0 + kq + e;
            float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), cv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q0 + r < n1 && k < ld1) qv = oea::ld4(e1 + (q0 + r) * ld1 + k);          // ld % 4 == 0; pad columns are zero
            if (c0 + r < n2 && k < ld2) cv = oea::ld4(e2 + (c0 + r) * ld2 + k);
            if (k + 0 >= dim) { qv.x = 0.f; cv.x = 0.f; }
            if (k + 1 >= dim) { qv.y = 0.f; cv.y = 0.f; }
            if (k + 2 >= dim) { qv.z = 0.f; cv.z = 0.f; }
            if (k + 3 >= dim) { qv.w = 0.f; cv.w = 0.f; }
            Qs[(kq + e + 0) * LT + r] = qv.x; Qs[(kq + e + 1) * LT + r] = qv.y; Qs[(kq + e + 2) * LT + r] = qv.z; Qs[(kq + e + 3) * LT + r] = qv.w;
            Cs[(kq + e + 0) * LT + r] = cv.x; Cs[(kq + e + 1) * LT + r] = cv.y; Cs[(kq + e + 2) * LT + r] = cv.z; Cs[(kq + e + 3) * LT + r] = cv.w;
        }
        __syncthreads();
        const int kk = min(LK, dim - k0);
        for (int k = 0; k < kk; ++k) {
            const float4 qa = *reinterpret_cast<const float4 *>(Qs + k * LT + ty * 4);
            const float4 qb = *reinterpret_cast<const float4 *>(Qs + k * LT + 64 + ty * 4);
            const float4 ca = *reinterpret_cast<const float4 *>(Cs + k * LT + tx * 4);
            const float4 cb = *reinterpret_cast<const float4 *>(Cs + k * LT + 64 + tx * 4);
            const float qv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
            const float cv[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[a][b] += fabsf(qv[a] - cv[b]);
        }
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int64_t i = q0 + (a < 4 ? ty * 4 + a : 64 + ty * 4 + (a - 4));
        if (i >= n1) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t j = c0 + h * 64 + tx * 4;
            float *o = out + i * ld_out + j;
            if (j + 3 < n2 && (ld_out & 3) == 0) {
                oea::st4(o, make_float4(1.0f - acc[a][h * 4 + 0], 1.0f - acc[a][h * 4 + 1], 1.0f - acc[a][h * 4 + 2], 1.0f - acc[a][h * 4 + 3]));
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (j + b < n2) o[b] = 1.0f - acc[a][h * 4 + b];
            }
        }
    }
}

// ---- fixed-point L1 pre-filter: rows on a common u16 grid, v_sad_u16 (two columns per instruction) -----------------------
// q = round((x - lo) * inv_step) in 0..65535 with lo / step from the table's own range: |x - (lo + q step)| <= step / 2, so
// the grid distance  step * sum_k |qa_k - qb_k|  is within dim * step of the true L1 distance -- a bound the caller turns
// into a certificate (approaches/rdgcn.py:get_neg): a candidate list is accepted only if no row outside it can beat its
// k-th exact distance.  The fp32 kernel above issues 2 vector instructions per (pair, column) and runs at the full
// unpacked issue rate (61 ms for 20,000 x 200,000 x 300); this one issues 1/2.
__global__ __launch_bounds__(256) void quantize_rows_u16_kernel(const float *__restrict__ src, int64_t n, int ld, int dim, float lo,
                                                                float inv_step, uint16_t *__restrict__ dst, int ldq) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per 8 columns (one 16-byte store)
    const int per_row = ldq >> 3;
    if (idx >= n * per_row) return;
    const int64_t row = idx / per_row;
    const int k0 = (int)(idx % per_row) * 8;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        uint32_t h[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = k0 + 2 * e + u;
            float v = 0.f;                                                   // pad columns: the same grid point in every row
            if (k < dim) v = fminf(fmaxf(rintf((src[row * ld + k] - lo) * inv_step), 0.f), 65535.f);
            h[u] = (uint32_t)v;
        }
        w[e] = h[0] | (h[1] << 16);
    }
    *reinterpret_cast<uint4 *>(dst + row * ldq + k0) = make_uint4(w[0], w[1], w[2], w[3]);
}

// out[i, j] = -(float) sum_k |q[i, k] - c[j, k]|  (larger = nearer, what oea_topk_rows selects).  Same tiling as
// sim_l1_f32_store_kernel: 128 x 128 per workgroup, 8 x 8 per thread, operands k-major in LDS as dwords of two columns.
constexpr int UK = 32;                                                       // dwords (64 columns) per staged chunk
__global__ __launch_bounds__(256) void l1_u16_strip_kernel(const uint16_t *__restrict__ q, int64_t nq,
                                                           const uint16_t *__restrict__ c, int64_t nc, int ldq,
                                                           float *__restrict__ out, int64_t ld_out) {
    __shared__ __attribute__((aligned(16))) uint32_t Qs[UK * LT];
    __shared__ __attribute__((aligned(16))) uint32_t Cs[UK * LT];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t q0 = (int64_t)blockIdx.y * LT, c0 = (int64_t)blockIdx.x * LT;
    uint32_t acc[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = 0u;
    const int r = tid >> 1, wq = (tid & 1) * 16;                              // staging: thread -> (row, 16 dwords of the chunk)
    const int ldw = ldq >> 1;                                                // dwords per row, a multiple of 4
    const uint32_t *qrow = reinterpret_cast<const uint32_t *>(q) + (q0 + r) * ldw;
    const uint32_t *crow = reinterpret_cast<const uint32_t *>(c) + (c0 + r) * ldw;
    const bool q_in = q0 + r < nq, c_in = c0 + r < nc;
    for (int w0 = 0; w0 < ldw; w0 += UK) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
            const int w = w0 + wq + e;
            uint4 qv = make_uint4(0u, 0u, 0u, 0u), cv = make_uint4(0u, 0u, 0u, 0u);
            if (q_in && w < ldw) qv = *reinterpret_cast<const uint4 *>(qrow + w);
            if (c_in && w < ldw) cv = *reinterpret_cast<const uint4 *>(crow + w);
            Qs[(wq + e + 0) * LT + r] = qv.x; Qs[(wq + e + 1) * LT + r] = qv.y; Qs[(wq + e + 2) * LT + r] = qv.z; Qs[(wq + e + 3) * LT + r] = qv.w;
            Cs[(wq + e + 0) * LT + r] = cv.x; Cs[(wq + e + 1) * LT + r] = cv.y; Cs[(wq + e + 2) * LT + r] = cv.z; Cs[(wq + e + 3) * LT + r] = cv.w;
        }
        __syncthreads();
        const int kk = min(UK, ldw - w0);
        for (int k = 0; k < kk; ++k) {
            const uint4 qa = *reinterpret_cast<const uint4 *>(Qs + k * LT + ty * 4);
            const uint4 qb = *reinterpret_cast<const uint4 *>(Qs + k * LT + 64 + ty * 4);
            const uint4 ca = *reinterpret_cast<const uint4 *>(Cs + k * LT + tx * 4);
            const uint4 cb = *reinterpret_cast<const uint4 *>(Cs + k * LT + 64 + tx * 4);
            const uint32_t qv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
            const uint32_t cv[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[a][b] = __builtin_amdgcn_sad_u16(qv[a], cv[b], acc[a][b]);
        }
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int64_t i = q0 + (a < 4 ? ty * 4 + a : 64 + ty * 4 + (a - 4));
        if (i >= nq) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t j = c0 + h * 64 + tx * 4;
            float *o = out + i * ld_out + j;
            if (j + 3 < nc && (ld_out & 3) == 0) {
                oea::st4(o, make_float4(-(float)acc[a][h * 4 + 0], -(float)acc[a][h * 4 + 1], -(float)acc[a][h * 4 + 2], -(float)acc[a][h * 4 + 3]));
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (j + b < nc) o[b] = -(float)acc[a][h * 4 + b];
            }
        }
    }
}

// ---- manhattan evaluation from the grid distances (RDGCN's metric, similarity.py:46-48 + alignment.py:146-168) -------------
// The reference ranks float32(1 - cdist(e1, e2, 'cityblock')): fp64 distances of EVERY pair (rank_valu_kernel: 125 ms at
// 70,000^2 x 300, the fp64 vector rate).  Here every pair only gets its 16-bit grid distance G (l1_u16_strip_kernel); the
// true distance lies within `err` of G step, so against the gold distance d_g a candidate is
//     certainly nearer   G step + err < d_g      -> counted,
//     certainly farther  G step - err > d_g      -> ignored,
//     in between                                  -> its EXACT similarity decides (sequential fp64 chain, the bits of
//                                                    rank_valu_kernel / scipy), tie rule included;
// the nearest candidate is the best exact similarity among the candidates within 2 err of the smallest grid distance.
// One workgroup per query row over its strip row [nc] of -G (two reads, the second out of L2); the few exact distances
// are one thread each.  A row whose candidate lists overflow (cannot happen unless thousands of candidates sit within
// `err` of the gold distance) evaluates every pair exactly.
constexpr int kGridAmb = 2048, kGridTop = 4096;

__device__ __forceinline__ float exact_l1_sim(const double *__restrict__ qs, const float *__restrict__ c, int dim) {
    double acc = 0.0;
#pragma unroll 4
    for (int k = 0; k < dim; ++k) acc += fabs(qs[k] - (double)c[k]);       // k ascending: valu_tile's order
    return (float)(1.0 - acc);
}

__global__ __launch_bounds__(256) void rank_l1_grid_rows_kernel(const float *__restrict__ strip, int64_t rows, int64_t row0, int64_t nc,
                                                                int64_t ld, const float *__restrict__ e1, int ld1,
                                                                const float *__restrict__ e2, int ld2, int dim, int64_t gold_off,
                                                                float step, float err, int32_t *__restrict__ rank,
                                                                int32_t *__restrict__ argmax, int32_t *__restrict__ n_exact_rows) {
    extern __shared__ double lds_d[];                              // qs [dim] (padded to even), then the lists
    double *qs = lds_d;
    int32_t *amb = reinterpret_cast<int32_t *>(qs + ((dim + 1) & ~1));
    int32_t *top = amb + kGridAmb;
    __shared__ int s_namb, s_ntop, s_cnt;
    __shared__ float s_gold, s_gmin;
    __shared__ unsigned long long s_best;
    const int tid = threadIdx.x;
    const int64_t r = blockIdx.x;
    if (r >= rows) return;
    const int64_t i = row0 + r, g = i + gold_off;
    for (int k = tid; k < dim; k += 256) qs[k] = (double)e1[i * ld1 + k];
    if (tid == 0) { s_namb = 0; s_ntop = 0; s_cnt = 0; s_best = 0ull; s_gmin = INFINITY; }
    __syncthreads();
    if (tid == 0) s_gold = exact_l1_sim(qs, e2 + g * ld2, dim);
    __syncthreads();
    const float sg = s_gold;
    // slack: the float rounding of the similarities around the gold one and of grid sums >= 2^24
    const float tol = err + 4.0e-7f * fmaxf(fabsf(sg), 1.0f) + 4.0f * step;
    const float dg = (float)(1.0 - (double)sg);
    const float g_lo = (dg - tol) / step, g_hi = (dg + tol) / step;          // in grid units; strip holds -G
    const float *srow = strip + r * ld;
    const float band = 2.0f * tol / step;
    int cnt = 0;
    float gmin = INFINITY;
    // ONE read of the strip row: the candidates for the nearest are collected against the thread's RUNNING minimum (a
    // superset of those within `band` of the row's minimum: ~ln(n / 256) records per thread on unordered data) and filtered
    // once the row minimum is known; only if that list overflows (candidates ordered by falling distance) the row is read again
    for (int64_t j4 = (int64_t)tid * 4; j4 < nc; j4 += 1024) {       // ld % 4 == 0: 16-byte reads; columns >= nc are unwritten
        const float4 v4 = oea::ld4(srow + j4);
        const float Gs[4] = {-v4.x, -v4.y, -v4.z, -v4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = j4 + u;
            const float G = Gs[u];
            if (j >= nc) continue;
            if (G <= gmin + band) {
                const int at = atomicAdd(&s_ntop, 1);
                if (at < kGridTop) top[at] = (int32_t)j;
            }
            gmin = fminf(gmin, G);
            if (j == g) continue;
            if (G < g_lo) ++cnt;
            else if (G <= g_hi) {
                const int at = atomicAdd(&s_namb, 1);
                if (at < kGridAmb) amb[at] = (int32_t)j;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_xor(cnt, off, 64);
        gmin = fminf(gmin, __shfl_xor(gmin, off, 64));
    }
    if ((tid & 63) == 0) {
        atomicAdd(&s_cnt, cnt);
        atomicMin(reinterpret_cast<int *>(&s_gmin), __float_as_int(gmin));     // G >= 0: the int order is the float order
    }
    __syncthreads();
    const float top_hi = s_gmin + band;
    if (s_ntop > kGridTop) {                                        // workgroup-uniform
        __syncthreads();
        if (tid == 0) s_ntop = 0;
        __syncthreads();
        for (int64_t j = tid; j < nc; j += 256) {
            if (-srow[j] <= top_hi) {
                const int at = atomicAdd(&s_ntop, 1);
                if (at < kGridTop) top[at] = (int32_t)j;
            }
        }
        __syncthreads();
    }
    const int namb = s_namb, ntop = s_ntop;
    int extra = 0;
    unsigned long long best = 0ull;
    if (namb > kGridAmb || ntop > kGridTop) {
        // every pair exactly (the lists overflowed); counted, so that the caller can leave the grid path when a table's
        // range makes its error bound useless (a few outliers stretch the grid: most candidates become doubtful)
        if (tid == 0 && n_exact_rows) atomicAdd(n_exact_rows, 1);
        for (int64_t j = tid; j < nc; j += 256) {
            const float v = exact_l1_sim(qs, e2 + j * ld2, dim);
            extra += (j != g) && (v > sg || (v == sg && j < g));
            const unsigned long long key = ((unsigned long long)f2ord(v) << 32) | (0xFFFFFFFFu - (uint32_t)j);
            best = key > best ? key : best;
        }
        __syncthreads();
        if (tid == 0) s_cnt = 0;                                    // the grid count is replaced, not extended
        __syncthreads();
    } else {
        for (int a = tid; a < namb; a += 256) {
            const int64_t j = amb[a];
            const float v = exact_l1_sim(qs, e2 + j * ld2, dim);
            extra += v > sg || (v == sg && j < g);
        }
        for (int a = tid; a < ntop; a += 256) {
            const int64_t j = top[a];
            if (-srow[j] > top_hi) continue;                        // a record of the running minimum only
            const float v = exact_l1_sim(qs, e2 + j * ld2, dim);
            const unsigned long long key = ((unsigned long long)f2ord(v) << 32) | (0xFFFFFFFFu - (uint32_t)j);
            best = key > best ? key : best;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        extra += __shfl_xor(extra, off, 64);
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    if ((tid & 63) == 0) {
        if (extra) atomicAdd(&s_cnt, extra);
        atomicMax(&s_best, best);
    }
    __syncthreads();
    if (tid == 0) {
        rank[i] = s_cnt;
        argmax[i] = (int32_t)(0xFFFFFFFFu - (uint32_t)(s_best & 0xFFFFFFFFull));
    }
}

// ---- the same with CSLS (basic_model.py:132-135: test() runs greedy_alignment a second time with csls = args.csls; for
// GCN-Align / RDGCN that is the manhattan metric) ------------------------------------------------------------------------------
// v_ij = (2 s_ij - r_i) - c_j in fp32 (rank_valu_kernel's expression; s_ij = float(1 - d_ij)).  From the grid distance G the
// similarity is known to +-err, so v~_ij = (2 (1 - G step) - r_i) - c_j is within tol_j = 2 err + rounding slack of v_ij:
// candidates whose interval [v~ - tol, v~ + tol] does not contain the gold value are decided by the strip, the others -- and
// the candidates whose upper bound reaches the row's best lower bound, for the nearest -- by their exact value.
__device__ __forceinline__ float csls_tol(float s_approx, float r, float c, float err, float step) {
    return 2.0f * err + 8.0f * step + 6.0e-7f * (2.0f * fabsf(s_approx) + fabsf(r) + fabsf(c) + 1.0f);
}

__global__ __launch_bounds__(256) void rank_l1_grid_rows_csls_kernel(const float *__restrict__ strip, int64_t rows, int64_t row0,
                                                                     int64_t nc, int64_t ld, const float *__restrict__ e1, int ld1,
                                                                     const float *__restrict__ e2, int ld2, int dim, int64_t gold_off,
                                                                     float step, float err, const float *__restrict__ csls_r,
                                                                     const float *__restrict__ csls_c, int32_t *__restrict__ rank,
                                                                     int32_t *__restrict__ argmax, int32_t *__restrict__ n_exact_rows) {
    extern __shared__ double lds_d[];
    double *qs = lds_d;
    int32_t *amb = reinterpret_cast<int32_t *>(qs + ((dim + 1) & ~1));
    int32_t *top = amb + kGridAmb;
    __shared__ int s_namb, s_ntop, s_cnt;
    __shared__ float s_gold;
    __shared__ unsigned s_lmax_ord;                                  // f2ord bits of the row's best lower bound
    __shared__ unsigned long long s_best;
    const int tid = threadIdx.x;
    const int64_t r = blockIdx.x;
    if (r >= rows) return;
    const int64_t i = row0 + r, g = i + gold_off;
    const float ri = csls_r[i];
    for (int k = tid; k < dim; k += 256) qs[k] = (double)e1[i * ld1 + k];
    if (tid == 0) { s_namb = 0; s_ntop = 0; s_cnt = 0; s_best = 0ull; s_lmax_ord = 0u; }
    __syncthreads();
    if (tid == 0) s_gold = (2.0f * exact_l1_sim(qs, e2 + g * ld2, dim) - ri) - csls_c[g];
    __syncthreads();
    const float vg = s_gold;
    const float *srow = strip + r * ld;
    int cnt = 0;
    float lmax = -INFINITY;                                          // running maximum of the lower bounds v~ - tol
    for (int64_t j4 = (int64_t)tid * 4; j4 < nc; j4 += 1024) {       // ld % 4 == 0: 16-byte reads; columns >= nc are unwritten
        const float4 v4 = oea::ld4(srow + j4);
        const float Gs[4] = {-v4.x, -v4.y, -v4.z, -v4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = j4 + u;
            if (j >= nc) continue;
            const float cj = csls_c[j];
            const float sa = fmaf(-Gs[u], step, 1.0f);
            const float va = (2.0f * sa - ri) - cj;
            const float tol = csls_tol(sa, ri, cj, err, step);
            if (va + tol >= lmax) {
                const int at = atomicAdd(&s_ntop, 1);
                if (at < kGridTop) top[at] = (int32_t)j;
            }
            lmax = fmaxf(lmax, va - tol);
            if (j == g) continue;
            if (va - tol > vg) ++cnt;
            else if (va + tol >= vg) {
                const int at = atomicAdd(&s_namb, 1);
                if (at < kGridAmb) amb[at] = (int32_t)j;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_xor(cnt, off, 64);
        lmax = fmaxf(lmax, __shfl_xor(lmax, off, 64));
    }
    if ((tid & 63) == 0) {
        atomicAdd(&s_cnt, cnt);
        atomicMax(&s_lmax_ord, f2ord(lmax));                         // order-preserving bits: the bound may be negative
    }
    __syncthreads();
    float row_lmax;
    {
        const unsigned o = s_lmax_ord;
        row_lmax = __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
    }
    auto upper = [&](int64_t j) {
        const float cj = csls_c[j];
        const float sa = fmaf(srow[j], step, 1.0f);                   // srow holds -G
        return ((2.0f * sa - ri) - cj) + csls_tol(sa, ri, cj, err, step);
    };
    if (s_ntop > kGridTop) {                                        // workgroup-uniform
        __syncthreads();
        if (tid == 0) s_ntop = 0;
        __syncthreads();
        for (int64_t j = tid; j < nc; j += 256) {
            if (upper(j) >= row_lmax) {
                const int at = atomicAdd(&s_ntop, 1);
                if (at < kGridTop) top[at] = (int32_t)j;
            }
        }
        __syncthreads();
    }
    const int namb = s_namb, ntop = s_ntop;
    int extra = 0;
    unsigned long long best = 0ull;
    if (namb > kGridAmb || ntop > kGridTop) {                        // the lists overflowed: every pair exactly
        if (tid == 0 && n_exact_rows) atomicAdd(n_exact_rows, 1);
        for (int64_t j = tid; j < nc; j += 256) {
            const float v = (2.0f * exact_l1_sim(qs, e2 + j * ld2, dim) - ri) - csls_c[j];
            extra += (j != g) && (v > vg || (v == vg && j < g));
            const unsigned long long key = ((unsigned long long)f2ord(v) << 32) | (0xFFFFFFFFu - (uint32_t)j);
            best = key > best ? key : best;
        }
        __syncthreads();
        if (tid == 0) s_cnt = 0;
        __syncthreads();
    } else {
        for (int a = tid; a < namb; a += 256) {
            const int64_t j = amb[a];
            const float v = (2.0f * exact_l1_sim(qs, e2 + j * ld2, dim) - ri) - csls_c[j];
            extra += v > vg || (v == vg && j < g);
        }
        for (int a = tid; a < ntop; a += 256) {
            const int64_t j = top[a];
            if (upper(j) < row_lmax) continue;                       // a record of the running bound only
            const float v = (2.0f * exact_l1_sim(qs, e2 + j * ld2, dim) - ri) - csls_c[j];
            const unsigned long long key = ((unsigned long long)f2ord(v) << 32) | (0xFFFFFFFFu - (uint32_t)j);
            best = key > best ? key : best;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        extra += __shfl_xor(extra, off, 64);
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    if ((tid & 63) == 0) {
        if (extra) atomicAdd(&s_cnt, extra);
        atomicMax(&s_best, best);
    }
    __syncthreads();
    if (tid == 0) {
        rank[i] = s_cnt;
        argmax[i] = (int32_t)(0xFFFFFFFFu - (uint32_t)(s_best & 0xFFFFFFFFull));
    }
}

// exact similarity float(1 - d) of every (query row, candidate) pair of a candidate list with the SEQUENTIAL fp64 chain
// (k ascending: the bits of valu_tile / scipy's cdist): one thread per pair.  The CSLS means of the manhattan metric are
// sums of these values, so the chain's order matters (pair_l1_f64_kernel below adds in a butterfly order).
__global__ __launch_bounds__(256) void pair_l1_sim_seq_kernel(const float *__restrict__ q, int64_t nq, int ldq,
                                                              const float *__restrict__ table, int ldt, int dim,
                                                              const int32_t *__restrict__ cand, int c, float *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nq * c) return;
    const float *a = q + (p / c) * ldq, *b = table + (int64_t)cand[p] * ldt;
    double acc = 0.0;
#pragma unroll 4
    for (int k = 0; k < dim; ++k) acc += fabs((double)a[k] - (double)b[k]);
    out[p] = (float)(1.0 - acc);
}

// exact fp64 L1 distance of every (query row, candidate) pair of a candidate list: one 16-lane group per pair, lane-strided
// columns, butterfly sum (a fixed order: equal rows give equal sums)
__global__ __launch_bounds__(256) void pair_l1_f64_kernel(const float *__restrict__ q, int64_t nq, int ldq,
                                                          const float *__restrict__ table, int ldt, int dim,
                                                          const int32_t *__restrict__ cand, int c, double *__restrict__ out) {
    const int lane = threadIdx.x & 15;
    const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (p >= nq * c) return;
    const int64_t i = p / c;
    const float *a = q + i * ldq, *b = table + (int64_t)cand[p] * ldt;
    double s = 0.0;
    for (int k = lane; k < dim; k += 16) s += fabs((double)a[k] - (double)b[k]);
    s = oea::group_sum_d<16>(s);
    if (lane == 0) out[p] = s;
}

template <int METRIC>
__global__ __launch_bounds__(256) void sim_valu_store_kernel(const float *__restrict__ e1, int64_t n1, int ld1,
                                                             const float *__restrict__ e2, int64_t n2, int ld2,
                                                             int dim, float *__restrict__ out, int64_t ld_out) {
    __shared__ double Qs[VK * (VT + 1)];
    __shared__ double Cs[VK * (VT + 1)];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t q0 = (int64_t)blockIdx.y * VT, c0 = (int64_t)blockIdx.x * VT;
    float simv[4][4];
    valu_tile<METRIC>(e1, n1, ld1, e2, n2, ld2, dim, q0, c0, Qs, Cs, simv);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int64_t i = q0 + ty * 4 + a, j = c0 + tx + 16 * b;
            if (i < n1 && j < n2) out[i * ld_out + j] = simv[a][b];
        }
}

// ---- integer metric reductions ---------------------------------------------------------------------
// hits[k] = #{rank < top_k[k]}, rank_sum = sum(rank + 1), rr_sum = sum 1/(rank+1) in a fixed order
// (one block, strided partials, tree reduce) -> deterministic.
__global__ __launch_bounds__(1024) void rank_metrics_kernel(const int32_t *__restrict__ rank, int64_t n, int4 tk0,
                                                            int4 tk1, int nk, long long *__restrict__ hits,
                                                            long long *__restrict__ rank_sum, double *__restrict__ rr_sum) {
    __shared__ long long s_i[1024];
    __shared__ double s_d[1024];
    const int tks[8] = {tk0.x, tk0.y, tk0.z, tk0.w, tk1.x, tk1.y, tk1.z, tk1.w};
    long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rs = 0;
    double rr = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = rank[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) h[k] += (k < nk && r < tks[k]);
        rs += r + 1;
        rr += 1.0 / (double)(r + 1);
    }
    // wave butterflies (fixed pairing), then the 16 wave results in wave order: deterministic, two barriers in all
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long v[9];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = h[k];
    v[8] = rs;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rr += __shfl_xor(rr, off, 64);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) s_i[wave * 9 + k] = v[k];
        s_d[wave] = rr;
    }
    __syncthreads();
    if (threadIdx.x <= 9) {
        const int nw = blockDim.x >> 6;
        if (threadIdx.x < 9) {
            long long t = 0;
            for (int w = 0; w < nw; ++w) t += s_i[w * 9 + threadIdx.x];
            if ((int)threadIdx.x < nk) hits[threadIdx.x] = t;
            else if (threadIdx.x == 8) *rank_sum = t;
        } else {
            double t = 0.0;
            for (int w = 0; w < nw; ++w) t += s_d[w];
            *rr_sum = t;
        }
    }
}

// ---- per-row top-k mean (CSLS) -------------------------------------------------------------------
// One wave per row, k <= 32.  Pass 1: every lane takes the maximum of its strided slice; the k-th largest of
// the 64 lane maxima is a lower bound L of the row's k-th largest value (they are k distinct entries >= L).
// Pass 2 (the row is L2-hot): the handful of entries >= L -- about k of them -- go to an LDS list.  The k
// largest of the list are then taken by k rounds of wave-wide "largest head" and summed in DESCENDING order
// exactly like oracle_topk_mean.  Rows with more than kCandMean entries >= L (constant / heavily tied rows)
// fall back to per-lane sorted insertion lists (the previous algorithm).
constexpr int kCandMean = 256;

template <int KMAX>
__device__ float topk_mean_by_insertion(const float *__restrict__ src, int64_t n2, int k, int lane) {
    float top[KMAX];
#pragma unroll
    for (int p = 0; p < KMAX; ++p) top[p] = -INFINITY;
    for (int64_t j = lane; j < n2; j += 64) {
        float v = src[j];
        if (v > top[KMAX - 1]) {
#pragma unroll
            for (int p = 0; p < KMAX; ++p) {
                const float hi = fmaxf(top[p], v), lo = fminf(top[p], v);
                top[p] = hi; v = lo;
            }
        }
    }
    float acc = 0.f;
    int taken = 0;    // how many of this lane's list were consumed
    for (int round = 0; round < k; ++round) {
        float head = -INFINITY;
#pragma unroll
        for (int p = 0; p < KMAX; ++p) if (p == taken) head = top[p];
        float m = head;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        const unsigned long long bal = __ballot(head == m);      // lowest lane holding the maximum advances
        const int winner = __ffsll((long long)bal) - 1;
        if (lane == winner) ++taken;
        acc += m;
    }
    return acc / (float)k;
}

__global__ __launch_bounds__(256) void row_topk_mean_kernel(const float *__restrict__ s, int64_t n1, int64_t n2,
                                                            int64_t ld, int k, float *__restrict__ out,
                                                            const int32_t *__restrict__ out_index /* may be NULL */,
                                                            const int32_t *__restrict__ n_active /* may be NULL */) {
    __shared__ float s_cand[4][kCandMean];
    __shared__ int s_cnt[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= n1 || (n_active && row >= *n_active)) return;  // whole waves leave: no block-wide barrier below
    const float *src = s + row * ld;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(s) & 15) == 0);
    const int64_t n4 = vec ? (n2 & ~(int64_t)3) : 0;        // [0, n4) by float4, the rest scalar
    // ---- pass 1: lane maxima -> L ------------------------------------------------------------------
    float mx = -INFINITY;
    constexpr int U = 4;                                     // 16-byte loads in flight per lane
    for (int64_t j0 = (int64_t)lane * 4; j0 < n4; j0 += 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = j0 + u * 256;
            v[u] = j < n4 ? *reinterpret_cast<const float4 *>(src + j) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) mx = fmaxf(fmaxf(mx, fmaxf(v[u].x, v[u].y)), fmaxf(v[u].z, v[u].w));
    }
    for (int64_t j = n4 + lane; j < n2; j += 64) mx = fmaxf(mx, src[j]);
    float head = mx, L = -INFINITY;
    for (int round = 0; round < k; ++round) {                // k <= 64 distinct lanes exist only if n2 >= 64 * ...; else L = -inf
        float m = head;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        const unsigned long long bal = __ballot(head == m);
        if (lane == __ffsll((long long)bal) - 1) head = -INFINITY;
        L = m;
    }
    // ---- pass 2: entries >= L -----------------------------------------------------------------------
    if (lane == 0) s_cnt[wave] = 0;
    __builtin_amdgcn_wave_barrier();
    float *cand = s_cand[wave];
    for (int64_t j0 = (int64_t)lane * 4; j0 < n4; j0 += 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = j0 + u * 256;
            v[u] = j < n4 ? *reinterpret_cast<const float4 *>(src + j) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j0 + u * 256 >= n4) continue;
            const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (vv[q] >= L) { const int at = atomicAdd(&s_cnt[wave], 1); if (at < kCandMean) cand[at] = vv[q]; }
        }
    }
    for (int64_t j = n4 + lane; j < n2; j += 64) {
        const float v = src[j];
        if (v >= L) { const int at = atomicAdd(&s_cnt[wave], 1); if (at < kCandMean) cand[at] = v; }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const int cnt = s_cnt[wave];
    float result;
    if (cnt > kCandMean || cnt < k) {                        // wave-uniform
        result = k <= 16 ? topk_mean_by_insertion<16>(src, n2, k, lane) : topk_mean_by_insertion<32>(src, n2, k, lane);
    } else {
        float c[kCandMean / 64];
#pragma unroll
        for (int u = 0; u < kCandMean / 64; ++u) c[u] = (u * 64 + lane) < cnt ? cand[u * 64 + lane] : -INFINITY;
        float acc = 0.f;
        for (int round = 0; round < k; ++round) {
            float h = c[0];
#pragma unroll
            for (int u = 1; u < kCandMean / 64; ++u) h = fmaxf(h, c[u]);
            float m = h;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            const unsigned long long bal = __ballot(h == m);
            if (lane == __ffsll((long long)bal) - 1) {       // remove ONE instance of the maximum
                bool done = false;
#pragma unroll
                for (int u = 0; u < kCandMean / 64; ++u)
                    if (!done && c[u] == m) { c[u] = -INFINITY; done = true; }
            }
            acc += m;
        }
        result = acc / (float)k;
    }
    if (lane == 0) out[out_index ? out_index[row] : row] = result;
}

__global__ void csls_apply_kernel(float *__restrict__ s, int64_t n1, int64_t n2, int64_t ld,
                                  const float *__restrict__ r, const float *__restrict__ c) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j < n2) s[i * ld + j] = (2.0f * s[i * ld + j] - r[i]) - c[j];
}

static int pick_chunks(int64_t q_tiles, int64_t c_tiles, int *tiles_per_chunk) {
    // enough workgroups to fill 256 CUs several times over, without splitting finer than a tile
    // (OEA_RANK_WGS overrides the target count: experiments)
    static const int64_t target = [] { const char *e = getenv("OEA_RANK_WGS"); return e ? (int64_t)atoi(e) : (int64_t)3072; }();
    int64_t want = std::max<int64_t>(1, (target + q_tiles - 1) / q_tiles);
    int64_t chunks = std::min<int64_t>(want, c_tiles);
    *tiles_per_chunk = (int)((c_tiles + chunks - 1) / chunks);
    return (int)((c_tiles + *tiles_per_chunk - 1) / *tiles_per_chunk);
}


// launch geometry of the XCD-aware item order (tile_grid_item): 8 * per blocks, XCD c = items [c * per, (c + 1) * per)
static TileGrid make_tile_grid(unsigned nx, unsigned ny) {
    static const bool on = [] { const char *e = getenv("OEA_XCD_MAP"); return !(e && e[0] == '0'); }();
    TileGrid g;
    g.nx = nx; g.ny = ny;
    g.per = on ? (nx * ny + 7u) / 8u : 0u;
    return g;
}
static unsigned tile_grid_blocks(const TileGrid &g) { return g.per ? 8u * g.per : g.nx * g.ny; }

// ---- CSLS means in ONE sweep (similarity.py:57-83 without S or S^T in HBM) ------------------------------------------------
// r_i = mean of the k largest of row i of S = e1 e2^T, c_j = the same for column j.  Both are top-k problems with k ~ 10, so
// a per-row / per-column threshold estimated from a strided sample (as in the strip-free neighbour search, topk.hip) lets
// the tile sweep keep the few hundred values that can matter: the query side in lane-private list segments (no atomics),
// the candidate side in one list per candidate (a returning atomic per survivor: ~1 % of the values).  The exact top-k
// means come from the lists, summed in DESCENDING order like row_topk_mean_kernel (bit-identical with oracle_topk_mean).
// Rows / columns whose list overflowed or fell short are redone from a recomputed strip (bulk: <= 128 of each).
// BF16 (round 4): q / c are the hi / lo split rows, the values v~ are within *tol_ptr of the exact ones, both cuts are lowered
// by that bound and every list entry is a PAIR (v~, index of the other side) -- list_mean_rows_kernel<true> finds the entries that
// can belong to the exact top k and recomputes them with the exact chain.
// NW = 8 (BF16, K > 128): 512 threads on 256-candidate tiles through the three-stage ring (tile_pipeline_bf16_big, dynamic LDS);
// the query lists then have 8 * chunks segments (one per chunk, wave row and half-wave)
template <bool PACKED, bool BF16, int NCH = 0, int NW = 4, int MODE = 1>
__global__ __launch_bounds__(NW * 64, 2) void csls_append_kernel(
    const float *__restrict__ q, int64_t nq, int ldq, const float *__restrict__ c, int64_t nc, int ldc, int dim,
    const float *__restrict__ thr_q, const float *__restrict__ thr_c, int tiles_per_chunk, int cap, int ccap,
    float *__restrict__ qlists, int32_t *__restrict__ qcounts, float *__restrict__ clists, int32_t *__restrict__ ccounts,
    const float *__restrict__ tol_ptr, TileGrid tg) {
    constexpr uint32_t ES = BF16 ? 8u : 4u;                 // bytes per list entry
    constexpr int NC = (NW == 8 && MODE == 2) ? SPEC_NC : NW;       // multiplying waves (MODE 2: waves 6 and 7 only load)
    constexpr int MT = NC * 32;                             // candidate rows of a tile
    extern __shared__ __attribute__((aligned(16))) float csls_dyn_lds[];          // NW == 8: BIG_LDS_BYTES
    __shared__ __attribute__((aligned(16))) float lds_static[NW == 8 ? 4 : (NCH > 0 ? 4 * TILE * PLD : 4 * TILE * LDS_LD)];
    float *lds = NW == 8 ? csls_dyn_lds : lds_static;
    float *As = lds, *Bs = lds + 2 * TILE * LDS_LD;
    unsigned bx, by;
    if (!tile_grid_item(tg, bx, by)) return;
    // candidate j owns 2 * nqt segments of ccap values: one per (query tile, wave column) -- written by ONE wave, whose
    // half-waves hold the same candidate for 32 queries each: slots = prefix counts of a wave ballot (no atomics, no barrier;
    // the first version took a returning LDS atomic per survivor between two barriers per tile)
    const int nqt = (int)tg.nx, qt = (int)bx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l32 = lane & 31;
    const int64_t q0 = (int64_t)bx * TILE;
    const int64_t nct = (nc + MT - 1) / MT;
    const int64_t ct_begin = (int64_t)by * tiles_per_chunk;
    const int64_t ct_end = (ct_begin + tiles_per_chunk < nct) ? ct_begin + tiles_per_chunk : nct;
    const int nseg = NC * (int)tg.ny;
    const int sidx = ((int)by * (NC / 2) + wm) * 2 + half;        // (a loader wave has no segment)
    float th[2];
    uint32_t boff[2], bbeg[2], blast[2];
    int64_t qi[2];
    const float tol = BF16 ? *tol_ptr : 0.f;
    char *__restrict__ vbase = reinterpret_cast<char *>(qlists) + (size_t)q0 * nseg * cap * ES;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int ql = wn * 64 + tn * 32 + l32;
        qi[tn] = q0 + ql;
        th[tn] = qi[tn] < nq ? thr_q[qi[tn]] - tol : INFINITY;
        bbeg[tn] = boff[tn] = ES * (uint32_t)((ql * nseg + sidx) * cap);
        blast[tn] = boff[tn] + ES * (uint32_t)(cap - 1);
    }
    const int jl0 = wm * 64 + 4 * half;
    const int my_jl = jl0 + (l32 >> 4) * 32 + (l32 & 3) + 8 * ((l32 & 15) >> 2);     // lane l32 = tm * 16 + r looks after that candidate
    const uint32_t below = (1u << l32) - 1u;
    auto epilogue = [&](int64_t t, f32x16 (&acc)[2][2]) {
            const int64_t c0 = (ct_begin + t) * MT;
            const int64_t my_j = c0 + my_jl;
            const float my_tc = my_j < nc ? thr_c[my_j] - tol : INFINITY;
            int my_cnt = 0;
            // k ~ 10: ~0.05 % of the values survive either threshold, so most accumulator registers hold none.  One bound per
            // lane -- the smaller of its own query thresholds and the smallest candidate threshold of this tile's wave --
            // lets a register be skipped with two compares and a ballot instead of the slot arithmetic below (which cost the
            // means sweep 12.3 ms against 9.5 ms for the plain rank sweep at 70,000^2)
            float tc_min = my_tc;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) tc_min = fminf(tc_min, __shfl_xor(tc_min, off, 64));
            const float lo0 = fminf(th[0], tc_min), lo1 = fminf(th[1], tc_min);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (__ballot(acc[tm][0][r] >= lo0 || acc[tm][1][r] >= lo1) == 0ull) continue;
                    const int jl = jl0 + tm * 32 + (r & 3) + 8 * (r >> 2);
                    const int j = (int)c0 + jl;
                    const bool jin = j < nc;
                    const float tc = __shfl(my_tc, (lane & 32) + tm * 16 + r, 64);
                    char *__restrict__ seg = reinterpret_cast<char *>(clists) + (size_t)((((int64_t)j * nqt + qt) * 2 + wn) * ccap) * ES;
                    int cnt = 0;
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        const float v = acc[tm][tn][r];
                        if (v >= th[tn] && jin) {
                            if constexpr (BF16) *reinterpret_cast<uint2 *>(vbase + min(boff[tn], blast[tn])) = make_uint2(__float_as_uint(v), (uint32_t)j);
                            else *reinterpret_cast<float *>(vbase + min(boff[tn], blast[tn])) = v;
                            boff[tn] += ES;
                        }
                        const bool pc = v >= tc && qi[tn] < nq;      // tc = +inf past the last candidate
                        const unsigned long long bal = __ballot(pc);
                        const uint32_t bh = half ? (uint32_t)(bal >> 32) : (uint32_t)bal;
                        const int slot = cnt + __popc(bh & below);
                        if (pc && slot < ccap) {
                            if constexpr (BF16) reinterpret_cast<uint2 *>(seg)[slot] = make_uint2(__float_as_uint(v), (uint32_t)qi[tn]);
                            else reinterpret_cast<float *>(seg)[slot] = v;
                        }
                        cnt += __popc(bh);
                    }
                    if (l32 == tm * 16 + r) my_cnt = cnt;
                }
            }
            if (my_j < nc) ccounts[(my_j * nqt + qt) * 2 + wn] = my_cnt;
        };
    auto m_tile = [=](int64_t t) { return (ct_begin + t) * MT; };
    const int64_t n_tiles = ct_end > ct_begin ? ct_end - ct_begin : 0;
    if constexpr (BF16 && NW == 8 && MODE == 2) {
        tile_pipeline_bf16_spec(c, ldc, q, dim, q0, n_tiles, m_tile, lds, epilogue);
        if (wave >= NC) return;                                // a loader
    } else if constexpr (BF16 && NW == 8) tile_pipeline_bf16_big<MODE == 1>(c, ldc, q, dim, q0, n_tiles, m_tile, lds, epilogue);
    else if constexpr (BF16 && NCH > 0) tile_pipeline_bf16_breg<NCH>(c, ldc, q, q0, n_tiles, m_tile, As, epilogue, dim);
    else if constexpr (BF16) tile_pipeline_bf16<true>(c, ldc, q, dim, q0, n_tiles, m_tile, As, Bs, epilogue);
    else run_tiles<PACKED>(c, nc, ldc, q, nq, ldq, dim, q0, n_tiles, m_tile, As, Bs, epilogue);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
        if (qi[tn] < nq) qcounts[qi[tn] * nseg + sidx] = (int32_t)((boff[tn] - bbeg[tn]) / ES);
}

constexpr int kMeanRegs = 16;                 // list values per lane: lists of up to 1,024 survivors
constexpr int kMeanSeg = 2048;               // segments per list (rows: 4 * chunks; columns: two per query tile)

// mean of the k largest of `cnt` values held kMeanRegs per lane (-inf padded): k rounds of wave-wide maximum, summed in
// descending order -- the arithmetic of row_topk_mean_kernel
__device__ __forceinline__ float wave_topk_mean(float (&c)[kMeanRegs], int k, int lane) {
    float acc = 0.f;
    for (int round = 0; round < k; ++round) {
        float h = c[0];
#pragma unroll
        for (int u = 1; u < kMeanRegs; ++u) h = fmaxf(h, c[u]);
        float m = h;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        const unsigned long long bal = __ballot(h == m);
        if (lane == __ffsll((long long)bal) - 1) {       // remove ONE instance of the maximum
            bool done = false;
#pragma unroll
            for (int u = 0; u < kMeanRegs; ++u)
                if (!done && c[u] == m) { c[u] = -INFINITY; done = true; }
        }
        acc += m;
    }
    return acc / (float)k;
}

// one wave per query row: its survivors sit in nseg segments of `cap`
// BF16: the entries are pairs (v~, index); with t~ = the k-th largest v~ every entry that can belong to the exact top k has
// v~ >= t~ - 2 tol (|t - t~| <= tol for the exact k-th value t); those -- k plus a few, at most 64 -- are recomputed with the exact
// k-ordered chain (row `row` of a against row `index` of b), and the mean is the sum of the k largest of THEM in descending
// order: row_topk_mean_kernel's arithmetic on row_topk_mean_kernel's values.  The band has to lie inside the list
// (t~ - 2 tol >= thr - tol, the sweep's cut); otherwise the row fails over to the strip fallback like an overflowed one.
template <bool BF16>
__global__ __launch_bounds__(256) void list_mean_rows_kernel(const float *__restrict__ lists, const int32_t *__restrict__ counts,
                                                             int nseg, int cap, int64_t n_rows, int k, float *__restrict__ out,
                                                             int32_t *__restrict__ fail_rows, int32_t *__restrict__ n_fail,
                                                             const float *__restrict__ a, int lda, const float *__restrict__ b, int ldb,
                                                             int dim, const float *__restrict__ thr, const float *__restrict__ tol_ptr) {
    __shared__ int s_off[4][kMeanSeg + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= n_rows) return;
    int *off = s_off[wave];
    bool bad = false;
    for (int sg = lane; sg < nseg; sg += 64) {
        const int c = counts[row * nseg + sg];
        off[sg + 1] = c;
        bad |= c > cap;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    {   // lengths -> offsets: every lane scans a contiguous run of segments, the runs are chained by a wave scan
        const int per = (nseg + 63) >> 6;
        int mine = 0;
        for (int u = 0; u < per; ++u) {
            const int sg = lane * per + u;
            if (sg < nseg) mine += off[sg + 1];
        }
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        int run = incl - mine;
        for (int u = 0; u < per; ++u) {
            const int sg = lane * per + u;
            if (sg < nseg) { run += off[sg + 1]; off[sg + 1] = run; }
        }
        if (lane == 0) off[0] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const int total = off[nseg];
    bad = __ballot(bad) != 0ull || total < k || total > kMeanRegs * 64;
    if (bad) {
        if (lane == 0) fail_rows[atomicAdd(n_fail, 1)] = (int32_t)row;
        return;
    }
    float c[kMeanRegs];
    uint32_t id[BF16 ? kMeanRegs : 1];
    const float *base = lists + row * nseg * (int64_t)cap * (BF16 ? 2 : 1);
#pragma unroll
    for (int u = 0; u < kMeanRegs; ++u) {
        const int i = u * 64 + lane;
        c[u] = -INFINITY;
        if (BF16) id[u] = 0u;
        if (i < total) {
            int lo = 0, hi = nseg;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (off[mid] <= i) lo = mid; else hi = mid;
            }
            if constexpr (BF16) {
                const uint2 pr = reinterpret_cast<const uint2 *>(base)[(int64_t)lo * cap + (i - off[lo])];
                c[u] = __uint_as_float(pr.x);
                id[u] = pr.y;
            } else {
                c[u] = base[(int64_t)lo * cap + (i - off[lo])];
            }
        }
    }
    if constexpr (!BF16) {
        const float res = wave_topk_mean(c, k, lane);
        if (lane == 0) out[row] = res;
    } else {
        const float tol = *tol_ptr;
        float tk = INFINITY;                                   // t~: the k-th largest approximate value (k rounds of wave maximum)
        {
            float cc[kMeanRegs];
#pragma unroll
            for (int u = 0; u < kMeanRegs; ++u) cc[u] = c[u];
            for (int round = 0; round < k; ++round) {
                float h = cc[0];
#pragma unroll
                for (int u = 1; u < kMeanRegs; ++u) h = fmaxf(h, cc[u]);
                float m = h;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                const unsigned long long bal = __ballot(h == m);
                if (lane == __ffsll((long long)bal) - 1) {
                    bool done = false;
#pragma unroll
                    for (int u = 0; u < kMeanRegs; ++u)
                        if (!done && cc[u] == m) { cc[u] = -INFINITY; done = true; }
                }
                tk = m;
            }
        }
        const float lo2 = tk - 2.0f * tol;
        int *cand = off;                                        // the offsets are dead: 64 candidate indices per wave
        int ncand = 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < kMeanRegs; ++u) {
            const bool in = c[u] >= lo2;
            const unsigned long long bal = __ballot(in);
            const int at = ncand + __popcll(bal & ((1ull << lane) - 1ull));
            if (in && at < 64) cand[at] = (int)id[u];
            ncand += __popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        if (ncand > 64 || lo2 < thr[row] - tol) {
            if (lane == 0) fail_rows[atomicAdd(n_fail, 1)] = (int32_t)row;
            return;
        }
        float e = -INFINITY;
        if (lane < ncand) {
            const float *__restrict__ x = a + row * lda, *__restrict__ y = b + (int64_t)cand[lane] * ldb;
            float acc = 0.f;
            int kk = 0;
            for (; kk + 32 <= dim; kk += 32) {                      // 16 loads in flight, then the chain in k order
                float4 xv[8], yv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { xv[u] = oea::ld4(x + kk + 4 * u); yv[u] = oea::ld4(y + kk + 4 * u); }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc = fmaf(xv[u].x, yv[u].x, acc); acc = fmaf(xv[u].y, yv[u].y, acc);
                    acc = fmaf(xv[u].z, yv[u].z, acc); acc = fmaf(xv[u].w, yv[u].w, acc);
                }
            }
            for (; kk + 4 <= dim; kk += 4) {
                const float4 xv = oea::ld4(x + kk), yv = oea::ld4(y + kk);
                acc = fmaf(xv.x, yv.x, acc); acc = fmaf(xv.y, yv.y, acc); acc = fmaf(xv.z, yv.z, acc); acc = fmaf(xv.w, yv.w, acc);
            }
            for (; kk < dim; ++kk) acc = fmaf(x[kk], y[kk], acc);
            e = acc;
        }
        float acc = 0.f;
        for (int round = 0; round < k; ++round) {
            float m = e;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            const unsigned long long bal = __ballot(e == m);
            if (lane == __ffsll((long long)bal) - 1) e = -INFINITY;
            acc += m;
        }
        if (lane == 0) out[row] = acc / (float)k;
    }
}

// failed rows beyond the bulk path (adversarial inputs): the row by the k-ordered fmaf chain into scratch, then the exact mean
__global__ __launch_bounds__(256) void slow_mean_rows_kernel(const float *__restrict__ a, int lda, const float *__restrict__ b, int64_t nb,
                                                             int ldb, int dim, int k, float *__restrict__ out,
                                                             const int32_t *__restrict__ fail_rows, const int32_t *__restrict__ n_fail,
                                                             int first, float *__restrict__ scratch, int64_t ld) {
    __shared__ float qs[2048];
    const int nf = *n_fail;
    float *srow = scratch + (int64_t)blockIdx.x * ld;
    for (int f = first + blockIdx.x; f < nf; f += gridDim.x) {
        const int64_t row = fail_rows[f];
        for (int i = threadIdx.x; i < dim; i += 256) qs[i] = a[row * lda + i];
        __syncthreads();
        for (int64_t j = threadIdx.x; j < nb; j += 256) {
            const float *br = b + j * ldb;
            float acc = 0.f;
            for (int kk = 0; kk < dim; ++kk) acc = fmaf(qs[kk], br[kk], acc);
            srow[j] = acc;
        }
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x < 64) {
            const float res = k <= 16 ? topk_mean_by_insertion<16>(srow, nb, k, threadIdx.x) : topk_mean_by_insertion<32>(srow, nb, k, threadIdx.x);
            if (threadIdx.x == 0) out[row] = res;
        }
        __syncthreads();
    }
}

// ---- rank of given gold columns on an explicit similarity block (calculate_rank, alignment.py:146-168)
// one workgroup per row: rank = #{S_ij > S_ig} + #{j < g : S_ij == S_ig}, argmax = smallest j of the max
__global__ __launch_bounds__(256) void rank_rows_kernel(const float *__restrict__ s, int64_t nc, int64_t ld,
                                                        const int32_t *__restrict__ gold_idx, int32_t *__restrict__ rank,
                                                        int32_t *__restrict__ argmax) {
    __shared__ int s_cnt[4];
    __shared__ unsigned long long s_key[4];
    const int64_t row = blockIdx.x;
    const float *src = s + row * ld;
    const int g = gold_idx[row];
    const float gv = src[g];
    int cnt = 0;
    unsigned long long best = 0ull;
    for (int64_t j = threadIdx.x; j < nc; j += 256) {
        const float v = src[j];
        cnt += (j != g) && (v > gv || (v == gv && j < g));
        const unsigned long long key = ((unsigned long long)f2ord(v) << 32) | (0xFFFFFFFFu - (uint32_t)j);
        best = key > best ? key : best;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        cnt += __shfl_xor(cnt, off, 64);
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) { s_cnt[threadIdx.x >> 6] = cnt; s_key[threadIdx.x >> 6] = best; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
        unsigned long long b = 0ull;
        for (int w = 0; w < 4; ++w) { c += s_cnt[w]; b = s_key[w] > b ? s_key[w] : b; }
        rank[row] = c;
        argmax[row] = (int32_t)(0xFFFFFFFFu - (uint32_t)(b & 0xFFFFFFFFull));
    }
}

// ---- host side of the packed path ---------------------------------------------------------------------------------
// OEA_TILE_GLDS=0 selects the register-staged pipeline (kept for A/B measurements and as the reference of the
// bit-exactness test between the two stagings).
static bool use_glds() {
    static const bool on = [] { const char *e = getenv("OEA_TILE_GLDS"); return !(e && e[0] == '0'); }();
    return on;
}

struct PackedOp {
    float *p = nullptr;      // [n_pad, kp] in the process-wide scratch slot
    int kp = 0;
};

// Two grow-only scratch slots (one per operand) instead of an allocation per call: at 10,500^2 two stream-ordered
// allocations + frees cost 0.12 ms on a 0.37 ms evaluation.  Reuse is ordered by the stream; a call on ANOTHER stream
// first waits for the event recorded after the previous use.  Growing a slot frees the old one (hipFree waits for the
// device), so kernels still reading it are done.
struct PackSlot {
    float *p = nullptr;
    size_t cap = 0;
    hipStream_t last = nullptr;
    hipEvent_t used = nullptr;
};
static PackSlot g_slot[6];       // 0 = queries, 1 = candidates, 2 / 3 = column / row samples (neighbour search, CSLS means),
                                 // 3 also = the bf16 split rows of the neighbour search, 4 / 5 = the bf16 split operands of the CSLS means

static int reserve_operand(int slot, int64_t n, int dim, hipStream_t st, PackedOp *out, int64_t *n_pad_out);

static int pack_operand(int slot, const float *src, int64_t n, int ld, int dim, hipStream_t st, PackedOp *out) {
    int64_t n_pad = 0;
    const int rc = reserve_operand(slot, n, dim, st, out, &n_pad);
    if (rc != OEA_OK) return rc;
    const int64_t total = n_pad * (out->kp / 4);
    pack_rows_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(total, 256), 16384), 256, 0, st>>>(src, n, ld, dim, out->p, n_pad,
                                                                                                   out->kp);
    return OEA_OK;
}

// the slot's buffer for an [n, dim] operand (grown if needed, ordered behind its previous use), without the pack launch
static int reserve_operand(int slot, int64_t n, int dim, hipStream_t st, PackedOp *out, int64_t *n_pad_out) {
    PackSlot &sl = g_slot[slot];
    const int64_t n_pad = (n + TILE - 1) / TILE * TILE;
    *n_pad_out = n_pad;
    out->kp = (dim + BK - 1) / BK * BK;
    // (+ two tiles of rows: the 256- / 192-candidate tiles of the big pipelines read up to 191 rows past n_pad; products discarded)
    const size_t need = sizeof(float) * (size_t)(n_pad + 2 * TILE) * out->kp;
    if (!sl.used) OEA_CHECK_HIP(hipEventCreateWithFlags(&sl.used, hipEventDisableTiming));
    if (need > sl.cap) {
        if (sl.p) OEA_CHECK_HIP(hipFree(sl.p));
        sl.p = nullptr;
        sl.cap = 0;
        const size_t cap = std::max<size_t>(need + need / 4, (size_t)64 << 20);   // 64 MB = 131,072 rows of K = 128: growth is rare
        OEA_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&sl.p), cap));
        sl.cap = cap;
    } else if (sl.last != st) {
        OEA_CHECK_HIP(hipStreamWaitEvent(st, sl.used, 0));
    }
    sl.last = st;
    out->p = sl.p;
    return OEA_OK;
}
// after the kernels that read the packed operands have been enqueued
static int release_packed(hipStream_t st) {
    for (int i = 0; i < 6; ++i)
        if (g_slot[i].used && g_slot[i].last == st) OEA_CHECK_HIP(hipEventRecord(g_slot[i].used, st));
    return OEA_OK;
}

static void launch_store_packed(const float *e1p, int64_t n1, const float *e2p, int64_t n2, int kp, int dim, float *out,
                                int64_t ld_out, hipStream_t st) {
    sim_inner_store_kernel<true><<<dim3((unsigned)oea::ceil_div(n2, TILE), (unsigned)oea::ceil_div(n1, TILE)), 256, 0, st>>>(
        e1p, n1, kp, e2p, n2, kp, dim, out, ld_out, nullptr);
}


// ---- certified bf16 prefilter for the inner-product rank sweep (round 4) ------------------------------------------------------
// The exact evaluation multiplies in fp32 (v_mfma_f32_32x32x2_f32: 1/16 of the bf16 matrix rate).  Here every operand is split
// x = hi + lo + r (hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-18 |x|) and the tile sweep accumulates hi.hi + hi.lo + lo.hi with
// v_mfma_f32_32x32x16_bf16 -- 3/16 of the fp32 pipe time, the same LDS bytes (hi + lo = 4 B per element).  The result v~ is
// within tol = eps(dim) |q|max |c|max of the exact k-ordered fmaf chain v (split residue 3 * 2^-18, fp32 accumulation of
// 3 * Kp products bounded term by term, the chain's own rounding).  So against the exact gold value g:
//     v~ > g + tol   certainly greater: counted in the sweep;      v~ < g - tol   certainly smaller: ignored;
//     |v~ - g| <= tol                 a RECORD (query, candidate): decided afterwards by the exact chain, tie rule included;
// and the nearest candidate: a candidate can only be the exact row maximum if v~ + tol >= (the best lower bound v~' - tol seen
// so far, gold included), so those are recorded too (the bound is shared by the lanes of a row through an atomic maximum).
// The fix-up kernel evaluates the records -- a few dozen per row -- exactly.  Ranks and nearest candidates are those of the
// fp32 sweep, bit for bit; if the record buffer overflows the caller takes the fp32 sweep.
// (the split operands' pack kernel, mma_chunk_bf16 and tile_pipeline_bf16 sit beside tile_pipeline_packed above: the neighbour
// search's append kernel uses them too)
__device__ __forceinline__ float ord2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o); }

constexpr uint32_t kRecTop = 0x80000000u;        // record kinds: bit 31 of the candidate word

// gold, tolerance, per-row state of the bf16 sweep
// nmax: float bits of max |e1 row|, max |e2 row|, max |csls_c|
__global__ void rank_bf16_init_kernel(const float *__restrict__ gold, int64_t n1, int64_t gold_off, const unsigned *__restrict__ nmax,
                                      float eps_rel, const float *__restrict__ csls_r, float *__restrict__ tol_out,
                                      int32_t *__restrict__ rank, unsigned long long *__restrict__ best_key,
                                      unsigned *__restrict__ lbrow, unsigned *__restrict__ rec_cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float smax = __uint_as_float(nmax[0]) * __uint_as_float(nmax[1]);
    // + an absolute slack for the ulp-level roundings of g +- tol, v~ - tol and of the comparisons themselves (values <= smax)
    const float tol0 = 1.05f * eps_rel * smax + 1.0e-6f * smax + 1e-30f;
    const float scale = 2.0f * smax + __uint_as_float(nmax[2]) + 2.0f * tol0;
    if (i == 0) { tol_out[0] = tol0; tol_out[1] = scale; rec_cnt[0] = 0u; rec_cnt[1] = 0u; }
    if (i >= n1) return;
    const float tol = csls_r ? 2.0f * tol0 + 3.0e-7f * (scale + fabsf(csls_r[i])) : tol0;      // == the sweep's per-lane tolerance
    rank[i] = 0;
    best_key[i] = ((unsigned long long)f2ord(gold[i]) << 32) | (0xFFFFFFFFu - (uint32_t)(i + gold_off));     // the gold is a candidate
    lbrow[i] = f2ord(gold[i] - tol);
}

// WARM: a first pass over the first `tiles_per_chunk` candidate tiles only raises the rows' lower bounds (lbrow = max v~ - tol
// over a few thousand candidates: about the (n2 / 2048)-th largest value of the row), so that the full sweep records as
// nearest-candidate suspects only the few dozen candidates above that -- without it a row whose gold is far from the top
// records every running maximum of every lane (hundreds per row).
// Records go to a slice of the record buffer PRIVATE to the wave (slot = count + prefix of a ballot; no returning atomic:
// the first version took one global round trip per recording wave instruction and ran 4x slower on rows whose gold is far
// from the top).  rec_cnt[2 + wave id] = the slice's length; a full slice raises the overflow flag.
// CSLS: the values compared are (2 s - r_i) - c_j (rank_inner_kernel's expression on the approximate s); tol[tn] then holds
// 2 tol + the rounding slack of the two extra operations for this lane's query.
// The three classes are COMPLEMENTARY comparisons against ghi = g + tol and glo = g - tol (counted: v > ghi; band: glo <= v <=
// ghi; ignored: v < glo) -- a first version tested the band as |v - g| <= tol, whose rounding left a gap of an ulp below ghi
// through which one candidate in ~10^7 fell.  The ulp-level roundings of g +- tol themselves sit inside the absolute slack
// of the tolerance (rank_bf16_init_kernel).
template <bool WARM, bool INTERIOR, bool CSLS, class Acc, class Rec>
__device__ __forceinline__ void rank_bf16_tile(Acc &acc, int jb, int n2, const float (&glo)[2], const float (&ghi)[2], float (&lb)[2],
                                               float (&lbm)[2], int (&cnt)[2], bool (&dirty)[2], const int64_t (&qi)[2], int64_t gold_off,
                                               const float (&tol)[2], const float (&rq)[2], float my_c, Rec &&record) {
    // my_c: the column mean of the candidate this lane looks after (lane l32 = tm * 16 + r of either half: one load per tile and
    // lane + a lane shuffle per register instead of 32 loaded values held in registers)
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jb + tm * 32 + (r & 3) + 8 * (r >> 2);
            const bool jin = INTERIOR || j < n2;
            const float cj = CSLS ? __shfl(my_c, (lane & 32) + tm * 16 + r, 64) : 0.f;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                float v = acc[tm][tn][r];
                if (CSLS) v = fmaf(2.0f, v, -rq[tn]) - cj;
                if (WARM) {
                    if (jin) lb[tn] = fmaxf(lb[tn], v - tol[tn]);
                    continue;
                }
                cnt[tn] += (v > ghi[tn]) & jin;
                const bool band = (v <= ghi[tn]) & (v >= glo[tn]);
                const bool hit = ((v >= lbm[tn]) | band) & jin;
                if (__ballot(hit)) {                                  // wave-uniform branch: rare
                    const bool other = hit && (int64_t)j != qi[tn] + gold_off;
                    record(other && v >= lbm[tn], tn, j, kRecTop);
                    record(other && band, tn, j, 0u);
                    if (hit && v - tol[tn] > lb[tn]) { lb[tn] = v - tol[tn]; lbm[tn] = lb[tn] - tol[tn]; dirty[tn] = true; }
                }
            }
        }
    }
}

// NCH = 0: both operands through LDS (tile_pipeline_bf16, any Kp); NCH = Kp / 32 in {1..4}: B in registers (tile_pipeline_bf16_breg);
// NW = 8: 512 threads on 256-candidate tiles through the three-stage ring (tile_pipeline_bf16_big; tiles_per_chunk counts THOSE tiles)
// MODE (NW = 8): 0 / 1 = tile_pipeline_bf16_big<PF = MODE> on 256-candidate tiles, all 8 waves multiply; 2 = tile_pipeline_bf16_spec on
// 192-candidate tiles: waves 0-5 multiply, waves 6-7 load
template <bool WARM, bool CSLS, int NCH, int NW = 4, int MODE = 1>
__device__ __forceinline__ void rank_bf16_body(
    const float *__restrict__ qp, int64_t n1, int kp, const float *__restrict__ cp, int64_t n2, int dim,
    const float *__restrict__ gold, const float *__restrict__ tol_ptr, const float *__restrict__ csls_r,
    const float *__restrict__ csls_c, int tiles_per_chunk, int64_t gold_off,
    int32_t *__restrict__ rank, unsigned *__restrict__ lbrow, uint2 *__restrict__ rec, unsigned *__restrict__ rec_cnt, unsigned slice_cap,
    TileGrid tg, float *As, float *Bs) {
    unsigned bx, by;
    if (!tile_grid_item(tg, bx, by)) return;               // (the whole workgroup: padding of the XCD-aware launch)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t q0 = (int64_t)bx * TILE;
    constexpr int NC = (NW == 8 && MODE == 2) ? SPEC_NC : NW;   // multiplying waves
    constexpr int MT = NC * 32;                              // candidate rows of a tile
    const int64_t nct = (n2 + MT - 1) / MT;
    const int64_t ct_begin = (int64_t)by * tiles_per_chunk;
    const int64_t ct_end = (ct_begin + tiles_per_chunk < nct) ? ct_begin + tiles_per_chunk : nct;
    const float tol0 = tol_ptr[0];                         // bound on |s~ - s|; tol_ptr[1] = slack scale of the CSLS expression
    const unsigned wid = (by * tg.nx + bx) * (unsigned)NW + (unsigned)wave;
    uint2 *__restrict__ my_rec = rec + (size_t)wid * slice_cap;
    unsigned nrec = 0;                                     // wave-uniform
    int64_t qi[2];
    float glo[2], ghi[2], lb[2], lbm[2], tol[2], rq[2];
    int cnt[2] = {0, 0};
    bool dirty[2] = {false, false};
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        qi[tn] = q0 + wn * 64 + tn * 32 + (lane & 31);
        const bool ok = qi[tn] < n1;
        const float g = ok ? gold[qi[tn]] : INFINITY;      // padding rows: nothing counts, nothing is recorded
        rq[tn] = (CSLS && ok) ? csls_r[qi[tn]] : 0.f;
        // (2 s~ - r) - c against (2 s - r) - c: twice the bound on s, plus the roundings of the fma and of the subtraction on
        // either side (each <= 2^-24 of a value <= 2 |s|max + |r| + |c|max = tol_ptr[1] + |r|)
        tol[tn] = CSLS ? 2.0f * tol0 + 3.0e-7f * (tol_ptr[1] + fabsf(rq[tn])) : tol0;
        ghi[tn] = g + tol[tn];
        glo[tn] = ok ? g - tol[tn] : INFINITY;
        lb[tn] = glo[tn];
        lbm[tn] = lb[tn] - tol[tn];
    }
    auto record = [&](bool pred, int tn, int j, uint32_t kind) {
        const unsigned long long m = __ballot(pred);
        if (m == 0ull) return;
        const unsigned at = nrec + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (pred && at < slice_cap) my_rec[at] = make_uint2((uint32_t)qi[tn], (uint32_t)j | kind);
        nrec += (unsigned)__popcll(m);
    };
    // what a tile's epilogue reads from memory -- the row's shared bound (other lanes / workgroups may have raised it) and, with CSLS,
    // the column mean this lane looks after -- is requested ONE TILE AHEAD (round 6): asked for at the top of the epilogue it was an
    // exposed L2 round trip per tile and wave.  A bound that is one tile old is still a lower bound (it only grows).
    const int l32c = lane & 31;
    const int my_off = wm * 64 + 4 * (lane >> 5) + (l32c >> 4) * 32 + (l32c & 3) + 8 * ((l32c & 15) >> 2);
    unsigned lb_pre[2] = {0u, 0u};
    float c_pre = 0.f;
    auto prefetch = [&](int64_t t) {                         // for tile t of this work item
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
            if (qi[tn] < n1) lb_pre[tn] = __hip_atomic_load(lbrow + qi[tn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (CSLS) {
            const int64_t my_j = (ct_begin + t) * MT + my_off;
            c_pre = (my_j < n2 && ct_begin + t < ct_end) ? csls_c[my_j] : 0.f;
        }
    };
    prefetch(0);
    auto epilogue = [&](int64_t t, f32x16 (&acc)[2][2]) {
            const int64_t c0 = (ct_begin + t) * MT;
            const int jb = (int)c0 + wm * 64 + 4 * (lane >> 5);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                if (qi[tn] < n1) {
                    const float o = ord2f(lb_pre[tn]);
                    if (o > lb[tn]) { lb[tn] = o; lbm[tn] = o - tol[tn]; }
                }
            const float my_c = CSLS ? c_pre : 0.f;
            prefetch(t + 1);
            if (c0 + MT <= n2) rank_bf16_tile<WARM, true, CSLS>(acc, jb, (int)n2, glo, ghi, lb, lbm, cnt, dirty, qi, gold_off, tol, rq, my_c, record);
            else rank_bf16_tile<WARM, false, CSLS>(acc, jb, (int)n2, glo, ghi, lb, lbm, cnt, dirty, qi, gold_off, tol, rq, my_c, record);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                if (dirty[tn]) { atomicMax(lbrow + qi[tn], f2ord(lb[tn])); dirty[tn] = false; }
        };
    auto m_tile = [=](int64_t t) { return (ct_begin + t) * MT; };
    const int64_t n_tiles = ct_end > ct_begin ? ct_end - ct_begin : 0;
    if constexpr (NW == 8 && MODE == 2) {
        tile_pipeline_bf16_spec(cp, kp, qp, dim, q0, n_tiles, m_tile, As, epilogue);
        if (wave >= NC) {                                  // a loader: no ranks, an empty record slice
            if (!WARM && lane == 0) rec_cnt[2 + wid] = 0u;
            return;
        }
    } else if constexpr (NW == 8) tile_pipeline_bf16_big<MODE == 1>(cp, kp, qp, dim, q0, n_tiles, m_tile, As, epilogue);
    else if constexpr (NCH > 0) tile_pipeline_bf16_breg<NCH>(cp, kp, qp, q0, n_tiles, m_tile, As, epilogue, dim);
    else tile_pipeline_bf16<true>(cp, kp, qp, dim, q0, n_tiles, m_tile, As, Bs, epilogue);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        if (WARM) {
            lb[tn] = fmaxf(lb[tn], __shfl_xor(lb[tn], 32, 64));
            if (lane < 32 && qi[tn] < n1) atomicMax(lbrow + qi[tn], f2ord(lb[tn]));
            continue;
        }
        cnt[tn] += __shfl_xor(cnt[tn], 32, 64);
        if (lane < 32 && qi[tn] < n1 && cnt[tn]) atomicAdd(rank + qi[tn], cnt[tn]);
    }
    if (!WARM && lane == 0) {
        rec_cnt[2 + wid] = min(nrec, slice_cap);
        if (nrec) atomicAdd(rec_cnt, nrec);                   // total (statistics)
        if (nrec > slice_cap) atomicMax(rec_cnt + 1, 1u);     // overflow: the caller falls back to the fp32 sweep
    }
}

#define OEA_RANK_BF16_PARAMS                                                                                                        \
    const float *__restrict__ qp, int64_t n1, int kp, const float *__restrict__ cp, int64_t n2, int dim,                            \
    const float *__restrict__ gold, const float *__restrict__ tol_ptr, const float *__restrict__ csls_r,                            \
    const float *__restrict__ csls_c, int tiles_per_chunk, int64_t gold_off, int32_t *__restrict__ rank,                            \
    unsigned *__restrict__ lbrow, uint2 *__restrict__ rec, unsigned *__restrict__ rec_cnt, unsigned slice_cap, TileGrid tg
#define OEA_RANK_BF16_ARGS qp, n1, kp, cp, n2, dim, gold, tol_ptr, csls_r, csls_c, tiles_per_chunk, gold_off, rank, lbrow, rec, rec_cnt, slice_cap, tg

template <bool WARM, bool CSLS>
__global__ __launch_bounds__(256, 2) void rank_bf16_kernel(OEA_RANK_BF16_PARAMS) {
    __shared__ __attribute__((aligned(16))) float As[2 * TILE * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2 * TILE * LDS_LD];
    rank_bf16_body<WARM, CSLS, 0>(OEA_RANK_BF16_ARGS, As, Bs);
}

template <bool WARM, bool CSLS, int NCH>
__global__ __launch_bounds__(256, 2) void rank_bf16_breg_kernel(OEA_RANK_BF16_PARAMS) {
    __shared__ __attribute__((aligned(16))) float As[4 * TILE * PLD];           // 2 slots x 2 chunks
    rank_bf16_body<WARM, CSLS, NCH>(OEA_RANK_BF16_ARGS, As, nullptr);
}

template <bool CSLS, int MODE>
__global__ __launch_bounds__(512, 2) void rank_bf16_big_kernel(OEA_RANK_BF16_PARAMS) {
    extern __shared__ __attribute__((aligned(16))) float big_lds[];                // BIG_LDS_BYTES
    rank_bf16_body<false, CSLS, 0, 8, MODE>(OEA_RANK_BF16_ARGS, big_lds, nullptr);
}

// ONE grid-stride prologue: both bf16 packs, the gold similarities (the exact k-ordered chain), the max row norms of both
// tables, zeroing of ranks / counters
__global__ __launch_bounds__(256) void rank_bf16_prologue_kernel(const float *__restrict__ e1, int64_t n1, int ld1,
                                                                 const float *__restrict__ e2, int64_t n2, int ld2, int dim,
                                                                 int64_t gold_off, uint4 *__restrict__ p1, int64_t n1_pad,
                                                                 uint4 *__restrict__ p2, int64_t n2_pad, int kp,
                                                                 float *__restrict__ gold, unsigned *__restrict__ nmax,
                                                                 int32_t *__restrict__ rank, const float *__restrict__ csls_r,
                                                                 const float *__restrict__ csls_c, int aug) {
    // aug (with the CSLS means): the packed rows carry two more coordinates, q' = [2 q, -r_i, -1] and c' = [c, 1, c_j], so that the
    // tile sweep's product IS 2 <q, c> - r_i - c_j and the sweep runs without the CSLS terms in its epilogue (same kernel, same
    // registers as the plain evaluation); the norms below are those of the augmented rows
    const int cpr = kp / 4;
    const int64_t a_end = n1_pad * cpr, b_end = a_end + n2_pad * cpr;
    // then: the gold chains (one thread per query row), then the row norms (16 lanes per row, table 2 on a wave boundary)
    const int64_t g_end = b_end + (n1 + 63) / 64 * 64;
    const int64_t m1_end = g_end + (16 * n1 + 63) / 64 * 64, m2_end = m1_end + (16 * n2 + 63) / 64 * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m2_end; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < b_end) {
            const bool second = i >= a_end;
            const int64_t li = second ? i - a_end : i;
            const float *src = second ? e2 : e1;
            const int64_t n = second ? n2 : n1;
            const int ld = second ? ld2 : ld1;
            const int64_t row = li / cpr;
            const int c = (int)(li - row * cpr);
            const int g = c & 7, k0 = 32 * (c >> 3) + 16 * (g >> 2) + 8 * ((g >> 1) & 1), lo = g & 1;
            uint32_t h[8];
            float x[8];
            if (row < n && k0 + 8 <= dim) {
                const float4 u = oea::ld4(src + row * ld + k0), w = oea::ld4(src + row * ld + k0 + 4);
                x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = w.x; x[5] = w.y; x[6] = w.z; x[7] = w.w;
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    x[t] = (row < n && k0 + t < dim) ? src[row * ld + k0 + t] : 0.f;
                    if (aug && row < n && k0 + t == dim) x[t] = second ? 1.0f : -csls_r[row];
                    if (aug && row < n && k0 + t == dim + 1) x[t] = second ? csls_c[row] : -1.0f;
                }
            }
            if (aug && !second) {
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (k0 + t < dim) x[t] *= 2.0f;                   // exact
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint32_t hi = bf16_rne(x[t]);
                h[t] = lo ? bf16_rne(x[t] - __uint_as_float(hi << 16)) : hi;
            }
            (second ? p2 : p1)[li] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
            continue;
        }
        if (i < g_end) {
            const int64_t row = i - b_end;
            if (row < n1) {
                const float *a = e1 + row * ld1, *b = e2 + (row + gold_off) * ld2;
                float acc = 0.f;
                int k = 0;
                for (; k + 4 <= dim; k += 4) {
                    const float4 x = oea::ld4(a + k), y = oea::ld4(b + k);
                    acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
                }
                for (; k < dim; ++k) acc = fmaf(a[k], b[k], acc);
                if (csls_r) acc = (2.0f * acc - csls_r[row]) - csls_c[row + gold_off];       // gold_inner_kernel's expression
                gold[row] = acc;
                rank[row] = 0;
            }
            continue;
        }
        const bool second = i >= m1_end;
        const int64_t li = second ? i - m1_end : i - g_end;
        const int64_t row = li >> 4, n = second ? n2 : n1;
        const int l16 = (int)(li & 15);
        float ss = 0.f;
        if (row < n) {
            const float *a = second ? e2 + row * ld2 : e1 + row * ld1;
            for (int k = l16; k < dim; k += 16) ss = fmaf(a[k], a[k], ss);
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        if (aug && row < n) {
            const float extra = second ? csls_c[row] : csls_r[row];
            ss = (second ? ss : 4.0f * ss) + fmaf(extra, extra, 1.0f);
        }
        float nr = sqrtf(ss) * 1.000001f;
        float cm = (second && csls_c && row < n && l16 == 0) ? fabsf(csls_c[row]) : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            nr = fmaxf(nr, __shfl_xor(nr, off, 64));
            cm = fmaxf(cm, __shfl_xor(cm, off, 64));
        }
        if ((threadIdx.x & 63) == 0 && cm > __uint_as_float(__hip_atomic_load(nmax + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
            atomicMax(nmax + 2, __float_as_uint(cm));
        // one atomic per wave, and only when it would raise the maximum (35,000 waves on two addresses took 0.3 ms)
        unsigned *dst = nmax + (second ? 1 : 0);
        if ((threadIdx.x & 63) == 0 && nr > __uint_as_float(__hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
            atomicMax(dst, __float_as_uint(nr));
    }
}

// argmax from the keys + Hits@k / sum(rank + 1) / sum 1 / (rank + 1) in a fixed order + the sweep's status, ONE block
__global__ __launch_bounds__(1024) void rank_bf16_finish_kernel(const int32_t *__restrict__ rank, const unsigned long long *__restrict__ best_key,
                                                                int64_t n, int4 tk0, int4 tk1, int nk, int32_t *__restrict__ argmax,
                                                                long long *__restrict__ out, const unsigned *__restrict__ rec_cnt) {
    __shared__ long long s_i[1024];
    __shared__ double s_d[1024];
    const int tks[8] = {tk0.x, tk0.y, tk0.z, tk0.w, tk1.x, tk1.y, tk1.z, tk1.w};
    long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rs = 0;
    double rr = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = rank[i];
        argmax[i] = (int32_t)(0xFFFFFFFFu - (uint32_t)(best_key[i] & 0xFFFFFFFFull));
#pragma unroll
        for (int k = 0; k < 8; ++k) h[k] += (k < nk && r < tks[k]);
        rs += r + 1;
        rr += 1.0 / (double)(r + 1);
    }
    // the reduction order of rank_metrics_kernel (wave butterflies, then the wave results in wave order): the same bits
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long v[9];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = h[k];
    v[8] = rs;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rr += __shfl_xor(rr, off, 64);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) s_i[wave * 9 + k] = v[k];
        s_d[wave] = rr;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        for (int k = 0; k < 9; ++k) {
            long long t = 0;
            for (int w = 0; w < nw; ++w) t += s_i[w * 9 + k];
            if (k < nk) out[k] = t;
            else if (k == 8) out[nk] = t;
        }
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += s_d[w];
        out[nk + 1] = __double_as_longlong(t);
        out[nk + 2] = rec_cnt[1];            // != 0: the record buffer overflowed, nothing above is valid
        out[nk + 3] = rec_cnt[0];
    }
}

// the records by the exact k-ordered fmaf chain (gold_inner_kernel's arithmetic): one thread per record, workgroups
// stride over the waves' slices
__global__ __launch_bounds__(256) void rank_bf16_fixup_kernel(const float *__restrict__ e1, int ld1, const float *__restrict__ e2, int ld2,
                                                              int dim, const float *__restrict__ gold, int64_t gold_off,
                                                              const float *__restrict__ csls_r, const float *__restrict__ csls_c,
                                                              const uint2 *__restrict__ rec, const unsigned *__restrict__ rec_cnt,
                                                              unsigned n_slices, unsigned slice_cap, int32_t *__restrict__ rank,
                                                              unsigned long long *__restrict__ best_key) {
    if (rec_cnt[0] == 0u) return;
    const int sub = threadIdx.x >> 6, lane = threadIdx.x & 63;          // one wave per slice at a time
    for (unsigned sl = blockIdx.x * 4u + sub; sl < n_slices; sl += gridDim.x * 4u) {
        const unsigned n = rec_cnt[2 + sl];
        for (unsigned p = lane; p < n; p += 64) {
            const uint2 rc = rec[(size_t)sl * slice_cap + p];
            const int64_t i = rc.x, j = rc.y & 0x7FFFFFFFu;
            const float *a = e1 + i * ld1, *b = e2 + j * ld2;
            float acc = 0.f;
            int k = 0;
            for (; k + 32 <= dim; k += 32) {                        // 16 loads in flight, then the chain in k order
                float4 x[8], y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { x[u] = oea::ld4(a + k + 4 * u); y[u] = oea::ld4(b + k + 4 * u); }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc = fmaf(x[u].x, y[u].x, acc); acc = fmaf(x[u].y, y[u].y, acc);
                    acc = fmaf(x[u].z, y[u].z, acc); acc = fmaf(x[u].w, y[u].w, acc);
                }
            }
            for (; k + 4 <= dim; k += 4) {
                const float4 x = oea::ld4(a + k), y = oea::ld4(b + k);
                acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
            }
            for (; k < dim; ++k) acc = fmaf(a[k], b[k], acc);
            if (csls_r) acc = fmaf(2.0f, acc, -csls_r[i]) - csls_c[j];                // rank_inner_kernel's expression
            if (rc.y & kRecTop) {
                atomicMax(best_key + i, ((unsigned long long)f2ord(acc) << 32) | (0xFFFFFFFFu - (uint32_t)j));
            } else {
                const float gi = gold[i];
                if (acc > gi || (acc == gi && j < i + gold_off)) atomicAdd(rank + i, 1);
            }
        }
    }
}

// the approximate similarities themselves (tests: the error bound; timing of the bare sweep)
__global__ __launch_bounds__(256, 2) void sim_bf16_store_kernel(const float *__restrict__ e1p, int64_t n1, int kp,
                                                               const float *__restrict__ e2p, int64_t n2, int dim,
                                                               float *__restrict__ out, int64_t ld_out) {
    __shared__ __attribute__((aligned(16))) float As[2 * TILE * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2 * TILE * LDS_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * TILE, c0 = (int64_t)blockIdx.x * TILE;
    tile_pipeline_bf16<true>(
        e1p, kp, e2p, dim, c0, 1, [=](int64_t) { return m0; }, As, Bs,
        [&](int64_t, f32x16 (&acc)[2][2]) {
            float *tile = out + m0 * ld_out + c0;
            const int rows_left = (int)(n1 - m0 < TILE ? n1 - m0 : TILE), cols_left = (int)(n2 - c0 < TILE ? n2 - c0 : TILE);
            const int col = wn * 64 + (lane & 31), rbase = wm * 64 + 4 * (lane >> 5);
            const bool ok0 = col < cols_left, ok1 = col + 32 < cols_left;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + tm * 32 + (r & 3) + 8 * (r >> 2);
                    if (row < rows_left) {
                        float *p = tile + (row * (int)ld_out + col);
                        if (ok0) p[0] = acc[tm][0][r];
                        if (ok1) p[32] = acc[tm][1][r];
                    }
                    asm volatile("" ::: "memory");
                }
        });
}

// workspace layout of oea_csls_means
// tol[0] = bound on |v~ - v| of the neighbour search's bf16 sweep from tol[1] = bits of the max row norm (as rank_bf16_init_kernel)
__global__ void knn_tol_kernel(float *tol, float eps_rel) {
    const float nm = tol[1], smax = nm * nm;
    tol[0] = 1.05f * eps_rel * smax + 1.0e-6f * smax + 1e-30f;
}

// tol[0] = bound on |v~ - v| of the CSLS means' bf16 sweep from tol[1], tol[2] = bits of the max row norms of the two tables
__global__ void csls_tol_kernel(float *tol, float eps_rel) {
    const float smax = tol[1] * tol[2];
    tol[0] = 1.05f * eps_rel * smax + 1.0e-6f * smax + 1e-30f;
}

struct CslsPlan {
    bool ok = false;
    int sample = 0, r1 = 0, r2 = 0, cap = 0, ccap = 0, chunks = 0, nseg = 0, tpc = 0, nqt = 0;
    int64_t ld1 = 0, ld2 = 0;
    size_t off_thr1, off_thr2, off_qcnt, off_ccnt, off_fail1, off_fail2, off_nfail, off_qlists, off_clists, off_strip, off_fbq,
        off_scratch, total;
};
constexpr int kCslsFb = 128, kCslsSlow = 64;

// big: the sweep runs on 256-candidate tiles with 8 waves (csls_append_kernel<.., 8>): chunks count those tiles, 8 segments per chunk
static CslsPlan plan_csls(int64_t n1, int64_t n2, int k, int big = 0 /* candidate rows of a big tile: 0, 256 or 192 */) {
    CslsPlan p;
    if (n1 < 4096 || n2 < 4096 || k > 32) return p;
    p.sample = std::max(n1, n2) >= 32768 ? 4096 : 1024;       // keeps r n / S, the survivors per row, in the low hundreds
    static const int env_sample = [] { const char *e = getenv("OEA_CSLS_SAMPLE"); return e ? atoi(e) : 0; }();       // experiments
    if (env_sample == 1024 || env_sample == 2048 || env_sample == 4096) p.sample = env_sample;
    auto rank_of = [&](int64_t n) { const double e = (double)k * p.sample / (double)n; return (int)(e + 3.5 * std::sqrt(e) + 8.0); };
    p.r1 = rank_of(n2);                       // thresholds of the rows of S (queries against sampled candidates)
    p.r2 = rank_of(n1);
    const double m1 = (double)p.r1 * n2 / p.sample, m2 = (double)p.r2 * n1 / p.sample;      // survivors per row / per column
    if (m1 * (1.0 + 5.0 / std::sqrt((double)p.r1)) > kMeanRegs * 64 || m2 * (1.0 + 5.0 / std::sqrt((double)p.r2)) > kMeanRegs * 64) return p;
    p.chunks = pick_chunks(oea::ceil_div(n1, TILE), oea::ceil_div(n2, big ? big : TILE), &p.tpc);
    p.nseg = (big ? big / 32 : 4) * p.chunks;
    if (p.nseg > kMeanSeg) return p;
    // the threshold is the r-th of a sample: the survivor count of a row scales with a factor of relative spread 1 / sqrt(r)
    // COMMON to its segments, on top of each segment's own sqrt(m) noise
    p.nqt = (int)oea::ceil_div(n1, TILE);
    if (2 * p.nqt > kMeanSeg) return p;
    const double ms = m1 / p.nseg, mc = m2 / (2 * p.nqt);        // column lists: one segment per (query tile, wave column)
    p.cap = ((int)(ms * (1.0 + 5.0 / std::sqrt((double)p.r1)) + 8.0 * std::sqrt(ms) + 16.0) + 7) / 8 * 8;
    p.ccap = ((int)(mc * (1.0 + 5.0 / std::sqrt((double)p.r2)) + 8.0 * std::sqrt(mc) + 8.0) + 3) / 4 * 4;
    if ((size_t)128 * p.nseg * p.cap * 8 >= ((size_t)1 << 32)) return p;
    p.ld1 = (n1 + 31) / 32 * 32;
    p.ld2 = (n2 + 31) / 32 * 32;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    p.off_thr1 = take(4 * (size_t)n1); p.off_thr2 = take(4 * (size_t)n2);
    p.off_qcnt = take(4 * (size_t)n1 * p.nseg); p.off_ccnt = take(4 * (size_t)n2 * 2 * p.nqt);
    p.off_fail1 = take(4 * (size_t)n1); p.off_fail2 = take(4 * (size_t)n2); p.off_nfail = take(256);
    p.off_qlists = take(8 * (size_t)n1 * p.nseg * p.cap);             // 8 B per entry: (value, index) pairs under the bf16 sweep
    p.off_clists = take(8 * (size_t)n2 * 2 * p.nqt * p.ccap);
    // sample strips; the fallback strip [kCslsFb, max ld] reuses the space after the thresholds are taken
    p.off_strip = take(4 * std::max<size_t>((size_t)std::max(n1, n2) * p.sample, (size_t)kCslsFb * std::max(p.ld1, p.ld2)));
    p.off_fbq = take(4 * (size_t)kCslsFb * 4096);
    p.off_scratch = take(4 * (size_t)kCslsSlow * std::max(p.ld1, p.ld2));
    p.total = off;
    p.ok = true;
    return p;
}

}  // namespace

static float bf16_eps_rel(int dim, bool k_blocks);
static int pack_operand_bf16(int slot, const float *src, int64_t n, int ld, int dim, hipStream_t st, PackedOp *out);

namespace oea {
// kNN strips (topk.hip): both operands packed once (slot 0 = queries, slot 1 = candidates), every strip produced from
// the packed copies; release_packed_rows() after the last strip
bool tile_glds_enabled() { return use_glds(); }
int pack_rows(int slot, const float *src, int64_t n, int ld, int dim, hipStream_t st, float **packed, int *kp) {
    PackedOp op;
    const int rc = pack_operand(slot, src, n, ld, dim, st, &op);
    *packed = op.p;
    *kp = op.kp;
    return rc;
}
int release_packed_rows(hipStream_t st) { return release_packed(st); }
void sim_inner_store_packed(const float *e1p, int64_t n1, const float *e2p, int64_t n2, int kp, int dim, float *out,
                            int64_t ld_out, hipStream_t st) {
    launch_store_packed(e1p, n1, e2p, n2, kp, dim, out, ld_out, st);
}
// strip-free neighbour search: survivors of the threshold sweep into per-query list segments (see topk_append_kernel);
// -> number of segments per query (4 * chunks); chunks is chosen here so that the grid fills the chip
void sim_inner_store_packed_gated(const float *e1p, int64_t n1, const float *e2p, int64_t n2, int kp, int dim, float *out,
                                  int64_t ld_out, const int32_t *gate, hipStream_t st) {
    sim_inner_store_kernel<true><<<dim3((unsigned)ceil_div(n2, TILE), (unsigned)ceil_div(n1, TILE)), 256, 0, st>>>(
        e1p, n1, kp, e2p, n2, kp, dim, out, ld_out, gate);
}
int topk_append_chunks(int64_t nq, int64_t nc) {
    int tpc;
    return pick_chunks(ceil_div(nq, TILE), ceil_div(nc, TILE), &tpc);
}
void topk_append_sym_packed(const float *ep, int64_t n, int kp, int dim, const float *thr, const void *items, int n_items, int nseg,
                            int cap, float *list_vals, int32_t *list_cols, int32_t *counts, int T, int ccap, void *clists,
                            uint8_t *ccounts, int32_t *spill_cnt, void *spill, int sp_cap, hipStream_t st) {
    topk_append_sym_kernel<true, false><<<(unsigned)n_items, 256, 0, st>>>(ep, n, kp, dim, thr, static_cast<const int4 *>(items), nseg,
                                                                           cap, list_vals, list_cols, counts, T, ccap,
                                                                           static_cast<uint2 *>(clists), ccounts, spill_cnt,
                                                                           static_cast<uint2 *>(spill), sp_cap, nullptr);
}
// stream form (topk_stream_sym_kernel): rows split and packed into slot 3, the bound into tol_dev[0], one launch
int topk_stream_sym_bf16(const float *src, int64_t n, int ld, int dim, const float *thr, const void *items, int n_items, void *row_streams,
                         int rcap, void *col_streams, int ccap, int32_t *row_cnt, int32_t *col_off, int lp1, uint8_t *row_fail,
                         float *tol_dev, void *ovf_pool, int32_t *ovf_alloc, int32_t *ovf_len, int ovf_chunks, int32_t *redo_cnt,
                         void *redo, int redo_cap, hipStream_t st, bool packed) {
    PackedOp op;
    int64_t n_pad_unused = 0;
    const int rc = packed ? reserve_operand(3, n, dim, st, &op, &n_pad_unused) : pack_operand_bf16(3, src, n, ld, dim, st, &op);   // packed: by sample_strip_bf16
    if (rc != OEA_OK) return rc;
    OEA_CHECK_HIP(hipMemsetAsync(tol_dev, 0, 8, st));
    row_norm_max_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(src, n, ld, dim, reinterpret_cast<unsigned *>(tol_dev) + 1);
    knn_tol_kernel<<<1, 1, 0, st>>>(tol_dev, bf16_eps_rel(dim, false));
    static const int nch_env = [] { const char *e = getenv("OEA_TOPK_STREAM_NCH"); return e ? atoi(e) : 1; }();
    // OEA_TOPK_STREAM_FAST=0: the branching epilogue (stream_tile) on every tile -- the ablation of stream_tile_fast
    static const int fast_env = [] { const char *e = getenv("OEA_TOPK_STREAM_FAST"); return (e && e[0] == '0') ? 0 : 1; }();
#define OEA_STREAM_LAUNCH(N, F)                                                                                                       \
    topk_stream_sym_kernel<N, F><<<(unsigned)n_items, 256, 0, st>>>(op.p, n, op.kp, dim, thr, static_cast<const int4 *>(items),         \
                                                                 static_cast<uint2 *>(row_streams), rcap, static_cast<uint2 *>(col_streams), \
                                                                 ccap, row_cnt, col_off, lp1, row_fail, tol_dev, redo_cnt,              \
                                                                 static_cast<int4 *>(redo), redo_cap, fast_env)
    // the query operand in registers (tile_pipeline_bf16_breg: the candidate stages alone travel through LDS, nothing is re-sent per chunk)
    // for 64 < dim <= 128 when the branch-free epilogue is compiled in alone: with the branching one the kernel spilled (13.3 -> 20.8 ms),
    // without it it fits (20 B of scratch) -- 11.3 -> 10.5 ms at 100,000^2 x 100.  OEA_TOPK_STREAM_NCH=0: both operands through LDS
    const bool breg = nch_env != 0 && fast_env;
    if (breg && op.kp == 128) OEA_STREAM_LAUNCH(4, true);
    else if (breg && op.kp == 96) OEA_STREAM_LAUNCH(3, true);
    else if (fast_env) OEA_STREAM_LAUNCH(0, true);
    else OEA_STREAM_LAUNCH(0, false);
#undef OEA_STREAM_LAUNCH
    topk_stream_redo_kernel<<<2048, 256, 0, st>>>(op.p, n, op.kp, dim, thr, static_cast<const int4 *>(items), rcap, ccap, row_fail, tol_dev,
                                                  redo_cnt, static_cast<const int4 *>(redo), redo_cap, static_cast<uint4 *>(ovf_pool), ovf_alloc,
                                                  ovf_len, ovf_chunks);
    return OEA_OK;
}
// the same sweep on the hi / lo split rows of `src` (packed here into slot 3); tol_dev[0] receives the bound on |v~ - v|
// (max row norm^2 x eps(dim), computed on the device) that the sweep cuts by and the select resolves with
int topk_append_sym_bf16(const float *src, int64_t n, int ld, int dim, const float *thr, const void *items, int n_items, int nseg,
                         int cap, float *list_vals, int32_t *list_cols, int32_t *counts, int T, int ccap, void *clists,
                         uint8_t *ccounts, int32_t *spill_cnt, void *spill, int sp_cap, float *tol_dev, hipStream_t st) {
    PackedOp op;
    const int rc = pack_operand_bf16(3, src, n, ld, dim, st, &op);
    if (rc != OEA_OK) return rc;
    OEA_CHECK_HIP(hipMemsetAsync(tol_dev, 0, 8, st));
    row_norm_max_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, st>>>(src, n, ld, dim, reinterpret_cast<unsigned *>(tol_dev) + 1);
    knn_tol_kernel<<<1, 1, 0, st>>>(tol_dev, bf16_eps_rel(dim, false));
    topk_append_sym_kernel<true, true><<<(unsigned)n_items, 256, 0, st>>>(op.p, n, op.kp, dim, thr, static_cast<const int4 *>(items), nseg,
                                                                          cap, list_vals, list_cols, counts, T, ccap,
                                                                          static_cast<uint2 *>(clists), ccounts, spill_cnt,
                                                                          static_cast<uint2 *>(spill), sp_cap, tol_dev);
    return OEA_OK;
}
void topk_append_packed(const float *qp, int64_t nq, const float *cp, int64_t nc, int kp, int dim, const float *thr, int cap,
                        int chunks, float *list_vals, int32_t *list_cols, int32_t *counts, int32_t *spill_cnt, void *spill, int sp_cap,
                        hipStream_t st) {
    const int tpc = (int)ceil_div(ceil_div(nc, TILE), chunks);       // chunks planned by topk_append_chunks
    topk_append_kernel<true><<<dim3((unsigned)ceil_div(nq, TILE), (unsigned)chunks), 256, 0, st>>>(
        qp, nq, kp, cp, nc, kp, dim, thr, tpc, cap, list_vals, list_cols, counts, spill_cnt, static_cast<uint2 *>(spill), sp_cap);
}
// the neighbour search's sample strip on the bf16 split (round 6; the CSLS means' strips since round 5): a threshold is an ESTIMATE of where
// the k-th value lies -- the lists are cut below it by the bound and a row whose list comes out short or long takes the exact fallback
// either way.  Packs the query rows (slot 3: the sweep that follows finds them there) and every stride-th candidate row (slot 2).
int sample_strip_bf16_pack(const float *q, int64_t nq, int ldq, const float *c, int64_t n_sample, int ld_sample, int dim, hipStream_t st,
                           const float **qs, const float **ss, int *kp) {
    PackedOp pq, ps;
    int rc = pack_operand_bf16(3, q, nq, ldq, dim, st, &pq);
    if (rc == OEA_OK) rc = pack_operand_bf16(2, c, n_sample, ld_sample, dim, st, &ps);
    if (rc != OEA_OK) return rc;
    *qs = pq.p; *ss = ps.p; *kp = pq.kp;
    return OEA_OK;
}
void sample_strip_bf16_launch(const float *qs, int64_t rows, const float *ss, int64_t n_sample, int kp, int dim, float *strip, hipStream_t st) {
    sim_bf16_store_kernel<<<dim3((unsigned)ceil_div(n_sample, TILE), (unsigned)ceil_div(rows, TILE)), 256, 0, st>>>(qs, rows, kp, ss, n_sample, dim,
                                                                                                                   strip, n_sample);
}
// the same on the bf16 hi / lo split of both tables (queries != candidates): _prepare packs them (slots 3 / 4) and leaves the bound on
// |v~ - v| in tol_dev[0] (max row norms of the two tables in tol_dev[1], [2]); _launch sweeps a block of query rows
int topk_append_bf16_prepare(const float *q, int64_t nq, int ldq, const float *c, int64_t nc, int ldc, int dim, float *tol_dev,
                             hipStream_t st, const float **qs, const float **cs, int *kp, bool q_packed) {
    PackedOp pq, pc;
    int64_t n_pad_unused = 0;
    int rc = q_packed ? reserve_operand(3, nq, dim, st, &pq, &n_pad_unused) : pack_operand_bf16(3, q, nq, ldq, dim, st, &pq);
    if (rc == OEA_OK) rc = pack_operand_bf16(4, c, nc, ldc, dim, st, &pc);
    if (rc != OEA_OK) return rc;
    OEA_CHECK_HIP(hipMemsetAsync(tol_dev, 0, 16, st));
    row_norm_max_kernel<<<(unsigned)ceil_div(nq, 256), 256, 0, st>>>(q, nq, ldq, dim, reinterpret_cast<unsigned *>(tol_dev) + 1);
    row_norm_max_kernel<<<(unsigned)ceil_div(nc, 256), 256, 0, st>>>(c, nc, ldc, dim, reinterpret_cast<unsigned *>(tol_dev) + 2);
    csls_tol_kernel<<<1, 1, 0, st>>>(tol_dev, bf16_eps_rel(dim, false));
    *qs = pq.p; *cs = pc.p; *kp = pq.kp;
    return OEA_OK;
}
void topk_append_bf16_launch(const float *qs, int64_t nq, const float *cs, int64_t nc, int kp, int dim, const float *thr, int cap,
                             int chunks, float *list_vals, int32_t *list_cols, int32_t *counts, int32_t *spill_cnt, void *spill,
                             int sp_cap, const float *tol_dev, hipStream_t st) {
    const int tpc = (int)ceil_div(ceil_div(nc, TILE), chunks);
    static const bool breg = [] { const char *e = getenv("OEA_TOPK_STREAM_NCH"); return !(e && e[0] == '0'); }();
    const dim3 grid((unsigned)ceil_div(nq, TILE), (unsigned)chunks);
#define OEA_APPEND_LAUNCH(N)                                                                                                             \
    topk_append_kernel<true, true, N><<<grid, 256, 0, st>>>(qs, nq, kp, cs, nc, kp, dim, thr, tpc, cap, list_vals, list_cols, counts, spill_cnt, \
                                                            static_cast<uint2 *>(spill), sp_cap, tol_dev)
    if (breg && kp == 128) OEA_APPEND_LAUNCH(4);
    else if (breg && kp == 96) OEA_APPEND_LAUNCH(3);
    else OEA_APPEND_LAUNCH(0);
#undef OEA_APPEND_LAUNCH
}
}  // namespace oea

extern "C" {

size_t oea_rank_workspace_bytes(int64_t n1) {
    return (size_t)n1 * (sizeof(float) + sizeof(unsigned long long)) + 256;
}

int oea_rank_eval(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2,
                  int32_t dim, int32_t metric, const float *csls_r, const float *csls_c, int64_t gold_offset,
                  int32_t *rank, int32_t *argmax, void *workspace, void *stream) {
    OEA_REQUIRE(e1 && e2 && rank && argmax && workspace, "null pointer");
    OEA_REQUIRE(n1 >= 0 && gold_offset >= 0 && n1 + gold_offset <= n2, "gold of row i is column gold_offset + i <= n2");
    OEA_REQUIRE(ld1 % 4 == 0 && ld2 % 4 == 0 && dim > 0 && dim <= ld1 && dim <= ld2, "ld % 4 == 0, dim <= ld");
    OEA_REQUIRE((csls_r == nullptr) == (csls_c == nullptr), "csls_r and csls_c go together");
    OEA_REQUIRE(n2 < 0x7fffffff, "n2 < 2^31");
    if (n1 == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
    unsigned long long *keys = static_cast<unsigned long long *>(workspace);
    float *gold = reinterpret_cast<float *>(keys + n1);
    OEA_CHECK_HIP(hipMemsetAsync(keys, 0, sizeof(unsigned long long) * (size_t)n1, st));
    OEA_CHECK_HIP(hipMemsetAsync(rank, 0, sizeof(int32_t) * (size_t)n1, st));
    const unsigned gb = (unsigned)oea::ceil_div(n1, 256);
    int tpc = 1;
    if (metric == OEA_METRIC_INNER) {
        gold_inner_kernel<<<(unsigned)oea::ceil_div(n1, 64), 64, 0, st>>>(e1, n1, ld1, e2 + gold_offset * ld2, ld2, dim, csls_r,
                                                                          csls_c ? csls_c + gold_offset : nullptr, gold);
        const int64_t qt = oea::ceil_div(n1, TILE), ctiles = oea::ceil_div(n2, TILE);
        const int chunks = pick_chunks(qt, ctiles, &tpc);
        const dim3 grid((unsigned)qt, (unsigned)chunks);
        if (use_glds()) {
            PackedOp p1, p2;
            int rc = pack_operand(0, e1, n1, ld1, dim, st, &p1);
            if (rc == OEA_OK) rc = pack_operand(1, e2, n2, ld2, dim, st, &p2);
            if (rc != OEA_OK) return rc;
            if (csls_r)
                rank_inner_kernel<true, true><<<grid, 256, 0, st>>>(p1.p, n1, p1.kp, p2.p, n2, p2.kp, dim, gold, csls_r, csls_c, tpc,
                                                                    gold_offset, rank, keys, EvalTail{});
            else
                rank_inner_kernel<false, true><<<grid, 256, 0, st>>>(p1.p, n1, p1.kp, p2.p, n2, p2.kp, dim, gold, csls_r, csls_c, tpc,
                                                                     gold_offset, rank, keys, EvalTail{});
            rc = release_packed(st);
            if (rc != OEA_OK) return rc;
        } else if (csls_r) {
            rank_inner_kernel<true, false><<<grid, 256, 0, st>>>(e1, n1, ld1, e2, n2, ld2, dim, gold, csls_r, csls_c, tpc, gold_offset,
                                                                 rank, keys, EvalTail{});
        } else {
            rank_inner_kernel<false, false><<<grid, 256, 0, st>>>(e1, n1, ld1, e2, n2, ld2, dim, gold, csls_r, csls_c, tpc, gold_offset,
                                                                  rank, keys, EvalTail{});
        }
    } else if (metric == OEA_METRIC_MANHATTAN || metric == OEA_METRIC_EUCLIDEAN) {
        const int64_t qt = oea::ceil_div(n1, VT), ctiles = oea::ceil_div(n2, VT);
        const int chunks = pick_chunks(qt, ctiles, &tpc);
        if (metric == OEA_METRIC_MANHATTAN) {
            gold_valu_kernel<OEA_METRIC_MANHATTAN><<<gb, 256, 0, st>>>(e1, n1, ld1, e2 + gold_offset * ld2, ld2, dim, csls_r, csls_c ? csls_c + gold_offset : nullptr, gold);
            rank_valu_kernel<OEA_METRIC_MANHATTAN><<<dim3((unsigned)qt, (unsigned)chunks), 256, 0, st>>>(
                e1, n1, ld1, e2, n2, ld2, dim, gold, csls_r, csls_c, tpc, gold_offset, rank, keys);
        } else {
            gold_valu_kernel<OEA_METRIC_EUCLIDEAN><<<gb, 256, 0, st>>>(e1, n1, ld1, e2 + gold_offset * ld2, ld2, dim, csls_r, csls_c ? csls_c + gold_offset : nullptr, gold);
            rank_valu_kernel<OEA_METRIC_EUCLIDEAN><<<dim3((unsigned)qt, (unsigned)chunks), 256, 0, st>>>(
                e1, n1, ld1, e2, n2, ld2, dim, gold, csls_r, csls_c, tpc, gold_offset, rank, keys);
        }
    } else {
        oea::set_error("unknown metric %d", metric);
        return OEA_EINVAL;
    }
    rank_finalize_kernel<<<gb, 256, 0, st>>>(keys, n1, argmax);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

// ---- inner-product evaluation through the certified bf16 prefilter (see rank_bf16_kernel) --------------------------------------
// record slots: the band around the gold value grows with the tolerance, i.e. with dim (measured on rows whose gold has a
// random rank: 30 records per row at dim 100, 114 at dim 300)
static unsigned bf16_rec_cap(int64_t n1, int dim) { return (unsigned)std::min<int64_t>((96 + (int64_t)dim) * n1 + (1 << 20), (int64_t)1 << 30); }
// an upper bound of the sweep's wave count: pick_chunks gives at most target / q_tiles + 1 chunks per query tile
static int64_t bf16_max_waves(int64_t n1) { return 8 * (oea::ceil_div(n1, TILE) + 16384); }

size_t oea_rank_eval_bf16_workspace_bytes(int64_t n1, int32_t dim) {
    auto a256 = [](size_t x) { return (x + 255) / 256 * 256; };
    return a256(8 * (size_t)n1) + 2 * a256(4 * (size_t)n1) + 256 + a256(4 * (size_t)(2 + bf16_max_waves(n1))) + 8 * (size_t)bf16_rec_cap(n1, dim);
}

static float bf16_eps_rel(int dim, bool k_blocks) {
    const int kp16 = (dim + 15) / 16 * 16;
    // (1) split residue: 3.02 * 2^-18 of |a||b|;
    // (2) fp32 accumulation of the products, bounded term by term with a chopping unit roundoff 2^-23 (the MFMA's internal order
    //     and rounding are not documented): the 3 * min(Kp, 128) products of ONE 128-k block (tile_pipeline_bf16 restarts the
    //     accumulators per block from 5 chunks on; the B-in-registers form only exists for Kp <= 128 = one block) + one rounding
    //     per block sum added;
    // (3) the exact chain's own roundings: fmaf rounds to nearest, <= 2^-24 of a partial sum <= |a||b| per step
    // k_blocks = false (the neighbour sweeps: tile_pipeline_bf16<false>): one accumulation chain over all 3 * Kp products
    const int blocks = k_blocks ? (kp16 + 127) / 128 : 1;
    const int n_acc = blocks > 1 ? 3 * 128 + blocks : 3 * kp16;
    return 1.02f * (3.02f * 3.814697265625e-06f + (float)n_acc * 1.1920928955078125e-07f + (float)(dim + 8) * 5.9604644775390625e-08f);
}

static int pack_operand_bf16(int slot, const float *src, int64_t n, int ld, int dim, hipStream_t st, PackedOp *out) {
    int64_t n_pad = 0;
    const int rc = reserve_operand(slot, n, dim, st, out, &n_pad);
    if (rc != OEA_OK) return rc;
    const int64_t total = n_pad * (out->kp / 4);
    pack_rows_bf16_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(total, 256), 16384), 256, 0, st>>>(
        src, n, ld, dim, reinterpret_cast<uint4 *>(out->p), n_pad, out->kp);
    return OEA_OK;
}

constexpr int kBf16WarmTiles = 16;          // the warm-up pass sees 2,048 candidates

// prologue | init | warm-up | sweep | fix-up | finish: six launches; metrics_out (may be NULL) = int64 [nk + 4] on the device
static int rank_eval_bf16_impl(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2, int32_t dim,
                               const float *csls_r, const float *csls_c, int64_t gold_offset, const int32_t *top_k_host, int32_t nk,
                               int32_t *rank, int32_t *argmax, long long *metrics_out, int32_t *status, void *workspace, hipStream_t st) {
    auto a256 = [](size_t x) { return (x + 255) / 256 * 256; };
    char *w = static_cast<char *>(workspace);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(w);
    float *gold = reinterpret_cast<float *>(w + a256(8 * (size_t)n1));
    unsigned *lbrow = reinterpret_cast<unsigned *>(w + a256(8 * (size_t)n1) + a256(4 * (size_t)n1));
    char *sc = w + a256(8 * (size_t)n1) + 2 * a256(4 * (size_t)n1);
    unsigned *nmax = reinterpret_cast<unsigned *>(sc);              // [2]: max row norms of e1 / e2 (float bits)
    float *tol = reinterpret_cast<float *>(sc + 16);
    long long *fin = reinterpret_cast<long long *>(sc + 64);        // [nk + 4] when the caller wants no metrics
    unsigned *rec_cnt = reinterpret_cast<unsigned *>(sc + 256);     // [0] records, [1] overflow flag, [2 + w] length of wave w's slice
    uint2 *rec = reinterpret_cast<uint2 *>(sc + 256 + a256(4 * (size_t)(2 + bf16_max_waves(n1))));
    const unsigned cap = bf16_rec_cap(n1, dim);
    OEA_CHECK_HIP(hipMemsetAsync(sc, 0, 64, st));                   // norms, tolerance
    PackedOp p1, p2;
    int64_t n1_pad = 0, n2_pad = 0;
    // CSLS means: the rows are packed with two more coordinates (see the prologue) when that costs no k chunk, and the sweep is
    // the plain one.  Its bound: split + accumulation over dim + 2 coordinates on the augmented norms, + the roundings of the
    // exact expression fl(fl(2 s - r) - c) it is compared with (<= 2 * 2^-24 * (2 |s| + |r| + |c|) <= 3.6e-7 * |q'|max |c'|max,
    // the chain's own error inside eps(dim + 2)); OEA_CSLS_AUG=0: the CSLS terms in the sweep's epilogue
    static const bool aug_on = [] { const char *e = getenv("OEA_CSLS_AUG"); return !(e && e[0] == '0'); }();
    const bool aug = csls_r && aug_on && (dim + 2 + BK - 1) / BK == (dim + BK - 1) / BK;
    const int dim_p = aug ? dim + 2 : dim;
    const float *sweep_r = aug ? nullptr : csls_r, *sweep_c = aug ? nullptr : csls_c;
    int rc = reserve_operand(0, n1, dim_p, st, &p1, &n1_pad);
    if (rc == OEA_OK) rc = reserve_operand(1, n2, dim_p, st, &p2, &n2_pad);
    if (rc != OEA_OK) return rc;
    const int64_t items = (n1_pad + n2_pad) * (p1.kp / 4) + 17 * (n1 + n2) + 256;
    rank_bf16_prologue_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(items, 256), 16384), 256, 0, st>>>(
        e1, n1, ld1, e2, n2, ld2, dim, gold_offset, reinterpret_cast<uint4 *>(p1.p), n1_pad, reinterpret_cast<uint4 *>(p2.p), n2_pad,
        p1.kp, gold, nmax, rank, csls_r, csls_c, aug ? 1 : 0);
    const unsigned gb = (unsigned)oea::ceil_div(n1, 256);
    rank_bf16_init_kernel<<<gb, 256, 0, st>>>(gold, n1, gold_offset, nmax, aug ? bf16_eps_rel(dim_p, true) + 1.0e-6f : bf16_eps_rel(dim, true), sweep_r, tol,
                                              rank, keys, lbrow, rec_cnt);
    int tpc = 1;
    const int64_t qt = oea::ceil_div(n1, TILE), ctiles = oea::ceil_div(n2, TILE);
    // Kp > 128: 256-candidate tiles, 512 threads, three-stage ring (tile_pipeline_bf16_big); OEA_BF16_BIG=0: the 128 x 128 kernel
    static const bool big_on = [] { const char *e = getenv("OEA_BF16_BIG"); return !(e && e[0] == '0'); }();
    const bool big = big_on && p1.kp > 128;
    const int nw = big ? 8 : 4;
    // OEA_BF16_BIG_MODE: 1 (default) = all 8 waves multiply, fragment reads one MFMA group ahead; 2 = 6 multiplying + 2 loading waves on
    // 192-candidate tiles (measured equal: 35.3 against 34.7 ms at 70,000^2 x 1,200); 0 = the first form; the
    // CSLS-terms-in-the-epilogue variant (rare) keeps mode 0
    static const int big_mode_env = [] { const char *e = getenv("OEA_BF16_BIG_MODE"); return e ? atoi(e) : 1; }();
    const int big_mode = sweep_r ? 0 : big_mode_env;
    const int big_mt = big_mode == 2 ? SPEC_MT : BIG_MT;
    const int chunks = pick_chunks(qt, big ? oea::ceil_div(n2, big_mt) : ctiles, &tpc);
    const int64_t n_waves = (int64_t)nw * qt * chunks;
    const unsigned slice_cap = (unsigned)(cap / std::max<int64_t>(n_waves, 1));
    if (n_waves > bf16_max_waves(n1) || slice_cap < 32) {
        // more workgroups than the workspace was sized for (OEA_RANK_WGS raised) or record slices too small: reported like a record
        // overflow -- the caller takes the fp32 entry point (this used to be a hard error)
        rc = release_packed(st);
        if (rc != OEA_OK) return rc;
        if (status) OEA_CHECK_HIP(hipMemsetAsync(status, 1, 2 * sizeof(int32_t), st));
        if (metrics_out) OEA_CHECK_HIP(hipMemsetAsync(metrics_out, 1, sizeof(long long) * (size_t)(nk + 4), st));
        return OEA_OK;
    }
    // warm-up over 1/16 of the candidate tiles (at most 16 = 2,048 candidates, 3 % of a 70,000-row sweep); a small candidate
    // set needs none: every workgroup sees most of it anyway
    const int warm = (int)std::min<int64_t>(kBf16WarmTiles, ctiles / 16);
    const TileGrid gw = make_tile_grid((unsigned)qt, 1), gs = make_tile_grid((unsigned)qt, (unsigned)chunks);
    // Kp <= 128 (dim <= 128): the candidates' operand stays in registers (OEA_BF16_BREG=0: both operands through LDS)
    static const bool breg_on = [] { const char *e = getenv("OEA_BF16_BREG"); return !(e && e[0] == '0'); }();
    // (with the CSLS terms the epilogue's registers + 32 NCH of B spill from NCH = 3 on: 6.5 -> 7.6 ms, stays on the LDS pipeline)
    const int nch = (breg_on && p1.kp <= (sweep_r ? 64 : 128)) ? p1.kp / 32 : 0;
#define OEA_BF16_LAUNCH(K, GRID, TPC) K<<<tile_grid_blocks(GRID), 256, 0, st>>>(p1.p, n1, p1.kp, p2.p, n2, dim_p, gold, tol, sweep_r, sweep_c, TPC, \
                                                                                gold_offset, rank, lbrow, rec, rec_cnt, slice_cap, GRID)
#define OEA_BF16_SWEEP(W, C, GRID, TPC)                                                                 \
    do {                                                                                                \
        if (nch == 4) OEA_BF16_LAUNCH((rank_bf16_breg_kernel<W, C, 4>), GRID, TPC);                     \
        else if (nch == 3) OEA_BF16_LAUNCH((rank_bf16_breg_kernel<W, C, 3>), GRID, TPC);                \
        else if (nch == 2) OEA_BF16_LAUNCH((rank_bf16_breg_kernel<W, C, 2>), GRID, TPC);                \
        else if (nch == 1) OEA_BF16_LAUNCH((rank_bf16_breg_kernel<W, C, 1>), GRID, TPC);                \
        else OEA_BF16_LAUNCH((rank_bf16_kernel<W, C>), GRID, TPC);                                      \
    } while (0)
#define OEA_BF16_BIG_SWEEP(C, M)                                                                                                     \
    do {                                                                                                                            \
        /* the attribute belongs to (function, DEVICE): set on every call (cheap), not once per process */             \
        OEA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(rank_bf16_big_kernel<C, M>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          BIG_LDS_BYTES));                                                                          \
        rank_bf16_big_kernel<C, M><<<tile_grid_blocks(gs), 512, BIG_LDS_BYTES, st>>>(p1.p, n1, p1.kp, p2.p, n2, dim_p, gold, tol, sweep_r, sweep_c, tpc, \
                                                                                     gold_offset, rank, lbrow, rec, rec_cnt, slice_cap, gs);    \
        OEA_CHECK_HIP(hipGetLastError());                                                                                           \
    } while (0)
    if (sweep_r) {
        if (warm >= 2) OEA_BF16_SWEEP(true, true, gw, warm);
        if (big) OEA_BF16_BIG_SWEEP(true, 0);          // (CSLS terms in the epilogue -- only when the two extra coordinates would cost a
        else OEA_BF16_SWEEP(false, true, gs, tpc);     //  chunk: the prefetched forms spill there)
    } else {
        if (warm >= 2) OEA_BF16_SWEEP(true, false, gw, warm);
        if (big && big_mode == 2) OEA_BF16_BIG_SWEEP(false, 2);
        else if (big && big_mode == 1) OEA_BF16_BIG_SWEEP(false, 1);
        else if (big) OEA_BF16_BIG_SWEEP(false, 0);
        else OEA_BF16_SWEEP(false, false, gs, tpc);
    }
#undef OEA_BF16_BIG_SWEEP
#undef OEA_BF16_SWEEP
#undef OEA_BF16_LAUNCH
    rc = release_packed(st);
    if (rc != OEA_OK) return rc;
    rank_bf16_fixup_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(n_waves, 4), 2048), 256, 0, st>>>(
        e1, ld1, e2, ld2, dim, gold, gold_offset, csls_r, csls_c, rec, rec_cnt, (unsigned)n_waves, slice_cap, rank, keys);
    int t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < nk; ++i) t[i] = top_k_host[i];
    long long *out = metrics_out ? metrics_out : fin;
    rank_bf16_finish_kernel<<<1, 1024, 0, st>>>(rank, keys, n1, make_int4(t[0], t[1], t[2], t[3]), make_int4(t[4], t[5], t[6], t[7]), nk,
                                                argmax, out, rec_cnt);
    if (status) {
        OEA_CHECK_HIP(hipMemcpyAsync(status, rec_cnt + 1, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        OEA_CHECK_HIP(hipMemcpyAsync(status + 1, rec_cnt, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    }
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_rank_eval_bf16(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2, int32_t dim,
                       int64_t gold_offset, int32_t *rank, int32_t *argmax, int32_t *status, void *workspace, void *stream) {
    OEA_REQUIRE(e1 && e2 && rank && argmax && status && workspace, "null pointer");
    OEA_REQUIRE(n1 >= 0 && gold_offset >= 0 && n1 + gold_offset <= n2, "gold of row i is column gold_offset + i <= n2");
    OEA_REQUIRE(ld1 % 4 == 0 && ld2 % 4 == 0 && dim > 0 && dim <= ld1 && dim <= ld2, "ld % 4 == 0, dim <= ld");
    OEA_REQUIRE(n2 < 0x7fffffff && n1 < 0x7fffffff, "n < 2^31");
    OEA_REQUIRE(use_glds(), "the bf16 prefilter runs on the packed (LDS-DMA) tile path");
    hipStream_t st = oea::as_stream(stream);
    if (n1 == 0) { OEA_CHECK_HIP(hipMemsetAsync(status, 0, 2 * sizeof(int32_t), st)); return OEA_OK; }
    return rank_eval_bf16_impl(e1, n1, ld1, e2, n2, ld2, dim, nullptr, nullptr, gold_offset, nullptr, 0, rank, argmax, nullptr, status,
                               workspace, st);
}

int oea_rank_eval_bf16_csls(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2, int32_t dim,
                            const float *csls_r, const float *csls_c, int64_t gold_offset, int32_t *rank, int32_t *argmax, int32_t *status,
                            void *workspace, void *stream) {
    OEA_REQUIRE(e1 && e2 && rank && argmax && status && workspace, "null pointer");
    OEA_REQUIRE((csls_r == nullptr) == (csls_c == nullptr), "csls_r and csls_c: both or neither");
    OEA_REQUIRE(n1 >= 0 && gold_offset >= 0 && n1 + gold_offset <= n2, "gold of row i is column gold_offset + i <= n2");
    OEA_REQUIRE(ld1 % 4 == 0 && ld2 % 4 == 0 && dim > 0 && dim <= ld1 && dim <= ld2, "ld % 4 == 0, dim <= ld");
    OEA_REQUIRE(n2 < 0x7fffffff && n1 < 0x7fffffff, "n < 2^31");
    OEA_REQUIRE(use_glds(), "the bf16 prefilter runs on the packed (LDS-DMA) tile path");
    hipStream_t st = oea::as_stream(stream);
    if (n1 == 0) { OEA_CHECK_HIP(hipMemsetAsync(status, 0, 2 * sizeof(int32_t), st)); return OEA_OK; }
    return rank_eval_bf16_impl(e1, n1, ld1, e2, n2, ld2, dim, csls_r, csls_c, gold_offset, nullptr, 0, rank, argmax, nullptr, status,
                               workspace, st);
}

int oea_rank_eval_metrics_bf16(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2, int32_t dim,
                               const float *csls_r, const float *csls_c, int64_t gold_offset, const int32_t *top_k_host, int32_t nk,
                               int32_t *rank, int32_t *argmax, int64_t *out_dev, void *workspace, void *stream) {
    OEA_REQUIRE(e1 && e2 && rank && argmax && workspace && top_k_host && out_dev, "null pointer");
    OEA_REQUIRE((csls_r == nullptr) == (csls_c == nullptr), "csls_r and csls_c go together");
    OEA_REQUIRE(n1 > 0 && gold_offset >= 0 && n1 + gold_offset <= n2, "gold of row i is column gold_offset + i <= n2");
    OEA_REQUIRE(ld1 % 4 == 0 && ld2 % 4 == 0 && dim > 0 && dim <= ld1 && dim <= ld2, "ld % 4 == 0, dim <= ld");
    OEA_REQUIRE(n2 < 0x7fffffff && n1 < 0x7fffffff && nk >= 1 && nk <= 8, "n < 2^31, 1 <= len(top_k) <= 8");
    OEA_REQUIRE(use_glds(), "the bf16 prefilter runs on the packed (LDS-DMA) tile path");
    return rank_eval_bf16_impl(e1, n1, ld1, e2, n2, ld2, dim, csls_r, csls_c, gold_offset, top_k_host, nk, rank, argmax,
                               reinterpret_cast<long long *>(out_dev), nullptr, workspace, oea::as_stream(stream));
}

int oea_sim_bf16_matrix(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2, int32_t dim, float *out,
                        int64_t ld_out, void *stream) {
    OEA_REQUIRE(e1 && e2 && out && ld_out >= n2 && dim > 0 && dim <= ld1 && dim <= ld2 && ld1 % 4 == 0 && ld2 % 4 == 0, "arguments");
    OEA_REQUIRE(use_glds(), "packed tile path only");
    if (n1 == 0 || n2 == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
    PackedOp p1, p2;
    int rc = pack_operand_bf16(0, e1, n1, ld1, dim, st, &p1);
    if (rc == OEA_OK) rc = pack_operand_bf16(1, e2, n2, ld2, dim, st, &p2);
    if (rc != OEA_OK) return rc;
    sim_bf16_store_kernel<<<dim3((unsigned)oea::ceil_div(n2, TILE), (unsigned)oea::ceil_div(n1, TILE)), 256, 0, st>>>(p1.p, n1, p1.kp, p2.p,
                                                                                                                     n2, dim, out, ld_out);
    rc = release_packed(st);
    if (rc != OEA_OK) return rc;
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_rank_metrics(const int32_t *rank, int64_t n, const int32_t *top_k_host, int32_t nk, int64_t *hits_dev,
                     int64_t *rank_sum_dev, double *rr_sum_dev, void *stream) {
    OEA_REQUIRE(rank && top_k_host && hits_dev && rank_sum_dev && rr_sum_dev, "null pointer");
    OEA_REQUIRE(nk >= 1 && nk <= 8, "1 <= len(top_k) <= 8");
    int t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < nk; ++i) t[i] = top_k_host[i];
    rank_metrics_kernel<<<1, 1024, 0, oea::as_stream(stream)>>>(rank, n, make_int4(t[0], t[1], t[2], t[3]),
                                                               make_int4(t[4], t[5], t[6], t[7]), nk,
                                                               (long long *)hits_dev, (long long *)rank_sum_dev, rr_sum_dev);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

size_t oea_rank_eval_metrics_workspace_bytes(int64_t n1) {
    const size_t tiles = (size_t)((n1 + TILE - 1) / TILE);
    return oea_rank_workspace_bytes(n1) + 256 + tiles * (10 * sizeof(long long) + sizeof(unsigned)) + 64;
}

int oea_rank_eval_metrics(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2, int32_t dim,
                          const float *csls_r, const float *csls_c, int64_t gold_offset, const int32_t *top_k_host, int32_t nk,
                          int32_t *rank, int32_t *argmax, int64_t *hits_and_rank_sum_dev, double *rr_sum_dev, void *workspace,
                          void *stream) {
    OEA_REQUIRE(e1 && e2 && rank && argmax && workspace && top_k_host && hits_and_rank_sum_dev && rr_sum_dev, "null pointer");
    OEA_REQUIRE(n1 > 0 && gold_offset >= 0 && n1 + gold_offset <= n2, "gold of row i is column gold_offset + i <= n2");
    OEA_REQUIRE(ld1 % 4 == 0 && ld2 % 4 == 0 && dim > 0 && dim <= ld1 && dim <= ld2, "ld % 4 == 0, dim <= ld");
    OEA_REQUIRE((csls_r == nullptr) == (csls_c == nullptr), "csls_r and csls_c go together");
    OEA_REQUIRE(n2 < 0x7fffffff && nk >= 1 && nk <= 8, "n2 < 2^31, 1 <= len(top_k) <= 8");
    if (!use_glds()) { oea::set_error("oea_rank_eval_metrics needs the packed-operand tiles (OEA_TILE_GLDS=0 is set)"); return OEA_EUNSUPPORTED; }
    hipStream_t st = oea::as_stream(stream);
    unsigned long long *keys = static_cast<unsigned long long *>(workspace);
    float *gold = reinterpret_cast<float *>(keys + n1);
    const int64_t qt = oea::ceil_div(n1, TILE), ctiles = oea::ceil_div(n2, TILE);
    char *extra = reinterpret_cast<char *>(workspace) + ((oea_rank_workspace_bytes(n1) + 15) / 16) * 16;
    long long *tile_part = reinterpret_cast<long long *>(extra);
    unsigned *done = reinterpret_cast<unsigned *>(extra + (size_t)qt * 10 * sizeof(long long));
    PackedOp p1, p2;
    int64_t n1_pad = 0, n2_pad = 0;
    int rc = reserve_operand(0, n1, dim, st, &p1, &n1_pad);
    if (rc == OEA_OK) rc = reserve_operand(1, n2, dim, st, &p2, &n2_pad);
    if (rc != OEA_OK) return rc;
    const int64_t work = (n1_pad + n2_pad) * (p1.kp / 4);
    eval_prologue_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(work, 256), 16384), 256, 0, st>>>(
        e1, n1, ld1, e2, n2, ld2, dim, p1.p, n1_pad, p2.p, n2_pad, p1.kp, csls_r, csls_c, gold_offset, gold, keys, rank, done,
        (int)qt + 1);
    EvalTail tail{};
    tail.done = done;
    tail.tile_part = tile_part;
    tail.argmax = argmax;
    tail.hits = reinterpret_cast<long long *>(hits_and_rank_sum_dev);
    tail.rr_sum = rr_sum_dev;
    tail.nk = nk;
    for (int i = 0; i < nk; ++i) tail.tk[i] = top_k_host[i];
    int tpc = 1;
    const int chunks = pick_chunks(qt, ctiles, &tpc);
    const dim3 grid((unsigned)qt, (unsigned)chunks);
    if (csls_r)
        rank_inner_kernel<true, true><<<grid, 256, 0, st>>>(p1.p, n1, p1.kp, p2.p, n2, p2.kp, dim, gold, csls_r, csls_c, tpc,
                                                            gold_offset, rank, keys, tail);
    else
        rank_inner_kernel<false, true><<<grid, 256, 0, st>>>(p1.p, n1, p1.kp, p2.p, n2, p2.kp, dim, gold, csls_r, csls_c, tpc,
                                                             gold_offset, rank, keys, tail);
    rc = release_packed(st);
    if (rc != OEA_OK) return rc;
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_sim_matrix(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2,
                   int32_t dim, int32_t metric, float *out, int64_t ld_out, void *stream) {
    OEA_REQUIRE(e1 && e2 && out, "null pointer");
    OEA_REQUIRE(ld1 % 4 == 0 && ld2 % 4 == 0 && dim > 0 && dim <= ld1 && dim <= ld2 && ld_out >= n2 && ld_out < (1 << 24), "shapes");
    if (n1 == 0 || n2 == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
    if (metric == OEA_METRIC_INNER && use_glds()) {
        PackedOp p1, p2;
        int rc = pack_operand(0, e1, n1, ld1, dim, st, &p1);
        if (rc == OEA_OK) rc = pack_operand(1, e2, n2, ld2, dim, st, &p2);
        if (rc != OEA_OK) return rc;
        launch_store_packed(p1.p, n1, p2.p, n2, p1.kp, dim, out, ld_out, st);
        rc = release_packed(st);
        if (rc != OEA_OK) return rc;
    } else if (metric == OEA_METRIC_INNER) {
        sim_inner_store_kernel<false><<<dim3((unsigned)oea::ceil_div(n2, TILE), (unsigned)oea::ceil_div(n1, TILE)), 256, 0, st>>>(
            e1, n1, ld1, e2, n2, ld2, dim, out, ld_out, nullptr);
    } else if (metric == OEA_METRIC_MANHATTAN) {
        sim_valu_store_kernel<OEA_METRIC_MANHATTAN><<<dim3((unsigned)oea::ceil_div(n2, VT), (unsigned)oea::ceil_div(n1, VT)), 256, 0, st>>>(
            e1, n1, ld1, e2, n2, ld2, dim, out, ld_out);
    } else if (metric == OEA_METRIC_EUCLIDEAN) {
        sim_valu_store_kernel<OEA_METRIC_EUCLIDEAN><<<dim3((unsigned)oea::ceil_div(n2, VT), (unsigned)oea::ceil_div(n1, VT)), 256, 0, st>>>(
            e1, n1, ld1, e2, n2, ld2, dim, out, ld_out);
    } else if (metric == OEA_METRIC_MANHATTAN_F32) {
        sim_l1_f32_store_kernel<<<dim3((unsigned)oea::ceil_div(n2, LT), (unsigned)oea::ceil_div(n1, LT)), 256, 0, st>>>(
            e1, n1, ld1, e2, n2, ld2, dim, out, ld_out);
    } else {
        oea::set_error("unknown metric %d", metric);
        return OEA_EINVAL;
    }
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_quantize_rows_u16(const float *src, int64_t n, int32_t ld, int32_t dim, float lo, float inv_step, uint16_t *dst,
                          int32_t ldq, void *stream) {
    OEA_REQUIRE(src && dst && dim > 0 && dim <= ld && ldq % 8 == 0 && dim <= ldq && n >= 0, "ldq: a multiple of 8 >= dim");
    if (n == 0) return OEA_OK;
    const int64_t items = n * (ldq / 8);
    quantize_rows_u16_kernel<<<(unsigned)oea::ceil_div(items, 256), 256, 0, oea::as_stream(stream)>>>(src, n, ld, dim, lo, inv_step,
                                                                                                 dst, ldq);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_l1_u16_strip(const uint16_t *q, int64_t nq, const uint16_t *c, int64_t nc, int32_t ldq, float *out, int64_t ld_out,
                     void *stream) {
    OEA_REQUIRE(q && c && out && ldq > 0 && ldq % 8 == 0 && ld_out >= nc, "ldq: a multiple of 8; ld_out >= nc");
    OEA_REQUIRE(ldq <= 32768, "at most 32768 columns (u32 sums)");   // sums >= 2^24 round to float: <= 2^-24 relative
    if (nq == 0 || nc == 0) return OEA_OK;
    l1_u16_strip_kernel<<<dim3((unsigned)oea::ceil_div(nc, LT), (unsigned)oea::ceil_div(nq, LT)), 256, 0, oea::as_stream(stream)>>>(
        q, nq, c, nc, ldq, out, ld_out);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_rank_l1_grid_rows(const float *strip, int64_t rows, int64_t row0, int64_t nc, int64_t ld, const float *e1, int32_t ld1,
                          const float *e2, int32_t ld2, int32_t dim, int64_t gold_offset, float step, float err, int32_t *rank,
                          int32_t *argmax, int32_t *n_exact_rows, void *stream) {
    OEA_REQUIRE(strip && e1 && e2 && rank && argmax && rows >= 0 && row0 >= 0 && nc > 0 && ld >= nc, "arguments");
    OEA_REQUIRE(ld % 4 == 0 && ((uintptr_t)strip & 15) == 0, "strip rows: 16-byte aligned (ld % 4 == 0)");
    OEA_REQUIRE(dim > 0 && dim <= ld1 && dim <= ld2 && dim <= 4096 && step > 0.f && err >= 0.f, "dim <= 4096, step > 0");
    OEA_REQUIRE(row0 + rows + gold_offset <= nc && gold_offset >= 0, "gold of row i is column gold_offset + i < nc");
    if (rows == 0) return OEA_OK;
    const size_t lds = sizeof(double) * (size_t)((dim + 1) & ~1) + sizeof(int32_t) * (kGridAmb + kGridTop);
    rank_l1_grid_rows_kernel<<<(unsigned)rows, 256, lds, oea::as_stream(stream)>>>(strip, rows, row0, nc, ld, e1, ld1, e2, ld2, dim,
                                                                                   gold_offset, step, err, rank, argmax, n_exact_rows);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_rank_l1_grid_rows_csls(const float *strip, int64_t rows, int64_t row0, int64_t nc, int64_t ld, const float *e1, int32_t ld1,
                               const float *e2, int32_t ld2, int32_t dim, int64_t gold_offset, float step, float err,
                               const float *csls_r, const float *csls_c, int32_t *rank, int32_t *argmax, int32_t *n_exact_rows,
                               void *stream) {
    OEA_REQUIRE(strip && e1 && e2 && rank && argmax && csls_r && csls_c && rows >= 0 && row0 >= 0 && nc > 0 && ld >= nc, "arguments");
    OEA_REQUIRE(ld % 4 == 0 && ((uintptr_t)strip & 15) == 0, "strip rows: 16-byte aligned (ld % 4 == 0)");
    OEA_REQUIRE(dim > 0 && dim <= ld1 && dim <= ld2 && dim <= 4096 && step > 0.f && err >= 0.f, "dim <= 4096, step > 0");
    OEA_REQUIRE(row0 + rows + gold_offset <= nc && gold_offset >= 0, "gold of row i is column gold_offset + i < nc");
    if (rows == 0) return OEA_OK;
    const size_t lds = sizeof(double) * (size_t)((dim + 1) & ~1) + sizeof(int32_t) * (kGridAmb + kGridTop);
    rank_l1_grid_rows_csls_kernel<<<(unsigned)rows, 256, lds, oea::as_stream(stream)>>>(strip, rows, row0, nc, ld, e1, ld1, e2, ld2, dim,
                                                                                        gold_offset, step, err, csls_r, csls_c, rank,
                                                                                        argmax, n_exact_rows);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_pair_l1_sim(const float *q, int64_t nq, int32_t ldq, const float *table, int64_t n, int32_t ldt, int32_t dim,
                    const int32_t *cand, int32_t c, float *out, void *stream) {
    OEA_REQUIRE(q && table && cand && out && dim > 0 && dim <= ldq && dim <= ldt && c > 0 && n > 0, "arguments");
    if (nq == 0) return OEA_OK;
    pair_l1_sim_seq_kernel<<<(unsigned)oea::ceil_div(nq * c, 256), 256, 0, oea::as_stream(stream)>>>(q, nq, ldq, table, ldt, dim, cand, c, out);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_pair_l1_f64(const float *q, int64_t nq, int32_t ldq, const float *table, int64_t n, int32_t ldt, int32_t dim,
                    const int32_t *cand, int32_t c, double *out, void *stream) {
    OEA_REQUIRE(q && table && cand && out && dim > 0 && dim <= ldq && dim <= ldt && c > 0 && n > 0, "arguments");
    if (nq == 0) return OEA_OK;
    const int64_t groups = nq * c;
    pair_l1_f64_kernel<<<(unsigned)oea::ceil_div(groups, 16), 256, 0, oea::as_stream(stream)>>>(q, nq, ldq, table, ldt, dim, cand, c, out);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_row_topk_mean(const float *s, int64_t n1, int64_t n2, int64_t ld, int32_t k, float *out, void *stream) {
    OEA_REQUIRE(s && out, "null pointer");
    OEA_REQUIRE(k >= 1 && k <= 32 && k <= n2, "1 <= k <= min(32, n2)");
    if (n1 == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
    const unsigned grid = (unsigned)oea::ceil_div(n1, 4);
    row_topk_mean_kernel<<<grid, 256, 0, st>>>(s, n1, n2, ld, k, out, nullptr, nullptr);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

size_t oea_csls_means_workspace_bytes(int64_t n1, int64_t n2, int32_t k) {
    const CslsPlan p = plan_csls(n1, n2, k), pb = plan_csls(n1, n2, k, BIG_MT), ps = plan_csls(n1, n2, k, SPEC_MT);   // (the call picks one by the row width)
    return p.ok ? std::max(p.total, std::max(pb.ok ? pb.total : 0, ps.ok ? ps.total : 0)) : 0;
}

int oea_csls_means(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2, int32_t dim, int32_t k,
                   float *r_out, float *c_out, void *workspace, size_t ws_bytes, void *stream) {
    OEA_REQUIRE(e1 && e2 && r_out && c_out && workspace, "null pointer");
    OEA_REQUIRE(ld1 % 4 == 0 && ld2 % 4 == 0 && dim > 0 && dim <= ld1 && dim <= ld2 && dim <= 2048, "ld % 4 == 0, dim <= min(ld, 2048)");
    OEA_REQUIRE(k >= 1 && k <= n1 && k <= n2, "1 <= k <= min(n1, n2)");
    // from 3e8 pairs on the sweep multiplies the bf16 hi / lo split (3/16 of the fp32 matrix time, see the prefilter section);
    // the means are still those of the exact values (list_mean_rows_kernel<true>).  OEA_CSLS_BF16=0 keeps the fp32 sweep.
    // (both read per call: the tests move the limit).  Rows wider than 128: 256-candidate tiles, 512 threads, three-stage ring.
    const char *env_on = getenv("OEA_CSLS_BF16"), *env_min = getenv("OEA_CSLS_BF16_MIN_PAIRS");
    const bool bf16_on = !(env_on && env_on[0] == '0');
    const double bf16_min = env_min ? atof(env_min) : 3e8;
    const bool bf16 = bf16_on && (double)n1 * (double)n2 >= bf16_min;
    static const bool big_on = [] { const char *e = getenv("OEA_BF16_BIG"); return !(e && e[0] == '0'); }();
    static const int big_mode = [] { const char *e = getenv("OEA_BF16_BIG_MODE"); return e ? atoi(e) : 1; }();
    const int big_mt = big_mode == 2 ? SPEC_MT : BIG_MT;
    const bool big = bf16 && big_on && (dim + BK - 1) / BK * BK > 128 && plan_csls(n1, n2, k, big_mt).ok;
    const CslsPlan p = plan_csls(n1, n2, k, big ? big_mt : 0);
    if (!p.ok || !use_glds()) { oea::set_error("oea_csls_means: shape not covered (n1, n2 >= 4096, k <= 32, packed tiles)"); return OEA_EUNSUPPORTED; }
    OEA_REQUIRE(ws_bytes >= p.total, "workspace smaller than oea_csls_means_workspace_bytes");
    hipStream_t st = oea::as_stream(stream);
    char *w = static_cast<char *>(workspace);
    float *thr1 = reinterpret_cast<float *>(w + p.off_thr1), *thr2 = reinterpret_cast<float *>(w + p.off_thr2);
    int32_t *qcnt = reinterpret_cast<int32_t *>(w + p.off_qcnt), *ccnt = reinterpret_cast<int32_t *>(w + p.off_ccnt);
    int32_t *fail1 = reinterpret_cast<int32_t *>(w + p.off_fail1), *fail2 = reinterpret_cast<int32_t *>(w + p.off_fail2);
    int32_t *nfail = reinterpret_cast<int32_t *>(w + p.off_nfail);          // [0] rows, [1] columns
    float *qlists = reinterpret_cast<float *>(w + p.off_qlists), *clists = reinterpret_cast<float *>(w + p.off_clists);
    float *strip = reinterpret_cast<float *>(w + p.off_strip), *fbq = reinterpret_cast<float *>(w + p.off_fbq);
    float *scratch = reinterpret_cast<float *>(w + p.off_scratch);
    PackedOp p1, p2, s1, s2, b1, b2;
    int rc = pack_operand(0, e1, n1, ld1, dim, st, &p1);          // (fp32 packs: the exact strips of the fallbacks)
    if (rc == OEA_OK) rc = pack_operand(1, e2, n2, ld2, dim, st, &p2);
    if (rc != OEA_OK) return rc;
    const int kp = p1.kp;
    OEA_REQUIRE(kp <= 4096, "dim <= 4096");
    // thresholds: rows of S against sampled candidates (every (n2 / S)-th), columns against sampled queries.  Under the bf16 sweep
    // the sample strips are bf16 products too (round 5): a threshold is an ESTIMATE of where the k-th value lies -- the lists are
    // cut below it by the bound and rows whose list comes out short or long take the exact fallback either way -- and at K = 1,200
    // the two fp32 strips cost 10.9 ms of a 90 ms evaluation
    static const bool bf16_strips = [] { const char *e = getenv("OEA_CSLS_BF16_STRIPS"); return !(e && e[0] == '0'); }();
    if (bf16) {
        rc = pack_operand_bf16(4, e1, n1, ld1, dim, st, &b1);
        if (rc == OEA_OK) rc = pack_operand_bf16(5, e2, n2, ld2, dim, st, &b2);
        if (rc != OEA_OK) return rc;
    }
    if (bf16 && bf16_strips) {
        rc = pack_operand_bf16(2, e2, p.sample, ld2 * (int)(n2 / p.sample), dim, st, &s2);
        if (rc == OEA_OK) rc = pack_operand_bf16(3, e1, p.sample, ld1 * (int)(n1 / p.sample), dim, st, &s1);
        if (rc != OEA_OK) return rc;
        const unsigned gs = (unsigned)oea::ceil_div(p.sample, TILE);
        sim_bf16_store_kernel<<<dim3(gs, (unsigned)oea::ceil_div(n1, TILE)), 256, 0, st>>>(b1.p, n1, kp, s2.p, p.sample, dim, strip, p.sample);
        rc = oea::kth_value(strip, n1, p.sample, p.r1, thr1, st);
        if (rc != OEA_OK) return rc;
        sim_bf16_store_kernel<<<dim3(gs, (unsigned)oea::ceil_div(n2, TILE)), 256, 0, st>>>(b2.p, n2, kp, s1.p, p.sample, dim, strip, p.sample);
        rc = oea::kth_value(strip, n2, p.sample, p.r2, thr2, st);
        if (rc != OEA_OK) return rc;
    } else {
        rc = pack_operand(2, e2, p.sample, ld2 * (int)(n2 / p.sample), dim, st, &s2);
        if (rc == OEA_OK) rc = pack_operand(3, e1, p.sample, ld1 * (int)(n1 / p.sample), dim, st, &s1);
        if (rc != OEA_OK) return rc;
        launch_store_packed(p1.p, n1, s2.p, p.sample, kp, dim, strip, p.sample, st);
        rc = oea::kth_value(strip, n1, p.sample, p.r1, thr1, st);
        if (rc != OEA_OK) return rc;
        launch_store_packed(p2.p, n2, s1.p, p.sample, kp, dim, strip, p.sample, st);
        rc = oea::kth_value(strip, n2, p.sample, p.r2, thr2, st);
        if (rc != OEA_OK) return rc;
    }
    // every (candidate, query tile) count is written by the sweep when chunks cover all candidate tiles -- they do
    OEA_CHECK_HIP(hipMemsetAsync(nfail, 0, 256, st));
    const TileGrid grid = make_tile_grid((unsigned)oea::ceil_div(n1, TILE), (unsigned)p.chunks);
    if (bf16) {
        float *tol = reinterpret_cast<float *>(nfail + 16);                // [0] the bound, [1] / [2] max row norms (zeroed above)
        row_norm_max_kernel<<<(unsigned)oea::ceil_div(n1, 256), 256, 0, st>>>(e1, n1, ld1, dim, reinterpret_cast<unsigned *>(tol) + 1);
        row_norm_max_kernel<<<(unsigned)oea::ceil_div(n2, 256), 256, 0, st>>>(e2, n2, ld2, dim, reinterpret_cast<unsigned *>(tol) + 2);
        csls_tol_kernel<<<1, 1, 0, st>>>(tol, bf16_eps_rel(dim, true));
        static const bool breg_on = [] { const char *e = getenv("OEA_BF16_BREG"); return !(e && e[0] == '0'); }();
#define OEA_CSLS_APPEND(N) csls_append_kernel<true, true, N><<<tile_grid_blocks(grid), 256, 0, st>>>(b1.p, n1, kp, b2.p, n2, kp, dim, thr1, thr2, p.tpc, \
                                                                                                   p.cap, p.ccap, qlists, qcnt, clists, ccnt, tol, grid)
        if (big) {
#define OEA_CSLS_BIG(M)                                                                                                              \
            do {                                                                                                                    \
                OEA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(csls_append_kernel<true, true, 0, 8, M>),           \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS_BYTES));   /* per device */   \
                csls_append_kernel<true, true, 0, 8, M><<<tile_grid_blocks(grid), 512, BIG_LDS_BYTES, st>>>(                        \
                    b1.p, n1, kp, b2.p, n2, kp, dim, thr1, thr2, p.tpc, p.cap, p.ccap, qlists, qcnt, clists, ccnt, tol, grid);      \
                OEA_CHECK_HIP(hipGetLastError());                                                                                   \
            } while (0)
            if (big_mode == 2) OEA_CSLS_BIG(2);
            else if (big_mode == 1) OEA_CSLS_BIG(1);
            else OEA_CSLS_BIG(0);
#undef OEA_CSLS_BIG
        } else switch ((breg_on && kp <= 128) ? kp / 32 : 0) {
            case 4: OEA_CSLS_APPEND(4); break;
            case 3: OEA_CSLS_APPEND(3); break;
            case 2: OEA_CSLS_APPEND(2); break;
            case 1: OEA_CSLS_APPEND(1); break;
            default: OEA_CSLS_APPEND(0); break;
        }
#undef OEA_CSLS_APPEND
        list_mean_rows_kernel<true><<<(unsigned)oea::ceil_div(n1, 4), 256, 0, st>>>(qlists, qcnt, p.nseg, p.cap, n1, k, r_out, fail1, nfail,
                                                                                    e1, ld1, e2, ld2, dim, thr1, tol);
        list_mean_rows_kernel<true><<<(unsigned)oea::ceil_div(n2, 4), 256, 0, st>>>(clists, ccnt, 2 * p.nqt, p.ccap, n2, k, c_out, fail2,
                                                                                    nfail + 1, e2, ld2, e1, ld1, dim, thr2, tol);
    } else {
        csls_append_kernel<true, false><<<tile_grid_blocks(grid), 256, 0, st>>>(p1.p, n1, kp, p2.p, n2, kp, dim, thr1, thr2, p.tpc, p.cap, p.ccap,
                                                                                 qlists, qcnt, clists, ccnt, nullptr, grid);
        list_mean_rows_kernel<false><<<(unsigned)oea::ceil_div(n1, 4), 256, 0, st>>>(qlists, qcnt, p.nseg, p.cap, n1, k, r_out, fail1, nfail,
                                                                                     nullptr, 0, nullptr, 0, 0, nullptr, nullptr);
        list_mean_rows_kernel<false><<<(unsigned)oea::ceil_div(n2, 4), 256, 0, st>>>(clists, ccnt, 2 * p.nqt, p.ccap, n2, k, c_out, fail2,
                                                                                     nfail + 1, nullptr, 0, nullptr, 0, 0, nullptr, nullptr);
    }
    // fallbacks (normally empty): bulk for the first kCslsFb failed rows / columns, slow kernel for the rest
    oea::gather_packed_rows(p1.p, kp, fail1, nfail, fbq, st);
    sim_inner_store_kernel<true><<<dim3((unsigned)oea::ceil_div(n2, TILE), 1), 256, 0, st>>>(fbq, kCslsFb, kp, p2.p, n2, kp, dim, strip, p.ld2, nfail);
    row_topk_mean_kernel<<<kCslsFb / 4, 256, 0, st>>>(strip, kCslsFb, n2, p.ld2, k, r_out, fail1, nfail);
    slow_mean_rows_kernel<<<kCslsSlow, 256, 0, st>>>(e1, ld1, e2, n2, ld2, dim, k, r_out, fail1, nfail, kCslsFb, scratch, p.ld2);
    oea::gather_packed_rows(p2.p, kp, fail2, nfail + 1, fbq, st);
    sim_inner_store_kernel<true><<<dim3((unsigned)oea::ceil_div(n1, TILE), 1), 256, 0, st>>>(fbq, kCslsFb, kp, p1.p, n1, kp, dim, strip, p.ld1, nfail + 1);
    row_topk_mean_kernel<<<kCslsFb / 4, 256, 0, st>>>(strip, kCslsFb, n1, p.ld1, k, c_out, fail2, nfail + 1);
    slow_mean_rows_kernel<<<kCslsSlow, 256, 0, st>>>(e2, ld2, e1, n1, ld1, dim, k, c_out, fail2, nfail + 1, kCslsFb, scratch, p.ld1);
    rc = release_packed(st);
    if (rc != OEA_OK) return rc;
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_rank_rows(const float *s, int64_t n_rows, int64_t nc, int64_t ld, const int32_t *gold_idx, int32_t *rank,
                  int32_t *argmax, void *stream) {
    OEA_REQUIRE(s && gold_idx && rank && argmax, "null pointer");
    OEA_REQUIRE(nc >= 1 && nc < 0x7fffffff && ld >= nc && n_rows < 0x7fffffff, "1 <= nc <= ld");
    if (n_rows == 0) return OEA_OK;
    rank_rows_kernel<<<(unsigned)n_rows, 256, 0, oea::as_stream(stream)>>>(s, nc, ld, gold_idx, rank, argmax);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_csls_apply(float *s, int64_t n1, int64_t n2, int64_t ld, const float *r, const float *c, void *stream) {
    OEA_REQUIRE(s && r && c, "null pointer");
    if (n1 == 0 || n2 == 0) return OEA_OK;
    OEA_REQUIRE(n1 <= 65535 * 1024ll, "n1 too large for one launch");
    // grid.y is limited to 65535: loop over row blocks
    for (int64_t i0 = 0; i0 < n1; i0 += 65535) {
        const int64_t rows = std::min<int64_t>(65535, n1 - i0);
        csls_apply_kernel<<<dim3((unsigned)oea::ceil_div(n2, 256), (unsigned)rows), 256, 0, oea::as_stream(stream)>>>(
            s + i0 * ld, rows, n2, ld, r + i0, c);
    }
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
