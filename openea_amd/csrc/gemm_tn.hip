// gemm_tn.hip -- C = A^T B for two TALL row-major operands: the weight gradients of the GNN approaches' dense layers.
//
// AliNet / RDGCN put a [E, d_in] x [d_in, d_out] product in front of every sparse aggregate (alinet.py:574-582, 656-660;
// rdgcn.py:250-256); its weight gradient is  dW = X^T dY  with X [E, d_in], dY [E, d_out], E = 200,000 entities and
// d <= 500: a tiny output reduced over a very long K.  The library runs that shape at 44-57 TFLOP/s (profiles/r03_*: its
// NN / NT kernels reach 85-105 on the forward products of the same layers); here the reduction is split over the rows:
//
//   grid = (output tiles, row chunks); a workgroup (4 waves, 2 x 2) owns a 128 x TJ output tile (TJ = 128 or 64) over one
//   chunk of rows and runs v_mfma_f32_32x32x2_f32 (exact fp32) down its rows: the reduction index is the ROW index of both
//   operands, so a 16-row slab of A and of B in LDS *as it lies in memory* is already the operand layout of the MFMA (lane l
//   holds column l % 32 of slab row 2 s + l / 32): no transposes, no k permutation, ds_read_b32 only, conflict-free with a
//   row stride of 160 floats (two consecutive slab rows land on the two halves of the 64 banks);
//   slabs double-buffered in LDS and two more in flight in registers (global -> registers two slabs ahead), one barrier per slab;
//   the chunk's tile goes to partials[chunk] and a second kernel adds the chunks in chunk order: a fixed summation order
//   (replicas of a data-parallel job get the same bits; the library's split-K order is its own business).
#include "common.h"

#include <algorithm>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TI = 128;          // output rows per workgroup (columns of A)
constexpr int SLAB = 16;         // operand rows per LDS slab
constexpr int LDT = 160;         // LDS row stride in floats: 160 % 64 == 32

template <int NJ>                // 32-column blocks of B per wave: TJ = 64 * NJ
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const float *__restrict__ a, int lda, int k1,
                                                         const float *__restrict__ b, int ldb, int k2, int64_t m,
                                                         int64_t rows_per_chunk, int tiles_j, float *__restrict__ dst,
                                                         int64_t chunk_stride, int ld_dst) {
    constexpr int TJ = 64 * NJ;
    __shared__ __attribute__((aligned(16))) float As[2][SLAB * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[2][SLAB * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ti = (int)blockIdx.x / tiles_j, tj = (int)blockIdx.x % tiles_j;
    const int i0 = ti * TI, j0 = tj * TJ;
    const int64_t m0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t m1 = m0 + rows_per_chunk < m ? m0 + rows_per_chunk : m;
    const int nslab = (int)((m1 - m0 + SLAB - 1) / SLAB);
    // staging: a slab is 16 rows x 128 (or TJ) columns; thread -> (row = tid / 32 [+ 8], 4 columns at (tid % 32) * 4)
    const int sr = tid >> 5, sc = (tid & 31) * 4;
    const bool a_col = i0 + sc < k1, b_col = sc < TJ && j0 + sc < k2;          // whole float4 inside: k1, k2 % 4 == 0
    // two register sets: slab s + 1 waits in one while slab s + 2 is in flight in the other (a slab is only 32 MFMAs = 2,048
    // cycles per wave, less than one trip to HBM; with one set the 128-wide kernel ran at 100 TFLOP/s, the 64-wide at 31)
    float4 ra[2][2], rb[2][2];
    auto load = [&](int set, int s) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t row = m0 + (int64_t)s * SLAB + sr + 8 * h;
            const bool in = row < m1;
            ra[set][h] = (in && a_col) ? oea::ld4(a + row * lda + i0 + sc) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[set][h] = (in && b_col) ? oea::ld4(b + row * ldb + j0 + sc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store = [&](int set, int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            oea::st4(&As[buf][(sr + 8 * h) * LDT + sc], ra[set][h]);
            if (sc < TJ) oea::st4(&Bs[buf][(sr + 8 * h) * LDT + sc], rb[set][h]);
        }
    };
    f32x16 acc[2][NJ];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < NJ; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    // Which 32 x 32 blocks of the tile a wave owns.  Interior tiles: 2 x 2 waves, 2 x NJ blocks each.  A tile that hangs over
    // the matrix edge with <= 64 valid columns (d = 300: 44, d = 400: 16) is cut 4 x 1 instead -- every wave keeps work and the
    // tile costs a half / a quarter of an interior one -- and likewise 1 x 4 for <= 64 valid rows; blocks entirely past the edge
    // are skipped (all wave-uniform).
    int i_off[2], j_off[NJ];
    bool on_i[2], on_j[NJ];
    const int vi = k1 - i0, vj = k2 - j0;
    if (NJ == 2 && vj <= 64 && vi > 64) {
        i_off[0] = wave * 32; i_off[1] = 0;
        on_i[0] = i_off[0] < vi; on_i[1] = false;
#pragma unroll
        for (int y = 0; y < NJ; ++y) { j_off[y] = y * 32; on_j[y] = j_off[y] < vj; }
    } else if (NJ == 2 && vi <= 64 && vj > 64) {
#pragma unroll
        for (int x = 0; x < 2; ++x) { i_off[x] = x * 32; on_i[x] = i_off[x] < vi; }
#pragma unroll
        for (int y = 0; y < NJ; ++y) { j_off[y] = wave * 32; on_j[y] = y == 0 && j_off[y] < vj; }
    } else {
#pragma unroll
        for (int x = 0; x < 2; ++x) { i_off[x] = wm * 64 + x * 32; on_i[x] = i_off[x] < vi; }
#pragma unroll
        for (int y = 0; y < NJ; ++y) { j_off[y] = wn * 32 * NJ + y * 32; on_j[y] = j_off[y] < vj; }
    }
    const int half = lane >> 5, l32 = lane & 31;
    auto compute = [&](int cur) {
        const float *ap = &As[cur][half * LDT + l32];
        const float *bp = &Bs[cur][half * LDT + l32];
#pragma unroll
        for (int ks = 0; ks < SLAB / 2; ++ks) {
            float av[2], bv[NJ];
#pragma unroll
            for (int x = 0; x < 2; ++x) av[x] = ap[2 * ks * LDT + i_off[x]];
#pragma unroll
            for (int y = 0; y < NJ; ++y) bv[y] = bp[2 * ks * LDT + j_off[y]];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < NJ; ++y)
                    if (on_i[x] && on_j[y]) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], bv[y], acc[x][y], 0, 0, 0);
        }
    };
    // invariant at the top of step s: LDS buffer s & 1 holds slab s, register set (s + 1) & 1 holds slab s + 1, set s & 1 is
    // receiving slab s + 2 (loads of rows past the chunk return zeros without touching memory)
    if (nslab > 0) {
        load(0, 0);
        store(0, 0);
        load(1, 1);
        load(0, 2);
    }
    __syncthreads();
    for (int s = 0; s < nslab; s += 2) {
        compute(0);
        store(1, 1);                                                  // slab s + 1
        load(1, s + 3);
        __syncthreads();
        if (s + 1 < nslab) {
            compute(1);
            store(0, 0);                                              // slab s + 2
            load(0, s + 4);
        }
        __syncthreads();
    }
    // acc[x][y][r]: output row i0 + i_off[x] + (r & 3) + 8 (r >> 2) + 4 half, column j0 + j_off[y] + l32
    float *out = dst + (int64_t)blockIdx.y * chunk_stride;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < NJ; ++y) {
            if (!(on_i[x] && on_j[y])) continue;
            const int j = j0 + j_off[y] + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + i_off[x] + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (i < k1 && j < k2) out[(int64_t)i * ld_dst + j] = acc[x][y][r];
            }
        }
}

// out[e] = partials[0][e] + partials[1][e] + ... in chunk order
__global__ __launch_bounds__(256) void add_chunks_kernel(const float *__restrict__ partials, int chunks, int64_t stride, int k1, int k2,
                                                         float *__restrict__ out, int ld_out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)k1 * k2) return;
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < chunks; ++c) s += partials[c * stride + e];
    out[(e / k2) * ld_out + (e % k2)] = s;
}

struct Plan { int tj, tiles_i, tiles_j, chunks; int64_t rows_per_chunk; };

Plan plan_tn(int64_t m, int k1, int k2) {
    Plan p;
    // 128-wide column tiles (4 MFMAs per 4 LDS reads); the 64-wide instantiation (2 per 3) only when the matrix is that narrow
    p.tj = k2 <= 64 ? 64 : 128;
    p.tiles_i = (int)oea::ceil_div(k1, TI);
    p.tiles_j = (int)oea::ceil_div(k2, p.tj);
    const int tiles = p.tiles_i * p.tiles_j;
    int64_t chunks = std::max<int64_t>(1, std::min<int64_t>(oea::ceil_div(2048, tiles), oea::ceil_div(m, 256)));
    p.rows_per_chunk = oea::ceil_div(oea::ceil_div(m, chunks), SLAB) * SLAB;
    p.chunks = (int)oea::ceil_div(m, p.rows_per_chunk);
    return p;
}

}  // namespace

extern "C" {

size_t oea_gemm_tn_workspace_floats(int64_t m, int32_t k1, int32_t k2) {
    if (m <= 0 || k1 <= 0 || k2 <= 0) return 0;
    const Plan p = plan_tn(m, k1, k2);
    return p.chunks > 1 ? (size_t)p.chunks * k1 * k2 : 0;
}

int oea_gemm_tn_f32(const float *a, int32_t lda, int32_t k1, const float *b, int32_t ldb, int32_t k2, int64_t m, float *out,
                    int32_t ld_out, float *workspace, void *stream) {
    OEA_REQUIRE(a && b && out && k1 > 0 && k2 > 0 && m >= 0 && k1 <= lda && k2 <= ldb && k2 <= ld_out, "shapes");
    OEA_REQUIRE(k1 % 4 == 0 && k2 % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0, "k1, k2, lda, ldb: multiples of 4 (16-byte row segments)");
    OEA_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0, "16-byte aligned operands");
    hipStream_t st = oea::as_stream(stream);
    if (m == 0) {
        OEA_CHECK_HIP(hipMemset2DAsync(out, sizeof(float) * ld_out, 0, sizeof(float) * k2, k1, st));
        return OEA_OK;
    }
    const Plan p = plan_tn(m, k1, k2);
    OEA_REQUIRE(p.chunks == 1 || workspace, "workspace of oea_gemm_tn_workspace_floats(m, k1, k2) floats");
    float *dst = p.chunks > 1 ? workspace : out;
    const int ld_dst = p.chunks > 1 ? k2 : ld_out;
    const int64_t stride = (int64_t)k1 * k2;
    const dim3 grid((unsigned)(p.tiles_i * p.tiles_j), (unsigned)p.chunks);
    if (p.tj == 128) gemm_tn_kernel<2><<<grid, 256, 0, st>>>(a, lda, k1, b, ldb, k2, m, p.rows_per_chunk, p.tiles_j, dst, stride, ld_dst);
    else gemm_tn_kernel<1><<<grid, 256, 0, st>>>(a, lda, k1, b, ldb, k2, m, p.rows_per_chunk, p.tiles_j, dst, stride, ld_dst);
    if (p.chunks > 1)
        add_chunks_kernel<<<(unsigned)oea::ceil_div(stride, 256), 256, 0, st>>>(workspace, p.chunks, stride, k1, k2, out, ld_out);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

// The same product cut at its row chunks, for a job whose ranks each own a block of the rows (models/graph_ops.py:
// the weight gradients of the GNN approaches' dense layers): rank r computes the partial products of ITS chunks into their
// slots of the [chunks][k1 * k2] workspace, the slots are all-gathered, and every rank adds all of them in chunk order --
// the single-process kernel's summation order, so the sharded job gives the same bits.
int oea_gemm_tn_plan(int64_t m, int32_t k1, int32_t k2, int32_t *chunks, int64_t *rows_per_chunk) {
    OEA_REQUIRE(chunks && rows_per_chunk && m > 0 && k1 > 0 && k2 > 0, "arguments");
    const Plan p = plan_tn(m, k1, k2);
    *chunks = p.chunks;
    *rows_per_chunk = p.rows_per_chunk;
    return OEA_OK;
}

int oea_gemm_tn_partial(const float *a, int32_t lda, int32_t k1, const float *b, int32_t ldb, int32_t k2, int64_t m,
                        int32_t chunk_begin, int32_t chunk_end, float *workspace, void *stream) {
    OEA_REQUIRE(a && b && workspace && k1 > 0 && k2 > 0 && m > 0 && k1 <= lda && k2 <= ldb, "shapes");
    OEA_REQUIRE(k1 % 4 == 0 && k2 % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0, "k1, k2, lda, ldb: multiples of 4 (16-byte row segments)");
    OEA_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0, "16-byte aligned operands");
    const Plan p = plan_tn(m, k1, k2);
    OEA_REQUIRE(0 <= chunk_begin && chunk_begin <= chunk_end && chunk_end <= p.chunks, "0 <= chunk_begin <= chunk_end <= chunks");
    if (chunk_begin == chunk_end) return OEA_OK;
    const int64_t r0 = (int64_t)chunk_begin * p.rows_per_chunk;
    const int64_t r1 = std::min<int64_t>(m, (int64_t)chunk_end * p.rows_per_chunk);
    const int64_t stride = (int64_t)k1 * k2;
    const dim3 grid((unsigned)(p.tiles_i * p.tiles_j), (unsigned)(chunk_end - chunk_begin));
    hipStream_t st = oea::as_stream(stream);
    // rows_per_chunk is a multiple of the 16-row slab: the offset operands stay 16-byte aligned
    const float *a0 = a + r0 * lda, *b0 = b + r0 * ldb;
    float *dst = workspace + (int64_t)chunk_begin * stride;
    if (p.tj == 128) gemm_tn_kernel<2><<<grid, 256, 0, st>>>(a0, lda, k1, b0, ldb, k2, r1 - r0, p.rows_per_chunk, p.tiles_j, dst, stride, k2);
    else gemm_tn_kernel<1><<<grid, 256, 0, st>>>(a0, lda, k1, b0, ldb, k2, r1 - r0, p.rows_per_chunk, p.tiles_j, dst, stride, k2);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_gemm_tn_reduce(const float *workspace, int32_t chunks, int32_t k1, int32_t k2, float *out, int32_t ld_out, void *stream) {
    OEA_REQUIRE(workspace && out && chunks >= 1 && k1 > 0 && k2 > 0 && k2 <= ld_out, "arguments");
    const int64_t stride = (int64_t)k1 * k2;
    add_chunks_kernel<<<(unsigned)oea::ceil_div(stride, 256), 256, 0, oea::as_stream(stream)>>>(workspace, chunks, stride, k1, k2, out, ld_out);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
