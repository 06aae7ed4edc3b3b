// sparse_attn.hip -- segment softmax + neighbour aggregate (sparse graph attention) for gfx950.
//
// Replaces, for AliNet (approaches/alinet.py:661-676) and RDGCN (approaches/rdgcn.py:202-215):
//     weights = leaky_relu(per-edge logits)            tf.nn.leaky_relu on SparseTensor.values
//     attention = tf.sparse_softmax(weights)           softmax over the entries of a row
//     out = tf.sparse_tensor_dense_matmul(attention, V)
// and the TF gradients of the three ops.
//
// The entries that are normalised together are DATA (TF1's CPU SparseSoftmax groups
// consecutive entries with equal leading index, and the reference feeds non-canonical index
// orders): a segment s owns edges [seg_ptr[s], seg_ptr[s+1]) and adds its aggregate to output
// row seg_row[s].  Whole-row semantics = one segment per row (seg_ptr = CSR rowptr).
//
// Forward: one wave per segment.  Edge logits are reduced lane-parallel (max, sum of exp) in
// chunks of 64 edges; the aggregate walks the edges with the 64 lanes across the feature
// columns (coalesced 256-B gathers of V rows), alpha broadcast by __shfl.
// Backward: (1) per segment d alpha_e = dOut_row . V_col (one wave reduction per edge), then
// d z_e = alpha_e (d alpha_e - sum_k alpha_k d alpha_k) * lrelu'(z_e); (2) dV = transposed
// aggregate with the stored alphas (edge permutation from the transposed CSR).
// Bytes per launch (forward): nnz*(12 + 4*d) + 4*N*d  (SpMM bytes + nnz*4 logits).
#include "common.h"

namespace {

constexpr int W = 64;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float lrelu(float x, float a) { return x > 0.f ? x : a * x; }

template <int IT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const int32_t *__restrict__ seg_ptr, const int32_t *__restrict__ seg_row,
                                                       int64_t n_seg, const int32_t *__restrict__ colidx,
                                                       const float *__restrict__ z, const float *__restrict__ v, int dim,
                                                       int ld, float slope, float *__restrict__ out,
                                                       float *__restrict__ alpha, int unique_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t seg = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (seg >= n_seg) return;
    const int e0 = seg_ptr[seg], e1 = seg_ptr[seg + 1];
    if (e1 <= e0) return;
    // softmax statistics over the segment
    float m = -INFINITY;
    for (int e = e0 + lane; e < e1; e += W) m = fmaxf(m, lrelu(z[e], slope));
    m = wave_max(m);
    float l = 0.f;
    for (int e = e0 + lane; e < e1; e += W) l += expf(lrelu(z[e], slope) - m);
    l = wave_sum(l);
    const float inv_l = 1.0f / l;
    float acc[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) acc[it] = 0.f;
    for (int base = e0; base < e1; base += W) {
        const int e = base + lane;
        float a = 0.f;
        int c = 0;
        if (e < e1) {
            a = expf(lrelu(z[e], slope) - m) * inv_l;
            c = colidx[e];
            alpha[e] = a;
        }
        const int cnt = min(W, e1 - base);
        for (int j = 0; j < cnt; ++j) {
            const float aj = __shfl(a, j, 64);
            const int cj = __shfl(c, j, 64);
            const float *vr = v + (int64_t)cj * ld;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int col = it * W + lane;
                if (col < dim) acc[it] = fmaf(aj, vr[col], acc[it]);
            }
        }
    }
    float *o = out + (int64_t)seg_row[seg] * ld;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int col = it * W + lane;
        if (col < dim) {
            if (unique_rows) o[col] = acc[it];
            else oea::atomic_add_f32(o + col, acc[it]);
        }
    }
}

// d z_e for every edge of a segment.  dz doubles as scratch for d alpha between the two passes.
template <int IT>
__global__ __launch_bounds__(256) void attn_bwd_edges_kernel(const int32_t *__restrict__ seg_ptr,
                                                             const int32_t *__restrict__ seg_row, int64_t n_seg,
                                                             const int32_t *__restrict__ colidx, const float *__restrict__ z,
                                                             const float *__restrict__ v, const float *__restrict__ alpha,
                                                             const float *__restrict__ dout, int dim, int ld, float slope,
                                                             float *__restrict__ dz) {
    const int lane = threadIdx.x & 63;
    const int64_t seg = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (seg >= n_seg) return;
    const int e0 = seg_ptr[seg], e1 = seg_ptr[seg + 1];
    if (e1 <= e0) return;
    const float *dor = dout + (int64_t)seg_row[seg] * ld;
    float d[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int col = it * W + lane;
        d[it] = col < dim ? dor[col] : 0.f;
    }
    float csum = 0.f;                      // lane partial of sum_e alpha_e * dalpha_e
    for (int base = e0; base < e1; base += W) {
        const int e = base + lane;
        const int c = e < e1 ? colidx[e] : 0;
        const int cnt = min(W, e1 - base);
        float mine = 0.f;
        for (int j = 0; j < cnt; ++j) {
            const int cj = __shfl(c, j, 64);
            const float *vr = v + (int64_t)cj * ld;
            float p = 0.f;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int col = it * W + lane;
                if (col < dim) p = fmaf(d[it], vr[col], p);
            }
            p = wave_sum(p);
            if (lane == j) mine = p;
        }
        if (e < e1) {
            dz[e] = mine;                  // d alpha_e, revisited by the same lane below
            csum += alpha[e] * mine;
        }
    }
    csum = wave_sum(csum);
    for (int e = e0 + lane; e < e1; e += W) {
        const float de = alpha[e] * (dz[e] - csum);
        dz[e] = de * (z[e] > 0.f ? 1.f : slope);
    }
}

// dV[j] = sum over edges (i -> j) of alpha_e * dOut[row_e]; transposed CSR gives, per column j, its
// edges as (source output row, edge id in segment order).
template <int IT>
__global__ __launch_bounds__(256) void attn_bwd_v_kernel(const int32_t *__restrict__ t_ptr, const int32_t *__restrict__ t_row,
                                                         const int32_t *__restrict__ t_edge, int64_t n_cols,
                                                         const float *__restrict__ alpha, const float *__restrict__ dout,
                                                         int dim, int ld, float *__restrict__ dv) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= n_cols) return;
    const int e0 = t_ptr[j], e1 = t_ptr[j + 1];
    float acc[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) acc[it] = 0.f;
    for (int base = e0; base < e1; base += W) {
        const int e = base + lane;
        float a = 0.f;
        int r = 0;
        if (e < e1) { a = alpha[t_edge[e]]; r = t_row[e]; }
        const int cnt = min(W, e1 - base);
        for (int q = 0; q < cnt; ++q) {
            const float aq = __shfl(a, q, 64);
            const int rq = __shfl(r, q, 64);
            const float *dr = dout + (int64_t)rq * ld;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int col = it * W + lane;
                if (col < dim) acc[it] = fmaf(aq, dr[col], acc[it]);
            }
        }
    }
    float *o = dv + j * ld;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int col = it * W + lane;
        if (col < ld) o[col] = col < dim ? acc[it] : 0.f;
    }
}

#define OEA_ATTN_DISPATCH(ld, CALL)                                      \
    do {                                                                 \
        if ((ld) <= 128) { CALL(2); }                                    \
        else if ((ld) <= 256) { CALL(4); }                               \
        else if ((ld) <= 512) { CALL(8); }                               \
        else if ((ld) <= 1280) { CALL(20); }                             \
        else { oea::set_error("ld %d > 1280 unsupported", (int)(ld)); return OEA_EUNSUPPORTED; } \
    } while (0)

}  // namespace

extern "C" {

int oea_sparse_attn_fwd(const int32_t *seg_ptr, const int32_t *seg_row, int64_t n_seg, const int32_t *colidx,
                        const float *z, const float *v, int32_t dim, int32_t ld, float lrelu_slope, int32_t unique_rows,
                        float *out, float *alpha, void *stream) {
    OEA_REQUIRE(seg_ptr && seg_row && colidx && z && v && out && alpha, "null pointer");
    OEA_REQUIRE(dim > 0 && dim <= ld && ld % 4 == 0, "dim <= ld, ld % 4 == 0");
    if (n_seg == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
    const unsigned grid = (unsigned)oea::ceil_div(n_seg, 4);
#define CALL(IT) attn_fwd_kernel<IT><<<grid, 256, 0, st>>>(seg_ptr, seg_row, n_seg, colidx, z, v, dim, ld, lrelu_slope, out, alpha, unique_rows)
    OEA_ATTN_DISPATCH(ld, CALL);
#undef CALL
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_sparse_attn_bwd(const int32_t *seg_ptr, const int32_t *seg_row, int64_t n_seg, const int32_t *colidx,
                        const float *z, const float *v, const float *alpha, const float *dout, int32_t dim, int32_t ld,
                        float lrelu_slope, const int32_t *t_ptr, const int32_t *t_row, const int32_t *t_edge,
                        int64_t n_cols, float *dz, float *dv, void *stream) {
    OEA_REQUIRE(seg_ptr && seg_row && colidx && z && v && alpha && dout && t_ptr && t_row && t_edge && dz && dv, "null pointer");
    OEA_REQUIRE(dim > 0 && dim <= ld && ld % 4 == 0, "dim <= ld, ld % 4 == 0");
    hipStream_t st = oea::as_stream(stream);
    if (n_seg > 0) {
        const unsigned grid = (unsigned)oea::ceil_div(n_seg, 4);
#define CALL(IT) attn_bwd_edges_kernel<IT><<<grid, 256, 0, st>>>(seg_ptr, seg_row, n_seg, colidx, z, v, alpha, dout, dim, ld, lrelu_slope, dz)
        OEA_ATTN_DISPATCH(ld, CALL);
#undef CALL
    }
    if (n_cols > 0) {
        const unsigned grid = (unsigned)oea::ceil_div(n_cols, 4);
#define CALL(IT) attn_bwd_v_kernel<IT><<<grid, 256, 0, st>>>(t_ptr, t_row, t_edge, n_cols, alpha, dout, dim, ld, dv)
        OEA_ATTN_DISPATCH(ld, CALL);
#undef CALL
    }
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
