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
// orders): segment s owns a consecutive edge range and adds its aggregate to output row
// seg_row[s].  Whole-row semantics = one segment per row.
//
// Degrees are power-law (hubs with 10^3..10^5 two-hop neighbours), so a segment is cut into
// SUB-SEGMENTS of at most kSubEdges edges on the host and every kernel works wave-per-sub-segment
// (bounded, balanced work); segment-wide quantities are combined by tiny per-segment kernels:
//   forward : K1 sub (max, sum exp)  ->  K2 segment (M, L)  ->  K3 sub: alpha = exp(e-M)/L,
//             partial aggregate with the 64 lanes across the feature columns (coalesced 256-B
//             gathers of V rows, alpha broadcast by __shfl), added to the output row (plain store
//             when the row has a single sub-segment, hardware fp32 atomics otherwise)
//   backward: B1 sub: d alpha_e = dOut_row . V_col (one wave reduction per edge), partial
//             c = sum alpha d alpha -> B2 segment c -> B3 sub: d z_e = alpha_e (d alpha_e - c) lrelu'(z_e);
//             B4: dV = transposed aggregate over column chunks with the stored alphas.
// Bytes per launch (forward K3): nnz*(12 + 4*d) + 4*N*d  (SpMM bytes + nnz*4 logits).
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

// K1: per sub-segment softmax statistics
__global__ __launch_bounds__(256) void attn_sub_stats_kernel(const int32_t *__restrict__ sub_ptr, int64_t n_sub,
                                                             const float *__restrict__ z, float slope,
                                                             float *__restrict__ sub_m, float *__restrict__ sub_l) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= n_sub) return;
    const int e0 = sub_ptr[s], e1 = sub_ptr[s + 1];
    float m = -INFINITY;
    for (int e = e0 + lane; e < e1; e += W) m = fmaxf(m, lrelu(z[e], slope));
    m = wave_max(m);
    float l = 0.f;
    for (int e = e0 + lane; e < e1; e += W) l += expf(lrelu(z[e], slope) - m);
    l = wave_sum(l);
    if (lane == 0) { sub_m[s] = m; sub_l[s] = l; }
}

// K2: per segment (M, L) from its consecutive sub-segments.  One lane per segment for the usual few sub-segments; a hub
// segment (power-law degrees: hundreds of sub-segments) is combined by the WHOLE wave, lanes striding over its
// sub-segments (it was one thread's serial loop: 270 us of a 1.7 ms attention on the hub graph).
constexpr int kSerialSubs = 8;
__global__ __launch_bounds__(256) void attn_seg_combine_kernel(const int32_t *__restrict__ seg_sub_ptr, int64_t n_seg,
                                                               const float *__restrict__ sub_m, const float *__restrict__ sub_l,
                                                               float *__restrict__ seg_m, float *__restrict__ seg_l) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = g < n_seg;
    const int s0 = valid ? seg_sub_ptr[g] : 0, s1 = valid ? seg_sub_ptr[g + 1] : 0;
    const bool hub = s1 - s0 > kSerialSubs;
    if (valid && !hub) {
        float m = -INFINITY;
        for (int s = s0; s < s1; ++s) m = fmaxf(m, sub_m[s]);
        float l = 0.f;
        for (int s = s0; s < s1; ++s) l += sub_l[s] * expf(sub_m[s] - m);
        seg_m[g] = m;
        seg_l[g] = l;
    }
    unsigned long long todo = __ballot(hub);
    while (todo) {                                               // wave-uniform loop over this wave's hub segments
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int a = __shfl(s0, src, 64), b = __shfl(s1, src, 64);
        float m = -INFINITY;
        for (int s = a + lane; s < b; s += W) m = fmaxf(m, sub_m[s]);
        m = wave_max(m);
        float l = 0.f;
        for (int s = a + lane; s < b; s += W) l += sub_l[s] * expf(sub_m[s] - m);
        l = wave_sum(l);
        if (lane == src) { seg_m[g] = m; seg_l[g] = l; }
    }
}

// K3: alpha + partial aggregate of one sub-segment
template <int IT>
__global__ __launch_bounds__(256) void attn_aggregate_kernel(const int32_t *__restrict__ sub_ptr,
                                                             const int32_t *__restrict__ sub_seg,
                                                             const int32_t *__restrict__ seg_sub_ptr,
                                                             const int32_t *__restrict__ seg_row, int64_t n_sub,
                                                             const int32_t *__restrict__ colidx, const float *__restrict__ z,
                                                             const float *__restrict__ v, int dim, int ld, float slope,
                                                             const float *__restrict__ seg_m, const float *__restrict__ seg_l,
                                                             float *__restrict__ out, float *__restrict__ alpha,
                                                             int unique_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= n_sub) return;
    const int e0 = sub_ptr[s], e1 = sub_ptr[s + 1];
    if (e1 <= e0) return;
    const int g = sub_seg[s];
    const float m = seg_m[g], inv_l = 1.0f / seg_l[g];
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
    const bool single = unique_rows && (seg_sub_ptr[g + 1] - seg_sub_ptr[g] == 1);
    float *o = out + (int64_t)seg_row[g] * ld;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int col = it * W + lane;
        if (col < dim) {
            if (single) o[col] = acc[it];
            else oea::atomic_add_f32(o + col, acc[it]);
        }
    }
}

// B1: d alpha_e (stored in dz) and the sub-segment's partial c = sum alpha_e d alpha_e
template <int IT>
__global__ __launch_bounds__(256) void attn_bwd_dalpha_kernel(const int32_t *__restrict__ sub_ptr,
                                                              const int32_t *__restrict__ sub_seg,
                                                              const int32_t *__restrict__ seg_row, int64_t n_sub,
                                                              const int32_t *__restrict__ colidx, const float *__restrict__ v,
                                                              const float *__restrict__ alpha, const float *__restrict__ dout,
                                                              int dim, int ld, float *__restrict__ dz,
                                                              float *__restrict__ sub_c) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= n_sub) return;
    const int e0 = sub_ptr[s], e1 = sub_ptr[s + 1];
    const float *dor = dout + (int64_t)seg_row[sub_seg[s]] * ld;
    float d[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int col = it * W + lane;
        d[it] = col < dim ? dor[col] : 0.f;
    }
    float csum = 0.f;
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
            dz[e] = mine;
            csum += alpha[e] * mine;
        }
    }
    csum = wave_sum(csum);
    if (lane == 0) sub_c[s] = csum;
}

// B2 + B3: segment c (fixed order over its sub-segments), then d z of this sub-segment
__global__ __launch_bounds__(256) void attn_bwd_dz_kernel(const int32_t *__restrict__ sub_ptr, const int32_t *__restrict__ sub_seg,
                                                          const int32_t *__restrict__ seg_sub_ptr, int64_t n_sub,
                                                          const float *__restrict__ z, const float *__restrict__ alpha,
                                                          const float *__restrict__ sub_c, float slope,
                                                          float *__restrict__ dz) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= n_sub) return;
    const int g = sub_seg[s];
    const int q0 = seg_sub_ptr[g], q1 = seg_sub_ptr[g + 1];
    float c = 0.f;
    if (q1 - q0 <= kSerialSubs) {
        for (int q = q0; q < q1; ++q) c += sub_c[q];
    } else {                                                     // hub segment: lanes stride over its sub-segments
        for (int q = q0 + lane; q < q1; q += W) c += sub_c[q];
        c = wave_sum(c);
    }
    for (int e = sub_ptr[s] + lane; e < sub_ptr[s + 1]; e += W) {
        const float de = alpha[e] * (dz[e] - c);
        dz[e] = de * (z[e] > 0.f ? 1.f : slope);
    }
}

// B4: dV[col] += sum over the column chunk's incoming edges of alpha_e * dOut[row_e]
template <int IT>
__global__ __launch_bounds__(256) void attn_bwd_v_kernel(const int32_t *__restrict__ t_sub_ptr,
                                                         const int32_t *__restrict__ t_sub_col, int64_t n_tsub,
                                                         const int32_t *__restrict__ t_row, const int32_t *__restrict__ t_edge,
                                                         const float *__restrict__ alpha, const float *__restrict__ dout,
                                                         int dim, int ld, float *__restrict__ dv, int any_split) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= n_tsub) return;
    const int e0 = t_sub_ptr[s], e1 = t_sub_ptr[s + 1];
    if (e1 <= e0) return;
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
    float *o = dv + (int64_t)t_sub_col[s] * ld;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int col = it * W + lane;
        if (col < dim) {
            if (any_split) oea::atomic_add_f32(o + col, acc[it]);
            else o[col] = acc[it];
        }
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

static int check_graph(const oea_attn_graph *g) {
    OEA_REQUIRE(g && (g->n_sub == 0 || (g->sub_ptr && g->sub_seg && g->seg_sub_ptr && g->seg_row && g->colidx)),
                "attention graph: null pointer");
    OEA_REQUIRE(g->n_sub >= 0 && g->n_seg >= 0 && g->n_sub >= g->n_seg, "n_sub >= n_seg >= 0");
    return OEA_OK;
}

}  // namespace

extern "C" {

size_t oea_sparse_attn_workspace_floats(int64_t n_sub, int64_t n_seg) { return (size_t)(2 * n_sub + 2 * n_seg + 256); }

int oea_sparse_attn_fwd(const oea_attn_graph *g, const float *z, const float *v, int32_t dim, int32_t ld,
                        float lrelu_slope, float *out, float *alpha, float *workspace, void *stream) {
    const int rc = check_graph(g);
    if (rc != OEA_OK) return rc;
    OEA_REQUIRE(z && v && out && alpha && workspace, "null pointer");
    OEA_REQUIRE(dim > 0 && dim <= ld && ld % 4 == 0, "dim <= ld, ld % 4 == 0");
    if (g->n_sub == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
    float *sub_m = workspace, *sub_l = sub_m + g->n_sub, *seg_m = sub_l + g->n_sub, *seg_l = seg_m + g->n_seg;
    const unsigned grid = (unsigned)oea::ceil_div(g->n_sub, 4);
    attn_sub_stats_kernel<<<grid, 256, 0, st>>>(g->sub_ptr, g->n_sub, z, lrelu_slope, sub_m, sub_l);
    attn_seg_combine_kernel<<<(unsigned)oea::ceil_div(g->n_seg, 256), 256, 0, st>>>(g->seg_sub_ptr, g->n_seg, sub_m, sub_l,
                                                                                   seg_m, seg_l);
#define CALL(IT)                                                                                                     \
    attn_aggregate_kernel<IT><<<grid, 256, 0, st>>>(g->sub_ptr, g->sub_seg, g->seg_sub_ptr, g->seg_row, g->n_sub,    \
                                                    g->colidx, z, v, dim, ld, lrelu_slope, seg_m, seg_l, out, alpha, \
                                                    g->unique_rows)
    OEA_ATTN_DISPATCH(ld, CALL);
#undef CALL
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_sparse_attn_bwd(const oea_attn_graph *g, const float *z, const float *v, const float *alpha, const float *dout,
                        int32_t dim, int32_t ld, float lrelu_slope, float *dz, float *dv, float *workspace,
                        void *stream) {
    const int rc = check_graph(g);
    if (rc != OEA_OK) return rc;
    // a rank of a row-sharded job passes two partial graphs: its segments only (n_tsub = 0 -> dz of its edges) and
    // the transposed lists of its column block only (n_sub = 0 -> dv rows of that block, alpha / dout of ALL edges)
    OEA_REQUIRE(g->n_tsub == 0 || (g->t_sub_ptr && g->t_sub_col && g->t_row && g->t_edge), "attention graph: transposed lists missing");
    OEA_REQUIRE(v && alpha && dout && dv && workspace && (g->n_sub == 0 || (z && dz)), "null pointer");
    OEA_REQUIRE(dim > 0 && dim <= ld && ld % 4 == 0, "dim <= ld, ld % 4 == 0");
    hipStream_t st = oea::as_stream(stream);
    if (g->n_sub > 0) {
        float *sub_c = workspace;
        const unsigned grid = (unsigned)oea::ceil_div(g->n_sub, 4);
#define CALL(IT)                                                                                                    \
    attn_bwd_dalpha_kernel<IT><<<grid, 256, 0, st>>>(g->sub_ptr, g->sub_seg, g->seg_row, g->n_sub, g->colidx, v, alpha, \
                                                     dout, dim, ld, dz, sub_c)
        OEA_ATTN_DISPATCH(ld, CALL);
#undef CALL
        attn_bwd_dz_kernel<<<grid, 256, 0, st>>>(g->sub_ptr, g->sub_seg, g->seg_sub_ptr, g->n_sub, z, alpha, sub_c, lrelu_slope, dz);
    }
    if (g->n_tsub > 0) {
        const unsigned grid = (unsigned)oea::ceil_div(g->n_tsub, 4);
#define CALL(IT)                                                                                                  \
    attn_bwd_v_kernel<IT><<<grid, 256, 0, st>>>(g->t_sub_ptr, g->t_sub_col, g->n_tsub, g->t_row, g->t_edge, alpha, \
                                                dout, dim, ld, dv, g->t_any_split)
        OEA_ATTN_DISPATCH(ld, CALL);
#undef CALL
    }
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
