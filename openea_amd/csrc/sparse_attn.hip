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
//   forward : K1 sub (max, sum exp; a segment that is ONE sub-segment gets its alphas right here)
//             ->  K2 segment (M, L)  ->  K3 sub: alpha = exp(e-M)/L (both only when some segment has
//             several sub-segments)  ->  out = P . V with P = the alphas as a CSR over the OUTPUT
//             rows (slots in edge order inside a row; the edge order itself when it is canonical):
//             csrc/spmm.hip's aggregate -- short rows serially in slot order, long rows by the
//             workgroup's groups combined in group order, hub rows in chunks whose partial sums are
//             added in chunk order.  NO atomics anywhere: two runs give identical bits, whatever
//             the grouping (round 3; the round-2 kernels added sub-segment / split-column partials
//             with fp32 atomics and differed from run to run).
//   backward: B1 sub: d alpha_e = dOut_row . V_col (one wave reduction per edge), partial
//             c = sum alpha d alpha -> B2 segment c -> B3 sub: d z_e = alpha_e (d alpha_e - c) lrelu'(z_e);
//             B4: dV = P^T . dOut, the same aggregate over the transposed CSR with the stored alphas.
// Bytes per launch (forward aggregate): nnz*(12 + 4*d) + 4*N*d  (SpMM bytes + nnz*4 logits).
//
// Row-sharded jobs (one process per GPU): every rank holds the whole graph and runs the PHASES on its
// own ranges -- softmax statistics / alphas / d z for a block of segments (a contiguous edge range in
// either grouping), the aggregate for a block of output rows, dV for a block of columns -- with one
// all-gather between the phases (models/graph_ops.py).
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
// reductions over the G consecutive lanes that share a sub-segment (G = 1: one thread per sub-segment -- graphs whose
// softmax groups are a handful of edges, e.g. TF1's run grouping on a column-major adjacency; G = 64: one wave)
template <int G>
__device__ __forceinline__ float grp_max(float v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
template <int G>
__device__ __forceinline__ float grp_sum(float v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// K1: per sub-segment softmax statistics.  A segment that consists of this one sub-segment is finished here:
// its (M, L) and its alphas are written, K2 / K3 never look at it.
template <int G>
__global__ __launch_bounds__(256) void attn_sub_stats_kernel(const int32_t *__restrict__ sub_ptr,
                                                             const int32_t *__restrict__ sub_seg,
                                                             const int32_t *__restrict__ seg_sub_ptr, int64_t sub0,
                                                             int64_t sub1, const float *__restrict__ z, float slope,
                                                             float *__restrict__ sub_m, float *__restrict__ sub_l,
                                                             float *__restrict__ seg_m, float *__restrict__ seg_l,
                                                             float *__restrict__ alpha) {
    const int lane = threadIdx.x % G;
    const int64_t s = sub0 + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (s >= sub1) return;                                      // whole groups leave together (256 % G == 0)
    const int e0 = sub_ptr[s], e1 = sub_ptr[s + 1];
    float m = -INFINITY;
    for (int e = e0 + lane; e < e1; e += G) m = fmaxf(m, lrelu(z[e], slope));
    m = grp_max<G>(m);
    float l = 0.f;
    for (int e = e0 + lane; e < e1; e += G) l += expf(lrelu(z[e], slope) - m);
    l = grp_sum<G>(l);
    const int g = sub_seg[s];
    const bool single = seg_sub_ptr[g + 1] - seg_sub_ptr[g] == 1;
    if (lane == 0) {
        sub_m[s] = m; sub_l[s] = l;
        if (single) { seg_m[g] = m; seg_l[g] = l; }
    }
    if (single) {
        const float inv_l = 1.0f / l;
        for (int e = e0 + lane; e < e1; e += G) alpha[e] = expf(lrelu(z[e], slope) - m) * inv_l;
    }
}

// K2: per segment (M, L) from its consecutive sub-segments.  One lane per segment for the usual few sub-segments; a hub
// segment (power-law degrees: hundreds of sub-segments) is combined by the WHOLE wave, lanes striding over its
// sub-segments (it was one thread's serial loop: 270 us of a 1.7 ms attention on the hub graph).
constexpr int kSerialSubs = 8;
__global__ __launch_bounds__(256) void attn_seg_combine_kernel(const int32_t *__restrict__ seg_sub_ptr, int64_t seg0,
                                                               int64_t seg1, const float *__restrict__ sub_m,
                                                               const float *__restrict__ sub_l, float *__restrict__ seg_m,
                                                               float *__restrict__ seg_l) {
    const int lane = threadIdx.x & 63;
    const int64_t g = seg0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = g < seg1;
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

// K3: alphas of the sub-segments of segments with SEVERAL sub-segments (the others were finished by K1)
template <int G>
__global__ __launch_bounds__(256) void attn_alpha_kernel(const int32_t *__restrict__ sub_ptr, const int32_t *__restrict__ sub_seg,
                                                         const int32_t *__restrict__ seg_sub_ptr, int64_t sub0, int64_t sub1,
                                                         const float *__restrict__ z, float slope,
                                                         const float *__restrict__ seg_m, const float *__restrict__ seg_l,
                                                         float *__restrict__ alpha) {
    const int lane = threadIdx.x % G;
    const int64_t s = sub0 + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (s >= sub1) return;
    const int g = sub_seg[s];
    if (seg_sub_ptr[g + 1] - seg_sub_ptr[g] == 1) return;
    const float m = seg_m[g], inv_l = 1.0f / seg_l[g];
    for (int e = sub_ptr[s] + lane; e < sub_ptr[s + 1]; e += G) alpha[e] = expf(lrelu(z[e], slope) - m) * inv_l;
}

__global__ __launch_bounds__(256) void attn_fill_kernel(float *__restrict__ dst, int64_t i0, int64_t i1, float v) {
    const int64_t i = i0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < i1) dst[i] = v;
}

// values of the aggregate's CSR slots [slot0, slot1): dst[slot] = alpha[edge_of_slot[slot]]
__global__ __launch_bounds__(256) void attn_slot_values_kernel(const int32_t *__restrict__ edge_of_slot, int64_t slot0,
                                                               int64_t slot1, const float *__restrict__ alpha,
                                                               float *__restrict__ dst) {
    const int64_t i = slot0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < slot1) dst[i] = alpha[edge_of_slot[i]];
}

// B1: d alpha_e (stored in dz) and the sub-segment's partial c = sum alpha_e d alpha_e
template <int IT>
__global__ __launch_bounds__(256) void attn_bwd_dalpha_kernel(const int32_t *__restrict__ sub_ptr,
                                                              const int32_t *__restrict__ sub_seg,
                                                              const int32_t *__restrict__ seg_sub_ptr,
                                                              const int32_t *__restrict__ seg_row, int64_t sub0,
                                                              int64_t sub1, const int32_t *__restrict__ colidx,
                                                              const float *__restrict__ v, const float *__restrict__ alpha,
                                                              const float *__restrict__ dout, int dim, int ld,
                                                              float *__restrict__ dz, float *__restrict__ sub_c) {
    const int lane = threadIdx.x & 63;
    const int64_t s = sub0 + (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= sub1) return;
    const int e0 = sub_ptr[s], e1 = sub_ptr[s + 1];
    const int g = sub_seg[s];
    // a segment of ONE edge: alpha = 1 exactly, c = alpha * d alpha = d alpha, d z = alpha (d alpha - c) = 0 exactly --
    // no need for the dot product (AliNet's column-major 2-hop adjacency under the 'runs' grouping is all such segments)
    if (e1 - e0 == 1 && seg_sub_ptr[g + 1] - seg_sub_ptr[g] == 1) {
        if (lane == 0) { dz[e0] = 0.f; sub_c[s] = 0.f; }
        return;
    }
    const float *dor = dout + (int64_t)seg_row[g] * ld;
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

// B1, rows as float4: SIXTEEN lanes per edge (four edges of the sub-segment in flight per wave), every lane 16 B of the gathered
// V row per load and IT4 loads in flight, the dot product finished by a 16-lane reduction.  The wave-per-edge form above issues
// ld / 64 four-byte loads and a six-step wave reduction for EVERY edge: at d = 400 (AliNet's middle layer, 4.1 M two-hop edges)
// it was 2.4 ms of the 3.6 ms backward beside a forward aggregate that gathers the same rows in 1.07 ms.
// MASKED: dim % 4 != 0 -- the float4 that straddles dim is cut per element on both operands.
template <int IT4, bool MASKED>
__global__ __launch_bounds__(256) void attn_bwd_dalpha_rows_kernel(const int32_t *__restrict__ sub_ptr,
                                                                   const int32_t *__restrict__ sub_seg,
                                                                   const int32_t *__restrict__ seg_sub_ptr,
                                                                   const int32_t *__restrict__ seg_row, int64_t sub0,
                                                                   int64_t sub1, const int32_t *__restrict__ colidx,
                                                                   const float *__restrict__ v, const float *__restrict__ alpha,
                                                                   const float *__restrict__ dout, int dim, int ld,
                                                                   float *__restrict__ dz, float *__restrict__ sub_c) {
    const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4;
    const int64_t s = sub0 + (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= sub1) return;
    const int e0 = sub_ptr[s], e1 = sub_ptr[s + 1];
    const int g = sub_seg[s];
    if (e1 - e0 == 1 && seg_sub_ptr[g + 1] - seg_sub_ptr[g] == 1) {     // a one-edge segment: d z = 0 exactly (see above)
        if (lane == 0) { dz[e0] = 0.f; sub_c[s] = 0.f; }
        return;
    }
    auto cut = [&](float4 x, int col) {
        if (MASKED) {
            if (col + 1 >= dim) x.y = 0.f;
            if (col + 2 >= dim) x.z = 0.f;
            if (col + 3 >= dim) x.w = 0.f;
        }
        return x;
    };
    const float *dor = dout + (int64_t)seg_row[g] * ld;
    float4 d[IT4];
#pragma unroll
    for (int it = 0; it < IT4; ++it) {
        const int col = (it * 16 + l16) * 4;
        d[it] = col < dim ? cut(oea::ld4(dor + col), col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float csum = 0.f;
    for (int chunk = e0; chunk < e1; chunk += W) {
        const int cc = chunk + lane < e1 ? colidx[chunk + lane] : 0;           // the chunk's columns: one coalesced read
        const int cnt = min(W, e1 - chunk);
        for (int base = 0; base < cnt; base += 4) {
            const int j = base + grp;
            const bool on = j < cnt;
            const int c = __shfl(cc, on ? j : 0, 64);
            float p = 0.f;
            if (on) {
                const float *vr = v + (int64_t)c * ld;
                float4 x[IT4];
#pragma unroll
                for (int it = 0; it < IT4; ++it) {
                    const int col = (it * 16 + l16) * 4;
                    x[it] = col < dim ? oea::ld4(vr + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int it = 0; it < IT4; ++it) {
                    const float4 y = cut(x[it], (it * 16 + l16) * 4);
                    p = fmaf(d[it].x, y.x, p); p = fmaf(d[it].y, y.y, p); p = fmaf(d[it].z, y.z, p); p = fmaf(d[it].w, y.w, p);
                }
            }
            p = grp_sum<16>(p);
            if (on && l16 == 0) {
                const int e = chunk + j;
                dz[e] = p;
                csum += alpha[e] * p;
            }
        }
    }
    csum = wave_sum(csum);                                                       // (only the groups' first lanes hold terms)
    if (lane == 0) sub_c[s] = csum;
}

// B2 + B3: segment c (fixed order over its sub-segments), then d z of this sub-segment
template <int G>
__global__ __launch_bounds__(256) void attn_bwd_dz_kernel(const int32_t *__restrict__ sub_ptr, const int32_t *__restrict__ sub_seg,
                                                          const int32_t *__restrict__ seg_sub_ptr, int64_t sub0,
                                                          int64_t sub1, const float *__restrict__ z,
                                                          const float *__restrict__ alpha, const float *__restrict__ sub_c,
                                                          float slope, float *__restrict__ dz) {
    const int lane = threadIdx.x % G;
    const int64_t s = sub0 + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (s >= sub1) return;
    const int g = sub_seg[s];
    const int q0 = seg_sub_ptr[g], q1 = seg_sub_ptr[g + 1];
    float c = 0.f;
    if (q1 - q0 <= kSerialSubs) {
        for (int q = q0; q < q1; ++q) c += sub_c[q];
    } else {                                                     // hub segment: the group's lanes stride over its sub-segments
        for (int q = q0 + lane; q < q1; q += G) c += sub_c[q];
        c = grp_sum<G>(c);
    }
    for (int e = sub_ptr[s] + lane; e < sub_ptr[s + 1]; e += G) {
        const float de = alpha[e] * (dz[e] - c);
        dz[e] = de * (z[e] > 0.f ? 1.f : slope);
    }
}

// B1': the sub-segment's partial c = sum alpha_e d alpha_e from d alpha GIVEN per edge (in dz) -- oea_sparse_attn_dz: the softmax
// groups' edges do not share an output row there (values re-attached to another pattern), so the dots come from oea_pair_dots
template <int G>
__global__ __launch_bounds__(256) void attn_sub_c_kernel(const int32_t *__restrict__ sub_ptr, int64_t sub0, int64_t sub1,
                                                         const float *__restrict__ alpha, const float *__restrict__ dalpha,
                                                         float *__restrict__ sub_c) {
    const int lane = threadIdx.x % G;
    const int64_t s = sub0 + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (s >= sub1) return;
    float c = 0.f;
    for (int e = sub_ptr[s] + lane; e < sub_ptr[s + 1]; e += G) c += alpha[e] * dalpha[e];
    c = grp_sum<G>(c);
    if (lane == 0) sub_c[s] = c;
}

#define OEA_ATTN_DISPATCH(ld, CALL)                                      \
    do {                                                                 \
        if ((ld) <= 128) { CALL(2); }                                    \
        else if ((ld) <= 256) { CALL(4); }                               \
        else if ((ld) <= 512) { CALL(8); }                               \
        else if ((ld) <= 1280) { CALL(20); }                             \
        else { oea::set_error("ld %d > 1280 unsupported", (int)(ld)); return OEA_EUNSUPPORTED; } \
    } while (0)

// lanes per sub-segment of the element-wise kernels, from the average sub-segment length (a property of the graph: the
// same on every rank of a sharded job and in every call, so the summation order -- and the bits -- never change)
static int group_width(const oea_attn_graph *g) {
    const double avg = g->n_sub > 0 ? (double)g->nnz / (double)g->n_sub : 0.0;
    return avg <= 3.0 ? 1 : (avg <= 24.0 ? 8 : 64);
}

static int check_graph(const oea_attn_graph *g) {
    OEA_REQUIRE(g, "attention graph: null pointer");
    OEA_REQUIRE(g->n_sub >= 0 && g->n_seg >= 0 && g->n_sub >= g->n_seg && g->nnz >= g->n_seg, "nnz >= n_sub >= n_seg >= 0");
    OEA_REQUIRE(g->n_sub == 0 || (g->sub_ptr && g->sub_seg && g->seg_sub_ptr && g->seg_row && g->colidx),
                "attention graph: null pointer");
    OEA_REQUIRE(0 <= g->sub0 && g->sub0 <= g->sub1 && g->sub1 <= g->n_sub && 0 <= g->seg0 && g->seg0 <= g->seg1 &&
                g->seg1 <= g->n_seg, "segment / sub-segment range");
    OEA_REQUIRE(0 <= g->agg_row0 && g->agg_row0 <= g->agg_row1 && g->agg_row1 <= g->agg_rows && 0 <= g->agg_slot0 &&
                g->agg_slot0 <= g->agg_slot1, "aggregate row / slot range");
    OEA_REQUIRE(0 <= g->t_row0 && g->t_row0 <= g->t_row1 && g->t_row1 <= g->t_rows && 0 <= g->t_slot0 &&
                g->t_slot0 <= g->t_slot1, "transposed row / slot range");
    return OEA_OK;
}

}  // namespace

extern "C" {

size_t oea_sparse_attn_workspace_floats(const oea_attn_graph *g) {
    if (!g) return 0;
    const int64_t slots = g->agg_slot1 > g->t_slot1 ? g->agg_slot1 : g->t_slot1;
    return (size_t)(2 * g->n_sub + 2 * g->n_seg + slots + 256);
}

int oea_sparse_attn_fwd(const oea_attn_graph *g, const float *z, const float *v, int32_t dim, int32_t ld,
                        float lrelu_slope, float *out, float *alpha, float *workspace, int32_t phases, void *stream) {
    const int rc = check_graph(g);
    if (rc != OEA_OK) return rc;
    OEA_REQUIRE(alpha && workspace, "null pointer");
    OEA_REQUIRE(dim > 0 && dim <= ld && ld % 4 == 0, "dim <= ld, ld % 4 == 0");
    OEA_REQUIRE(phases & (OEA_ATTN_ALPHA | OEA_ATTN_AGGREGATE), "phases: OEA_ATTN_ALPHA | OEA_ATTN_AGGREGATE");
    hipStream_t st = oea::as_stream(stream);
    float *sub_m = workspace, *sub_l = sub_m + g->n_sub, *seg_m = sub_l + g->n_sub, *seg_l = seg_m + g->n_seg;
    float *slot_vals = seg_l + g->n_seg;
    if ((phases & OEA_ATTN_ALPHA) && g->sub1 > g->sub0) {
        OEA_REQUIRE(z, "null pointer");
        const int64_t ns = g->sub1 - g->sub0;
        if (g->n_seg == g->nnz) {           // every softmax group is ONE edge: alpha = exp(0) / 1 = 1, nothing to compute
            OEA_REQUIRE(g->sub_ptr, "null pointer");
            attn_fill_kernel<<<(unsigned)oea::ceil_div(ns, 256), 256, 0, st>>>(alpha, g->sub0, g->sub1, 1.0f);   // sub-segment id == edge id
        } else {
            const int G = group_width(g);
#define CALL(GW)                                                                                                        \
    do {                                                                                                                \
        const unsigned grid = (unsigned)oea::ceil_div(ns * GW, 256);                                                    \
        attn_sub_stats_kernel<GW><<<grid, 256, 0, st>>>(g->sub_ptr, g->sub_seg, g->seg_sub_ptr, g->sub0, g->sub1, z,    \
                                                        lrelu_slope, sub_m, sub_l, seg_m, seg_l, alpha);               \
        if (g->n_sub > g->n_seg) { /* some segment has several sub-segments */                                         \
            attn_seg_combine_kernel<<<(unsigned)oea::ceil_div(g->seg1 - g->seg0, 256), 256, 0, st>>>(                   \
                g->seg_sub_ptr, g->seg0, g->seg1, sub_m, sub_l, seg_m, seg_l);                                          \
            attn_alpha_kernel<GW><<<grid, 256, 0, st>>>(g->sub_ptr, g->sub_seg, g->seg_sub_ptr, g->sub0, g->sub1, z,    \
                                                        lrelu_slope, seg_m, seg_l, alpha);                              \
        }                                                                                                               \
    } while (0)
            if (G == 1) CALL(1); else if (G == 8) CALL(8); else CALL(64);
#undef CALL
        }
        OEA_CHECK_HIP(hipGetLastError());
    }
    if ((phases & OEA_ATTN_AGGREGATE) && g->agg_row1 > g->agg_row0) {
        OEA_REQUIRE(v && out && g->agg_rowptr && g->agg_colidx, "null pointer");
        const float *vals = alpha;
        if (g->agg_edge) {
            const int64_t n = g->agg_slot1 - g->agg_slot0;
            if (n > 0)
                attn_slot_values_kernel<<<(unsigned)oea::ceil_div(n, 256), 256, 0, st>>>(g->agg_edge, g->agg_slot0, g->agg_slot1,
                                                                                       alpha, slot_vals);
            vals = slot_vals;
        }
        return oea_spmm_csr(g->agg_rowptr + g->agg_row0, g->agg_colidx, vals, g->agg_row1 - g->agg_row0, v, dim, ld, 0, nullptr,
                            out + g->agg_row0 * (int64_t)ld, ld, g->agg_split, stream);
    }
    return OEA_OK;
}

int oea_sparse_attn_bwd(const oea_attn_graph *g, const float *z, const float *v, const float *alpha, const float *dout,
                        int32_t dim, int32_t ld, float lrelu_slope, float *dz, float *dv, float *workspace,
                        int32_t phases, void *stream) {
    const int rc = check_graph(g);
    if (rc != OEA_OK) return rc;
    OEA_REQUIRE(alpha && dout && workspace, "null pointer");
    OEA_REQUIRE(dim > 0 && dim <= ld && ld % 4 == 0, "dim <= ld, ld % 4 == 0");
    OEA_REQUIRE(phases & (OEA_ATTN_DZ | OEA_ATTN_DV), "phases: OEA_ATTN_DZ | OEA_ATTN_DV");
    hipStream_t st = oea::as_stream(stream);
    float *sub_c = workspace, *slot_vals = workspace + 2 * g->n_sub + 2 * g->n_seg;
    if ((phases & OEA_ATTN_DZ) && g->sub1 > g->sub0 && g->n_seg == g->nnz) {
        // every softmax group is ONE edge: d z = alpha (d alpha - alpha d alpha) = 0 exactly
        OEA_REQUIRE(dz, "null pointer");
        attn_fill_kernel<<<(unsigned)oea::ceil_div(g->sub1 - g->sub0, 256), 256, 0, st>>>(dz, g->sub0, g->sub1, 0.0f);
        OEA_CHECK_HIP(hipGetLastError());
    } else if ((phases & OEA_ATTN_DZ) && g->sub1 > g->sub0) {
        OEA_REQUIRE(z && v && dz, "null pointer");
        const unsigned grid = (unsigned)oea::ceil_div(g->sub1 - g->sub0, 4);
        static const bool rows16 = [] { const char *e = getenv("OEA_ATTN_BWD_ROWS"); return !(e && e[0] == '0'); }();
        const int it4 = (int)oea::ceil_div((int64_t)oea::ceil_div((int64_t)dim, 4), 16);
        if (rows16 && it4 <= 20 && (((uintptr_t)v | (uintptr_t)dout) & 15) == 0) {
#define ROWS(N, M)                                                                                                                  \
    attn_bwd_dalpha_rows_kernel<N, M><<<grid, 256, 0, st>>>(g->sub_ptr, g->sub_seg, g->seg_sub_ptr, g->seg_row, g->sub0, g->sub1, g->colidx, \
                                                            v, alpha, dout, dim, ld, dz, sub_c)
#define ROWS_IT(N) do { if (dim & 3) ROWS(N, true); else ROWS(N, false); } while (0)
            if (it4 <= 1) ROWS_IT(1); else if (it4 <= 2) ROWS_IT(2); else if (it4 <= 4) ROWS_IT(4); else if (it4 <= 5) ROWS_IT(5);
            else if (it4 <= 7) ROWS_IT(7); else if (it4 <= 8) ROWS_IT(8); else if (it4 <= 10) ROWS_IT(10); else if (it4 <= 13) ROWS_IT(13);
            else if (it4 <= 16) ROWS_IT(16); else ROWS_IT(20);
#undef ROWS_IT
#undef ROWS
        } else {
#define CALL(IT)                                                                                                      \
    attn_bwd_dalpha_kernel<IT><<<grid, 256, 0, st>>>(g->sub_ptr, g->sub_seg, g->seg_sub_ptr, g->seg_row, g->sub0, g->sub1, g->colidx, v, alpha, \
                                                     dout, dim, ld, dz, sub_c)
            OEA_ATTN_DISPATCH(ld, CALL);
#undef CALL
        }
        const int G = group_width(g);
        const unsigned gz = (unsigned)oea::ceil_div((g->sub1 - g->sub0) * G, 256);
        if (G == 1) attn_bwd_dz_kernel<1><<<gz, 256, 0, st>>>(g->sub_ptr, g->sub_seg, g->seg_sub_ptr, g->sub0, g->sub1, z, alpha, sub_c, lrelu_slope, dz);
        else if (G == 8) attn_bwd_dz_kernel<8><<<gz, 256, 0, st>>>(g->sub_ptr, g->sub_seg, g->seg_sub_ptr, g->sub0, g->sub1, z, alpha, sub_c, lrelu_slope, dz);
        else attn_bwd_dz_kernel<64><<<gz, 256, 0, st>>>(g->sub_ptr, g->sub_seg, g->seg_sub_ptr, g->sub0, g->sub1, z, alpha, sub_c, lrelu_slope, dz);
        OEA_CHECK_HIP(hipGetLastError());
    }
    if ((phases & OEA_ATTN_DV) && g->t_row1 > g->t_row0) {
        OEA_REQUIRE(dv && g->t_rowptr && g->t_row && g->t_edge, "attention graph: transposed lists missing");
        const int64_t n = g->t_slot1 - g->t_slot0;
        if (n > 0)
            attn_slot_values_kernel<<<(unsigned)oea::ceil_div(n, 256), 256, 0, st>>>(g->t_edge, g->t_slot0, g->t_slot1, alpha,
                                                                                   slot_vals);
        return oea_spmm_csr(g->t_rowptr + g->t_row0, g->t_row, slot_vals, g->t_row1 - g->t_row0, dout, dim, ld, 0, nullptr,
                            dv + g->t_row0 * (int64_t)ld, ld, g->t_split, stream);
    }
    return OEA_OK;
}

int oea_sparse_attn_dz(const oea_attn_graph *g, const float *z, const float *alpha, float *dz, float lrelu_slope, float *workspace,
                       void *stream) {
    const int rc = check_graph(g);
    if (rc != OEA_OK) return rc;
    OEA_REQUIRE(z && alpha && dz && workspace, "null pointer");
    if (g->sub1 <= g->sub0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
    float *sub_c = workspace;
    const int G = group_width(g);
    const unsigned gz = (unsigned)oea::ceil_div((g->sub1 - g->sub0) * G, 256);
#define CALL(GW)                                                                                                                          \
    do {                                                                                                                                  \
        attn_sub_c_kernel<GW><<<gz, 256, 0, st>>>(g->sub_ptr, g->sub0, g->sub1, alpha, dz, sub_c);                                        \
        attn_bwd_dz_kernel<GW><<<gz, 256, 0, st>>>(g->sub_ptr, g->sub_seg, g->seg_sub_ptr, g->sub0, g->sub1, z, alpha, sub_c, lrelu_slope, dz); \
    } while (0)
    if (G == 1) CALL(1); else if (G == 8) CALL(8); else CALL(64);
#undef CALL
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
