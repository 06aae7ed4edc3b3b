// gnn_fused.hip -- the row-wise glue of the GNN approaches, fused (forward + hand-written backward).
//
// AliNet's epoch at the EN-DE-100K shape was ~45 % torch element-wise kernels (rocprofv3, profiles/r03_*): every
// l2_normalize / concat / gate / loss term of the reference's graph (approaches/alinet.py) was 3-6 passes over
// [200,000, 300..1,200] fp32 tensors, and autograd doubled them.  The kernels here do each of those blocks in ONE pass
// forward and ONE pass backward, one wave per row, no atomics (fixed summation order -> reproducible bits):
//
//   concat_l2n   emb = l2n(concat(l2n(x_0), ..., l2n(x_{K-1})))      alinet.py:835-840 (training) / :932-943 (evaluation)
//   pair_loss    sum ||e_i - e_j||^2 over positive links + balance * sum w relu(margin - ||e_i - e_j||^2) over negative
//                links, and its gradient w.r.t. the embedding rows    alinet.py:828-850 (compute_loss)
//   highway      gate = relu(tanh(p)); out = tanh(b' (1 - gate) + a' gate), a' / b' = BatchNorm-affine of a / b
//                                                                     alinet.py:597-622 (HighwayLayer.call)
//   bias_tanh    y = tanh(x + bias)                                   alinet.py:583-590 (GraphConvolution.call tail)
//
// l2_normalize(x) = x * rsqrt(max(sum x^2, 1e-12)) (TF1); relu'(0) = 0.
#include "common.h"

namespace {

constexpr int W = 64;

__device__ __forceinline__ float wsum(float v) { return oea::group_sum<64>(v); }

struct Blocks {                     // up to 4 row blocks that are concatenated
    const float *x[4];
    float *dx[4];
    int dim[4], ld[4], off[4];
    int k;
};

// ---- concat_l2n ---------------------------------------------------------------------------------------------------
// one wave per row; the row's K blocks are normalised one by one, the concatenation is normalised again
template <int IT>
__global__ __launch_bounds__(256) void concat_l2n_fwd_kernel(Blocks b, int64_t n, float *__restrict__ out, int ld_out,
                                                             float *__restrict__ inv_blk, float *__restrict__ inv_all) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float y[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) y[it] = 0.f;
    float tot = 0.f;
    for (int k = 0; k < b.k; ++k) {
        const float *xr = b.x[k] + row * b.ld[k];
        float ss = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * W + lane - b.off[k];
            if (c >= 0 && c < b.dim[k]) { const float v = xr[c]; y[it] = v; ss += v * v; }
        }
        ss = wsum(ss);
        const float inv = rsqrtf(fmaxf(ss, 1e-12f));
        if (lane == 0) inv_blk[row * 4 + k] = inv;
        float s2 = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * W + lane - b.off[k];
            if (c >= 0 && c < b.dim[k]) { y[it] *= inv; s2 += y[it] * y[it]; }
        }
        tot += s2;
    }
    tot = wsum(tot);
    const float inv = rsqrtf(fmaxf(tot, 1e-12f));
    if (lane == 0) inv_all[row] = inv;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = it * W + lane;
        if (c < ld_out) out[row * ld_out + c] = y[it] * inv;       // pad columns: y = 0
    }
}

// dz -> dx_k:  z = y s (s = inv_all),  dy = s (g - z (z . g))  [0 if the clamp was active];  y_k = x_k t_k:
// dx_k = t_k (dy_k - y_k (y_k . dy_k))
template <int IT>
__global__ __launch_bounds__(256) void concat_l2n_bwd_kernel(Blocks b, int64_t n, const float *__restrict__ z, const float *__restrict__ g,
                                                             int ld_out, const float *__restrict__ inv_blk,
                                                             const float *__restrict__ inv_all) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float zv[IT], dy[IT];
    float zg = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = it * W + lane;
        zv[it] = c < ld_out ? z[row * ld_out + c] : 0.f;
        dy[it] = c < ld_out ? g[row * ld_out + c] : 0.f;
        zg += zv[it] * dy[it];
    }
    zg = wsum(zg);
    const float s = inv_all[row];
    const bool clamped_all = s >= 1e6f;                           // rsqrt(1e-12) = 1e6: the clamp was active, z = y * 1e6, d/dy = 1e6
#pragma unroll
    for (int it = 0; it < IT; ++it) dy[it] = clamped_all ? s * dy[it] : s * (dy[it] - zv[it] * zg);
    for (int k = 0; k < b.k; ++k) {
        const float t = inv_blk[row * 4 + k];
        const bool clamped = t >= 1e6f;
        float yd = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * W + lane - b.off[k];
            if (c >= 0 && c < b.dim[k]) yd += (zv[it] / s) * dy[it];              // y = z / s
        }
        yd = wsum(yd);
        float *dr = b.dx[k] + row * b.ld[k];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * W + lane - b.off[k];
            if (c >= 0 && c < b.dim[k]) dr[c] = clamped ? t * dy[it] : t * (dy[it] - (zv[it] / s) * yd);
        }
    }
}

// ---- pair loss ------------------------------------------------------------------------------------------------------
// one wave per pair p = (i, j): s = ||e_i - e_j||^2 (lane-strided columns, butterfly sum);
// p < n_pos: term = s, coef = 1;  else h = margin - s: term = balance w relu(h), coef = h > 0 ? -balance w : 0
template <int IT>
__global__ __launch_bounds__(256) void pair_loss_fwd_kernel(const float *__restrict__ emb, int dim, int ld,
                                                            const int32_t *__restrict__ pairs, int64_t m, int64_t n_pos,
                                                            const float *__restrict__ weight, float margin, float balance,
                                                            float *__restrict__ coef, float *__restrict__ terms) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= m) return;
    const float *a = emb + (int64_t)pairs[2 * p] * ld, *b = emb + (int64_t)pairs[2 * p + 1] * ld;
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = it * W + lane;
        if (c < dim) { const float d = a[c] - b[c]; s = fmaf(d, d, s); }
    }
    s = wsum(s);
    if (lane == 0) {
        if (p < n_pos) { terms[p] = s; coef[p] = 1.f; }
        else {
            const float w = weight ? weight[p - n_pos] : 1.f, h = margin - s;
            terms[p] = h > 0.f ? balance * w * h : 0.f;
            coef[p] = h > 0.f ? -balance * w : 0.f;
        }
    }
}

// one wave per embedding row r: grad[r] = gscale * sum over the row's pair slots (in slot order) of
//   L2: 2 coef (e_r - e_other)        (d / d e_r of coef ||e_r - e_other||^2)
//   L1: coef sign(e_r - e_other)      (d / d e_r of coef |e_r - e_other|_1; sign(0) = 0)
// -- the same expression whichever end of the pair r is.  Slots with coef == 0 (hinge inactive) are skipped; rows without
// slots get zeros.  No atomics.
template <int IT, bool L1>
__global__ __launch_bounds__(256) void pair_loss_bwd_kernel(const float *__restrict__ emb, int64_t n, int dim, int ld,
                                                            const int32_t *__restrict__ rowptr, const int32_t *__restrict__ other,
                                                            const int32_t *__restrict__ slot_pair, const float *__restrict__ coef,
                                                            const float *__restrict__ gscale, float *__restrict__ grad) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    float acc[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) acc[it] = 0.f;
    if (e1 > e0) {
        float me[IT];
        bool loaded = false;
        for (int base = e0; base < e1; base += W) {
            const int e = base + lane;
            float c = 0.f;
            int o = 0;
            if (e < e1) { c = coef[slot_pair[e]]; o = other[e]; }
            unsigned long long live = __ballot(c != 0.f);
            if (live && !loaded) {
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int col = it * W + lane;
                    me[it] = col < dim ? emb[row * ld + col] : 0.f;
                }
                loaded = true;
            }
            // the active slots of this batch of 64, in slot order, FOUR per trip: all their rows are requested before the
            // first is used (one slot per trip waited a full L2 round trip per pair: a seed row of RDGCN has 251 pairs --
            // 11.6 ms for the 100K loss); an empty place points at the row itself with coefficient 0, which adds exact zeros
            while (live) {
                float cq[4];
                const float *orow[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    cq[u] = 0.f;
                    orow[u] = emb + row * ld;
                    if (live) {
                        const int q = __ffsll((long long)live) - 1;
                        live &= live - 1;
                        cq[u] = (L1 ? 1.f : 2.f) * __shfl(c, q, 64);
                        orow[u] = emb + (int64_t)__shfl(o, q, 64) * ld;
                    }
                }
                float ov[4][IT];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int it = 0; it < IT; ++it) {
                        const int col = it * W + lane;
                        ov[u][it] = col < dim ? orow[u][col] : 0.f;
                    }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int it = 0; it < IT; ++it) {
                        const float df = me[it] - ov[u][it];
                        if (L1) acc[it] += df > 0.f ? cq[u] : (df < 0.f ? -cq[u] : 0.f);
                        else acc[it] = fmaf(cq[u], df, acc[it]);
                    }
            }
        }
    }
    const float gs = gscale ? *gscale : 1.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int col = it * W + lane;
        if (col < ld) grad[row * ld + col] = col < dim ? gs * acc[it] : 0.f;
    }
}

// ---- highway gate ---------------------------------------------------------------------------------------------------
// a' = a ga + be, b' = b ga + be (the layer's BatchNorm affine, applied to both inputs), gate = relu(tanh(p)),
// out = tanh(b' (1 - gate) + a' gate).  Element-wise: 256 threads x 4 columns per step.
__global__ __launch_bounds__(256) void highway_fwd_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          const float *__restrict__ p, const float *__restrict__ ga,
                                                          const float *__restrict__ be, int64_t n, int d,
                                                          float *__restrict__ out) {
    const int64_t total = n * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % d);
        const float av = fmaf(a[i], ga[c], be[c]), bv = fmaf(b[i], ga[c], be[c]);
        const float gate = fmaxf(tanhf(p[i]), 0.f);
        out[i] = tanhf(bv * (1.f - gate) + av * gate);
    }
}

// backward: rows [r0, r1) of a block; the column sums for d ga / d be are kept per block (partials [blocks, 2, d]) and
// added in block order by the caller (torch.sum over dim 0: fixed order)
__global__ __launch_bounds__(256) void highway_bwd_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          const float *__restrict__ p, const float *__restrict__ ga,
                                                          const float *__restrict__ be, const float *__restrict__ out,
                                                          const float *__restrict__ go, int64_t n, int d, int rows_per_block,
                                                          float *__restrict__ da, float *__restrict__ db,
                                                          float *__restrict__ dp, float *__restrict__ partials) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {           // a thread owns columns c, c + 256, ...: its sums need no reduction
        const float g = ga[c], bb = be[c];
        float sg = 0.f, sb = 0.f;
#pragma unroll 4
        for (int64_t r = r0; r < r1; ++r) {
            const int64_t i = r * d + c;
            const float av = fmaf(a[i], g, bb), bv = fmaf(b[i], g, bb);
            const float th = tanhf(p[i]), gate = fmaxf(th, 0.f);
            const float o = out[i];
            const float du = go[i] * (1.f - o * o);
            const float dav = du * gate, dbv = du * (1.f - gate);
            da[i] = dav * g;
            db[i] = dbv * g;
            dp[i] = th > 0.f ? du * (av - bv) * (1.f - th * th) : 0.f;
            sg += dav * a[i] + dbv * b[i];
            sb += dav + dbv;
        }
        partials[((int64_t)blockIdx.x * 2 + 0) * d + c] = sg;
        partials[((int64_t)blockIdx.x * 2 + 1) * d + c] = sb;
    }
}

// ---- bias + tanh ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_tanh_fwd_kernel(const float *__restrict__ x, const float *__restrict__ bias, int64_t n,
                                                            int d, float *__restrict__ y) {
    const int64_t total = n * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = tanhf(x[i] + bias[(int)(i % d)]);
}

__global__ __launch_bounds__(256) void bias_tanh_bwd_kernel(const float *__restrict__ y, const float *__restrict__ gy, int64_t n,
                                                            int d, int rows_per_block, float *__restrict__ gx,
                                                            float *__restrict__ partials) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float s = 0.f;
#pragma unroll 4
        for (int64_t r = r0; r < r1; ++r) {
            const int64_t i = r * d + c;
            const float o = y[i], g = gy[i] * (1.f - o * o);
            gx[i] = g;
            s += g;
        }
        partials[(int64_t)blockIdx.x * d + c] = s;
    }
}

// ---- RDGCN's highway (rdgcn.py:250-256) and residual (rdgcn.py:330-333) ------------------------------------------------
// gate = sigmoid(p + bias), out = gate b + (1 - gate) a.  b_relu != 0: b is the output of a relu (the diagonal GCN layer,
// rdgcn.py:184-191) and db comes out already gated by b > 0, i.e. as the gradient of the relu's INPUT.
__global__ __launch_bounds__(256) void sigmoid_mix_fwd_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                              const float *__restrict__ p, const float *__restrict__ bias,
                                                              int64_t n, int d, float *__restrict__ out) {
    const int64_t total = n * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float gate = 1.f / (1.f + __expf(-(p[i] + bias[(int)(i % d)])));
        out[i] = gate * b[i] + (1.f - gate) * a[i];
    }
}

__global__ __launch_bounds__(256) void sigmoid_mix_bwd_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                              const float *__restrict__ p, const float *__restrict__ bias,
                                                              const float *__restrict__ go, int64_t n, int d, int rows_per_block,
                                                              int b_relu, float *__restrict__ da, float *__restrict__ db,
                                                              float *__restrict__ dp, float *__restrict__ partials) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {           // a thread owns its columns: the bias sums need no reduction
        const float bb = bias[c];
        float sb = 0.f;
#pragma unroll 4
        for (int64_t r = r0; r < r1; ++r) {
            const int64_t i = r * d + c;
            const float gate = 1.f / (1.f + __expf(-(p[i] + bb)));
            const float g = go[i], av = a[i], bv = b[i];
            da[i] = g * (1.f - gate);
            db[i] = (b_relu && !(bv > 0.f)) ? 0.f : g * gate;
            const float dpv = g * (bv - av) * gate * (1.f - gate);
            dp[i] = dpv;
            sb += dpv;
        }
        partials[(int64_t)blockIdx.x * d + c] = sb;
    }
}

// out = x + alpha relu(y);  backward: dy = alpha go where y > 0 (dx = go needs no kernel)
__global__ __launch_bounds__(256) void relu_axpy_fwd_kernel(const float *__restrict__ x, const float *__restrict__ y, float alpha,
                                                            int64_t total, float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = x[i] + alpha * fmaxf(y[i], 0.f);
}

__global__ __launch_bounds__(256) void relu_axpy_bwd_kernel(const float *__restrict__ y, const float *__restrict__ go, float alpha,
                                                            int64_t total, float *__restrict__ dy) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        dy[i] = y[i] > 0.f ? alpha * go[i] : 0.f;
}

// column sums of x * y over row blocks (d w0 of the diagonal layer: sum_rows dxs * x): partials [blocks, d]
__global__ __launch_bounds__(256) void colsum_prod_kernel(const float *__restrict__ x, const float *__restrict__ y, int64_t n, int d,
                                                          int rows_per_block, float *__restrict__ partials) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float s = 0.f;
#pragma unroll 4
        for (int64_t r = r0; r < r1; ++r) s = fmaf(x[r * d + c], y[r * d + c], s);
        partials[(int64_t)blockIdx.x * d + c] = s;
    }
}

// ---- segment sums -------------------------------------------------------------------------------------------------------
// out[s] = sum of vals[order[e]] over e in [seg_ptr[s], seg_ptr[s + 1]): one wave per segment, lanes stride over it, butterfly
// sum -- the gradient of a gather z = src[idx] with FEW distinct indices (RDGCN: the logit of an attention edge is a
// per-RELATION scalar, rdgcn.py:202-215: 568,000 edges gather from 700 values; torch's sorted index_put backward walks each
// relation's ~800 duplicates serially: 2.2 ms per call, profiles/r03_*).  Fixed order: reproducible bits.
__global__ __launch_bounds__(256) void segment_sum_kernel(const float *__restrict__ vals, const int32_t *__restrict__ order,
                                                          const int32_t *__restrict__ seg_ptr, int64_t n_seg,
                                                          float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t sgm = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sgm >= n_seg) return;
    float acc = 0.f;
    for (int e = seg_ptr[sgm] + lane; e < seg_ptr[sgm + 1]; e += W) acc += vals[order ? order[e] : e];
    acc = wsum(acc);
    if (lane == 0) out[sgm] = acc;
}

#define OEA_ROW_DISPATCH(cols, CALL)                                     \
    do {                                                                 \
        if ((cols) <= 128) { CALL(2); }                                  \
        else if ((cols) <= 256) { CALL(4); }                             \
        else if ((cols) <= 320) { CALL(5); }                             \
        else if ((cols) <= 512) { CALL(8); }                             \
        else if ((cols) <= 1280) { CALL(20); }                           \
        else { oea::set_error("%d columns > 1280 unsupported", (int)(cols)); return OEA_EUNSUPPORTED; } \
    } while (0)

static int make_blocks(const float *const *x, float *const *dx, const int32_t *dims, const int32_t *lds, int32_t k, Blocks *b,
                       int *total) {
    OEA_REQUIRE(k >= 1 && k <= 4 && dims && lds, "1 <= blocks <= 4");
    int off = 0;
    b->k = k;
    for (int i = 0; i < 4; ++i) {
        b->x[i] = i < k && x ? x[i] : nullptr;
        b->dx[i] = i < k && dx ? dx[i] : nullptr;
        b->dim[i] = i < k ? dims[i] : 0;
        b->ld[i] = i < k ? lds[i] : 0;
        b->off[i] = off;
        if (i < k) {
            OEA_REQUIRE(dims[i] > 0 && dims[i] <= lds[i], "0 < dim <= ld");
            off += dims[i];
        }
    }
    *total = off;
    return OEA_OK;
}

}  // namespace

extern "C" {

int oea_concat_l2n_fwd(const float *const *x, const int32_t *dims, const int32_t *lds, int32_t k, int64_t n, float *out,
                       int32_t ld_out, float *inv_blk, float *inv_all, void *stream) {
    Blocks b;
    int total = 0;
    const int rc = make_blocks(x, nullptr, dims, lds, k, &b, &total);
    if (rc != OEA_OK) return rc;
    OEA_REQUIRE(x && out && inv_blk && inv_all && total <= ld_out, "null pointer / ld_out < sum of dims");
    for (int i = 0; i < k; ++i) OEA_REQUIRE(x[i], "null pointer");
    if (n == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
#define CALL(IT) concat_l2n_fwd_kernel<IT><<<(unsigned)oea::ceil_div(n, 4), 256, 0, st>>>(b, n, out, ld_out, inv_blk, inv_all)
    OEA_ROW_DISPATCH(ld_out, CALL);
#undef CALL
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_concat_l2n_bwd(float *const *dx, const int32_t *dims, const int32_t *lds, int32_t k, int64_t n, const float *z,
                       const float *dz, int32_t ld_out, const float *inv_blk, const float *inv_all, void *stream) {
    Blocks b;
    int total = 0;
    const int rc = make_blocks(nullptr, dx, dims, lds, k, &b, &total);
    if (rc != OEA_OK) return rc;
    OEA_REQUIRE(dx && z && dz && inv_blk && inv_all && total <= ld_out, "null pointer / ld_out < sum of dims");
    for (int i = 0; i < k; ++i) OEA_REQUIRE(dx[i], "null pointer");
    if (n == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
#define CALL(IT) concat_l2n_bwd_kernel<IT><<<(unsigned)oea::ceil_div(n, 4), 256, 0, st>>>(b, n, z, dz, ld_out, inv_blk, inv_all)
    OEA_ROW_DISPATCH(ld_out, CALL);
#undef CALL
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_pair_loss_l2_fwd(const float *emb, int64_t n, int32_t dim, int32_t ld, const int32_t *pairs, int64_t m, int64_t n_pos,
                         const float *weight, float margin, float balance, float *coef, float *terms, void *stream) {
    OEA_REQUIRE(emb && pairs && coef && terms && n > 0 && dim > 0 && dim <= ld && n_pos >= 0 && n_pos <= m, "arguments");
    if (m == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
#define CALL(IT) pair_loss_fwd_kernel<IT><<<(unsigned)oea::ceil_div(m, 4), 256, 0, st>>>(emb, dim, ld, pairs, m, n_pos, weight, margin, balance, coef, terms)
    OEA_ROW_DISPATCH(dim, CALL);
#undef CALL
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_pair_grad_rows(const float *emb, int64_t n, int32_t dim, int32_t ld, const int32_t *rowptr, const int32_t *other,
                       const int32_t *slot_pair, const float *coef, const float *gscale, int32_t norm, float *grad, void *stream) {
    OEA_REQUIRE(emb && rowptr && other && slot_pair && coef && grad && n > 0 && dim > 0 && dim <= ld, "arguments");
    OEA_REQUIRE(norm == 1 || norm == 2, "norm: 1 (L1) or 2 (squared L2)");
    hipStream_t st = oea::as_stream(stream);
#define CALL(IT)                                                                                                                  \
    do {                                                                                                                          \
        if (norm == 1) pair_loss_bwd_kernel<IT, true><<<(unsigned)oea::ceil_div(n, 4), 256, 0, st>>>(emb, n, dim, ld, rowptr, other, slot_pair, coef, gscale, grad); \
        else pair_loss_bwd_kernel<IT, false><<<(unsigned)oea::ceil_div(n, 4), 256, 0, st>>>(emb, n, dim, ld, rowptr, other, slot_pair, coef, gscale, grad); \
    } while (0)
    OEA_ROW_DISPATCH(ld, CALL);
#undef CALL
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_segment_sum_f32(const float *vals, const int32_t *order, const int32_t *seg_ptr, int64_t n_seg, float *out, void *stream) {
    OEA_REQUIRE(vals && seg_ptr && out && n_seg >= 0, "arguments");
    if (n_seg == 0) return OEA_OK;
    segment_sum_kernel<<<(unsigned)oea::ceil_div(n_seg, 4), 256, 0, oea::as_stream(stream)>>>(vals, order, seg_ptr, n_seg, out);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

// row blocks of the column-sum kernels: 64 rows each (a 200,000-row layer gives 3,125 workgroups; with 256 rows the 782
// workgroups of sigmoid_mix_bwd streamed 1.7 GB at 2.9 TB/s), at most 4,096 partial rows for the caller to add
int32_t oea_colsum_blocks(int64_t n) { return (int32_t)(n < 1024 ? (n > 0 ? 1 : 0) : (n + 63) / 64 > 4096 ? 4096 : (n + 63) / 64); }

int oea_highway_fwd(const float *a, const float *b, const float *p, const float *gamma, const float *beta, int64_t n, int32_t d,
                    float *out, void *stream) {
    OEA_REQUIRE(a && b && p && gamma && beta && out && d > 0, "arguments");
    if (n == 0) return OEA_OK;
    const int64_t total = n * d;
    highway_fwd_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(total, 256), 1 << 16), 256, 0, oea::as_stream(stream)>>>(
        a, b, p, gamma, beta, n, d, out);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_highway_bwd(const float *a, const float *b, const float *p, const float *gamma, const float *beta, const float *out,
                    const float *gout, int64_t n, int32_t d, float *da, float *db, float *dp, float *partials, void *stream) {
    OEA_REQUIRE(a && b && p && gamma && beta && out && gout && da && db && dp && partials && d > 0, "arguments");
    if (n == 0) return OEA_OK;
    const int nb = oea_colsum_blocks(n);
    const int rpb = (int)oea::ceil_div(n, nb);
    highway_bwd_kernel<<<nb, 256, 0, oea::as_stream(stream)>>>(a, b, p, gamma, beta, out, gout, n, d, rpb, da, db, dp, partials);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_sigmoid_mix_fwd(const float *a, const float *b, const float *p, const float *bias, int64_t n, int32_t d, float *out,
                        void *stream) {
    OEA_REQUIRE(a && b && p && bias && out && d > 0, "arguments");
    if (n == 0) return OEA_OK;
    sigmoid_mix_fwd_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(n * d, 256), 1 << 16), 256, 0, oea::as_stream(stream)>>>(
        a, b, p, bias, n, d, out);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_sigmoid_mix_bwd(const float *a, const float *b, const float *p, const float *bias, const float *gout, int64_t n, int32_t d,
                        int32_t b_relu, float *da, float *db, float *dp, float *partials, void *stream) {
    OEA_REQUIRE(a && b && p && bias && gout && da && db && dp && partials && d > 0, "arguments");
    if (n == 0) return OEA_OK;
    const int nb = oea_colsum_blocks(n);
    const int rpb = (int)oea::ceil_div(n, nb);
    sigmoid_mix_bwd_kernel<<<nb, 256, 0, oea::as_stream(stream)>>>(a, b, p, bias, gout, n, d, rpb, b_relu, da, db, dp, partials);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_relu_axpy_fwd(const float *x, const float *y, float alpha, int64_t total, float *out, void *stream) {
    OEA_REQUIRE(x && y && out && total >= 0, "arguments");
    if (total == 0) return OEA_OK;
    relu_axpy_fwd_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(total, 256), 1 << 16), 256, 0, oea::as_stream(stream)>>>(
        x, y, alpha, total, out);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_relu_axpy_bwd(const float *y, const float *gout, float alpha, int64_t total, float *dy, void *stream) {
    OEA_REQUIRE(y && gout && dy && total >= 0, "arguments");
    if (total == 0) return OEA_OK;
    relu_axpy_bwd_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(total, 256), 1 << 16), 256, 0, oea::as_stream(stream)>>>(
        y, gout, alpha, total, dy);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_colsum_prod(const float *x, const float *y, int64_t n, int32_t d, float *partials, void *stream) {
    OEA_REQUIRE(x && y && partials && d > 0, "arguments");
    if (n == 0) return OEA_OK;
    const int nb = oea_colsum_blocks(n);
    const int rpb = (int)oea::ceil_div(n, nb);
    colsum_prod_kernel<<<nb, 256, 0, oea::as_stream(stream)>>>(x, y, n, d, rpb, partials);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_bias_tanh_fwd(const float *x, const float *bias, int64_t n, int32_t d, float *y, void *stream) {
    OEA_REQUIRE(x && bias && y && d > 0, "arguments");
    if (n == 0) return OEA_OK;
    bias_tanh_fwd_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(n * d, 256), 1 << 16), 256, 0, oea::as_stream(stream)>>>(
        x, bias, n, d, y);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_bias_tanh_bwd(const float *y, const float *gy, int64_t n, int32_t d, float *gx, float *partials, void *stream) {
    OEA_REQUIRE(y && gy && gx && partials && d > 0, "arguments");
    if (n == 0) return OEA_OK;
    const int nb = oea_colsum_blocks(n);
    const int rpb = (int)oea::ceil_div(n, nb);
    bias_tanh_bwd_kernel<<<nb, 256, 0, oea::as_stream(stream)>>>(y, gy, n, d, rpb, gx, partials);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

}  // extern "C"
