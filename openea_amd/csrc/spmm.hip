// spmm.hip -- CSR neighbour aggregation + GCN-Align loss / SGD kernels.
//
// Replaces tf.sparse_tensor_dense_matmul(A, X) (approaches/gcn_align.py:83,259;
// alinet.py:581; rdgcn.py:187,196), the L1 alignment hinge (gcn_align.py:298-320) and the
// GradientDescentOptimizer step through trunc_normal's l2_normalize (gcn_align.py:52-56,511).
//
// Aggregate: one G-lane group per output row, lanes across the feature columns (float4 per
// lane, coalesced 16-B gathers of X rows); the row's (col, val) pairs are read once and
// broadcast.  Rows are short and skewed (avg degree 6-12), so a group walks its row
// sequentially in CSR order: the sum order is fixed (deterministic, equals the oracle's COO
// order when the COO is row-sorted) and no atomics are needed.  The backward pass is the same
// kernel on the transposed CSR with the relu gate fused in.
// Bytes per launch: nnz*(8 + 4*d) + 4*N*d.
#include "common.h"
#include <stdlib.h>

namespace {

using oea::group_sum;

// accumulate edges e0, e0+stride, ... < e1 of one row into acc (4 independent 16-B gathers in flight)
template <int G, int IT>
__device__ __forceinline__ void row_accumulate(const int32_t *__restrict__ colidx, const float *__restrict__ vals,
                                               const float *__restrict__ x, int ldx, int lane, int e0, int e1, int stride,
                                               float4 (&acc)[IT]) {
    int e = e0;
    for (; e + 3 * stride < e1; e += 4 * stride) {
        int c[4];
        float v[4];
        float4 xv[4][IT];
#pragma unroll
        for (int u = 0; u < 4; ++u) { c[u] = colidx[e + u * stride]; v[u] = vals[e + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int col = (it * G + lane) * 4;
                xv[u][it] = col < ldx ? oea::ld4(x + (int64_t)c[u] * ldx + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)           // fixed order u = 0..3 -> the serial CSR order when stride == 1
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                acc[it].x = fmaf(v[u], xv[u][it].x, acc[it].x); acc[it].y = fmaf(v[u], xv[u][it].y, acc[it].y);
                acc[it].z = fmaf(v[u], xv[u][it].z, acc[it].z); acc[it].w = fmaf(v[u], xv[u][it].w, acc[it].w);
            }
    }
    for (; e < e1; e += stride) {
        const int c0 = colidx[e];
        const float v0 = vals[e];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int col = (it * G + lane) * 4;
            const float4 x0 = col < ldx ? oea::ld4(x + (int64_t)c0 * ldx + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc[it].x = fmaf(v0, x0.x, acc[it].x); acc[it].y = fmaf(v0, x0.y, acc[it].y);
            acc[it].z = fmaf(v0, x0.z, acc[it].z); acc[it].w = fmaf(v0, x0.w, acc[it].w);
        }
    }
}

template <int G, int IT>
__device__ __forceinline__ void row_store(float4 (&acc)[IT], int64_t row, int lane, int dim, int act,
                                          const float *__restrict__ mask_from, float *__restrict__ y, int ldy) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = (it * G + lane) * 4;
        if (c < ldy) {
            float4 o = acc[it];
            if (act == 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            if (mask_from) {
                const float4 m = oea::ld4(mask_from + row * ldy + c);
                o.x = m.x > 0.f ? o.x : 0.f; o.y = m.y > 0.f ? o.y : 0.f;
                o.z = m.z > 0.f ? o.z : 0.f; o.w = m.w > 0.f ? o.w : 0.f;
            }
            if (c + 0 >= dim) o.x = 0.f;
            if (c + 1 >= dim) o.y = 0.f;
            if (c + 2 >= dim) o.z = 0.f;
            if (c + 3 >= dim) o.w = 0.f;
            oea::st4(y + row * ldy + c, o);
        }
    }
}

// Rows are short and skewed (average degree 6-12, hubs with thousands of neighbours at the low ids,
// read.py:64-79).  A workgroup takes NG = 256/G consecutive rows per iteration: every group sums its
// own row if it is short (<= kLongRow nonzeros, serial CSR order: deterministic, equals the oracle);
// long rows of the batch are then summed by ALL groups of the workgroup together (strided edges,
// partials combined through LDS in group order: still deterministic), so a hub row costs nnz/NG
// dependent steps instead of nnz (a 2,000-neighbour hub made the first version 1 ms per launch).
constexpr int kLongRow = 96;

template <int G, int IT>
__global__ __launch_bounds__(256) void spmm_csr_kernel(const int32_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ colidx,
                                                       const float *__restrict__ vals, int64_t n_rows,
                                                       const float *__restrict__ x, int dim, int ldx, int act,
                                                       const float *__restrict__ mask_from, float *__restrict__ y,
                                                       int ldy, int huge, oea_csr_split split, int n_chunk_blocks) {
    constexpr int NG = 256 / G;
    extern __shared__ __attribute__((aligned(16))) float s_part[];      // [NG][ldy] partial rows
    const int lane = threadIdx.x % G, gid = threadIdx.x / G;
    // the first n_chunk_blocks workgroups take one chunk of a hub row each (the heaviest items start first) and leave its
    // partial sum in split.partials[chunk]: no atomics, no zeroing; spmm_rows_epilogue_kernel adds a row's chunks in order
    if ((int)blockIdx.x < n_chunk_blocks) {
        const int ch = blockIdx.x;
        float4 acc[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) acc[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        row_accumulate<G, IT>(colidx, vals, x, ldx, lane, split.chunk_e0[ch] + gid, split.chunk_e1[ch], NG, acc);
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * G + lane) * 4;
            if (c < ldy) oea::st4(s_part + gid * ldy + c, acc[it]);
        }
        __syncthreads();
        if (gid == 0) {
            float *o = split.partials + (int64_t)ch * ldy;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = (it * G + lane) * 4;
                if (c < ldy) {
                    for (int g2 = 1; g2 < NG; ++g2) {
                        const float4 p = oea::ld4(s_part + g2 * ldy + c);
                        acc[it].x += p.x; acc[it].y += p.y; acc[it].z += p.z; acc[it].w += p.w;
                    }
                    oea::st4(o + c, acc[it]);
                }
            }
        }
        if (split.tickets) {
            // the row's LAST chunk to finish adds the row's chunks in chunk order and stores the row (round 3: the separate
            // epilogue launch -- ~5 us behind every aggregate with hub rows -- is gone; the order of the sum, and with it
            // the bits, do not depend on which chunk arrives last)
            __shared__ int s_sr, s_last;
            // the partial row was written with plain stores: release at device scope = write it back from this XCD's L2
            // (no invalidate -- the reader below uses device-coherent loads); only the few chunk workgroups pay for it
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (threadIdx.x == 0) {
                int lo = 0, hi = split.n_rows;                       // split row sr: row_chunk0[sr] <= ch < row_chunk0[sr + 1]
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (split.row_chunk0[mid] <= ch) lo = mid; else hi = mid;
                }
                const unsigned n = (unsigned)(split.row_chunk0[lo + 1] - split.row_chunk0[lo]);
                s_sr = lo;
                s_last = atomicAdd(split.tickets + lo, 1u) == n - 1u;
                if (s_last) split.tickets[lo] = 0u;                  // ready for the next launch on this operand
            }
            __syncthreads();
            if (s_last && gid == 0) {
                float4 sum[IT];
#pragma unroll
                for (int it = 0; it < IT; ++it) sum[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int c2 = split.row_chunk0[s_sr]; c2 < split.row_chunk0[s_sr + 1]; ++c2) {
                    const unsigned *pp = reinterpret_cast<const unsigned *>(split.partials + (int64_t)c2 * ldy);
#pragma unroll
                    for (int it = 0; it < IT; ++it) {
                        const int c = (it * G + lane) * 4;
                        if (c < ldy) {               // device-coherent loads: the other chunks were written through other XCDs' L2
                            sum[it].x += __uint_as_float(__hip_atomic_load(pp + c + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            sum[it].y += __uint_as_float(__hip_atomic_load(pp + c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            sum[it].z += __uint_as_float(__hip_atomic_load(pp + c + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            sum[it].w += __uint_as_float(__hip_atomic_load(pp + c + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        }
                    }
                }
                row_store<G, IT>(sum, (int64_t)split.rows[s_sr], lane, dim, act, mask_from, y, ldy);
            }
        }
        return;
    }
    // rows are dealt out CYCLICALLY: group gid of block b takes rows b + nb * (gid + NG * i).  Ids are
    // frequency-ordered, so consecutive rows would hand all hubs to the first few workgroups.
    const int64_t nb = gridDim.x - n_chunk_blocks, bid = blockIdx.x - n_chunk_blocks;
    for (int64_t base = 0; base < n_rows; base += nb * NG) {
        const int64_t row = base + (int64_t)gid * nb + bid;
        int e0 = 0, e1 = 0;
        if (row < n_rows) { e0 = rowptr[row]; e1 = rowptr[row + 1]; }
        if (row < n_rows && e1 - e0 <= kLongRow) {
            float4 acc[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) acc[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            row_accumulate<G, IT>(colidx, vals, x, ldx, lane, e0, e1, 1, acc);
            row_store<G, IT>(acc, row, lane, dim, act, mask_from, y, ldy);
        }
        // long rows of this batch: workgroup-cooperative (block-uniform control flow)
        for (int r = 0; r < NG; ++r) {
            const int64_t lrow = base + (int64_t)r * nb + bid;
            if (lrow >= n_rows) break;
            const int l0 = rowptr[lrow], l1 = rowptr[lrow + 1];
            if (l1 - l0 <= kLongRow || l1 - l0 > huge) continue;     // huge rows: split across workgroups below
            float4 acc[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) acc[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            row_accumulate<G, IT>(colidx, vals, x, ldx, lane, l0 + gid, l1, NG, acc);
            __syncthreads();
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = (it * G + lane) * 4;
                if (c < ldy) oea::st4(s_part + gid * ldy + c, acc[it]);
            }
            __syncthreads();
            if (gid == 0) {
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int c = (it * G + lane) * 4;
                    if (c < ldy) {
                        for (int g2 = 1; g2 < NG; ++g2) {
                            const float4 p = oea::ld4(s_part + g2 * ldy + c);
                            acc[it].x += p.x; acc[it].y += p.y; acc[it].z += p.z; acc[it].w += p.w;
                        }
                    }
                }
                row_store<G, IT>(acc, lrow, lane, dim, act, mask_from, y, ldy);
            }
        }
    }
}

// ---- hub rows (nnz > split->threshold): chunks of the row go to different workgroups ----------------
template <int G, int IT>
__global__ __launch_bounds__(256) void spmm_zero_rows_kernel(const int32_t *__restrict__ rows, int n, float *__restrict__ y, int ldy) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (grp >= n) return;
    float *o = y + (int64_t)rows[grp] * ldy;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = (it * G + lane) * 4;
        if (c < ldy) oea::st4(o + c, make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

template <int G, int IT>
__global__ __launch_bounds__(256) void spmm_chunk_kernel(const int32_t *__restrict__ chunk_row, const int32_t *__restrict__ chunk_e0,
                                                         const int32_t *__restrict__ chunk_e1,
                                                         const int32_t *__restrict__ colidx, const float *__restrict__ vals,
                                                         const float *__restrict__ x, int ldx, float *__restrict__ y, int ldy) {
    constexpr int NG = 256 / G;
    extern __shared__ __attribute__((aligned(16))) float s_part[];
    const int lane = threadIdx.x % G, gid = threadIdx.x / G;
    const int ch = blockIdx.x;
    float4 acc[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) acc[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    row_accumulate<G, IT>(colidx, vals, x, ldx, lane, chunk_e0[ch] + gid, chunk_e1[ch], NG, acc);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = (it * G + lane) * 4;
        if (c < ldy) oea::st4(s_part + gid * ldy + c, acc[it]);
    }
    __syncthreads();
    if (gid == 0) {
        float *o = y + (int64_t)chunk_row[ch] * ldy;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * G + lane) * 4;
            if (c < ldy) {
                for (int g2 = 1; g2 < NG; ++g2) {
                    const float4 p = oea::ld4(s_part + g2 * ldy + c);
                    acc[it].x += p.x; acc[it].y += p.y; acc[it].z += p.z; acc[it].w += p.w;
                }
                oea::atomic_add_f32(o + c, acc[it].x); oea::atomic_add_f32(o + c + 1, acc[it].y);
                oea::atomic_add_f32(o + c + 2, acc[it].z); oea::atomic_add_f32(o + c + 3, acc[it].w);
            }
        }
    }
}

// partials == NULL: y holds the atomically summed row (legacy path); else the row's chunks are added in chunk order
template <int G, int IT>
__global__ __launch_bounds__(256) void spmm_rows_epilogue_kernel(const int32_t *__restrict__ rows, int n, int dim, int act,
                                                                 const float *__restrict__ mask_from, float *__restrict__ y, int ldy,
                                                                 const float *__restrict__ partials, const int32_t *__restrict__ row_chunk0) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (grp >= n) return;
    const int64_t row = rows[grp];
    float4 acc[IT];
    if (partials) {
#pragma unroll
        for (int it = 0; it < IT; ++it) acc[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ch = row_chunk0[grp]; ch < row_chunk0[grp + 1]; ++ch) {
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = (it * G + lane) * 4;
                if (c < ldy) {
                    const float4 p = oea::ld4(partials + (int64_t)ch * ldy + c);
                    acc[it].x += p.x; acc[it].y += p.y; acc[it].z += p.z; acc[it].w += p.w;
                }
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * G + lane) * 4;
            acc[it] = c < ldy ? oea::ld4(y + row * ldy + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    row_store<G, IT>(acc, row, lane, dim, act, mask_from, y, ldy);
}

__device__ __forceinline__ float sgnf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// One WORKGROUP per seed link a: every G-lane group computes A = |x_l - x_r|_1 (the two rows are shared
// through L1/L2) and takes the negatives b = group, group + NG, ... of the link's 2k (k is 5 for GCN-Align,
// 125 for RDGCN: a serial loop over them was the whole kernel); the hinge-active count is combined through
// LDS and group 0 adds the positive pair's gradient once.
// Rows are held LANE-STRIDED here (lane owns columns lane, lane + G, ...): every load and every fp32 atomic of a
// wave instruction covers one contiguous 128-byte segment per row (a float4-per-lane layout spreads each atomic
// instruction over four lines and was 2x slower; the kernel is bound by the atomic rate).
template <int G, int IT>
__global__ __launch_bounds__(256) void align_loss_l1_kernel(const float *__restrict__ emb, int dim, int ld,
                                                            const int32_t *__restrict__ ill, int64_t t, int k, float gamma,
                                                            const int32_t *__restrict__ neg_left,
                                                            const int32_t *__restrict__ neg_right,
                                                            const int32_t *__restrict__ neg2_left,
                                                            const int32_t *__restrict__ neg2_right,
                                                            float *__restrict__ grad, double *__restrict__ loss_accum,
                                                            float *__restrict__ coef_out) {
    // coef_out != NULL: no gradient here -- the signed coefficient of every pair ([0, t): the links, + scale * #active
    // hinges; [t + a 2k + i]: negative i of link a, - scale if its hinge is active) goes out and oea_pair_grad_rows sums
    // each row's pairs in a fixed order (no atomics, reproducible bits)
    constexpr int NG = 256 / G;
    __shared__ int s_active;
    __shared__ double s_loss[4];
    const int lane = threadIdx.x % G, gid = threadIdx.x / G;
    const float scale = 1.0f / (2.0f * (float)k * (float)t);
    double loss_local = 0.0;
    for (int64_t a = blockIdx.x; a < t; a += gridDim.x) {
        if (threadIdx.x == 0) s_active = 0;
        __syncthreads();
        const int l = ill[2 * a], r = ill[2 * a + 1];
        float dp[IT];
        float A = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * G + lane;
            dp[it] = c < ld ? emb[(int64_t)l * ld + c] - emb[(int64_t)r * ld + c] : 0.f;
            A += fabsf(dp[it]);
        }
        A = group_sum<G>(A);
        const float D = A + gamma;
        int active = 0;
        // the negative lists of GCN-Align / RDGCN pair the link's OWN entity with k others (neg_left = repeat(l),
        // neg2_right = repeat(r)): contributions to rows l and r are summed in registers, one atomic row per group
        float own_l[IT], own_r[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) own_l[it] = own_r[it] = 0.f;
        for (int i = gid; i < 2 * k; i += NG) {
            const int side = i >= k, b = side ? i - k : i;
            const int32_t *nlp = side ? neg2_left : neg_left, *nrp = side ? neg2_right : neg_right;
            const int nl = nlp[a * k + b], nr = nrp[a * k + b];
            float dn[IT];
            float B = 0.f;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = it * G + lane;
                dn[it] = c < ld ? emb[(int64_t)nl * ld + c] - emb[(int64_t)nr * ld + c] : 0.f;
                B += fabsf(dn[it]);
            }
            B = group_sum<G>(B);
            const float L = D - B;
            if (coef_out) {
                if (lane == 0) coef_out[t + a * 2 * k + i] = L > 0.f ? -scale : 0.f;
                if (L > 0.f) {
                    ++active;
                    if (lane == 0) loss_local += (double)L;
                }
                continue;
            }
            if (L > 0.f) {
                ++active;
                if (lane == 0) loss_local += (double)L;
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int c = it * G + lane;
                    const float g = -scale * sgnf(dn[it]);
                    if (c < ld && g != 0.f) {
                        if (nl == l) own_l[it] += g;
                        else if (nl == r) own_r[it] += g;
                        else oea::atomic_add_f32(grad + (int64_t)nl * ld + c, g);
                        if (nr == r) own_r[it] -= g;
                        else if (nr == l) own_l[it] -= g;
                        else oea::atomic_add_f32(grad + (int64_t)nr * ld + c, -g);
                    }
                }
            }
        }
        if (active && !coef_out) {
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = it * G + lane;
                if (c < ld && own_l[it] != 0.f) oea::atomic_add_f32(grad + (int64_t)l * ld + c, own_l[it]);
                if (c < ld && own_r[it] != 0.f) oea::atomic_add_f32(grad + (int64_t)r * ld + c, own_r[it]);
            }
        }
        if (lane == 0 && active) atomicAdd(&s_active, active);
        __syncthreads();
        const int total = s_active;
        if (coef_out && threadIdx.x == 0) coef_out[a] = scale * (float)total;
        if (gid == 0 && total && !coef_out) {
            const float ca = scale * (float)total;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int c = it * G + lane;
                const float g = ca * sgnf(dp[it]);
                if (c < ld && g != 0.f) {
                    oea::atomic_add_f32(grad + (int64_t)l * ld + c, g);
                    oea::atomic_add_f32(grad + (int64_t)r * ld + c, -g);
                }
            }
        }
        __syncthreads();
    }
    const double w = oea::wave_sum_d(loss_local);
    if ((threadIdx.x & 63) == 0) s_loss[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3];
        if (tot != 0.0) atomicAdd(loss_accum, tot * (double)scale);
    }
}

// The coefficient pass for SMALL k (GCN-Align's 5): one G-lane group per seed link instead of one workgroup -- at
// t = 4,500 links x 10 negatives the workgroup-per-link kernel spent 43 us on two barriers and an LDS counter per link
// (the whole GCN-Align epoch is 0.22 ms).  A group holds x_l - x_r, walks the link's 2k negatives (both rows of a negative
// pair are requested before the first is used), writes their coefficients and the link's own; no LDS, no barrier inside
// the loop; the loss leaves through one double atomic per workgroup.
template <int G, int IT>
__global__ __launch_bounds__(256) void align_coef_groups_kernel(const float *__restrict__ emb, int dim, int ld,
                                                                const int32_t *__restrict__ ill, int64_t t, int k, float gamma,
                                                                const int32_t *__restrict__ neg_left, const int32_t *__restrict__ neg_right,
                                                                const int32_t *__restrict__ neg2_left, const int32_t *__restrict__ neg2_right,
                                                                double *__restrict__ loss_accum, float *__restrict__ coef_out) {
    constexpr int NG = 256 / G;
    __shared__ double s_loss[4];
    const int lane = threadIdx.x % G, gid = threadIdx.x / G;
    const float scale = 1.0f / (2.0f * (float)k * (float)t);
    double loss_local = 0.0;
    for (int64_t a = (int64_t)blockIdx.x * NG + gid; a < t; a += (int64_t)gridDim.x * NG) {
        const int l = ill[2 * a], r = ill[2 * a + 1];
        float A = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * G + lane;
            A += c < ld ? fabsf(emb[(int64_t)l * ld + c] - emb[(int64_t)r * ld + c]) : 0.f;
        }
        A = group_sum<G>(A);
        const float D = A + gamma;
        int active = 0;
        for (int i = 0; i < 2 * k; i += 2) {                       // two negatives per trip: four rows in flight
            float B[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int iu = min(i + u, 2 * k - 1);
                const int side = iu >= k, b = side ? iu - k : iu;
                const int nl = (side ? neg2_left : neg_left)[a * k + b], nr = (side ? neg2_right : neg_right)[a * k + b];
                float acc = 0.f;
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int c = it * G + lane;
                    acc += c < ld ? fabsf(emb[(int64_t)nl * ld + c] - emb[(int64_t)nr * ld + c]) : 0.f;
                }
                B[u] = acc;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float L = D - group_sum<G>(B[u]);
                if (i + u < 2 * k) {
                    if (lane == 0) coef_out[t + a * 2 * k + i + u] = L > 0.f ? -scale : 0.f;
                    if (L > 0.f) {
                        ++active;
                        if (lane == 0) loss_local += (double)L;
                    }
                }
            }
        }
        if (lane == 0) coef_out[a] = scale * (float)active;
    }
    const double w = oea::wave_sum_d(loss_local);
    if ((threadIdx.x & 63) == 0) s_loss[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3];
        if (tot != 0.0) atomicAdd(loss_accum, tot * (double)scale);
    }
}

template <int G, int IT>
__global__ __launch_bounds__(256) void sgd_rows_kernel(float *__restrict__ w, const float *__restrict__ grad, int64_t rows,
                                                       int dim, int ld, int normalize, float lr) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t row = grp; row < rows; row += ngrp) {
        float4 v[IT], g[IT];
        float ss = 0.f, dot = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * G + lane) * 4;
            v[it] = c < ld ? oea::ld4(w + row * ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            g[it] = c < ld ? oea::ld4(grad + row * ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            ss += v[it].x * v[it].x + v[it].y * v[it].y + v[it].z * v[it].z + v[it].w * v[it].w;
            dot += v[it].x * g[it].x + v[it].y * g[it].y + v[it].z * g[it].z + v[it].w * g[it].w;
        }
        float inv = 1.f, ydg = 0.f;
        if (normalize) {
            ss = group_sum<G>(ss);
            dot = group_sum<G>(dot);
            inv = rsqrtf(fmaxf(ss, 1e-12f));
            ydg = ss > 1e-12f ? dot * inv : 0.f;
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * G + lane) * 4;
            if (c < ld) {
                float4 o;
                o.x = v[it].x - lr * (normalize ? (g[it].x - v[it].x * inv * ydg) * inv : g[it].x);
                o.y = v[it].y - lr * (normalize ? (g[it].y - v[it].y * inv * ydg) * inv : g[it].y);
                o.z = v[it].z - lr * (normalize ? (g[it].z - v[it].z * inv * ydg) * inv : g[it].z);
                o.w = v[it].w - lr * (normalize ? (g[it].w - v[it].w * inv * ydg) * inv : g[it].w);
                oea::st4(w + row * ld + c, o);
            }
        }
    }
}

#define OEA_DISPATCH_LD(ld, CALL)                                       \
    do {                                                                \
        if ((ld) <= 64) { CALL(16, 1); }                                \
        else if ((ld) <= 128) { CALL(32, 1); }                          \
        else if ((ld) <= 256) { CALL(64, 1); }                          \
        else if ((ld) <= 512) { CALL(64, 2); }                          \
        else if ((ld) <= 1280) { CALL(64, 5); }                         \
        else { oea::set_error("ld %d > 1280 unsupported", (int)(ld)); return OEA_EUNSUPPORTED; } \
    } while (0)

}  // namespace

extern "C" {

int oea_spmm_csr(const int32_t *rowptr, const int32_t *colidx, const float *vals, int64_t n_rows,
                 const float *x, int32_t dim, int32_t ldx, int32_t act, const float *mask_from, float *y,
                 int32_t ldy, const oea_csr_split *split, void *stream) {
    OEA_REQUIRE(rowptr && colidx && vals && x && y, "null pointer");
    OEA_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && dim > 0 && dim <= ldx && dim <= ldy && ldx == ldy, "ldx == ldy, % 4 == 0");
    OEA_REQUIRE(act == 0 || act == 1, "act: 0 none, 1 relu");
    OEA_REQUIRE(!split || split->n_chunks == 0 || (split->chunk_row && split->chunk_e0 && split->chunk_e1 && split->rows &&
                                                   split->threshold >= kLongRow), "csr split: null pointer / threshold");
    if (n_rows == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
    const bool use_split = split && split->n_chunks > 0;
    const int huge = use_split ? split->threshold : 0x7fffffff;
    // hub chunks inside the row kernel when the split carries a partial-sum buffer (1 + 1 launches, fixed summation order);
    // else the legacy path: zero the hub rows, chunk kernel with atomics, epilogue (1 + 3 launches)
    const bool fused = use_split && split->partials && split->row_chunk0 &&
                       split->partials_floats >= (int64_t)split->n_chunks * ldy;
    oea_csr_split sp{};
    if (use_split) sp = *split;
#define CALL(G, IT)                                                                                                    \
    do {                                                                                                               \
        const size_t lds = sizeof(float) * (256 / G) * (size_t)ldy;                                                    \
        const int ncb = fused ? split->n_chunks : 0;                                                                   \
        spmm_csr_kernel<G, IT><<<(unsigned)(ncb + std::min<int64_t>(oea::ceil_div(n_rows, 256 / G), 1 << 20)), 256, lds, st>>>( \
            rowptr, colidx, vals, n_rows, x, dim, ldx, act, mask_from, y, ldy, huge, sp, ncb);                         \
        if (use_split) {                                                                                               \
            const unsigned gr = (unsigned)oea::ceil_div(split->n_rows, 256 / G);                                       \
            if (!fused) {                                                                                              \
                spmm_zero_rows_kernel<G, IT><<<gr, 256, 0, st>>>(split->rows, split->n_rows, y, ldy);                  \
                spmm_chunk_kernel<G, IT><<<(unsigned)split->n_chunks, 256, lds, st>>>(split->chunk_row, split->chunk_e0, \
                                                                                      split->chunk_e1, colidx, vals, x, ldx, y, ldy); \
            }                                                                                                          \
            if (!(fused && split->tickets))                                                                            \
                spmm_rows_epilogue_kernel<G, IT><<<gr, 256, 0, st>>>(split->rows, split->n_rows, dim, act, mask_from, y, ldy, \
                                                                     fused ? split->partials : nullptr, split->row_chunk0); \
        }                                                                                                              \
    } while (0)
    // 64 < ld <= 128 (the d = 100 tables of GCN-Align / the translational models): 16-lane groups with two float4 per lane --
    // 16 rows per workgroup instead of 8, half the waves for the same rows: 40.8 -> 36.1 us per aggregate at the D-W-15K shape
    // (gpurun_out r03k).  OEA_SPMM_G = 8 / 16 / 32 overrides (experiments).
    static const int env_g = [] { const char *e = getenv("OEA_SPMM_G"); return e ? atoi(e) : 16; }();
    if (ldx > 64 && ldx <= 128 && env_g == 16) { CALL(16, 2); }
    else if (ldx > 64 && ldx <= 128 && env_g == 8) { CALL(8, 4); }
    else OEA_DISPATCH_LD(ldx, CALL);
#undef CALL
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_align_loss_l1(const float *out_emb, int64_t n, int32_t dim, int32_t ld, const int32_t *ill, int64_t t,
                      int32_t k, float gamma, const int32_t *neg_left, const int32_t *neg_right,
                      const int32_t *neg2_left, const int32_t *neg2_right, float *grad, double *loss_accum,
                      void *stream) {
    return oea_align_loss_l1_coef(out_emb, n, dim, ld, ill, t, k, gamma, neg_left, neg_right, neg2_left, neg2_right, grad, loss_accum,
                                  nullptr, stream);
}

int oea_align_loss_l1_coef(const float *out_emb, int64_t n, int32_t dim, int32_t ld, const int32_t *ill, int64_t t,
                           int32_t k, float gamma, const int32_t *neg_left, const int32_t *neg_right,
                           const int32_t *neg2_left, const int32_t *neg2_right, float *grad, double *loss_accum,
                           float *coef_out, void *stream) {
    OEA_REQUIRE(out_emb && ill && neg_left && neg_right && neg2_left && neg2_right && (grad || coef_out) && loss_accum, "null pointer");
    OEA_REQUIRE(ld % 4 == 0 && dim > 0 && dim <= ld && k >= 1 && n > 0, "shapes");
    if (t == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
    if (coef_out && !grad && k <= 16 && ld <= 128) {                 // small k: a lane group per link (GCN-Align)
#define CALLG(IT)                                                                                                          \
    align_coef_groups_kernel<32, IT><<<(unsigned)std::min<int64_t>(oea::ceil_div(t, 8), 4096), 256, 0, st>>>(                 \
        out_emb, dim, ld, ill, t, k, gamma, neg_left, neg_right, neg2_left, neg2_right, loss_accum, coef_out)
        if (ld <= 32) CALLG(1);
        else if (ld <= 64) CALLG(2);
        else if (ld <= 96) CALLG(3);
        else CALLG(4);
#undef CALLG
        OEA_CHECK_HIP(hipGetLastError());
        return OEA_OK;
    }
#define CALL(G, IT)                                                                                             \
    align_loss_l1_kernel<G, IT><<<(unsigned)std::min<int64_t>(t, 65535), 256, 0, st>>>(                           \
        out_emb, dim, ld, ill, t, k, gamma, neg_left, neg_right, neg2_left, neg2_right, grad, loss_accum, coef_out)
    if (ld <= 32) CALL(32, 1);          // lane-strided rows: G * IT >= ld
    else if (ld <= 64) CALL(32, 2);
    else if (ld <= 96) CALL(32, 3);
    else if (ld <= 128) CALL(32, 4);
    else if (ld <= 256) CALL(64, 4);
    else if (ld <= 512) CALL(64, 8);
    else if (ld <= 1280) CALL(64, 20);
    else { oea::set_error("ld %d > 1280 unsupported", (int)ld); return OEA_EUNSUPPORTED; }
#undef CALL
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_sgd_rows(float *w, const float *grad_t, int64_t rows, int32_t dim, int32_t ld, int32_t normalize, float lr,
                 void *stream) {
    OEA_REQUIRE(w && grad_t, "null pointer");
    OEA_REQUIRE(ld % 4 == 0 && dim > 0 && dim <= ld, "ld % 4 == 0, dim <= ld");
    if (rows == 0) return OEA_OK;
    hipStream_t st = oea::as_stream(stream);
#define CALL(G, IT)                                                                                       \
    sgd_rows_kernel<G, IT><<<(unsigned)std::min<int64_t>(oea::ceil_div(rows, 256 / G), 65535), 256, 0, st>>>( \
        w, grad_t, rows, dim, ld, normalize, lr)
    OEA_DISPATCH_LD(ld, CALL);
#undef CALL
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

// ---- one full-batch epoch of a GCN_Align_Unit (gcn_align.py:498-539, 737-785) enqueued by ONE call -------------------------
// At the 15K shapes the epoch is ~10 kernels of 5-40 us: driven op by op from Python it was bound by the host (0.245 ms per
// epoch whatever the kernels took, gpurun_out r03f); the call below enqueues the same kernels back to back.
int oea_gcn_unit_epoch(const oea_gcn_unit *u, float *w, const oea_gcn_unit_buffers *b, double *loss_accum, void *stream) {
    OEA_REQUIRE(u && w && b && loss_accum, "null pointer");
    OEA_REQUIRE(u->a_rowptr && u->a_colidx && u->a_vals && u->at_rowptr && u->at_colidx && u->at_vals && u->row_ids && u->ill &&
                u->neg_left && u->neg_right && u->neg2_left && u->neg2_right, "gcn unit: null pointer");
    OEA_REQUIRE(b->t && b->h1 && b->out && b->g_out && b->g_pre1 && b->g_x, "gcn unit buffers: null pointer");
    OEA_REQUIRE(u->n > 0 && u->w_rows > 0 && u->dim > 0 && u->dim <= u->ld && u->ld % 4 == 0, "shapes");
    const bool feat = u->f_rowptr != nullptr;
    OEA_REQUIRE(!feat || (u->f_colidx && u->f_vals && u->ft_rowptr && u->ft_colidx && u->ft_vals && b->x && b->g_t), "gcn unit: feature operand");
    OEA_REQUIRE(feat || u->w_rows == u->n, "featureless unit: the weight IS the [n, dim] table");
    const int d = u->dim, ld = u->ld;
    int rc;
#define OEA_TRY(call) do { rc = (call); if (rc != OEA_OK) return rc; } while (0)
    // T = l2_normalize(W) (trunc_normal returns the normalised tensor, gcn_align.py:52-56); X = T | F . T
    OEA_TRY(oea_gather_rows(w, d, ld, u->row_ids, u->w_rows, 1, b->t, ld, stream));
    const float *x = b->t;
    if (feat) {
        OEA_TRY(oea_spmm_csr(u->f_rowptr, u->f_colidx, u->f_vals, u->n, b->t, d, ld, 0, nullptr, b->x, ld, u->f_split, stream));
        x = b->x;
    }
    OEA_TRY(oea_spmm_csr(u->a_rowptr, u->a_colidx, u->a_vals, u->n, x, d, ld, 1, nullptr, b->h1, ld, u->a_split, stream));        // relu(A X)
    OEA_TRY(oea_spmm_csr(u->a_rowptr, u->a_colidx, u->a_vals, u->n, b->h1, d, ld, 0, nullptr, b->out, ld, u->a_split, stream));   // A H1
    if (u->pair_rowptr) {        // hinge coefficients, then every row adds its pairs in a fixed order (no atomics)
        OEA_REQUIRE(u->pair_other && u->pair_slot && b->coef, "gcn unit: pair lists");
        OEA_TRY(oea_align_loss_l1_coef(b->out, u->n, d, ld, u->ill, u->t, u->k, u->gamma, u->neg_left, u->neg_right, u->neg2_left,
                                       u->neg2_right, nullptr, loss_accum, b->coef, stream));
        OEA_TRY(oea_pair_grad_rows(b->out, u->n, d, ld, u->pair_rowptr, u->pair_other, u->pair_slot, b->coef, nullptr, 1, b->g_out, stream));
    } else {
        OEA_CHECK_HIP(hipMemsetAsync(b->g_out, 0, sizeof(float) * (size_t)u->n * ld, oea::as_stream(stream)));
        OEA_TRY(oea_align_loss_l1(b->out, u->n, d, ld, u->ill, u->t, u->k, u->gamma, u->neg_left, u->neg_right, u->neg2_left,
                                  u->neg2_right, b->g_out, loss_accum, stream));
    }
    OEA_TRY(oea_spmm_csr(u->at_rowptr, u->at_colidx, u->at_vals, u->n, b->g_out, d, ld, 0, b->h1, b->g_pre1, ld, u->at_split, stream));   // relu gate fused
    OEA_TRY(oea_spmm_csr(u->at_rowptr, u->at_colidx, u->at_vals, u->n, b->g_pre1, d, ld, 0, nullptr, b->g_x, ld, u->at_split, stream));
    const float *g_t = b->g_x;
    if (feat) {
        OEA_TRY(oea_spmm_csr(u->ft_rowptr, u->ft_colidx, u->ft_vals, u->w_rows, b->g_x, d, ld, 0, nullptr, b->g_t, ld, u->ft_split, stream));
        g_t = b->g_t;
    }
    OEA_TRY(oea_sgd_rows(w, g_t, u->w_rows, d, ld, 1, u->lr, stream));       // through the row normalisation
#undef OEA_TRY
    return OEA_OK;
}

}  // extern "C"
