// mapping.hip -- MTransE's mapping step, fused.
//
// Replaces add_mapping_module + mapping_loss + AdagradOptimizer.minimize (modules/base/mapping.py:9-19,
// modules/base/losses.py:76-80, approaches/mtranse.py:84-96):
//     loss = alpha * ( sum_n || e2_n - e1_n M ||^2  +  || M M^T - I ||_F^2 ),   e = l2_normalize(ent)[seed ids]
// d is 75..300 and a batch a few hundred links, so the reference's five small matmuls (and their autodiff) are
// launch-bound; here one step is three kernels:
//   K1  one wave per link: normalise both rows, p = e1 M, diff = e2 - p, entity-row gradients
//       g2 = 2a diff, g1 = -2a diff M^T straight into the translational step's gradient scratch (so that
//       apply_rows pulls them through the normalisation and the optimiser), e1 / diff kept for K3,
//   K2  orth = M M^T - I (one thread per entry) + its share of the loss,
//   K3  g_M = a (-2 E1^T Diff + 4 orth M), fixed summation order (replicas of a data-parallel job get the same
//       bits), Adagrad / SGD on M into a second buffer that is copied back (K3 reads all of M).
#include "common.h"

namespace {

// workspace: [e1: n x ld][diff: n x ld][orth: d x d][m_new: d x d]
// M_LDS: the workgroup first copies M into LDS (d <= 120: 57.6 KB + the per-wave rows; every element is then read 2 x 4 times
// at LDS latency instead of L2 latency -- 166 links at d = 100: 78 us with one global load per fma, 23.5 us with 16 loads in
// flight, M_LDS below that)
template <bool M_LDS>
__global__ __launch_bounds__(256) void mapping_links_kernel(const float *__restrict__ ent, int ld, int dim, int l2norm,
                                                            const int32_t *__restrict__ ids1, const int32_t *__restrict__ ids2,
                                                            int64_t n, const float *__restrict__ Mg, float alpha,
                                                            oea::grad_t *__restrict__ ent_grad, oea::flag_t *__restrict__ ent_touched,
                                                            float *__restrict__ e1_out, float *__restrict__ diff_out,
                                                            double *__restrict__ loss_accum) {
    extern __shared__ float lds[];                       // per wave: y1 [dim], diff [dim]; then M [dim x dim] if M_LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *y1 = lds + wave * 2 * dim, *df = y1 + dim;
    const float *M = Mg;
    if (M_LDS) {
        float *Ms = lds + 8 * dim;
        for (int i = threadIdx.x; i < dim * dim; i += 256) Ms[i] = Mg[i];
        __syncthreads();
        M = Ms;
    }
    double loss_local = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < n; i += (int64_t)gridDim.x * 4) {
        const int a = ids1[i], b = ids2[i];
        const float *r1 = ent + (int64_t)a * ld, *r2 = ent + (int64_t)b * ld;
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < dim; c += 64) { const float u = r1[c], v = r2[c]; s1 += u * u; s2 += v * v; }
        s1 = oea::group_sum<64>(s1);
        s2 = oea::group_sum<64>(s2);
        const float i1 = l2norm ? rsqrtf(fmaxf(s1, 1e-12f)) : 1.f, i2 = l2norm ? rsqrtf(fmaxf(s2, 1e-12f)) : 1.f;
        for (int c = lane; c < dim; c += 64) y1[c] = r1[c] * i1;
        __builtin_amdgcn_wave_barrier();
        float sq = 0.f;
        // p[c] = sum_k y1[k] M[k][c]: M rows read coalesced across lanes, two columns per lane and eight rows per trip so
        // that 16 loads are in flight (one load per fma was L2-latency bound: 78 us for 166 links at d = 100).
        for (int c0 = lane; c0 < dim; c0 += 128) {
            const int c1 = c0 + 64;
            const bool two = c1 < dim;
            const float *m0 = M + c0, *m1 = M + (two ? c1 : c0);
            float p0 = 0.f, p1 = 0.f;
#pragma unroll 8
            for (int k = 0; k < dim; ++k) {
                const float y = y1[k];
                p0 = fmaf(y, m0[(int64_t)k * dim], p0);
                p1 = fmaf(y, m1[(int64_t)k * dim], p1);
            }
            const float d0 = r2[c0] * i2 - p0;
            df[c0] = d0;
            sq += d0 * d0;
            diff_out[i * ld + c0] = d0;
            e1_out[i * ld + c0] = y1[c0];
            oea::grad_add(ent_grad + (int64_t)b * ld + c0, 2.f * alpha * d0);
            if (two) {
                const float d1 = r2[c1] * i2 - p1;
                df[c1] = d1;
                sq += d1 * d1;
                diff_out[i * ld + c1] = d1;
                e1_out[i * ld + c1] = y1[c1];
                oea::grad_add(ent_grad + (int64_t)b * ld + c1, 2.f * alpha * d1);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // g1[k] = -2a sum_c diff[c] M[k][c]: lanes across the columns of row k (coalesced), one wave sum per row, eight rows
        // per trip; the sums land in y1 (its values are already stored) and the atomics go out one row element per lane.
        for (int k0 = 0; k0 < dim; k0 += 8) {
            float part[8];
            const float *mr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { part[u] = 0.f; mr[u] = M + (int64_t)min(k0 + u, dim - 1) * dim; }
            for (int c = lane; c < dim; c += 64) {
                const float d = df[c];
#pragma unroll
                for (int u = 0; u < 8; ++u) part[u] = fmaf(d, mr[u][c], part[u]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float g = oea::group_sum<64>(part[u]);
                if (lane == 0 && k0 + u < dim) y1[k0 + u] = g;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < dim; k += 64) oea::grad_add(ent_grad + (int64_t)a * ld + k, -2.f * alpha * y1[k]);
        if (lane == 0) { ent_touched[a] = 1.f; ent_touched[b] = 1.f; }
        sq = oea::group_sum<64>(sq);
        if (lane == 0) loss_local += (double)sq;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0 && loss_local != 0.0) atomicAdd(loss_accum, (double)alpha * loss_local);
}

__global__ __launch_bounds__(256) void mapping_orth_kernel(const float *__restrict__ M, int dim, float alpha,
                                                           float *__restrict__ orth, double *__restrict__ loss_accum) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    double sq = 0.0;
    if (idx < dim * dim) {
        const int k = idx / dim, j = idx % dim;
        const float *a = M + (int64_t)k * dim, *b = M + (int64_t)j * dim;
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < dim; ++c) s = fmaf(a[c], b[c], s);
        s -= (k == j) ? 1.f : 0.f;
        orth[idx] = s;
        sq = (double)s * (double)s;
    }
    sq = oea::wave_sum_d(sq);
    if ((threadIdx.x & 63) == 0 && sq != 0.0) atomicAdd(loss_accum, (double)alpha * sq);
}

__global__ __launch_bounds__(256) void mapping_update_kernel(const float *__restrict__ M, float *__restrict__ M_acc, int dim,
                                                             const float *__restrict__ e1, const float *__restrict__ diff,
                                                             int64_t n, int ld, const float *__restrict__ orth, float alpha,
                                                             float lr, int opt_kind, float *__restrict__ M_new) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= dim * dim) return;
    const int k = idx / dim, c = idx % dim;
    float ed = 0.f;                                      // loads eight links ahead of the fma chain (was latency bound)
#pragma unroll 8
    for (int64_t i = 0; i < n; ++i) ed = fmaf(e1[i * ld + k], diff[i * ld + c], ed);      // (E1^T Diff)[k][c]
    float om = 0.f;
#pragma unroll 8
    for (int j = 0; j < dim; ++j) om = fmaf(orth[k * dim + j], M[(int64_t)j * dim + c], om);
    const float g = alpha * (-2.f * ed + 4.f * om);
    const float m = M[idx];
    if (opt_kind == OEA_OPT_ADAGRAD) {
        const float acc = M_acc[idx] + g * g;
        M_acc[idx] = acc;
        M_new[idx] = m - lr * g / sqrtf(acc);
    } else {
        M_new[idx] = m - lr * g;
    }
}

}  // namespace

extern "C" {

size_t oea_mapping_workspace_floats(int64_t n_links, int32_t ld, int32_t dim) {
    return (size_t)(2 * n_links * ld + 2 * (int64_t)dim * dim + 64);
}

int oea_mapping_step(const float *ent, int32_t ld, int32_t dim, int32_t ent_l2_norm, const int32_t *ids1,
                     const int32_t *ids2, int64_t n, float *M, float *M_acc, float alpha, float lr, int32_t opt_kind,
                     void *ent_grad_, void *ent_touched_, float *work, double *loss_accum, void *stream) {
    oea::grad_t *ent_grad = static_cast<oea::grad_t *>(ent_grad_);
    oea::flag_t *ent_touched = static_cast<oea::flag_t *>(ent_touched_);
    OEA_REQUIRE(ent && ids1 && ids2 && M && ent_grad && ent_touched && work && loss_accum, "null pointer");
    OEA_REQUIRE(dim > 0 && dim <= ld && n >= 0, "dim <= ld");
    OEA_REQUIRE(opt_kind == OEA_OPT_SGD || (opt_kind == OEA_OPT_ADAGRAD && M_acc), "Adagrad needs the accumulator of M");
    OEA_REQUIRE(dim <= 2000, "dim <= 2000 (per-wave rows live in the default 64 KB of LDS)");
    hipStream_t st = oea::as_stream(stream);
    float *e1 = work, *diff = e1 + n * ld, *orth = diff + n * ld, *m_new = orth + (int64_t)dim * dim;
    if (n > 0 && dim <= 120)
        mapping_links_kernel<true><<<(unsigned)std::min<int64_t>(oea::ceil_div(n, 4), 1024), 256, sizeof(float) * (8 * dim + dim * dim), st>>>(
            ent, ld, dim, ent_l2_norm, ids1, ids2, n, M, alpha, ent_grad, ent_touched, e1, diff, loss_accum);
    else if (n > 0)
        mapping_links_kernel<false><<<(unsigned)std::min<int64_t>(oea::ceil_div(n, 4), 4096), 256, sizeof(float) * 8 * dim, st>>>(
            ent, ld, dim, ent_l2_norm, ids1, ids2, n, M, alpha, ent_grad, ent_touched, e1, diff, loss_accum);
    const unsigned nb = (unsigned)oea::ceil_div((int64_t)dim * dim, 256);
    mapping_orth_kernel<<<nb, 256, 0, st>>>(M, dim, alpha, orth, loss_accum);
    mapping_update_kernel<<<nb, 256, 0, st>>>(M, M_acc, dim, e1, diff, n, ld, orth, alpha, lr, opt_kind, m_new);
    OEA_CHECK_HIP(hipMemcpyAsync(M, m_new, sizeof(float) * (size_t)dim * dim, hipMemcpyDeviceToDevice, st));
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

// A whole mapping epoch of MTransE (approaches/mtranse.py:84-96: `steps` steps on n random seed links each) enqueued by ONE
// call: per step the fused mapping step above + the apply phase of the step engine (the optimiser on the entity rows the
// links touched).  batches: device int32 [steps, 2, n] (ids1 row, ids2 row per step).  Driven step by step from Python the
// epoch was bound by the host (two library calls + their argument marshalling per 50 us of kernels).
int oea_mapping_epoch(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel, int32_t dim,
                      int32_t ld, int32_t ent_l2_norm, const int32_t *batches, int32_t steps, int64_t n, float *M, float *M_acc,
                      float alpha, float lr, int32_t opt_kind, const oea_step_cfg *cfg, void *workspace, float *work,
                      double *mapping_loss_accum, double *step_loss_accum, void *stream) {
    OEA_REQUIRE(ent && rel && batches && M && cfg && workspace && work && mapping_loss_accum && step_loss_accum, "null pointer");
    OEA_REQUIRE(steps >= 0 && n >= 0, "steps, n >= 0");
    void *eg = nullptr, *et = nullptr;
    int rc = oea_step_entity_scratch(workspace, n_ent, n_rel, ld, &eg, &et);
    if (rc != OEA_OK) return rc;
    oea_step_cfg step_cfg = *cfg;
    for (int32_t s = 0; s < steps; ++s) {
        const int32_t *ids1 = batches + (int64_t)s * 2 * n, *ids2 = ids1 + n;
        rc = oea_mapping_step(ent, ld, dim, ent_l2_norm, ids1, ids2, n, M, M_acc, alpha, lr, opt_kind, eg, et, work,
                              mapping_loss_accum, stream);
        if (rc != OEA_OK) return rc;
        rc = oea_triple_step_phase(ent, ent_acc, n_ent, rel, rel_acc, n_rel, dim, ld, nullptr, 0, nullptr, 0, &step_cfg, workspace,
                                   step_loss_accum, OEA_PHASE_APPLY, stream);
        if (rc != OEA_OK) return rc;
        ++step_cfg.opt_t;
    }
    return OEA_OK;
}

}  // extern "C"
