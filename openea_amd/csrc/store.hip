// store.hip -- embedding store, gathers, row normalisation, error plumbing.
// Replaces tf.Variable tables + tf.nn.embedding_lookup + .eval() round trips
// (reference: models/basic_model.py:73-121,184-204; modules/base/initializers.py:26).
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "common.h"

namespace oea {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static bool g_prof = false;
static std::vector<hipEvent_t> g_events;
static size_t g_nmarks = 0;
bool prof_enabled() { return g_prof; }
static int g_stride = 1;
static long g_calls = 0;
static bool g_sampled = false;
// first thing a profiled entry point does: decides whether THIS call records its marks
void prof_call() { g_sampled = g_prof && (g_calls++ % g_stride == 0); }
void prof_mark(hipStream_t st) {
    if (!g_sampled) return;
    if (g_nmarks == g_events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        g_events.push_back(e);
    }
    (void)hipEventRecord(g_events[g_nmarks++], st);
}
bool prof_pair(hipEvent_t *start, hipEvent_t *stop) {
    if (!g_sampled) return false;
    while (g_events.size() < g_nmarks + 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return false;
        g_events.push_back(e);
    }
    *start = g_events[g_nmarks++];
    *stop = g_events[g_nmarks++];
    return true;
}
}  // namespace oea

struct oea_store {
    int64_t rows;
    int32_t dim, ld;
    float *dev;
};

namespace {
__global__ __launch_bounds__(256) void pair_dots_kernel(const float *__restrict__ e1, int ld1, const float *__restrict__ e2, int ld2,
                                                        int dim, const int32_t *__restrict__ ii, const int32_t *__restrict__ jj,
                                                        int64_t n, float *__restrict__ out) {
    const int lane = threadIdx.x & 15;                                   // 16 lanes per pair
    const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (p >= n) return;
    const float *a = e1 + (int64_t)ii[p] * ld1, *b = e2 + (int64_t)jj[p] * ld2;
    float s = 0.f;
    for (int c = lane; c < dim; c += 16) s = fmaf(a[c], b[c], s);
    s = oea::group_sum<16>(s);
    if (lane == 0) out[p] = s;
}
}  // namespace

extern "C" {

int oea_version(void) { return 100; }
const char *oea_last_error(void) { return oea::g_err; }
int oea_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

int oea_profile_begin(int32_t stride) {
    OEA_REQUIRE(stride >= 1, "stride >= 1");
    oea::g_prof = true;
    oea::g_nmarks = 0;
    oea::g_stride = stride;
    oea::g_calls = 0;
    return OEA_OK;
}
/* marks come in groups of `group` consecutive events per profiled call; out_ms[j] receives the
 * SUM over calls of the time between mark j and mark j+1 of each group (j < group-1). */
int oea_profile_end(int32_t group, double *out_ms_host, int32_t *n_calls_host) {
    oea::g_prof = false;
    oea::g_sampled = false;
    OEA_REQUIRE(group >= 2 && out_ms_host && n_calls_host, "group >= 2");
    const size_t n = oea::g_nmarks / (size_t)group;
    for (int j = 0; j < group - 1; ++j) out_ms_host[j] = 0.0;
    if (n > 0) OEA_CHECK_HIP(hipEventSynchronize(oea::g_events[n * group - 1]));
    for (size_t i = 0; i < n; ++i)
        for (int j = 0; j < group - 1; ++j) {
            float ms = 0.f;
            // odd j: from the stop event of one dispatch to the start event of the next (the gap between the kernels);
            // the runtime may refuse that pairing for dispatch-bound events: the gap is then left out
            const hipError_t e = hipEventElapsedTime(&ms, oea::g_events[i * group + j], oea::g_events[i * group + j + 1]);
            if (e != hipSuccess) {
                if ((j & 1) == 0) OEA_CHECK_HIP(e);
                (void)hipGetLastError();
                ms = 0.f;
            }
            out_ms_host[j] += (double)ms;
        }
    *n_calls_host = (int32_t)n;
    oea::g_nmarks = 0;
    return OEA_OK;
}

int oea_pair_dots(const float *e1, int32_t ld1, const float *e2, int32_t ld2, int32_t dim, const int32_t *ii,
                  const int32_t *jj, int64_t n, float *out, void *stream) {
    OEA_REQUIRE(e1 && e2 && (n == 0 || (ii && jj && out)) && dim > 0 && dim <= ld1 && dim <= ld2, "shapes");
    if (n == 0) return OEA_OK;
    pair_dots_kernel<<<(unsigned)oea::ceil_div(n * 16, 256), 256, 0, oea::as_stream(stream)>>>(e1, ld1, e2, ld2, dim, ii, jj, n, out);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_store_create(int64_t rows, int32_t dim, oea_store_t *out) {
    OEA_REQUIRE(rows > 0 && dim > 0 && out, "rows, dim > 0");
    oea_store *s = new oea_store;
    s->rows = rows;
    s->dim = dim;
    s->ld = (dim + 3) / 4 * 4;
    s->dev = nullptr;
    hipError_t e = hipMalloc(&s->dev, sizeof(float) * (size_t)rows * s->ld);
    if (e != hipSuccess) {
        oea::set_error("hipMalloc(%zu) failed: %s", sizeof(float) * (size_t)rows * s->ld, hipGetErrorString(e));
        delete s;
        return OEA_ENOMEM;
    }
    e = hipMemset(s->dev, 0, sizeof(float) * (size_t)rows * s->ld);
    if (e != hipSuccess) { (void)hipFree(s->dev); delete s; oea::set_error("hipMemset failed"); return OEA_EHIP; }
    *out = s;
    return OEA_OK;
}
int oea_store_destroy(oea_store_t s) {
    if (!s) return OEA_OK;
    (void)hipFree(s->dev);
    delete s;
    return OEA_OK;
}
int64_t oea_store_rows(oea_store_t s) { return s->rows; }
int32_t oea_store_dim(oea_store_t s) { return s->dim; }
int32_t oea_store_ld(oea_store_t s) { return s->ld; }
float *oea_store_rows_ptr(oea_store_t s) { return s->dev; }

int oea_store_load_host(oea_store_t s, const float *src_host, void *stream) {
    OEA_REQUIRE(s && src_host, "null");
    OEA_CHECK_HIP(hipMemcpy2DAsync(s->dev, sizeof(float) * s->ld, src_host, sizeof(float) * s->dim,
                                   sizeof(float) * s->dim, (size_t)s->rows, hipMemcpyHostToDevice,
                                   oea::as_stream(stream)));
    return OEA_OK;
}
int oea_store_save_host(oea_store_t s, float *dst_host, void *stream) {
    OEA_REQUIRE(s && dst_host, "null");
    OEA_CHECK_HIP(hipMemcpy2DAsync(dst_host, sizeof(float) * s->dim, s->dev, sizeof(float) * s->ld,
                                   sizeof(float) * s->dim, (size_t)s->rows, hipMemcpyDeviceToHost,
                                   oea::as_stream(stream)));
    OEA_CHECK_HIP(hipStreamSynchronize(oea::as_stream(stream)));
    return OEA_OK;
}

}  // extern "C"

namespace {

// One G-lane group per row, float4 per lane per iteration; rows of ld floats (ld % 4 == 0).
template <int G>
__global__ void gather_rows_kernel(const float *__restrict__ table, int dim, int ld,
                                   const int32_t *__restrict__ ids, int64_t n, int normalize,
                                   float *__restrict__ out, int out_ld) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t i = grp; i < n; i += ngrp) {
        const float *src = table + (int64_t)ids[i] * ld;
        float ss = 0.f;
        if (normalize) {
            for (int c = lane * 4; c < ld; c += G * 4) {
                float4 v = oea::ld4(src + c);
                ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
            ss = oea::group_sum<G>(ss);
        }
        const float inv = normalize ? rsqrtf(fmaxf(ss, 1e-12f)) : 1.0f;
        for (int c = lane * 4; c < out_ld; c += G * 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < ld) v = oea::ld4(src + c);
            float e[4] = {v.x * inv, v.y * inv, v.z * inv, v.w * inv};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c + q >= dim) e[q] = 0.f;
            oea::st4(out + i * out_ld + c, make_float4(e[0], e[1], e[2], e[3]));
        }
    }
}

template <int G>
__global__ void normalize_rows_kernel(float *__restrict__ table, int64_t rows, int dim, int ld, int sk) {
    const int lane = threadIdx.x % G;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngrp = (int64_t)gridDim.x * blockDim.x / G;
    for (int64_t i = grp; i < rows; i += ngrp) {
        float *row = table + i * ld;
        float ss = 0.f;
        for (int c = lane * 4; c < ld; c += G * 4) {
            float4 v = oea::ld4(row + c);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        ss = oea::group_sum<G>(ss);
        float inv;
        if (sk) {  // sklearn.preprocessing.normalize: x / ||x||, zero rows unchanged
            const float nrm = sqrtf(ss);
            inv = nrm > 0.f ? 1.0f / nrm : 1.0f;
        } else {
            inv = rsqrtf(fmaxf(ss, 1e-12f));
        }
        for (int c = lane * 4; c < ld; c += G * 4) {
            float4 v = oea::ld4(row + c);
            oea::st4(row + c, make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv));
        }
    }
}

__global__ void fill_kernel(float *p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = v;
}

}  // namespace

extern "C" {

int oea_gather_rows(const float *table, int32_t dim, int32_t ld, const int32_t *ids, int64_t n,
                    int32_t normalize, float *out, int32_t out_ld, void *stream) {
    OEA_REQUIRE(table && ids && out, "null pointer");
    OEA_REQUIRE(ld % 4 == 0 && out_ld % 4 == 0 && dim <= ld && dim <= out_ld, "ld % 4 == 0 and dim <= ld");
    if (n == 0) return OEA_OK;
    const int block = 256;
    if (ld <= 64) {
        const int64_t grid = oea::ceil_div(n, block / 16);
        gather_rows_kernel<16><<<dim3((unsigned)std::min<int64_t>(grid, 65535)), block, 0, oea::as_stream(stream)>>>(
            table, dim, ld, ids, n, normalize, out, out_ld);
    } else if (ld <= 128) {
        const int64_t grid = oea::ceil_div(n, block / 32);
        gather_rows_kernel<32><<<dim3((unsigned)std::min<int64_t>(grid, 65535)), block, 0, oea::as_stream(stream)>>>(
            table, dim, ld, ids, n, normalize, out, out_ld);
    } else {
        const int64_t grid = oea::ceil_div(n, block / 64);
        gather_rows_kernel<64><<<dim3((unsigned)std::min<int64_t>(grid, 65535)), block, 0, oea::as_stream(stream)>>>(
            table, dim, ld, ids, n, normalize, out, out_ld);
    }
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_normalize_rows(float *table, int64_t rows, int32_t dim, int32_t ld, int32_t sk, void *stream) {
    OEA_REQUIRE(table && ld % 4 == 0 && dim <= ld, "table, ld % 4 == 0");
    if (rows == 0) return OEA_OK;
    const int block = 256;
    const int64_t grid = std::min<int64_t>(oea::ceil_div(rows, block / 32), 65535);
    normalize_rows_kernel<32><<<dim3((unsigned)grid), block, 0, oea::as_stream(stream)>>>(table, rows, dim, ld, sk);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

int oea_fill_f32(float *p, int64_t n, float value, void *stream) {
    OEA_REQUIRE(p || n == 0, "null");
    if (n == 0) return OEA_OK;
    const int64_t grid = std::min<int64_t>(oea::ceil_div(n, 256), 4096);
    fill_kernel<<<dim3((unsigned)grid), 256, 0, oea::as_stream(stream)>>>(p, n, value);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}

// device <-> host copies ordered on `stream` and complete on return (the host side of a collective callback stages its
// buffers with these: oea_comm_init_callbacks)
int oea_copy_to_host(const void *dev, void *host, size_t bytes, void *stream) {
    OEA_REQUIRE((dev && host) || bytes == 0, "null");
    if (bytes == 0) return OEA_OK;
    OEA_CHECK_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, oea::as_stream(stream)));
    OEA_CHECK_HIP(hipStreamSynchronize(oea::as_stream(stream)));
    return OEA_OK;
}

int oea_copy_from_host(void *dev, const void *host, size_t bytes, void *stream) {
    OEA_REQUIRE((dev && host) || bytes == 0, "null");
    if (bytes == 0) return OEA_OK;
    OEA_CHECK_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, oea::as_stream(stream)));
    OEA_CHECK_HIP(hipStreamSynchronize(oea::as_stream(stream)));
    return OEA_OK;
}

}  // extern "C"
