// optim.hip -- dense Adam with TensorFlow-1 semantics (tf.train.AdamOptimizer, used by AliNet
// alinet.py:871 and RDGCN rdgcn.py:332):
//     lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
//     m = beta1*m + (1-beta1)*g ;  v = beta2*v + (1-beta2)*g^2 ;  p -= lr_t * m / (sqrt(v) + eps)
// (epsilon is TF's "epsilon hat": added to sqrt(v) without bias correction.)  One fused pass:
// 16 B read + 12 B written per element.
#include "common.h"

namespace {
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, int64_t n, float lr_t, float b1, float b2, float eps) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}
}  // namespace

extern "C" int oea_adam_dense(float *param, const float *grad, float *m, float *v, int64_t n, float lr, float beta1,
                              float beta2, float eps, int64_t t, void *stream) {
    OEA_REQUIRE(param && grad && m && v && t >= 1, "null pointer / t >= 1");
    if (n == 0) return OEA_OK;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t));
    adam_kernel<<<(unsigned)std::min<int64_t>(oea::ceil_div(n, 256), 8192), 256, 0, oea::as_stream(stream)>>>(
        param, grad, m, v, n, (float)lr_t, beta1, beta2, eps);
    OEA_CHECK_HIP(hipGetLastError());
    return OEA_OK;
}
