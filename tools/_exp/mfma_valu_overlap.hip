// Does VALU work overlap with v_mfma_f32_32x32x16_bf16 on gfx950 -- (a) inside one wave (independent VALU ops between MFMAs),
// (b) between two waves of the same SIMD (one issuing MFMAs, the other VALU)?   hipcc --offload-arch=gfx950 -O3 -o ovl this.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NV>
__global__ __launch_bounds__(256) void same_wave(float *out, int iters) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(1.0f + i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NV; ++u) v[u & 7] = v[u & 7] * 1.0001f + 0.5f;       // one v_fma each, 8 independent chains
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 512 threads: waves 0-3 (one per SIMD) MFMA only, waves 4-7 VALU only (mode 3); mode 1: only the MFMA waves work; mode 2: only VALU
__global__ __launch_bounds__(512) void two_waves(float *out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc[4];
            for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
            bf16x8 x, y;
            for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(1.0f + i); }
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int m = 0; m < 12; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m & 3], 0, 0, 0);
            for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
        }
    } else if (mode & 2) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 96; ++u) v[u & 7] = v[u & 7] * 1.0001f + 0.5f;      // 96 VALU ops = 384 cycles = the 12 MFMAs' pipe time
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static float timed(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * 512 * 8);
    const int iters = 20000, grid = 256;              // one workgroup per CU: one wave per SIMD (same_wave), two per SIMD (two_waves)
    printf("same wave, 12 MFMA (384 pipe cycles) + 12 x NV VALU per iteration, %d iterations:\n", iters);
    printf("  NV=0  %.3f ms\n", timed([&] { same_wave<0><<<grid, 256>>>(out, iters); }));
    printf("  NV=2  %.3f ms\n", timed([&] { same_wave<2><<<grid, 256>>>(out, iters); }));
    printf("  NV=4  %.3f ms\n", timed([&] { same_wave<4><<<grid, 256>>>(out, iters); }));
    printf("  NV=6  %.3f ms\n", timed([&] { same_wave<6><<<grid, 256>>>(out, iters); }));
    printf("  NV=8  %.3f ms\n", timed([&] { same_wave<8><<<grid, 256>>>(out, iters); }));
    printf("  NV=12 %.3f ms\n", timed([&] { same_wave<12><<<grid, 256>>>(out, iters); }));
    printf("two waves per SIMD (one MFMA, one VALU of equal pipe time):\n");
    printf("  MFMA wave alone  %.3f ms\n", timed([&] { two_waves<<<grid, 512>>>(out, iters, 1); }));
    printf("  VALU wave alone  %.3f ms\n", timed([&] { two_waves<<<grid, 512>>>(out, iters, 2); }));
    printf("  both             %.3f ms\n", timed([&] { two_waves<<<grid, 512>>>(out, iters, 3); }));
    return 0;
}
