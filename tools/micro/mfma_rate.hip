// Micro-benchmark: v_mfma_f32_32x32x2_f32 issue rate vs number of independent accumulators and waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // a0 < 0: random-looking operands (hash of the lane), magnitude |a0|: data-dependent power -> sustained clock
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    if (a0 < 0.f) {
        unsigned h = (threadIdx.x + 1u) * 2654435761u + blockIdx.x * 40503u;
        a = -a0 * ((int)(h >> 8) - (1 << 23)) * (1.f / (1 << 23));
        h = h * 1664525u + 1013904223u;
        b = b0 * ((int)(h >> 8) - (1 << 23)) * (1.f / (1 << 23));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, float* out, float a0 = 1.0001f) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<256 * blocks_per_cu, 256>>>(out, 16, 1.f, 1.f);
    hipEventRecord(e0);
    k<NACC><<<256 * blocks_per_cu, 256>>>(out, iters, a0, 0.9999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = 256.0 * blocks_per_cu * 4 * iters * 16;          // MFMAs
    printf("%s acc=%d blocks/CU=%d  %.3f ms  %.1f TF/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", a0 < 0 ? "random  " : "constant", NACC, blocks_per_cu, ms,
           mf * 4096 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 16.0 * blocks_per_cu));
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int b = 1; b <= 4; b *= 2) { run<1>(b, out); run<2>(b, out); run<4>(b, out); }
    for (int b = 1; b <= 2; b *= 2) { run<4>(b, out, -1.f); run<4>(b, out, -0.01f); }
    return 0;
}
