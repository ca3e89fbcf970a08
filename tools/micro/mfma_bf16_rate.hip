// Micro-benchmark: v_mfma_f32_32x32x16_bf16 issue rate (cycles per MFMA per SIMD from s_memtime AND from wall time), for
// register operands with constant data, random-like data, and one / two waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, const bf16x8* src) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[3], b[3];
    for (int p = 0; p < 3; ++p) { a[p] = src[threadIdx.x * 6 + p]; b[p] = src[threadIdx.x * 6 + 3 + p]; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            f32x16 c = acc[i];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
            acc[i] = c;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
void run(int blocks_per_cu, float* out, unsigned long long* cyc, const bf16x8* src, const char* what) {
    const int iters = 2048;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<256 * blocks_per_cu, 256>>>(out, cyc, 16, src);
    hipEventRecord(e0);
    k<NACC><<<256 * blocks_per_cu, 256>>>(out, cyc, iters, src);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * NACC * 6;                  // MFMAs per wave
    printf("%-8s acc=%d blocks/CU=%d  %.3f ms  %.0f TF/s  %.1f counter ticks/MFMA  %.1f ns/MFMA/wave\n", what, NACC, blocks_per_cu, ms,
           256.0 * blocks_per_cu * 4 * n * 32768 / ms / 1e9, (double)c / n, ms * 1e6 / n);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    unsigned long long* cyc; hipMalloc(&cyc, 8);
    unsigned short* h = (unsigned short*)malloc(256 * 6 * 16);
    bf16x8* src; hipMalloc(&src, 256 * 6 * 16);
    for (int mode = 0; mode < 2; ++mode) {
        unsigned x = 12345;
        for (int i = 0; i < 256 * 6 * 8; ++i) {
            x = x * 1664525u + 1013904223u;
            h[i] = mode ? (unsigned short)(0x3c00 + ((x >> 12) & 0x3ff) + ((x >> 2) & 0x8000)) : (unsigned short)0x3f80;   // random ~[0.008,0.03] with sign : 1.0
        }
        hipMemcpy(src, h, 256 * 6 * 16, hipMemcpyHostToDevice);
        for (int b = 1; b <= 2; ++b) { run<1>(b, out, cyc, src, mode ? "random" : "ones"); run<4>(b, out, cyc, src, mode ? "random" : "ones"); }
    }
    return 0;
}
