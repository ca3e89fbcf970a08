// Does v_mfma_f32_32x32x16_bf16 overlap VALU work -- of the same wave (instructions placed between dependent MFMAs) or of a partner
// wave on the same SIMD?  Round 5: the split-bf16 Winograd kernel's ablations add up (MFMA + split + loads = total), which says "no
// overlap"; this measures the bare mechanism.  Per iteration and wave: 6 dependent MFMAs (one accumulator, like one transform position)
// and NX VALU instructions per MFMA of kind KIND (1: v_fma_f32, 2: the split's mix -- cvt_pk / and / shift / sub).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

template <int KIND, int NX, int NM>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a0) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a0 + i + threadIdx.x * 1e-6f;
    unsigned acc_u = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            if (NM) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
#pragma unroll
            for (int x = 0; x < NX; ++x) {
                if (KIND == 1) v[x & 7] = v[x & 7] * 1.0001f + 0.5f;
                if (KIND == 2) {                         // one level of the split on a pair of values: cvt_pk, shift, and, two subs (5 VALU)
                    if (x % 5 == 0) {
                        const unsigned p = cvt_pk(v[x & 7], v[(x + 1) & 7]);
                        acc_u ^= p;
                        v[x & 7] -= __uint_as_float(p << 16);
                        v[(x + 1) & 7] -= __uint_as_float(p & 0xffff0000u);
                    }
                }
            }
        }
    }
    float s = __uint_as_float(acc_u);
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND, int NX, int NM>
void run(const char* name, int threads, float* out) {
    const int iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, NX, NM><<<256, threads>>>(out, 8, 1.f);
    hipEventRecord(e0);
    k<KIND, NX, NM><<<256, threads>>>(out, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int wps = threads / 256;                       // waves per SIMD
    printf("%-22s VALU/MFMA %2d  waves/SIMD %d : %7.1f ns per iteration of 6 MFMA-slots per wave = %6.1f cycles @2.4GHz per SIMD-iteration-pair\n", name, NX, wps,
           ms * 1e6 / iters, ms * 1e-3 * 2.4e9 / iters);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    for (int t = 256; t <= 512; t += 256) {
        run<0, 0, 1>("mfma only", t, out);
        run<1, 6, 0>("fma only", t, out); run<1, 9, 0>("fma only", t, out);
        run<1, 6, 1>("mfma + fma", t, out); run<1, 9, 1>("mfma + fma", t, out); run<1, 12, 1>("mfma + fma", t, out);
        run<2, 10, 0>("split mix only", t, out);
        run<2, 10, 1>("mfma + split mix", t, out);
    }
    return 0;
}
