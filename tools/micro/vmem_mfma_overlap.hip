// Do 16-byte-per-lane global loads of an L2-resident stream overlap with bf16 MFMAs -- of the same wave, of a partner wave on the
// same SIMD?  Per iteration and wave: NM x v_mfma_f32_32x32x16_bf16 (independent accumulators) and NL x global_load_dwordx4 (1 KiB per
// wave-instruction, ring of RING iterations ahead), the loaded values feed the next iteration's MFMA operands (so the loads are
// really waited for).  Layout of an iteration: MODE 0 = all loads first, then the MFMAs (burst); 1 = one load after every NM / NL MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NL, int MODE, int RING>
__global__ __launch_bounds__(512, 1) void k(const u32x4* __restrict__ w, float* out, int iters, int kib) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    constexpr int NLR = NL > 0 ? NL : 1;
    u32x4 ring[RING][NLR];
    const int per_wave = kib / nw;
    int pos = 0;
    auto ld = [&](u32x4 (&dst)[NLR]) {
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            dst[u] = w[(long)(wave * per_wave + pos) * 64 + lane];
            pos = pos + 1 == per_wave ? 0 : pos + 1;
        }
    };
#pragma unroll
    for (int r = 0; r < RING - 1; ++r) ld(ring[r]);
    u32x4 b = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    for (int it = 0; it < iters; it += RING) {
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            u32x4 (&cur)[NLR] = ring[r];
            u32x4 (&nxt)[NLR] = ring[(r + RING - 1) % RING];
            if (MODE == 0) {
                ld(nxt);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < NM; ++m)
                    acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, NL ? cur[m % NLR] : b), __builtin_bit_cast(bf16x8, b), acc[m & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                constexpr int PER = NL > 0 ? NM / NLR : NM;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, NL ? cur[m % NLR] : b), __builtin_bit_cast(bf16x8, b), acc[m & 3], 0, 0, 0);
                    if (NL > 0 && m % PER == PER - 1) {
                        const int u = m / PER;
                        nxt[u] = w[(long)(wave * per_wave + pos) * 64 + lane];
                        pos = pos + 1 == per_wave ? 0 : pos + 1;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (NM == 0) for (int r = 0; r < RING; ++r) for (int u = 0; u < NLR; ++u) s += __uint_as_float(ring[r][u][0] ^ ring[r][u][1] ^ ring[r][u][2] ^ ring[r][u][3]);
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static u32x4* W; static float* OUT;
template <int NM, int NL, int MODE, int RING>
float run(int threads) {
    const int iters = 1536, kib = 1536;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NM, NL, MODE, RING><<<256, threads>>>(W, OUT, 12, kib);
    hipEventRecord(e0);
    k<NM, NL, MODE, RING><<<256, threads>>>(W, OUT, iters, kib);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters;                        // ns per iteration
}

int main() {
    hipMalloc(&W, 1536 * 1024); hipMalloc(&OUT, 256 * 512 * 4);
    hipMemset(W, 0, 1536 * 1024);
    for (int threads = 256; threads <= 512; threads += 256) {
        printf("== %d waves per SIMD (MFMA 32x32x16 bf16 = 32 cycles = 13.3 ns at 2.4 GHz)\n", threads / 256);
        const float m24 = run<24, 0, 0, 3>(threads), l12 = run<0, 12, 0, 3>(threads), b = run<24, 12, 0, 3>(threads), i = run<24, 12, 1, 3>(threads);
        printf("24 MFMA : 12 loads (the chain kernel's layer 1)   MFMAs only %6.1f ns   loads only %6.1f   burst %6.1f   interleaved %6.1f\n", m24, l12, b, i);
        const float m6 = run<6, 0, 0, 6>(threads), l3 = run<0, 3, 0, 6>(threads), b2 = run<6, 3, 0, 6>(threads), i2 = run<6, 3, 1, 6>(threads);
        printf(" 6 MFMA :  3 loads (its layer 2)                   MFMAs only %6.1f ns   loads only %6.1f   burst %6.1f   interleaved %6.1f\n", m6, l3, b2, i2);
        const float m24b = run<24, 0, 0, 3>(threads), l6 = run<0, 6, 0, 3>(threads), b3 = run<24, 6, 0, 3>(threads), i3 = run<24, 6, 1, 3>(threads);
        printf("24 MFMA :  6 loads (half the weight bytes)        MFMAs only %6.1f ns   loads only %6.1f   burst %6.1f   interleaved %6.1f\n", m24b, l6, b3, i3);
    }
    return 0;
}
