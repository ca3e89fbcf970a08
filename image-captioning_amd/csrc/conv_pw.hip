// conv_pw.hip -- streaming kernel for the short-K pointwise convolutions of the encoder (1x1 / stride 1, Cin = 64, 128 or 256:
// ResNet's `2c` expansions, the stage-2 branches, the FPN C2 lateral).
//
// The general implicit-GEMM kernel is the wrong shape for these layers: with K = 64..256 a block lives for 2-8 K-tiles, so its
// start-up (first-tile latency), LDS transpose and output burst are most of its life, and a grid of equal blocks runs them in
// lock-step.  Here there is no K loop and no barrier after start-up:
//   * a block keeps a 64 KB slice of the weights (16384 / Cin output channels x Cin) in LDS for its whole life;
//   * every wave walks over 32-pixel strips on its own: the strip's activations go straight from global memory into the
//     registers that ARE the MFMA B operands (lane = pixel, 16-byte loads of the lane's contiguous half of the K run; the K
//     order is permuted the same way on the weight side), the A operands are one ds_read_b128 per four MFMAs;
//   * channels sit on the MFMA's M side, so an accumulator quad is four consecutive output channels of the lane's pixel: the
//     folded-BN scale/shift, residual, ReLU and the store are 16-byte accesses straight from the accumulators (no LDS
//     transpose);
//   * two blocks (= two waves per SIMD) per CU at independent phases: one wave's loads / stores overlap the other's MFMAs.
// v_mfma_f32_32x32x2_f32: D[channel][pixel] += W[channel][k] * X[pixel][k]; lane (i, h): k = (Cin/2) h + s at step s.
#include "igemm_core.h"
#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace dcap {

template <int K>
constexpr size_t pw_lds_bytes() {
    return ((size_t)(16384 / K) * (K + 4) + 2 * (size_t)(16384 / K)) * sizeof(float);
}

template <int K, int RES>
__global__ __launch_bounds__(256, 2) void pwconv_stream_kernel(const float* __restrict__ x, const float* __restrict__ w, Epilogue ep, int M,
                                                              int N, int gx, int gy, int xcd_map) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NC = 16384 / K, LDW = K + 4, KH = K / 2;
    float* const Ws = smem;
    float* const scs = smem + NC * LDW;
    float* const shs = scs + NC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    // 1-D grid of gx * gy blocks, dealt round-robin over the 8 XCDs: the gx weight slices of one pixel group (which read the same
    // strips in the same order) are given consecutive slots of ONE XCD, so the strips are fetched from HBM once and shared through
    // that XCD's L2 (with slice = blockIdx.x on a 2-D grid the four slices of fpn_c2p2 sat on four XCDs: 554 MB fetched for 134 MB).
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int slice = xcd_map ? slot % gx : bid % gx;
    const int grp = xcd_map ? (slot / gx) * 8 + xcd : bid / gx;        // grp >= gy: a padding block of the last round
    if (grp >= gy) return;
    const int n0 = slice * NC;
    const int nc = min(NC, N - n0);                     // host-checked: a multiple of 64
    for (int idx = tid; idx < nc * (K / 4); idx += 256) {
        const int c = idx / (K / 4), k4 = idx - c * (K / 4);
        *reinterpret_cast<f4*>(&Ws[c * LDW + 4 * k4]) = *reinterpret_cast<const f4*>(w + (long)(n0 + c) * K + 4 * k4);
    }
    for (int c = tid; c < nc; c += 256) {
        scs[c] = ep.scale ? ep.scale[n0 + c] : 1.f;
        shs[c] = ep.shift ? ep.shift[n0 + c] : 0.f;
    }
    __syncthreads();
    const int strips = (M + 31) / 32;
    for (int st = grp * 4 + wave; st < strips; st += gy * 4) {
        const int prow = st * 32 + i;
        const bool pv = prow < M;
        const int p = min(prow, M - 1);
        const float* xr = x + (long)p * K + KH * h;
        f4 xs[KH / 4];
#pragma unroll
        for (int j = 0; j < KH / 4; ++j) xs[j] = *reinterpret_cast<const f4*>(xr + 4 * j);
        const float* rrow = nullptr;
        if constexpr (RES != 0) rrow = ep.res_row(p) + n0 + 4 * h;
        float* const yrow = ep.C + (long)p * ep.ldc + n0 + 4 * h;
        for (int cb = 0; cb < nc; cb += 64) {
            f4 r0[4], r1[4];
            if constexpr (RES != 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    r0[q] = *reinterpret_cast<const f4*>(rrow + cb + 8 * q);
                    r1[q] = *reinterpret_cast<const f4*>(rrow + cb + 32 + 8 * q);
                }
            }
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            const float* wa0 = &Ws[(cb + i) * LDW + KH * h];
            const float* wa1 = wa0 + 32 * LDW;
#pragma unroll
            for (int j = 0; j < KH / 4; ++j) {
                const f4 a0 = *reinterpret_cast<const f4*>(wa0 + 4 * j);
                const f4 a1 = *reinterpret_cast<const f4*>(wa1 + 4 * j);
                const f4 b = xs[j];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
            }
            // accumulator quad q of lane (i, h) = channels cb + 8q + 4h .. +3 of pixel i
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4 sc0 = *reinterpret_cast<const f4*>(&scs[cb + 8 * q + 4 * h]), sh0 = *reinterpret_cast<const f4*>(&shs[cb + 8 * q + 4 * h]);
                const f4 sc1 = *reinterpret_cast<const f4*>(&scs[cb + 32 + 8 * q + 4 * h]), sh1 = *reinterpret_cast<const f4*>(&shs[cb + 32 + 8 * q + 4 * h]);
                f4 v0 = f4{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]} * sc0 + sh0;
                f4 v1 = f4{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]} * sc1 + sh1;
                if constexpr (RES != 0) { v0 += r0[q]; v1 += r1[q]; }
                if (ep.relu) {
                    v0 = f4{fmaxf(v0.x, 0.f), fmaxf(v0.y, 0.f), fmaxf(v0.z, 0.f), fmaxf(v0.w, 0.f)};
                    v1 = f4{fmaxf(v1.x, 0.f), fmaxf(v1.y, 0.f), fmaxf(v1.z, 0.f), fmaxf(v1.w, 0.f)};
                }
                if (pv) {
                    *reinterpret_cast<f4*>(yrow + cb + 8 * q) = v0;
                    *reinterpret_cast<f4*>(yrow + cb + 32 + 8 * q) = v1;
                }
            }
        }
    }
}

template <int K>
static int launch_pw(const dc_conv_desc* d, const Epilogue& ep, int M, int N, hipStream_t s) {
    constexpr int NC = 16384 / K;
    constexpr size_t lds = pw_lds_bytes<K>();
    const int gx = (N + NC - 1) / NC;
    const int strips = (M + 31) / 32;
    const int gy = std::max(1, std::min((2 * kNumCU) / gx, (strips + 3) / 4));
    constexpr int xcd_map = 1;                           // the slices of one pixel group share an XCD (round 2: dealt over the XCDs, fpn_c2p2 fetched its input 4x)
    const dim3 grid(gx * ((gy + 7) / 8) * 8);            // whole rounds of 8 XCDs; the kernel drops the padding blocks
#define PW_LAUNCH(RES_)                                                                                                          \
    do {                                                                                                                              \
        DC_ENSURE_DYN_LDS((&pwconv_stream_kernel<K, RES_>), 160 * 1024);                                                              \
        hipLaunchKernelGGL((pwconv_stream_kernel<K, RES_>), grid, dim3(256), lds, s, d->x, d->w, ep, M, N, gx, gy, xcd_map);                           \
    } while (0)
    if (ep.res_mode == 0) PW_LAUNCH(0);
    else if (ep.res_mode == 1) PW_LAUNCH(1);
    else PW_LAUNCH(2);
#undef PW_LAUNCH
    return check_launch("pwconv_stream_kernel");
}

// true when the streaming kernel takes this (already validated, f32-math, pointwise) convolution
bool conv_pw_stream_supported(const dc_conv_desc* d, const Epilogue& ep) {
    if (d->Cin != 64 && d->Cin != 128 && d->Cin != 256) return false;
    if (d->Cout < 64 || (d->Cout & 63) != 0 || !ep.vec4) return false;
    // measured (tools/conv_bench.py, 2 x 1024^2): 64>64 20.5 vs 22.0 us, 64>256 +res 84.2 vs 92.8, 128>512 +res 55.3 vs 62.0,
    // 256>256 +up 165.9 vs 182.2 -- but 256>64 53.4 vs 51.6 and 256>1024 50.8 vs 45.2: with Cin = 256 a slice is only 64 channels,
    // so sixteen slices re-read every strip and the start-up burst (one strip = 32 KB per wave) costs more than the loop saves
    if (d->Cin == 256 && (d->Cout < 128 || d->Cout > 256)) return false;
    if (ep.res_mode < 0 || ep.res_mode > 2 || ep.accumulate || ep.Cb) return false;
    if (((uintptr_t)d->x & 15) != 0 || ((uintptr_t)d->w & 15) != 0) return false;
    return true;
}

int conv2d_pointwise_stream(const dc_conv_desc* d, const Epilogue& ep, int M, int N, hipStream_t s) {
    switch (d->Cin) {
        case 64: return launch_pw<64>(d, ep, M, N, s);
        case 128: return launch_pw<128>(d, ep, M, N, s);
        default: return launch_pw<256>(d, ep, M, N, s);
    }
}

}  // namespace dcap
