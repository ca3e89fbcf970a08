// conv_chain.hip -- two chained pointwise convolutions in ONE launch (round 5): the last convolution of a ResNet bottleneck
// (`2c`: 1x1, Cin -> 4 Cin, + shortcut, ReLU) and the first of the next block (`2a`: 1x1, 4 Cin -> Cin, ReLU)
// (feature_generation/dense_model.py:85-100: identity_block's branch2c / the next block's branch2a, fp32, frozen BN folded).
//
// Why: at the benchmark's two images a stage-4 `2c` + `2a` pair is 2 x 27.3 us of fp32 MFMA work that ran as 44 + 38.5 us -- each
// launch pays its prologue, its epilogue traffic (the 32 MB residual read and the 32 MB output write of `2c` after its MFMAs) and a
// one-round tail, and `2a` re-reads from L2 / HBM what `2c` just wrote (DESIGN section 10 sized it).
// Here a block owns 32 pixels for both layers: it computes their 32 x N1 rows of `2c`, keeps them in LDS (128 KiB for N1 = 1024),
// writes them out once, and contracts them straight away with the `2a` weights -- the intermediate never comes back from memory,
// the residual rows are requested before the first MFMA and are in registers when the epilogue needs them, the output stores drain
// behind the second layer's MFMAs.
//
// Structure: 512 threads = 8 waves (two per SIMD: one wave's vector-memory issue and waits run in the other's MFMA shadow -- the
// first version, four waves with 512 registers each, lost 18 % to exactly that), ONE block per CU, grid = pixels / 32.  Both layers put the output
// CHANNELS on the MFMA's rows (A operand = weights) and the 32 pixels on its columns (B operand = activations from LDS), so a lane
// ends with four consecutive channels of its pixel per accumulator quad (16-byte epilogue accesses, the Winograd kernels' trick).
//   weights: packed once per frozen kernel in FRAGMENT order (dc_pw_chain_pack_f32): f4 index ((cb * K/8 + g) * 64 + lane),
//     lane = 32 h + i, component e  <->  w[cout = 32 cb + i][k = 8 g + 4 h + e]: one wave-load = 1 KiB contiguous, four MFMAs
//     (v_mfma_f32_32x32x2_f32: step e contracts the k pair {8 g + e, 8 g + 4 + e}), loaded through a buffer resource with a
//     block-uniform SGPR offset (no address VALU beside the MFMAs), three (layer 1) / fifteen (layer 2) groups ahead;
//   activations: the 32 x K1 input rows and then the 32 x N1 intermediate rows sit in LDS with row stride K + 4 floats (lane p reads
//     16 bytes at row p: the 16 lanes of a read group hit 16 different bank quads); one ds_read_b128 feeds 4 x CB MFMAs.
// Layer 1: wave w owns the column blocks w * CB1 .. + CB1 - 1 (CB1 = N1 / 256: 4 accumulator tiles for N1 = 1024), layer 2 the
// column block w (N2 / 32 <= 8 blocks).
#include "igemm_bf16s.h"
#include <algorithm>

namespace dcap {
namespace chain {

struct Args {
    const float* x;
    const f4* w1;
    const float* scale1;
    const float* shift1;
    const float* residual;
    float* y;
    const f4* w2;
    const float* scale2;
    const float* shift2;
    float* z;
    int M, K1, relu1, relu2;
    unsigned w1_bytes, w2_bytes;
};

__global__ void chain_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int N, int K) {
    const long total = (long)N * K;
    const int NG = K >> 3;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cout = (int)(idx / K), k = (int)(idx - (long)cout * K);
        const int cb = cout >> 5, i = cout & 31, g = k >> 3, h = (k >> 2) & 1, e = k & 3;
        out[(((long)cb * NG + g) * 64 + (h * 32 + i)) * 4 + e] = w[idx];
    }
}

// Split-bf16 variant of the pack (pw_chain_kernel<.., true>): every weight as three bf16 pieces (x = p0 + p1 + p2, round to nearest even,
// exact remainders) in the fragment order of v_mfma_f32_32x32x16_bf16: ushort index
//   ((((cb * K/16 + g) * 3 + piece) * 64 + lane) * 8 + jj),  lane = 32 h + i  <->  w[cout = 32 cb + i][k = 16 g + 8 h + jj]
__global__ void chain_pack_b3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int N, int K) {
    const long total = (long)N * K;
    const int NG = K >> 4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cout = (int)(idx / K), k = (int)(idx - (long)cout * K);
        const int cb = cout >> 5, i = cout & 31, g = k >> 4, h = (k >> 3) & 1, jj = k & 7;
        float x = w[idx];
#pragma unroll
        for (int piece = 0; piece < 3; ++piece) {
            const unsigned short pb = __builtin_bit_cast(unsigned short, (__bf16)x);
            x -= __uint_as_float((unsigned)pb << 16);
            out[((((long)cb * NG + g) * 3 + piece) * 64 + (h * 32 + i)) * 8 + jj] = pb;
        }
    }
}

__device__ __forceinline__ void split8(const f4& va, const f4& vb, u32x4& p0, u32x4& p1, u32x4& p2) {
    unsigned a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;
    split_pair(va[0], va[1], a0, a1, a2);
    split_pair(va[2], va[3], b0, b1, b2);
    split_pair(vb[0], vb[1], c0, c1, c2);
    split_pair(vb[2], vb[3], d0, d1, d2);
    p0 = u32x4{a0, b0, c0, d0};
    p1 = u32x4{a1, b1, c1, d1};
    p2 = u32x4{a2, b2, c2, d2};
}
__device__ __forceinline__ f32x16 mfma_b(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// six products of the three-piece split, small terms first (igemm_bf16s.h's order): fp32-grade
__device__ __forceinline__ f32x16 mfma_b3(const u32x4 (&w)[3], const u32x4 (&q)[3], f32x16 c) {
    c = mfma_b(w[0], q[2], c);
    c = mfma_b(w[1], q[1], c);
    c = mfma_b(w[2], q[0], c);
    c = mfma_b(w[1], q[0], c);
    c = mfma_b(w[0], q[1], c);
    return mfma_b(w[0], q[0], c);
}

// CB1 = column blocks (of 32 channels) per wave in layer 1 = N1 / 256; NB2 = column blocks of layer 2 = N2 / 32 (<= 8: one per wave)
// B3: both layers' products on the BF16 matrix pipe in split arithmetic (weights pre-split into three bf16 pieces by
// dc_pw_chain_pack_b3, activations split in registers as they leave LDS; six v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32
// accumulation: DC_MATH_BF16X3's arithmetic) -- per 16 channels and column block 6 MFMAs of 32 cycles where the fp32 form issues 8 of 64.
template <int CB1, int NB2, bool B3>
__global__ __launch_bounds__(512, 1) void pw_chain_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = 512, NW = 8;
    constexpr int N1 = CB1 * 32 * NW, N2 = NB2 * 32, LDY = N1 + 4, NG2 = N1 / 8;
    static_assert(NB2 <= NW, "layer 2: one column block per wave");
    constexpr int RQ = N1 / 4;                             // float4 per intermediate row
    constexpr int RPT = 32 * RQ / NT;                      // float4 of the 32 x N1 tile per thread (row-major walk)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32;
    const int K1 = a.K1, LDX = K1 + 4, NG1 = K1 >> 3;

    // ---- the block's 32 input rows into LDS; the residual rows into registers (in flight during layer 1)
    const int xq = K1 >> 2;
    for (int idx = tid; idx < 32 * xq; idx += NT) {
        const int r = idx / xq, c4 = idx - r * xq;
        const int row = min(m0 + r, a.M - 1);
        *reinterpret_cast<f4*>(smem + r * LDX + 4 * c4) = *reinterpret_cast<const f4*>(a.x + (long)row * K1 + 4 * c4);
    }
    f4 resv[RPT];
    auto load_res = [&]() {
        if (a.residual) {
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int idx = j * NT + tid, r = idx / RQ, c4 = idx - r * RQ;
                resv[j] = *reinterpret_cast<const f4*>(a.residual + (long)min(m0 + r, a.M - 1) * N1 + 4 * c4);
            }
        }
    };
    if constexpr (!B3) load_res();                         // (B3: its deeper weight ring needs the registers; the rows are requested behind layer 1's loop)
    const unsigned lane16 = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4*>(a.w1), 0, (int)a.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4*>(a.w2), 0, (int)a.w2_bytes, 0x00020000);

    // ---- layer 1: acc[cb] (channels x pixels) += W1 fragment x input fragment
    f32x16 acc[CB1];
#pragma unroll
    for (int cb = 0; cb < CB1; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    auto ldw1 = [&](f4 (&dst)[CB1], int g) {
#pragma unroll
        for (int cb = 0; cb < CB1; ++cb) dst[cb] = buf_f4s(rsrc1, lane16, (unsigned)(((wave * CB1 + cb) * NG1 + g) * 1024));
    };
    if constexpr (B3) {
        // groups of 16 channels: the lane's eight activations (channels 16 g + 8 h .. + 7 of its pixel) -> three bf16x8 pieces -> for every
        // column block three weight fragments (48 bytes per lane) and six MFMAs.  Weight ring: R groups, requested R - 1 groups ahead.
        const int NGB = K1 >> 4;
        constexpr int R = 3;
        u32x4 wr[R][CB1][3];
        auto ldwb = [&](u32x4 (&dst)[CB1][3], int g) {
#pragma unroll
            for (int cb = 0; cb < CB1; ++cb) {
                const unsigned so = (unsigned)(((wave * CB1 + cb) * NGB + g) * 3072);
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) dst[cb][pc] = __builtin_bit_cast(u32x4, buf_f4s(rsrc1, lane16, so + 1024u * pc));
            }
        };
#pragma unroll
        for (int u = 0; u < R - 1; ++u) ldwb(wr[u], min(u, NGB - 1));
        __syncthreads();                                   // the input rows are in LDS
        const float* xrow = smem + p * LDX + 8 * h;
        f4 xa = *reinterpret_cast<const f4*>(__builtin_assume_aligned(xrow, 16)), xc = *reinterpret_cast<const f4*>(__builtin_assume_aligned(xrow + 4, 16));
        for (int g = 0; g < NGB; g += R) {                 // K1 % 48 == 0 is not required: the tail groups re-run clamped loads but
#pragma unroll
            for (int u = 0; u < R; ++u) {                  // only groups < NGB issue MFMAs (uniform branch)
                if (g + u < NGB) {
                    ldwb(wr[(u + R - 1) % R], min(g + u + R - 1, NGB - 1));
                    const int gn = min(g + u + 1, NGB - 1);
                    const f4 na = *reinterpret_cast<const f4*>(__builtin_assume_aligned(xrow + 16 * gn, 16));
                    const f4 nc = *reinterpret_cast<const f4*>(__builtin_assume_aligned(xrow + 16 * gn + 4, 16));
                    u32x4 q[3];
                    split8(xa, xc, q[0], q[1], q[2]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int cb = 0; cb < CB1; ++cb) acc[cb] = mfma_b3(wr[u][cb], q, acc[cb]);
                    xa = na;
                    xc = nc;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        load_res();
    } else {
    constexpr int R1 = 4;                                  // fragments requested R1 - 1 groups (48 MFMAs of this wave at CB1 = 4) ahead
    f4 wa[R1][CB1];
#pragma unroll
    for (int u = 0; u < R1 - 1; ++u) ldw1(wa[u], min(u, NG1 - 1));
    __syncthreads();                                       // the input rows are in LDS
    const float* xrow = smem + p * LDX + 4 * h;            // (LDX % 4 == 0: 16-byte aligned fragment reads)
    f4 xb = *reinterpret_cast<const f4*>(__builtin_assume_aligned(xrow, 16));
    for (int g = 0; g < NG1; g += R1) {                    // K1 % 32 == 0
#pragma unroll
        for (int u = 0; u < R1; ++u) {
            // (the fences pin the order the loop is written in: the compiler otherwise gathers a whole iteration's loads into one burst
            // right in front of their first use, and the wave then sits out an L2 round trip per iteration)
            ldw1(wa[(u + R1 - 1) % R1], min(g + u + R1 - 1, NG1 - 1));       // the last ones re-read: uniform counts
            const f4 xn = *reinterpret_cast<const f4*>(__builtin_assume_aligned(xrow + 8 * min(g + u + 1, NG1 - 1), 16));   // next group's activations
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int cb = 0; cb < CB1; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[u][cb][e], xb[e], acc[cb], 0, 0, 0);
            xb = xn;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    }
    // ---- epilogue 1: scale / shift into the LDS image of the intermediate rows (it takes the input rows' place), then one row-major
    // pass adds the residual, applies the ReLU, rewrites the image and stores the rows (1 KiB contiguous per wave-instruction)
    __syncthreads();                                       // every wave is done with the input rows
#pragma unroll
    for (int cb = 0; cb < CB1; ++cb)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int c0 = (wave * CB1 + cb) * 32 + 8 * gq + 4 * h;
            f4 v = {acc[cb][4 * gq], acc[cb][4 * gq + 1], acc[cb][4 * gq + 2], acc[cb][4 * gq + 3]};
            if (a.scale1) v = v * *reinterpret_cast<const f4*>(a.scale1 + c0);
            v = v + *reinterpret_cast<const f4*>(a.shift1 + c0);
            *reinterpret_cast<f4*>(smem + p * LDY + c0) = v;
        }
    // layer 2's first weight fragments: requested before the pass below, so that they are there when it ends
    const bool active2 = wave < NB2;                       // (N2 = 128: four column blocks, waves 4..7 sit layer 2 out)
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    auto finish_pass = [&]() {                             // residual + ReLU over the LDS image, the intermediate rows out to memory
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int idx = j * NT + tid, r = idx / RQ, c4 = idx - r * RQ;
            f4 v = *reinterpret_cast<const f4*>(smem + r * LDY + 4 * c4);
            if (a.residual) v = v + resv[j];
            if (a.relu1) v = f4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            *reinterpret_cast<f4*>(smem + r * LDY + 4 * c4) = v;
            if (m0 + r < a.M) *reinterpret_cast<f4*>(a.y + (long)(m0 + r) * N1 + 4 * c4) = v;
        }
        __syncthreads();                                   // the intermediate rows are final in LDS
    };
    if constexpr (B3) {
        constexpr int NGB2 = N1 / 16, R = 8;               // seven groups (42 MFMAs of this wave + their splits) ahead
        u32x4 wr[R][3];
        auto ldwb = [&](u32x4 (&dst)[3], int g) {
            const unsigned so = (unsigned)((wave * NGB2 + g) * 3072);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) dst[pc] = __builtin_bit_cast(u32x4, buf_f4s(rsrc2, lane16, so + 1024u * pc));
        };
        if (active2) {
#pragma unroll
            for (int u = 0; u < R - 1; ++u) ldwb(wr[u], u);
        }
        finish_pass();
        if (!active2) return;
        const float* yrow = smem + p * LDY + 8 * h;
        f4 ya = *reinterpret_cast<const f4*>(__builtin_assume_aligned(yrow, 16)), yc = *reinterpret_cast<const f4*>(__builtin_assume_aligned(yrow + 4, 16));
        for (int g = 0; g < NGB2; g += R) {                // NGB2 = N1 / 16 is a multiple of 8
#pragma unroll
            for (int u = 0; u < R; ++u) {
                ldwb(wr[(u + R - 1) % R], min(g + u + R - 1, NGB2 - 1));
                const int gn = min(g + u + 1, NGB2 - 1);
                const f4 na = *reinterpret_cast<const f4*>(__builtin_assume_aligned(yrow + 16 * gn, 16));
                const f4 nc = *reinterpret_cast<const f4*>(__builtin_assume_aligned(yrow + 16 * gn + 4, 16));
                u32x4 q[3];
                split8(ya, yc, q[0], q[1], q[2]);
                __builtin_amdgcn_sched_barrier(0);
                acc2 = mfma_b3(wr[u], q, acc2);
                ya = na;
                yc = nc;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
    constexpr int R2 = 16;                                 // R2 - 1 groups (60 MFMAs of this wave) ahead
    f4 wb[R2];
    auto ldw2 = [&](f4& dst, int g) { dst = buf_f4s(rsrc2, lane16, (unsigned)((wave * NG2 + g) * 1024)); };
    if (active2) {
#pragma unroll
        for (int u = 0; u < R2 - 1; ++u) ldw2(wb[u], u);
    }
    finish_pass();
    if (!active2) return;
    // ---- layer 2: acc2 += W2 fragment x intermediate fragment (from LDS)
    const float* yrow = smem + p * LDY + 4 * h;
    f4 yb = *reinterpret_cast<const f4*>(__builtin_assume_aligned(yrow, 16));
    for (int g = 0; g < NG2; g += R2) {                    // NG2 = N1 / 8 is a multiple of 16
#pragma unroll
        for (int u = 0; u < R2; ++u) {
            ldw2(wb[(u + R2 - 1) % R2], min(g + u + R2 - 1, NG2 - 1));
            const f4 yn = *reinterpret_cast<const f4*>(__builtin_assume_aligned(yrow + 8 * min(g + u + 1, NG2 - 1), 16));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[u][e], yb[e], acc2, 0, 0, 0);
            yb = yn;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    }
    if (m0 + p < a.M) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int c0 = wave * 32 + 8 * gq + 4 * h;
            f4 v = {acc2[4 * gq], acc2[4 * gq + 1], acc2[4 * gq + 2], acc2[4 * gq + 3]};
            if (a.scale2) v = v * *reinterpret_cast<const f4*>(a.scale2 + c0);
            v = v + *reinterpret_cast<const f4*>(a.shift2 + c0);
            if (a.relu2) v = f4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            *reinterpret_cast<f4*>(a.z + (long)(m0 + p) * N2 + c0) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The stage-2 seam (64 -> 256 -> 64 channels at 256 x 256 pixels per image): HBM-bound -- 470 MB as two launches, 335 MB chained (the
// 256-channel intermediate is written once and never read back) -- and both weight sets fit LDS (64 + 64 KiB), so it runs in the streaming
// form of conv_pw.hip rather than the tile form above: no K loop, no barrier after start-up, every wave walks over 32-pixel strips on its
// own, activations go from global memory straight into the registers that are the MFMA B operands, channels sit on the MFMA's M side.
// The chaining costs NO data movement at all: layer 1's accumulators (after scale / shift / shortcut / ReLU in registers) ARE layer 2's
// B operands -- lane (pixel i, half h) holds channels 8 q + 4 h + r of its pixel, which v_mfma_f32_32x32x2_f32 reads as the k pair
// {8 q + r, 8 q + 4 + r} of that pixel when the weight side is read in the same order (one ds_read_b128 of W2 per four MFMAs).
// 512 threads, one block per CU, fp32 MFMA products (the layer pair is bandwidth-bound: 8.6 GF against 335 MB).
// ------------------------------------------------------------------------------------------------
template <int K1, int N1, int N2>
__global__ __launch_bounds__(512, 1) void pw_chain_stream_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDW1 = K1 + 4, LDW2 = N1 + 4, KH = K1 / 2, NG1 = K1 / 8, NG2 = N1 / 8;
    float* const W1s = smem;                               // [N1][LDW1]
    float* const W2s = W1s + N1 * LDW1;                    // [N2][LDW2]
    float* const sc1 = W2s + N2 * LDW2;
    float* const sh1 = sc1 + N1;
    float* const sc2 = sh1 + N1;
    float* const sh2 = sc2 + N2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    // weights out of the fragment order of dc_pw_chain_pack_f32 (f4 ((cb * K/8 + g) * 64 + 32 h + i) = w[32 cb + i][8 g + 4 h .. + 3]) into rows
    for (int idx = tid; idx < N1 * (K1 / 4); idx += 512) {
        const int c = idx / (K1 / 4), k4 = idx - c * (K1 / 4);
        *reinterpret_cast<f4*>(&W1s[c * LDW1 + 4 * k4]) = a.w1[((c >> 5) * NG1 + (k4 >> 1)) * 64 + 32 * (k4 & 1) + (c & 31)];
    }
    for (int idx = tid; idx < N2 * (N1 / 4); idx += 512) {
        const int c = idx / (N1 / 4), k4 = idx - c * (N1 / 4);
        *reinterpret_cast<f4*>(&W2s[c * LDW2 + 4 * k4]) = a.w2[((c >> 5) * NG2 + (k4 >> 1)) * 64 + 32 * (k4 & 1) + (c & 31)];
    }
    for (int c = tid; c < N1; c += 512) {
        sc1[c] = a.scale1 ? a.scale1[c] : 1.f;
        sh1[c] = a.shift1[c];
    }
    for (int c = tid; c < N2; c += 512) {
        sc2[c] = a.scale2 ? a.scale2[c] : 1.f;
        sh2[c] = a.shift2[c];
    }
    __syncthreads();
    const int strips = (a.M + 31) / 32;
    for (int st = blockIdx.x * 8 + wave; st < strips; st += gridDim.x * 8) {      // (requesting the next strip's activations a strip ahead measured slower: 123 vs 117 us)
        const int prow = st * 32 + i;
        const bool pv = prow < a.M;
        const int p = min(prow, a.M - 1);
        const float* xr = a.x + (long)p * K1 + KH * h;
        f4 xs[KH / 4];
#pragma unroll
        for (int j = 0; j < KH / 4; ++j) xs[j] = *reinterpret_cast<const f4*>(xr + 4 * j);
        const float* rrow = a.residual ? a.residual + (long)p * N1 + 4 * h : nullptr;
        float* const yrow = a.y + (long)p * N1 + 4 * h;
        f32x16 acc2[N2 / 32];
#pragma unroll
        for (int nb = 0; nb < N2 / 32; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[nb][r] = 0.f;
        for (int cb = 0; cb < N1; cb += 64) {
            f4 r0[4], r1[4];
            if (rrow) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    r0[q] = *reinterpret_cast<const f4*>(rrow + cb + 8 * q);
                    r1[q] = *reinterpret_cast<const f4*>(rrow + cb + 32 + 8 * q);
                }
            }
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            const float* wa0 = &W1s[(cb + i) * LDW1 + KH * h];
            const float* wa1 = wa0 + 32 * LDW1;
#pragma unroll
            for (int j = 0; j < KH / 4; ++j) {
                const f4 a0 = *reinterpret_cast<const f4*>(wa0 + 4 * j);
                const f4 a1 = *reinterpret_cast<const f4*>(wa1 + 4 * j);
                const f4 b = xs[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b[e], acc1, 0, 0, 0);
                }
            }
            // layer 1's epilogue in registers: quad q of lane (i, h) = channels cb (+ 32) + 8 q + 4 h .. + 3 of pixel i
            f4 y0[4], y1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = cb + 8 * q + 4 * h;
                f4 v0 = f4{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]} * *reinterpret_cast<const f4*>(&sc1[c0]) + *reinterpret_cast<const f4*>(&sh1[c0]);
                f4 v1 = f4{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]} * *reinterpret_cast<const f4*>(&sc1[c0 + 32]) +
                        *reinterpret_cast<const f4*>(&sh1[c0 + 32]);
                if (rrow) { v0 += r0[q]; v1 += r1[q]; }
                if (a.relu1) {
                    v0 = f4{fmaxf(v0[0], 0.f), fmaxf(v0[1], 0.f), fmaxf(v0[2], 0.f), fmaxf(v0[3], 0.f)};
                    v1 = f4{fmaxf(v1[0], 0.f), fmaxf(v1[1], 0.f), fmaxf(v1[2], 0.f), fmaxf(v1[3], 0.f)};
                }
                if (pv) {
                    *reinterpret_cast<f4*>(yrow + cb + 8 * q) = v0;
                    *reinterpret_cast<f4*>(yrow + cb + 32 + 8 * q) = v1;
                }
                y0[q] = v0;
                y1[q] = v1;
            }
            // layer 2: those registers as the B operand (k pair {8 q + r, 8 q + 4 + r} of the lane's pixel); A = W2 rows in the same order
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nb = 0; nb < N2 / 32; ++nb) {
                    const float* w2r = &W2s[(nb * 32 + i) * LDW2 + cb + 8 * q + 4 * h];
                    const f4 wa = *reinterpret_cast<const f4*>(w2r);
                    const f4 wb = *reinterpret_cast<const f4*>(w2r + 32);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[e], y0[q][e], acc2[nb], 0, 0, 0);
                        acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[e], y1[q][e], acc2[nb], 0, 0, 0);
                    }
                }
        }
        if (pv) {
            float* const zrow = a.z + (long)p * N2 + 4 * h;
#pragma unroll
            for (int nb = 0; nb < N2 / 32; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = nb * 32 + 8 * q + 4 * h;
                    f4 v = f4{acc2[nb][4 * q], acc2[nb][4 * q + 1], acc2[nb][4 * q + 2], acc2[nb][4 * q + 3]} * *reinterpret_cast<const f4*>(&sc2[c0]) +
                           *reinterpret_cast<const f4*>(&sh2[c0]);
                    if (a.relu2) v = f4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                    *reinterpret_cast<f4*>(zrow + nb * 32 + 8 * q) = v;
                }
        }
    }
}

static int launch_stream_64_256_64(const Args& a, hipStream_t s) {
    constexpr int K1 = 64, N1 = 256, N2 = 64;
    constexpr size_t lds = ((size_t)N1 * (K1 + 4) + (size_t)N2 * (N1 + 4) + 2 * N1 + 2 * N2) * sizeof(float);
    DC_ENSURE_DYN_LDS((&pw_chain_stream_kernel<K1, N1, N2>), 160 * 1024);
    const int strips = (a.M + 31) / 32;
    const int grid = std::max(1, std::min(kNumCU, (strips + 7) / 8));
    hipLaunchKernelGGL((pw_chain_stream_kernel<K1, N1, N2>), dim3(grid), dim3(512), lds, s, a);
    return check_launch("pw_chain_stream_kernel");
}

template <int CB1, int NB2, bool B3>
static int launch(const Args& a, hipStream_t s) {
    constexpr size_t lds = (size_t)32 * (CB1 * 256 + 4) * sizeof(float);
    DC_ENSURE_DYN_LDS((&pw_chain_kernel<CB1, NB2, B3>), 160 * 1024);
    hipLaunchKernelGGL((pw_chain_kernel<CB1, NB2, B3>), dim3((a.M + 31) / 32), dim3(512), lds, s, a);
    return check_launch("pw_chain_kernel");
}

}  // namespace chain
}  // namespace dcap

using namespace dcap;

extern "C" int dc_pw_chain_supported(int K1, int N1, int N2) {
    if (K1 == 64 && N1 == 256 && N2 == 64) return 1;      // the stage-2 seam: streaming form (fp32 MFMA products only)
    return (K1 >= 32 && K1 % 32 == 0 && K1 <= N1 && ((N1 == 1024 && N2 == 256) || (N1 == 512 && N2 == 128))) ? 1 : 0;
}

extern "C" int dc_pw_chain_pack_f32(const float* w, float* out, int N, int K, void* stream) {
    DC_REQUIRE(w && out && N > 0 && K > 0 && N % 32 == 0 && K % 8 == 0, DC_EINVAL, "dc_pw_chain_pack: N %% 32 == 0 and K %% 8 == 0");
    DC_REQUIRE(aligned16(out), DC_EALIGN, "dc_pw_chain_pack: out must be 16-byte aligned");
    const long total = (long)N * K;
    hipLaunchKernelGGL(chain::chain_pack_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, static_cast<hipStream_t>(stream), w, out, N, K);
    return check_launch("dc_pw_chain_pack_f32");
}

extern "C" int dc_pw_chain_pack_b3(const float* w, uint16_t* out, int N, int K, void* stream) {
    DC_REQUIRE(w && out && N > 0 && K > 0 && N % 32 == 0 && K % 16 == 0, DC_EINVAL, "dc_pw_chain_pack_b3: N %% 32 == 0 and K %% 16 == 0");
    DC_REQUIRE(aligned16(out), DC_EALIGN, "dc_pw_chain_pack_b3: out must be 16-byte aligned");
    const long total = (long)N * K;
    hipLaunchKernelGGL(chain::chain_pack_b3_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, static_cast<hipStream_t>(stream), w, out, N, K);
    return check_launch("dc_pw_chain_pack_b3");
}

extern "C" int dc_pw_chain_kernel_name(const dc_pw_chain_desc* d, char* buf, size_t buf_bytes) {
    DC_REQUIRE(d && buf && buf_bytes >= 40, DC_EINVAL, "dc_pw_chain_kernel_name: bad arguments");
    if (d->N1 == 256) snprintf(buf, buf_bytes, "pw_chain_stream_kernel<%d, %d, %d>", d->K1, d->N1, d->N2);
    else snprintf(buf, buf_bytes, "pw_chain_kernel<%d, %d, %s>", d->N1 / 256, d->N2 / 32, (d->w1_b3 && d->w2_b3) ? "true" : "false");
    return DC_OK;
}

extern "C" int dc_pw_chain_f32(const dc_pw_chain_desc* d, void* stream) {
    const bool b3 = d && d->w1_b3 && d->w2_b3;
    DC_REQUIRE(d && d->x && (b3 || (d->w1 && d->w2)) && d->shift1 && d->y && d->shift2 && d->z && d->M > 0, DC_EINVAL, "dc_pw_chain: bad arguments");
    DC_REQUIRE(dc_pw_chain_supported(d->K1, d->N1, d->N2), DC_EINVAL,
               "dc_pw_chain: (K1, N1, N2) = (%d, %d, %d) is not a covered shape (K1 %% 32 == 0; N1 -> N2 = 1024 -> 256 or 512 -> 128; or 64 -> 256 -> 64)", d->K1, d->N1, d->N2);
    DC_REQUIRE(d->N1 != 256 || !b3, DC_EINVAL, "dc_pw_chain: the 64 -> 256 -> 64 seam takes fp32 products (dc_pw_chain_pack_f32 weights)");
    DC_REQUIRE(aligned16(d->x) && aligned16(b3 ? (const void*)d->w1_b3 : (const void*)d->w1) && aligned16(b3 ? (const void*)d->w2_b3 : (const void*)d->w2) && aligned16(d->y) && aligned16(d->z) && aligned16(d->shift1) && aligned16(d->shift2) &&
                   (!d->scale1 || aligned16(d->scale1)) && (!d->scale2 || aligned16(d->scale2)) && (!d->residual || aligned16(d->residual)),
               DC_EALIGN, "dc_pw_chain: every pointer must be 16-byte aligned");
    chain::Args a;
    a.x = d->x; a.w1 = b3 ? reinterpret_cast<const f4*>(d->w1_b3) : reinterpret_cast<const f4*>(d->w1); a.scale1 = d->scale1; a.shift1 = d->shift1; a.residual = d->residual; a.y = d->y;
    a.w2 = b3 ? reinterpret_cast<const f4*>(d->w2_b3) : reinterpret_cast<const f4*>(d->w2); a.scale2 = d->scale2; a.shift2 = d->shift2; a.z = d->z;
    a.M = d->M; a.K1 = d->K1; a.relu1 = d->relu1; a.relu2 = d->relu2;
    a.w1_bytes = (unsigned)((size_t)d->N1 * d->K1 * (b3 ? 6 : 4));
    a.w2_bytes = (unsigned)((size_t)d->N2 * d->N1 * (b3 ? 6 : 4));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (d->N1 == 256) return chain::launch_stream_64_256_64(a, s);
    if (b3) return d->N1 == 1024 ? chain::launch<4, 8, true>(a, s) : chain::launch<2, 4, true>(a, s);
    return d->N1 == 1024 ? chain::launch<4, 8, false>(a, s) : chain::launch<2, 4, false>(a, s);
}
