// vocab_ce.hip -- dc_vocab_ce: the vocabulary projection FUSED with softmax + Keras cross-entropy (forward and d/dlogits).
//
// Replaces  Dense(V, activation='softmax') + K.categorical_crossentropy / roi_caption_loss
//   (dense_img_cap_separate_models/text_generation_model.py:153-154, :286-294; text_generation_model_v2.py:164, :267)
// and the joint model's masked K.sparse_categorical_crossentropy (dense_img_cap/dense_model.py:936-946).
//
// The [rows, V] logits are never written: 600 MB of fp32 at 3000 caption rows x 50 000 words, read three times by an
// unfused softmax.  Instead the GEMM main loop (fp32 MFMA: igemm_core.h; bf16: bgemm_core.h) runs once per PASS over
// the 128 x 128 output tiles and each pass ends in a reduction epilogue on the tile while it sits in LDS:
//   pass STATS  per row and column tile: max and sum of exp (online softmax partials), and the target's logit;
//               a row kernel then combines the tiles: m, s, p_t, the loss row (categorical flavour) and the row's
//               gradient scale;
//   pass CLIP   (keras_sparse only) K.sparse_categorical_crossentropy clips the probabilities to [1e-7, 1-1e-7] and
//               renormalises: S = sum clip(p), UP = sum of the unclipped p, per row and tile; a row kernel combines them
//               into the loss row and the two row constants of the gradient.  LAZY since round 5: the STATS pass also keeps the
//               smallest logit of every row, so the row kernel knows which rows have a probability outside the clip range at all
//               (p is monotone in z: the smallest and the largest p of a row decide); a row tile without such a row leaves the
//               CLIP pass before its GEMM (S = UP = 1 there: nothing is clipped, the row's probabilities sum to s / s) -- with
//               50 000 words and ordinary logits that is every tile, and the step runs two GEMM passes instead of three;
//   pass DL     recomputes the tile and writes d(loss)/d(logits) (fp32, or bf16 as the operand of the bf16 weight /
//               data gradient GEMMs) plus per-tile column sums (the bias gradient, combined in a fixed order).
//   MATERIALISED bf16 logits (round 6; dc_vocab_ce_desc.materialize_bf16, the joint model's bf16 arithmetic on the 256-square tile):
//               pass STATSZ is pass STATS on logits ROUNDED to bf16, which it also parks in the gradient's own [M][lddl] bf16
//               buffer; the clip sums (rows that need them only) and the gradient are then ELEMENTWISE passes over that buffer, the
//               gradient in place: one GEMM pass (307 GFLOP at 3000 x 50 000 x 1024) + 600 MB of streaming instead of two GEMM
//               passes, the second of which spent a third of its time storing the same 300 MB from its epilogue.  Loss and
//               gradient are those of the rounded logits (consistent with each other: every row of the gradient still sums to 0).
// The recomputation costs one extra GEMM pass (two for keras_sparse); in exchange nothing of size rows x V is written
// but the gradient itself.  All reductions are wavefront shuffles over the 32 lanes that share a row of the tile.
#include "bgemm256_core.h"
#include <algorithm>

namespace dcap {

enum { CE_STATS = 0, CE_CLIP = 1, CE_DL = 2, CE_STATSZ = 3 };
constexpr int CE_RI = 8;                 // floats of row info per row
constexpr int CE_ST = 4;                 // floats of per-(row, column tile) partials: STATS (max, sum exp, min, -), CLIP (S, UP, -, -)

struct CeArgs {
    int M, V, tiles_n;
    const float* bias;                   // [V] or null
    const int32_t* targets;              // [M]
    float* stats;                        // [M][tiles_n][CE_ST]
    int* needs_clip;                     // [M]: the row has a probability outside [1e-7, 1 - 1e-7] (written by ce_rows_kernel, phase 0)
    int tile_rows;                       // rows per tile of the launched kernel (128 / 256)
    float* zt;                           // [M]
    const float* rowinfo;                // [M][CE_RI]: m, 1/s, gs, 1/S, c, tq  (see ce_rows_kernel)
    float* dl_f32;
    unsigned short* dl_bf16;
    long lddl;
    float* dbias_part;                   // [tiles_m][V] or null
    int keras_sparse;
};

__device__ __forceinline__ float half_max(float v) {          // over the 32 lanes that share a tile row
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float half_min(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
// CLIP pass: does any row of this block's row tile need it?  Every wave answers for itself from the same flags (a ballot: no LDS, no
// barrier -- __syncthreads_or would add a static LDS word to kernels that ask for the full 160 KiB dynamically), so the answer is
// block-uniform; called before the GEMM main loop.
__device__ __forceinline__ bool ce_tile_needs_clip(const CeArgs& ce, int m0) {
    int any = 0;
    for (int r = threadIdx.x & 63; r < ce.tile_rows; r += 64)
        if (m0 + r < ce.M) any |= ce.needs_clip[m0 + r];
    return __any(any) != 0;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Epilogue of one 128 x 128 tile (accumulators in MFMA layout, wave origin wm, wn).
template <int MODE>
__device__ __forceinline__ void ce_epilogue(f32x16 (&acc)[2][2], float* Cs, const CeArgs& ce, int m0, int n0, int wm, int wn, int tile_m, int tile_n) {
    constexpr int LDC = 128 + 4;
    const int tid = threadIdx.x, lane = tid & 63;
    {
        const int i = lane & 31, h = lane >> 5;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) Cs[(wm + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDC + wn + tn * 32 + i] = acc[tm][tn][r];
    }
    __syncthreads();
    const int c4 = tid & 31, rp = tid >> 5;                    // 32 lanes x 4 columns per row, 8 rows per pass
    const int col = n0 + 4 * c4;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ce.bias && col + 3 < ce.V) b4 = *reinterpret_cast<const float4*>(ce.bias + col);
    else if (ce.bias) {
        if (col < ce.V) b4.x = ce.bias[col];
        if (col + 1 < ce.V) b4.y = ce.bias[col + 1];
        if (col + 2 < ce.V) b4.z = ce.bias[col + 2];
    }
    const bool v0 = col < ce.V, v1 = col + 1 < ce.V, v2 = col + 2 < ce.V, v3 = col + 3 < ce.V;
    float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f;          // DL: column sums over this thread's rows
#pragma unroll 2
    for (int p = 0; p < 16; ++p) {
        const int lr = p * 8 + rp, row = m0 + lr;
        const bool rv = row < ce.M;                            // uniform over the 32 lanes of the row
        float4 z = *reinterpret_cast<const float4*>(&Cs[lr * LDC + 4 * c4]);
        z.x += b4.x; z.y += b4.y; z.z += b4.z; z.w += b4.w;
        const int t = rv ? ce.targets[row] : -1;
        if constexpr (MODE == CE_STATS) {
            float mx = fmaxf(fmaxf(v0 ? z.x : -INFINITY, v1 ? z.y : -INFINITY), fmaxf(v2 ? z.z : -INFINITY, v3 ? z.w : -INFINITY));
            mx = half_max(mx);
            float s = (v0 ? __expf(z.x - mx) : 0.f) + (v1 ? __expf(z.y - mx) : 0.f) + (v2 ? __expf(z.z - mx) : 0.f) + (v3 ? __expf(z.w - mx) : 0.f);
            s = half_sum(s);
            float mn = fminf(fminf(v0 ? z.x : INFINITY, v1 ? z.y : INFINITY), fminf(v2 ? z.z : INFINITY, v3 ? z.w : INFINITY));
            mn = half_min(mn);
            if (rv) {
                if (c4 == 0) *reinterpret_cast<float4*>(ce.stats + ((long)row * ce.tiles_n + tile_n) * CE_ST) = make_float4(mx, s, mn, 0.f);
                if (t >= col && t < col + 4 && t < ce.V) ce.zt[row] = t == col ? z.x : (t == col + 1 ? z.y : (t == col + 2 ? z.z : z.w));
            }
        } else {
            const float* ri = ce.rowinfo + (long)min(row, ce.M - 1) * CE_RI;
            const float m = ri[0], inv_s = ri[1];
            const float p0 = __expf(z.x - m) * inv_s, p1 = __expf(z.y - m) * inv_s, p2 = __expf(z.z - m) * inv_s, p3 = __expf(z.w - m) * inv_s;
            auto unclipped = [](float q) { return q >= 1e-7f && q <= 1.f - 1e-7f; };
            auto clip = [](float q) { return fminf(fmaxf(q, 1e-7f), 1.f - 1e-7f); };
            if constexpr (MODE == CE_CLIP) {
                float S = (v0 ? clip(p0) : 0.f) + (v1 ? clip(p1) : 0.f) + (v2 ? clip(p2) : 0.f) + (v3 ? clip(p3) : 0.f);
                float U = ((v0 && unclipped(p0)) ? p0 : 0.f) + ((v1 && unclipped(p1)) ? p1 : 0.f) + ((v2 && unclipped(p2)) ? p2 : 0.f) +
                          ((v3 && unclipped(p3)) ? p3 : 0.f);
                S = half_sum(S);
                U = half_sum(U);
                if (rv && c4 == 0) *reinterpret_cast<float2*>(ce.stats + ((long)row * ce.tiles_n + tile_n) * CE_ST) = make_float2(S, U);
            } else {
                const float gs = ri[2];
                float g0, g1, g2, g3;
                if (ce.keras_sparse) {
                    const float invS = ri[3], c = ri[4], tq = ri[5];
                    g0 = gs * p0 * ((unclipped(p0) ? invS : 0.f) - (t == col ? tq : 0.f) - c);
                    g1 = gs * p1 * ((unclipped(p1) ? invS : 0.f) - (t == col + 1 ? tq : 0.f) - c);
                    g2 = gs * p2 * ((unclipped(p2) ? invS : 0.f) - (t == col + 2 ? tq : 0.f) - c);
                    g3 = gs * p3 * ((unclipped(p3) ? invS : 0.f) - (t == col + 3 ? tq : 0.f) - c);
                } else {
                    g0 = gs * (p0 - (t == col ? 1.f : 0.f));
                    g1 = gs * (p1 - (t == col + 1 ? 1.f : 0.f));
                    g2 = gs * (p2 - (t == col + 2 ? 1.f : 0.f));
                    g3 = gs * (p3 - (t == col + 3 ? 1.f : 0.f));
                }
                if (rv) {
                    cs0 += g0; cs1 += g1; cs2 += g2; cs3 += g3;
                    if (ce.dl_f32) {
                        float* o = ce.dl_f32 + (long)row * ce.lddl + col;
                        if (v3) *reinterpret_cast<float4*>(o) = make_float4(g0, g1, g2, g3);
                        else { if (v0) o[0] = g0; if (v1) o[1] = g1; if (v2) o[2] = g2; }
                    } else {
                        typedef unsigned short us4 __attribute__((ext_vector_type(4)));
                        unsigned short* o = ce.dl_bf16 + (long)row * ce.lddl + col;
                        // columns V..lddl-1 of the gradient are zero-filled: they are the K padding of the bf16 GEMMs that read it
                        const us4 q{Epilogue::bf16_bits(v0 ? g0 : 0.f), Epilogue::bf16_bits(v1 ? g1 : 0.f), Epilogue::bf16_bits(v2 ? g2 : 0.f),
                                    Epilogue::bf16_bits(v3 ? g3 : 0.f)};
                        if (col + 3 < ce.lddl) *reinterpret_cast<us4*>(o) = q;
                        else { if (col < ce.lddl) o[0] = q[0]; if (col + 1 < ce.lddl) o[1] = q[1]; if (col + 2 < ce.lddl) o[2] = q[2]; }
                    }
                }
            }
        }
    }
    if constexpr (MODE == CE_DL) {
        if (ce.dbias_part) {                                   // column sums of the tile: 8 row lanes per column, fixed order
            __syncthreads();                                   // every thread is done reading Cs
            float* red = Cs;                                   // [8][128]
            *reinterpret_cast<float4*>(&red[rp * 128 + 4 * c4]) = make_float4(cs0, cs1, cs2, cs3);
            __syncthreads();
            if (tid < 128 && n0 + tid < ce.V) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) s += red[q * 128 + tid];
                ce.dbias_part[(long)tile_m * ce.V + n0 + tid] = s;
            }
        }
    }
}

using CeA32 = DenseKCT<true>;
using CeB32 = DenseMCT<true>;

template <int MODE>
__global__ __launch_bounds__(256, 2) void vocab_ce_f32_kernel(CeA32 al, CeB32 bl, CeArgs ce, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = lid / ce.tiles_n, tile_n = lid % ce.tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    if constexpr (MODE == CE_CLIP) {
        if (!ce_tile_needs_clip(ce, m0)) return;
    }
    f32x16 acc[2][2];
    igemm_mainloop<128, 128, CeA32, CeB32>(al, bl, smem, m0, n0, 0, K, acc, wm, wn);
    ce_epilogue<MODE>(acc, smem, ce, m0, n0, wm, wn, tile_m, tile_n);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void vocab_ce_bf16_kernel(BOperand a, BOperand b, CeArgs ce, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = lid / ce.tiles_n, tile_n = lid % ce.tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    if constexpr (MODE == CE_CLIP) {
        if (!ce_tile_needs_clip(ce, m0)) return;
    }
    f32x16 acc[2][2];
    bgemm_mainloop<true, false>(a, b, reinterpret_cast<char*>(smem), m0, n0, 0, K, acc, wm, wn);
    ce_epilogue<MODE>(acc, smem, ce, m0, n0, wm, wn, tile_m, tile_n);
}

// ------------------------------------------------------------------------------------------------
// The same three passes on the 256 x 256 bf16 tile (bgemm256_core.h).  The accumulators are never transposed through LDS:
// lane l of wave (group g, column c) holds, for mt = 0..7 and nt = 0..3, the four consecutive logits
//   row  m0 + 128 g + 64 (mt >> 2) + 16 (mt & 3) + (l & 15),   columns  n0 + 64 c + 32 (nt >> 1) + 16 (nt & 1) + 4 (l >> 4) .. + 3,
// so a row reduction over the tile's 256 columns is 16 in-lane terms, two shuffles (lanes l ^ 16, l ^ 32 hold the other column
// quads of the same row) and a 4-entry combine across the wave columns through LDS; column sums (the bias gradient) are 8 in-lane
// rows, four shuffles over the 16 row lanes and a 2-entry combine across the wave groups -- all in a fixed order.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float quad_max(float v) { return fmaxf(fmaxf(v, __shfl_xor(v, 16, 64)), fmaxf(__shfl_xor(v, 32, 64), __shfl_xor(v, 48, 64))); }
__device__ __forceinline__ float quad_min(float v) { return fminf(fminf(v, __shfl_xor(v, 16, 64)), fminf(__shfl_xor(v, 32, 64), __shfl_xor(v, 48, 64))); }
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

template <int MODE>
__device__ __forceinline__ void ce_epilogue256(b256::f32x4 (&acc)[8][4], float* lds, const CeArgs& ce, int m0, int n0, int tile_m, int tile_n) {
    using b256::f32x4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int group = wave >> 2, wcol = wave & 3, i = lane & 15, q = lane >> 4;
    const int cbase = n0 + 64 * wcol + 4 * q;
    f32x4 bias[4];
    bool cv[4];
    int col[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        col[nt] = cbase + 32 * (nt >> 1) + 16 * (nt & 1);
        cv[nt] = col[nt] < ce.V;                                  // V % 4 == 0: a lane's four columns are all inside or all outside
        bias[nt] = (ce.bias && cv[nt]) ? *reinterpret_cast<const f32x4*>(ce.bias + col[nt]) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float4* part = reinterpret_cast<float4*>(lds);               // [4 wave columns][256 tile rows]
    f32x4 cs[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) cs[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int lr = 128 * group + 64 * (mt >> 2) + 16 * (mt & 3) + i, row = m0 + lr;
        const bool rv = row < ce.M;
        const int t = rv ? ce.targets[row] : -1;
        f32x4 z[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) z[nt] = acc[mt][nt] + bias[nt];
        if constexpr (MODE == CE_STATSZ) {                     // the logits this call works with are the bf16-rounded ones: park them
            typedef unsigned short us4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const us4 zb{Epilogue::bf16_bits(z[nt][0]), Epilogue::bf16_bits(z[nt][1]), Epilogue::bf16_bits(z[nt][2]), Epilogue::bf16_bits(z[nt][3])};
#pragma unroll
                for (int j = 0; j < 4; ++j) z[nt][j] = __builtin_bit_cast(float, (unsigned)zb[j] << 16);
                if (rv && cv[nt]) *reinterpret_cast<us4*>(ce.dl_bf16 + (long)row * ce.lddl + col[nt]) = zb;
            }
        }
        if constexpr (MODE == CE_STATS || MODE == CE_STATSZ) {
            float mx = -INFINITY;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                if (cv[nt]) mx = fmaxf(mx, fmaxf(fmaxf(z[nt][0], z[nt][1]), fmaxf(z[nt][2], z[nt][3])));
            mx = quad_max(mx);
            float sm = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                if (cv[nt]) sm += __expf(z[nt][0] - mx) + __expf(z[nt][1] - mx) + __expf(z[nt][2] - mx) + __expf(z[nt][3] - mx);
            sm = quad_sum(sm);
            float mn = INFINITY;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                if (cv[nt]) mn = fminf(mn, fminf(fminf(z[nt][0], z[nt][1]), fminf(z[nt][2], z[nt][3])));
            mn = quad_min(mn);
            if (q == 0) part[wcol * 256 + lr] = make_float4(mx, sm, mn, 0.f);       // mx = -inf, sm = 0, mn = +inf when the wave's 64 columns lie past V
            if (rv && t < ce.V)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int d = t - col[nt];
                    if (d >= 0 && d < 4) ce.zt[row] = d == 0 ? z[nt][0] : (d == 1 ? z[nt][1] : (d == 2 ? z[nt][2] : z[nt][3]));
                }
        } else {
            const float* ri = ce.rowinfo + (long)min(row, ce.M - 1) * CE_RI;
            const float m = ri[0], inv_s = ri[1];
            auto unclipped = [](float p) { return p >= 1e-7f && p <= 1.f - 1e-7f; };
            auto clip = [](float p) { return fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f); };
            f32x4 p[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) p[nt][j] = __expf(z[nt][j] - m) * inv_s;
            if constexpr (MODE == CE_CLIP) {
                float S = 0.f, U = 0.f;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    if (cv[nt])
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            S += clip(p[nt][j]);
                            U += unclipped(p[nt][j]) ? p[nt][j] : 0.f;
                        }
                S = quad_sum(S);
                U = quad_sum(U);
                if (q == 0) part[wcol * 256 + lr] = make_float4(S, U, 0.f, 0.f);
            } else {
                const float gs = ri[2], invS = ri[3], c = ri[4], tq = ri[5];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    f32x4 g;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool hit = t == col[nt] + j;
                        g[j] = ce.keras_sparse ? gs * p[nt][j] * ((unclipped(p[nt][j]) ? invS : 0.f) - (hit ? tq : 0.f) - c) : gs * (p[nt][j] - (hit ? 1.f : 0.f));
                        if (!cv[nt]) g[j] = 0.f;               // columns V .. lddl-1 of the gradient are the zero K padding of the GEMMs that read it
                    }
                    if (rv) {
                        cs[nt] += g;
                        if (col[nt] < ce.lddl) {                   // lddl % 4 == 0
                            if (ce.dl_f32) {
                                *reinterpret_cast<f32x4*>(ce.dl_f32 + (long)row * ce.lddl + col[nt]) = g;
                            } else {
                                typedef unsigned short us4 __attribute__((ext_vector_type(4)));
                                *reinterpret_cast<us4*>(ce.dl_bf16 + (long)row * ce.lddl + col[nt]) =
                                    us4{Epilogue::bf16_bits(g[0]), Epilogue::bf16_bits(g[1]), Epilogue::bf16_bits(g[2]), Epilogue::bf16_bits(g[3])};
                            }
                        }
                    }
                }
            }
        }
    }
    if constexpr (MODE != CE_DL) {
        __syncthreads();
        if (tid < 256 && m0 + tid < ce.M) {                          // one thread per tile row: combine the four wave columns in order
            const float4 a0 = part[tid], a1 = part[256 + tid], a2 = part[512 + tid], a3 = part[768 + tid];
            float4 r;
            if constexpr (MODE == CE_STATS || MODE == CE_STATSZ) {
                const float m = fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x));            // finite: column 0 of every tile is a real word
                r = make_float4(m, a0.y * __expf(a0.x - m) + a1.y * __expf(a1.x - m) + a2.y * __expf(a2.x - m) + a3.y * __expf(a3.x - m),
                                fminf(fminf(a0.z, a1.z), fminf(a2.z, a3.z)), 0.f);
            } else {
                r = make_float4(a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y, 0.f, 0.f);
            }
            *reinterpret_cast<float4*>(ce.stats + ((long)(m0 + tid) * ce.tiles_n + tile_n) * CE_ST) = r;
        }
    } else if (ce.dbias_part) {
        float* red = lds;                                            // [2 wave groups][256 tile columns]
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = cs[nt][j];
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 4, 64);
                v += __shfl_xor(v, 8, 64);
                if (i == 0) red[group * 256 + 64 * wcol + 32 * (nt >> 1) + 16 * (nt & 1) + 4 * q + j] = v;
            }
        __syncthreads();
        if (tid < 256 && n0 + tid < ce.V) ce.dbias_part[(long)tile_m * ce.V + n0 + tid] = red[tid] + red[256 + tid];
    }
}

template <int MODE>
__global__ __launch_bounds__(b256::NTHREADS, 2) void vocab_ce_bf16_256_kernel(BOperand a, BOperand b, CeArgs ce, int K, int tiles_m) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_m, tile_n;
    b256::tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, ce.tiles_n, tile_m, tile_n);
    const int m0 = tile_m * b256::BM, n0 = tile_n * b256::BN;
    b256::Load<true, true> la;
    b256::Load<false, false> lb;
    la.init(a, m0, lane, wave);
    lb.init(b, n0, lane, wave);
    if constexpr (MODE == CE_CLIP) {
        if (!ce_tile_needs_clip(ce, m0)) return;
    }
    b256::f32x4 acc[8][4];
    b256::mainloop(la, lb, reinterpret_cast<char*>(smem), 0, K, acc);
    ce_epilogue256<MODE>(acc, smem, ce, m0, n0, tile_m, tile_n);
}

// One wave per row: combines the per-tile partials.
//   phase 0 (after STATS): m = max m_j, s = sum s_j exp(m_j - m), p_t = exp(z_t - m)/s.
//       categorical: loss = rw * -log(clip(p_t)); gs = grad_scale * rw when p_t is inside the clip range, else 0
//       (K.categorical_crossentropy's clip has zero gradient outside; its renormalisation is the identity on a softmax row).
//       keras_sparse: only m, 1/s, p_t are final; the CLIP pass follows.
//   phase 1 (after CLIP, keras_sparse): S, UP -> q_t = clip(p_t); loss = rw * (-log q_t + log S);
//       d/dz_k = gs * p_k * (u_k / S - [k == t] * tq - c),  tq = live / q_t,  c = UP / S - live * p_t / q_t,  gs = grad_scale * rw.
__global__ __launch_bounds__(256) void ce_rows_kernel(int phase, int M, int tiles_n, const float* __restrict__ stats, const float* __restrict__ zt,
                                                      const int32_t* __restrict__ targets, int V, const float* __restrict__ row_weights,
                                                      float grad_scale, int keras_sparse, float* __restrict__ rowinfo,
                                                      float* __restrict__ loss_rows, int* __restrict__ needs_clip, int tile_rows,
                                                      const unsigned short* __restrict__ z_bf16, long ldz) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= M) return;
    const float4* st = reinterpret_cast<const float4*>(stats) + (long)row * tiles_n;
    float* ri = rowinfo + (long)row * CE_RI;
    const float rw = row_weights ? row_weights[row] : 1.f;
    const int t = targets[row];
    if (phase == 0) {
        float m = -INFINITY;
        for (int j = lane; j < tiles_n; j += 64) m = fmaxf(m, st[j].x);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float s = 0.f, mn = INFINITY;
        for (int j = lane; j < tiles_n; j += 64) { const float4 q = st[j]; s += q.y * __expf(q.x - m); mn = fminf(mn, q.z); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); mn = fminf(mn, __shfl_xor(mn, o, 64)); }
        if (lane != 0) return;
        const float inv_s = 1.f / s;
        // the passes' own expression for a probability, at the row's smallest and largest logit (p is monotone in z): is anything clipped?
        const float pmin = __expf(mn - m) * inv_s, pmax = __expf(0.f) * inv_s;
        needs_clip[row] = (pmin >= 1e-7f && pmax <= 1.f - 1e-7f) ? 0 : 1;
        const float pt = (t >= 0 && t < V) ? __expf(zt[row] - m) * inv_s : 1.f;
        const bool live = pt >= 1e-7f && pt <= 1.f - 1e-7f;
        ri[0] = m; ri[1] = inv_s; ri[6] = pt;
        if (!keras_sparse) {
            ri[2] = live ? grad_scale * rw : 0.f;
            if (loss_rows) loss_rows[row] = rw * -logf(fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f));
        }
        return;
    }
    // decided per ROW: a row without a clipped probability takes S = UP = 1 exactly whether or not a neighbour made its tile run the
    // CLIP pass -- a row's loss and gradient do not depend on the batch it sits in or on the tile size (a row that needs the pass has
    // had it: its tile's blocks tested the same flags; tile_rows only sizes that test)
    (void)tile_rows;
    const int any = needs_clip[row];
    float S = 1.f, U = 1.f;                                    // no probability of this row is clipped: S = UP = sum p = s / s
    if (any && z_bf16) {
        // materialised logits: the row's clip sums straight from the parked bf16 logits (the row's own wave, 8 logits per lane and step;
        // V % 8 == 0).  Only rows that need the sums come here: with ordinary logits that is none of them.
        typedef unsigned short us8 __attribute__((ext_vector_type(8)));
        const float m = ri[0], inv_s = ri[1];
        const unsigned short* zr = z_bf16 + (long)row * ldz;
        S = 0.f; U = 0.f;
        for (int c = 8 * lane; c < V; c += 512) {
            const us8 zb = *reinterpret_cast<const us8*>(zr + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float p = __expf(__builtin_bit_cast(float, (unsigned)zb[j] << 16) - m) * inv_s;
                S += fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
                U += (p >= 1e-7f && p <= 1.f - 1e-7f) ? p : 0.f;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { S += __shfl_xor(S, o, 64); U += __shfl_xor(U, o, 64); }
    } else if (any) {
        S = 0.f; U = 0.f;
        for (int j = lane; j < tiles_n; j += 64) { const float4 q = st[j]; S += q.x; U += q.y; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { S += __shfl_xor(S, o, 64); U += __shfl_xor(U, o, 64); }
    }
    if (lane != 0) return;
    const float pt = ri[6];
    const bool live = pt >= 1e-7f && pt <= 1.f - 1e-7f;
    const float qt = fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f), invS = 1.f / S;
    ri[2] = grad_scale * rw;
    ri[3] = invS;
    ri[4] = U * invS - (live ? pt / qt : 0.f);
    ri[5] = live ? 1.f / qt : 0.f;
    if (loss_rows) loss_rows[row] = rw * (-logf(qt) + logf(S));
}

// Materialised logits -> gradient, IN PLACE (dl_bf16 holds the bf16 logits on entry, the bf16 gradient on exit; columns V .. lddl-1
// become the zero K padding of the GEMMs that read it).  Block = 256 rows x 256 columns: 32 lanes x 8 columns (one 16-byte access) per
// row, 8 row groups, 32 rows each; the column sums of the block's rows (the bias gradient's partials, [row tile][V]) meet in LDS in
// row-group order -- fixed summation order, like the GEMM epilogue's.  HBM-bound: 4 bytes per logit.
__global__ __launch_bounds__(256) void ce_dl_from_logits_kernel(CeArgs ce) {
    typedef unsigned short us8 __attribute__((ext_vector_type(8)));
    __shared__ float red[8][256];
    const int tid = threadIdx.x, cl = tid & 31, rg = tid >> 5;
    const int m0 = blockIdx.y * 256, col = blockIdx.x * 256 + 8 * cl;
    const bool inside = col < ce.lddl;                            // lddl % 8 == 0 (host-checked): a lane's 8 columns are all inside or all outside
    float cs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 0.f;
    auto unclipped = [](float p) { return p >= 1e-7f && p <= 1.f - 1e-7f; };
#pragma unroll 4
    for (int rr = 0; rr < 32; ++rr) {
        const int row = m0 + rg + 8 * rr;
        if (row >= ce.M || !inside) continue;
        unsigned short* zp = ce.dl_bf16 + (long)row * ce.lddl + col;
        const us8 zb = *reinterpret_cast<const us8*>(zp);
        const float* ri = ce.rowinfo + (long)row * CE_RI;
        const float m = ri[0], inv_s = ri[1], gs = ri[2], invS = ri[3], c = ri[4], tq = ri[5];
        const int t = ce.targets[row];
        us8 gb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float g = 0.f;
            if (col + j < ce.V) {
                const float p = __expf(__builtin_bit_cast(float, (unsigned)zb[j] << 16) - m) * inv_s;
                const bool hit = t == col + j;
                g = ce.keras_sparse ? gs * p * ((unclipped(p) ? invS : 0.f) - (hit ? tq : 0.f) - c) : gs * (p - (hit ? 1.f : 0.f));
            }
            cs[j] += g;
            gb[j] = Epilogue::bf16_bits(g);
        }
        *reinterpret_cast<us8*>(zp) = gb;
    }
    if (!ce.dbias_part) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rg][8 * cl + j] = cs[j];
    __syncthreads();
    const int oc = blockIdx.x * 256 + tid;
    if (oc < ce.V) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) v += red[g][tid];
        ce.dbias_part[(long)blockIdx.y * ce.V + oc] = v;
    }
}

// defined in loss.hip: partial [chunks][N] -> out[N] in a fixed order
__global__ void colsum_finish_kernel(const float* __restrict__ partial, int chunks, int N, float* __restrict__ out, int accumulate);

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct CePlan {
    int tiles_m, tiles_n;
    size_t off_stats, off_zt, off_ri, off_nc, off_db, total;
    int tile_rows;
};
// bf16 problems whose 256-square grid covers the chip run on the large tile (bgemm256_core.h)
static bool ce_big(const dc_vocab_ce_desc* d) { return d->bf16 && b256::prefer(d->M, d->V, d->K, 1); }
static CePlan ce_plan(const dc_vocab_ce_desc* d) {
    CePlan p;
    const int T = ce_big(d) ? 256 : 128;
    p.tiles_m = (d->M + T - 1) / T;
    p.tiles_n = (d->V + T - 1) / T;
    p.off_stats = 0;
    p.tile_rows = T;
    p.off_zt = align256((size_t)d->M * p.tiles_n * CE_ST * sizeof(float));
    p.off_ri = p.off_zt + align256((size_t)d->M * sizeof(float));
    p.off_nc = p.off_ri + align256((size_t)d->M * CE_RI * sizeof(float));
    p.off_db = p.off_nc + align256((size_t)d->M * sizeof(int));
    p.total = p.off_db + (d->dbias ? align256((size_t)p.tiles_m * d->V * sizeof(float)) : 0);
    return p;
}

static int ce_validate(const dc_vocab_ce_desc* d) {
    DC_REQUIRE(d != nullptr, DC_EINVAL, "dc_vocab_ce: null descriptor");
    DC_REQUIRE(d->M > 0 && d->V > 0 && d->K > 0 && d->X && d->W && d->targets, DC_EINVAL, "dc_vocab_ce: bad arguments");
    DC_REQUIRE(d->ldx >= d->K && d->ldw >= d->V, DC_EINVAL, "dc_vocab_ce: leading dimension smaller than the row length");
    DC_REQUIRE(aligned16(d->X) && aligned16(d->W) && (!d->bias || aligned16(d->bias)), DC_EALIGN, "dc_vocab_ce: X, W, bias must be 16-byte aligned");
    if (d->bf16) {
        DC_REQUIRE((d->K & 7) == 0 && (d->ldx & 7) == 0 && (d->ldw & 7) == 0 && (d->V & 7) == 0, DC_EALIGN,
                   "dc_vocab_ce (bf16): K, V, ldx, ldw must be multiples of 8");
        DC_REQUIRE((size_t)d->M * d->ldx * 2 < (size_t)0x7FFFFFF0u && (size_t)d->K * d->ldw * 2 < (size_t)0x7FFFFFF0u, DC_EINVAL,
                   "dc_vocab_ce (bf16): operands must span < 2 GiB");
    } else {
        DC_REQUIRE((d->K & 31) == 0 && (d->ldx & 3) == 0 && (d->ldw & 3) == 0 && (d->V & 3) == 0 && d->V >= 4, DC_EALIGN,
                   "dc_vocab_ce (f32): K must be a multiple of 32 and V, ldx, ldw multiples of 4");
        DC_REQUIRE((size_t)d->M * d->ldx * 4 < (size_t)0xFFFFFFF0u && (size_t)d->K * d->ldw * 4 < (size_t)0xFFFFFFF0u, DC_EINVAL,
                   "dc_vocab_ce (f32): operands must span < 4 GiB");
    }
    if (d->dlogits) {
        DC_REQUIRE(d->lddl >= d->V, DC_EINVAL, "dc_vocab_ce: lddl smaller than V");
        DC_REQUIRE(d->dl_bf16 ? ((d->lddl & 3) == 0 && (reinterpret_cast<uintptr_t>(d->dlogits) & 7u) == 0) : ((d->lddl & 3) == 0 && aligned16(d->dlogits)),
                   DC_EALIGN, "dc_vocab_ce: dlogits rows must be 16-byte (fp32) / 8-byte (bf16) aligned, lddl a multiple of 4");
    }
    DC_REQUIRE(!d->dbias || d->dlogits, DC_EINVAL, "dc_vocab_ce: dbias is a by-product of the dlogits pass");
    return DC_OK;
}

template <int MODE>
static int ce_launch(const dc_vocab_ce_desc* d, const CeArgs& ce, const CePlan& p, hipStream_t s) {
    const int tiles = p.tiles_m * p.tiles_n;
    if (ce_big(d)) {
        DC_ENSURE_DYN_LDS((&vocab_ce_bf16_256_kernel<MODE>), 160 * 1024);
        BOperand a{static_cast<const unsigned short*>(d->X), d->ldx, d->M, nullptr, (unsigned)((size_t)d->M * d->ldx * 2)};
        BOperand b{static_cast<const unsigned short*>(d->W), d->ldw, d->V, nullptr, (unsigned)((size_t)d->K * d->ldw * 2)};
        hipLaunchKernelGGL((vocab_ce_bf16_256_kernel<MODE>), dim3(tiles), dim3(b256::NTHREADS), b256::LDS_BYTES, s, a, b, ce, d->K, p.tiles_m);
    } else if (d->bf16) {
        DC_ENSURE_DYN_LDS((&vocab_ce_bf16_kernel<MODE>), 160 * 1024);
        BOperand a{static_cast<const unsigned short*>(d->X), d->ldx, d->M, nullptr, (unsigned)((size_t)d->M * d->ldx * 2)};
        BOperand b{static_cast<const unsigned short*>(d->W), d->ldw, d->V, nullptr, (unsigned)((size_t)d->K * d->ldw * 2)};
        hipLaunchKernelGGL((vocab_ce_bf16_kernel<MODE>), dim3(tiles), dim3(256), bgemm_lds_bytes(), s, a, b, ce, d->K);
    } else {
        DC_ENSURE_DYN_LDS((&vocab_ce_f32_kernel<MODE>), 160 * 1024);
        CeA32 al{static_cast<const float*>(d->X), d->ldx, d->M, nullptr};
        CeB32 bl{static_cast<const float*>(d->W), d->ldw, d->V, nullptr};
        constexpr size_t lds = igemm_lds_bytes<128, 128, CeA32, CeB32>();
        hipLaunchKernelGGL((vocab_ce_f32_kernel<MODE>), dim3(tiles), dim3(256), lds, s, al, bl, ce, d->K);
    }
    return check_launch("vocab_ce_kernel");
}

}  // namespace dcap

using namespace dcap;

extern "C" size_t dc_vocab_ce_workspace_bytes(const dc_vocab_ce_desc* d) {
    if (!d || d->M <= 0 || d->V <= 0) return 0;
    return ce_plan(d).total;
}

extern "C" int dc_vocab_ce(const dc_vocab_ce_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = ce_validate(d);
    if (rc) return rc;
    const CePlan p = ce_plan(d);
    DC_REQUIRE(workspace && workspace_bytes >= p.total, DC_EWORKSPACE, "dc_vocab_ce: needs %zu workspace bytes, got %zu", p.total, workspace_bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    CeArgs ce{};
    ce.M = d->M; ce.V = d->V; ce.tiles_n = p.tiles_n;
    ce.bias = d->bias; ce.targets = d->targets;
    ce.stats = reinterpret_cast<float*>(ws + p.off_stats);
    ce.zt = reinterpret_cast<float*>(ws + p.off_zt);
    float* rowinfo = reinterpret_cast<float*>(ws + p.off_ri);
    ce.rowinfo = rowinfo;
    ce.needs_clip = reinterpret_cast<int*>(ws + p.off_nc);
    ce.tile_rows = p.tile_rows;
    ce.keras_sparse = d->keras_sparse;
    const int row_blocks = (d->M + 3) / 4;
    // materialised bf16 logits: the 256-square bf16 tile with a bf16 gradient buffer whose rows take 16-byte accesses
    const bool mat = d->materialize_bf16 && d->bf16 && d->dl_bf16 && d->dlogits && ce_big(d) && (d->lddl & 7) == 0 && (d->V & 7) == 0 &&
                     aligned16(d->dlogits);
    if (mat) {
        ce.dl_bf16 = static_cast<unsigned short*>(d->dlogits);
        ce.lddl = d->lddl;
        rc = ce_launch<CE_STATSZ>(d, ce, p, s);
    } else {
        rc = ce_launch<CE_STATS>(d, ce, p, s);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(ce_rows_kernel, dim3(row_blocks), dim3(256), 0, s, 0, d->M, p.tiles_n, ce.stats, ce.zt, d->targets, d->V, d->row_weights,
                       d->grad_scale, d->keras_sparse, rowinfo, d->loss_rows, ce.needs_clip, p.tile_rows, (const unsigned short*)nullptr, 0L);
    rc = check_launch("ce_rows_kernel");
    if (rc) return rc;
    if (d->keras_sparse) {
        if (!mat) {
            rc = ce_launch<CE_CLIP>(d, ce, p, s);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(ce_rows_kernel, dim3(row_blocks), dim3(256), 0, s, 1, d->M, p.tiles_n, ce.stats, ce.zt, d->targets, d->V, d->row_weights,
                           d->grad_scale, d->keras_sparse, rowinfo, d->loss_rows, ce.needs_clip, p.tile_rows,
                           mat ? static_cast<const unsigned short*>(d->dlogits) : (const unsigned short*)nullptr, (long)d->lddl);
        rc = check_launch("ce_rows_kernel");
        if (rc) return rc;
    }
    if (mat) {
        ce.dbias_part = d->dbias ? reinterpret_cast<float*>(ws + p.off_db) : nullptr;
        hipLaunchKernelGGL(ce_dl_from_logits_kernel, dim3((d->lddl + 255) / 256, p.tiles_m), dim3(256), 0, s, ce);
        rc = check_launch("ce_dl_from_logits_kernel");
        if (rc || !d->dbias) return rc;
        hipLaunchKernelGGL(colsum_finish_kernel, dim3((d->V + 15) / 16), dim3(256), 0, s, ce.dbias_part, p.tiles_m, d->V, d->dbias, 0);
        return check_launch("colsum_finish_kernel");
    }
    if (!d->dlogits) return DC_OK;
    if (d->dl_bf16) ce.dl_bf16 = static_cast<unsigned short*>(d->dlogits);
    else ce.dl_f32 = static_cast<float*>(d->dlogits);
    ce.lddl = d->lddl;
    ce.dbias_part = d->dbias ? reinterpret_cast<float*>(ws + p.off_db) : nullptr;
    rc = ce_launch<CE_DL>(d, ce, p, s);
    if (rc || !d->dbias) return rc;
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((d->V + 15) / 16), dim3(256), 0, s, ce.dbias_part, p.tiles_m, d->V, d->dbias, 0);
    return check_launch("colsum_finish_kernel");
}
