// lstm.hip -- Keras-2.1 LSTM recurrence over a whole sequence (forward and backward), time-major.
// One launch per timestep in each direction (U % 32 == 0 forward, U % 16 == 0 backward, no recurrent dropout): the skinny
// recurrent product (h_{t-1} * U_rec forward, dz_{t+1} * U_rec^T backward) streams its operands straight from L2 into MFMA
// fragments, the K split over the block's waves meets in LDS, and the same threads finish the hard-sigmoid / tanh gate math,
// the cell update and the Keras mask carry.  Other shapes and the recurrent-dropout path use the split-K MFMA GEMM plus
// pointwise gate kernels.  The caller batches the x * kernel + bias projection for all T steps into one GEMM beforehand
// (dcap.h); the recurrent weight gradient is one GEMM over all steps at the end.
#include "dcap_internal.h"
#include <algorithm>
#include <cstdlib>
#include <string>

namespace dcap {


__device__ __forceinline__ float hard_sigmoid(float z) { return fminf(fmaxf(0.2f * z + 0.5f, 0.f), 1.f); }
__device__ __forceinline__ float hard_sigmoid_grad(float z) {
    const float y = 0.2f * z + 0.5f;
    return (y >= 0.f && y <= 1.f) ? 0.2f : 0.f;   // tf.clip_by_value passes the gradient on [min, max]
}

// z_t holds the full pre-activation (x-projection + bias + h*U).  One thread per (b, unit).
__global__ void lstm_gate_fwd_kernel(const float* __restrict__ z_t, const float* __restrict__ h_prev,
                                     const float* __restrict__ c_prev, const uint8_t* __restrict__ mask_t,
                                     float* __restrict__ h_t, float* __restrict__ c_t, int B, int U) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * U) return;
    const int b = idx / U, u = idx - b * U;
    const float* z = z_t + (long)b * 4 * U;
    const float i = hard_sigmoid(z[u]), f = hard_sigmoid(z[U + u]), g = tanhf(z[2 * U + u]), o = hard_sigmoid(z[3 * U + u]);
    const float hp = h_prev ? h_prev[idx] : 0.f, cp = c_prev ? c_prev[idx] : 0.f;
    const float cn = f * cp + i * g;
    const float hn = o * tanhf(cn);
    const bool m = mask_t ? (mask_t[b] != 0) : true;
    h_t[idx] = m ? hn : hp;
    c_t[idx] = m ? cn : cp;
}

// dh_io: in = gradient arriving at h_t from step t+1 (recurrence + mask pass-through); out = the part
// of it that passes straight to h_{t-1} through masked rows (the GEMM then adds dz*U^T on top).
// dc_io: same for the cell state.
__global__ void lstm_gate_bwd_kernel(const float* __restrict__ z_t, const float* __restrict__ c_prev,
                                     const uint8_t* __restrict__ mask_t, const float* __restrict__ dh_out_t,
                                     const float* __restrict__ dh_last, float* __restrict__ dh_io, float* __restrict__ dc_io,
                                     float* __restrict__ dz_t, int B, int U) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * U) return;
    const int b = idx / U, u = idx - b * U;
    const float* z = z_t + (long)b * 4 * U;
    const float zi = z[u], zf = z[U + u], zc = z[2 * U + u], zo = z[3 * U + u];
    const float i = hard_sigmoid(zi), f = hard_sigmoid(zf), g = tanhf(zc), o = hard_sigmoid(zo);
    const float cp = c_prev ? c_prev[idx] : 0.f;
    const float tc = tanhf(f * cp + i * g);
    float dh = dh_io[idx];
    if (dh_out_t) dh += dh_out_t[idx];
    if (dh_last) dh += dh_last[idx];
    const float dc = dc_io[idx];
    const bool m = mask_t ? (mask_t[b] != 0) : true;
    float* dz = dz_t + (long)b * 4 * U;
    if (m) {
        const float dcn = dc + dh * o * (1.f - tc * tc);
        dz[u] = dcn * g * hard_sigmoid_grad(zi);
        dz[U + u] = dcn * cp * hard_sigmoid_grad(zf);
        dz[2 * U + u] = dcn * i * (1.f - g * g);
        dz[3 * U + u] = dh * tc * hard_sigmoid_grad(zo);
        dh_io[idx] = 0.f;
        dc_io[idx] = dcn * f;
    } else {
        dz[u] = 0.f; dz[U + u] = 0.f; dz[2 * U + u] = 0.f; dz[3 * U + u] = 0.f;
        dh_io[idx] = dh;
        dc_io[idx] = dc;
    }
}

// ------------------------------------------------------------------------------------------------
// Fused forward timestep (U % 32 == 0): z_t = zx_t + h_{t-1} * U_rec, gates, cell update, mask carry in ONE
// launch.  Block = 64 batch rows x 8 units (= 32 columns: 4 gates x 8 units), 4 waves that split K = U
// four ways; every wave streams its K range straight from global/L2 into MFMA fragments (one 16-byte load
// of h per 4 MFMAs, one dword of U_rec per MFMA, prefetched 4 chunks ahead; SGPR-advanced bases: no
// per-chunk VALU), the four partial 32x32 tiles meet in LDS and the 256 threads finish the gate math.
// Replaces {split-K GEMM + slab reduce + gate kernel} = 3 launches and ~16 MB of slab traffic per step.
// ------------------------------------------------------------------------------------------------
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// The weights: the first version of this kernel read U_rec in place: a block's 32 columns are 4 gate segments of 8 floats,
// i.e. 32 useful bytes per 128-byte line, and every weight was fetched once per 32-row block -- 128 MB through L2 per
// timestep at U = 1024, B = 64 (41-60 us: L2-bandwidth-bound).  Now dc_lstm_seq_fwd_f32 first repacks U_rec (once per
// call, 4U^2 floats into the workspace) so that a block's 32 columns are ONE contiguous 128-byte line per k row
// (Upk[k][u/8][gate][u%8]), and a block covers 64 batch rows (two MFMA row blocks share every weight fragment).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_pack_urec_kernel(const float* __restrict__ U_rec, float* __restrict__ Upk, int U) {
    const long total = (long)U * 4 * U;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int within = (int)(idx & 31), g = within >> 3, j = within & 7;
        const long blk = idx >> 5;
        const int ub = (int)(blk % (U / 8));
        const long k = blk / (U / 8);
        Upk[idx] = U_rec[k * (4L * U) + (long)g * U + ub * 8 + j];
    }
}

// RT = 32-row MFMA blocks per workgroup, NW = waves that split K = U.  The gate phase's operands (zx, h, c of the block's own
// (row, unit) items) are requested before the K loop, so their latency hides behind it.  LDS: NW/2 partial tiles (waves
// NW/2.. publish, waves 0..NW/2-1 add theirs on top): 16.9 KB at <1,8> and <2,4> -- in the training pipeline this kernel runs
// beside the encoder's convolutions, whose two resident blocks leave ~19 KB of a CU's LDS; with 33.8 KB its blocks could only
// start at conv-kernel boundaries (78-100 us per step in the pipeline against 18 us alone).
// (The recurrent-dropout variant of the step is lstm_step_masked_kernel below.)
template <int RT, int NW>
__global__ __launch_bounds__(NW * 64) void lstm_step_fused_kernel(float* __restrict__ z_t, const float* __restrict__ Upk,
                                                                  const float* __restrict__ h_prev, const float* __restrict__ c_prev,
                                                                  const uint8_t* __restrict__ mask_t, float* __restrict__ h_t,
                                                                  float* __restrict__ c_t, int B, int U) {
    constexpr int HALF = NW / 2, THREADS = NW * 64, ITEMS = RT * 256, IT = (ITEMS + THREADS - 1) / THREADS;
    constexpr int NG = 1;
    __shared__ float part[HALF][RT * 32][33];
    // In the training pipeline these waves share their SIMDs with the encoder's convolution waves, which issue 64-clock fp32 MFMAs
    // back to back.  The recurrence is the decoder's serial chain and a step's MFMA work is tiny: ask for the issue slots first.
    // Measured in the pipeline: 91 -> 71 us per step launch (12 us alone); the step time itself does not move (8.44 ms either way), and
    // a register diet (84 instead of 135 VGPRs) changes nothing -- most of the remaining wait is workgroup placement, not issue.
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ub = blockIdx.x, u0 = ub * 8, r0 = blockIdx.y * (RT * 32);
    const int i = lane & 31, h = lane >> 5;
    const int kq = U / NW, kbeg = wave * kq;
    const int nch = kq / 8;                                                   // 8 k values per chunk (4 MFMAs of K = 2)
    f32x16_t acc[NG][RT];
    const float* ap[NG][RT];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][rt][r] = 0.f;
            ap[g][rt] = h_prev + (long)min(r0 + 32 * rt + i, B - 1) * U + kbeg + 4 * h;
        }
    const long kstride = (long)(U / 8) * 32;                                  // floats between consecutive k rows of Upk
    const float* bp = Upk + (long)(kbeg + 4 * h) * kstride + (long)ub * 32 + i;
    constexpr int PF = 8;
    f4_t a[PF][NG][RT];
    float b[PF][4];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int c = min(p, nch - 1);
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[p][g][rt] = *reinterpret_cast<const f4_t*>(ap[g][rt] + 8 * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[p][j] = bp[(long)(8 * c + j) * kstride];
    }
    float gz[IT][4], ghp[IT], gcp[IT];
    bool gmk[IT];
#pragma unroll
    for (int q = 0; q < IT; ++q) {
        const int e = tid + q * THREADS, row = e >> 3, uu = e & 7;
        const int brow = min(r0 + row, B - 1);
        const float* zrow = z_t + (long)brow * 4 * U + u0 + uu;
#pragma unroll
        for (int g = 0; g < 4; ++g) gz[q][g] = zrow[(long)g * U];
        const long o = (long)brow * U + u0 + uu;
        ghp[q] = h_prev[o];
        gcp[q] = c_prev[o];
        gmk[q] = mask_t ? (mask_t[brow] != 0) : true;
    }
    for (int c0 = 0; c0 < nch; c0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            f4_t x[NG][RT];
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) x[g][rt] = a[p][g][rt];
            const float b0 = b[p][0], b1 = b[p][1], b2 = b[p][2], b3 = b[p][3];
            if (c0 + PF < nch) {                                              // the next ring of chunks (clamped at the end)
                const int cn = min(c0 + p + PF, nch - 1);
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) a[p][g][rt] = *reinterpret_cast<const f4_t*>(ap[g][rt] + 8 * cn);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[p][j] = bp[(long)(8 * cn + j) * kstride];
            }
            if (c0 + p < nch) {
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        acc[g][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[g][rt].x, b0, acc[g][rt], 0, 0, 0);
                        acc[g][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[g][rt].y, b1, acc[g][rt], 0, 0, 0);
                        acc[g][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[g][rt].z, b2, acc[g][rt], 0, 0, 0);
                        acc[g][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[g][rt].w, b3, acc[g][rt], 0, 0, 0);
                    }
            }
        }
    }
    f32x16_t mine[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) mine[rt] = acc[0][rt];
    if (wave >= HALF) {                                // round 1: the upper waves publish their K share's partial tiles
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[wave - HALF][32 * rt + (r & 3) + 8 * (r >> 2) + 4 * h][i] = mine[rt][r];
    }
    __syncthreads();
    if (wave < HALF) {                                 // round 2: the lower waves add theirs on top (same lane owns the same elements)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[wave][32 * rt + (r & 3) + 8 * (r >> 2) + 4 * h][i] += mine[rt][r];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < IT; ++q) {
        const int e = tid + q * THREADS, row = e >> 3, uu = e & 7;
        const int brow = r0 + row;
        if (e >= ITEMS || brow >= B) continue;
        float zg[4];
        float* zrow = z_t + (long)brow * 4 * U + u0 + uu;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = g * 8 + uu;
            float rec = 0.f;
#pragma unroll
            for (int w = 0; w < HALF; ++w) rec += part[w][row][col];
            zg[g] = gz[q][g] + rec;
            zrow[(long)g * U] = zg[g];
        }
        const long o = (long)brow * U + u0 + uu;
        const float ig = hard_sigmoid(zg[0]), fg = hard_sigmoid(zg[1]), gg = tanhf(zg[2]), og = hard_sigmoid(zg[3]);
        const float cn = fg * gcp[q] + ig * gg;
        const float hn = og * tanhf(cn);
        h_t[o] = gmk[q] ? hn : ghp[q];
        c_t[o] = gmk[q] ? cn : gcp[q];
    }
}

// ------------------------------------------------------------------------------------------------
// Fused forward timestep WITH recurrent-dropout masks (round 5; U % 32 == 0).  The masked step above kept the plain step's block
// (32 rows x 8 units x 4 gates = one 32-column MFMA tile) and computed that tile once per gate, because every gate has its own A
// operand h * m_g: 4 x the MFMAs for the same output, 6.8 us of matrix-pipe time per block at U = 512 (18.4 us per step against 7.5
// for the plain step -- the comment that called it irrelevant was wrong).  Here a wave owns ONE gate: block = 16 RT rows x 16 units,
// waves = 4 gates x 2 halves of K, v_mfma_f32_16x16x4_f32 tiles (the backward step's shape), so no product is computed twice:
// U / 8 MFMAs of 32 cycles per wave.  The masks are applied to the A fragments in registers (h and m_g each one 16-byte load per
// four MFMAs): no pre-masked copies of h, no mask kernel in front of the first step, no four extra stores per element in the epilogue.
// B operand: UpkT[gate * U + unit][k] = U_rec[k][gate * U + unit] (lstm_pack_urec_t_kernel, once per call): a lane reads 16 bytes along k.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_pack_urec_t_kernel(const float* __restrict__ U_rec, float* __restrict__ UpkT, int U) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, k0 = blockIdx.y * 32;            // columns of U_rec (gate * U + unit), rows k
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = U_rec[(long)(k0 + r) * (4L * U) + c0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) UpkT[(long)(c0 + r) * U + k0 + tx] = tile[tx][r];
}

template <int RT>
__global__ __launch_bounds__(512) void lstm_step_masked_kernel(float* __restrict__ z_t, const float* __restrict__ UpkT, const float* __restrict__ h_prev,
                                                               const float* __restrict__ c_prev, const uint8_t* __restrict__ mask_t,
                                                               float* __restrict__ h_t, float* __restrict__ c_t, int B, int U,
                                                               const float* __restrict__ rec_masks) {
    constexpr int NW = 8, THREADS = NW * 64, ITEMS = RT * 256, IT = (ITEMS + THREADS - 1) / THREADS;
    __shared__ float part[NW][RT * 16][17];
    __builtin_amdgcn_s_setprio(3);                       // see lstm_step_fused_kernel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gate = wave & 3, kp = wave >> 2;
    const int u0 = blockIdx.x * 16, r0 = blockIdx.y * (RT * 16);
    const int m = lane & 15, kk = lane >> 4;
    const int kq = U / 2, kbeg = kp * kq, nch = kq / 16;            // 16 k values per chunk (4 MFMAs of K = 4)
    const long gs = (long)B * U;
    const float* bp = UpkT + ((long)gate * U + u0 + m) * U + kbeg + 4 * kk;
    const float* ap[RT];
    const float* mp[RT];
    f32x4_t acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const long o = (long)min(r0 + 16 * rt + m, B - 1) * U + kbeg + 4 * kk;
        ap[rt] = h_prev + o;
        mp[rt] = rec_masks + gate * gs + o;
        acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    constexpr int PF = (RT == 1) ? 16 : 8;               // U = 512: the wave's whole K share is requested before the first MFMA (one latency, not two)
    f4_t a[PF][RT], mk[PF][RT], b[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int c = min(p, nch - 1);
        b[p] = *reinterpret_cast<const f4_t*>(bp + 16 * c);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            a[p][rt] = *reinterpret_cast<const f4_t*>(ap[rt] + 16 * c);
            mk[p][rt] = *reinterpret_cast<const f4_t*>(mp[rt] + 16 * c);
        }
    }
    // the gate phase's operands: issued now, consumed after the K loop
    float gz[IT][4], ghp[IT], gcp[IT];
    bool gmk[IT];
#pragma unroll
    for (int q = 0; q < IT; ++q) {
        const int e = tid + q * THREADS, row = e >> 4, uu = e & 15;
        const int brow = min(r0 + row, B - 1);
        const float* zrow = z_t + (long)brow * 4 * U + u0 + uu;
#pragma unroll
        for (int g = 0; g < 4; ++g) gz[q][g] = zrow[(long)g * U];
        const long o = (long)brow * U + u0 + uu;
        ghp[q] = h_prev[o];
        gcp[q] = c_prev[o];
        gmk[q] = mask_t ? (mask_t[brow] != 0) : true;
    }
    for (int c0 = 0; c0 < nch; c0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            f4_t x[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) x[rt] = a[p][rt] * mk[p][rt];
            const f4_t y = b[p];
            if (c0 + PF < nch) {                                // block-uniform: the next ring of chunks (clamped at the end)
                const int cn = min(c0 + p + PF, nch - 1);
                b[p] = *reinterpret_cast<const f4_t*>(bp + 16 * cn);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    a[p][rt] = *reinterpret_cast<const f4_t*>(ap[rt] + 16 * cn);
                    mk[p][rt] = *reinterpret_cast<const f4_t*>(mp[rt] + 16 * cn);
                }
            }
            if (c0 + p < nch) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].x, y.x, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].y, y.y, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].z, y.z, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].w, y.w, acc[rt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][16 * rt + 4 * kk + r][m] = acc[rt][r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < IT; ++q) {
        const int e = tid + q * THREADS, row = e >> 4, uu = e & 15;
        const int brow = r0 + row;
        if (e >= ITEMS || brow >= B) continue;
        float zg[4];
        float* zrow = z_t + (long)brow * 4 * U + u0 + uu;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            zg[g] = gz[q][g] + (part[g][row][uu] + part[4 + g][row][uu]);       // the two halves of K
            zrow[(long)g * U] = zg[g];
        }
        const long o = (long)brow * U + u0 + uu;
        const float ig = hard_sigmoid(zg[0]), fg = hard_sigmoid(zg[1]), gg = tanhf(zg[2]), og = hard_sigmoid(zg[3]);
        const float cn = fg * gcp[q] + ig * gg;
        const float hn = og * tanhf(cn);
        h_t[o] = gmk[q] ? hn : ghp[q];
        c_t[o] = gmk[q] ? cn : gcp[q];
    }
}

// ------------------------------------------------------------------------------------------------
// Fused backward timestep (U % 16 == 0, t < T-1): the recurrent gradient dz_{t+1} * U_rec^T, the gate derivatives and the
// mask carry in ONE launch (was: gate kernel + split-K GEMM + slab reducer = 3 launches, 27.6 us per step at B = 64, U = 512).
// Block = RT*16 batch rows x 16 units; the K = 4U reduction is split over the 4 waves, each streaming its quarter of the 16 rows
// of U_rec (a row of U_rec IS the B operand's K run: no repack) and of the dz rows straight from L2 into v_mfma_f32_16x16x4_f32
// fragments (one 16-byte load per operand per 4 MFMAs, PF chunks in flight); the four partial tiles meet in LDS and the block's
// threads finish the gate math for their (row, unit).  Every (row, unit) belongs to one thread of one block, so the carried
// dh/dc state is updated in place.
// ------------------------------------------------------------------------------------------------

template <int RT, int NW>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_step_fused_kernel(const float* __restrict__ z_t, const float* __restrict__ c_prev,
                                                                      const uint8_t* __restrict__ mask_t, const float* __restrict__ dh_out_t,
                                                                      const float* __restrict__ dz_next, const float* __restrict__ U_rec,
                                                                      float* __restrict__ dh_io, float* __restrict__ dc_io,
                                                                      float* __restrict__ dz_t, int B, int U,
                                                                      const float* __restrict__ rec_masks = nullptr) {
    // rec_masks [4][B][U] (recurrent dropout): dh_t += sum_g m_g * (dz_{t+1,g} U_g^T).  A wave's share of K = 4U lies inside ONE gate
    // (NW = 4: wave = gate; NW = 8: two waves per gate), so its partial tile IS that gate's product: the masks apply where the tiles meet.
    __shared__ float part[NW][RT * 16][17];
    constexpr int THREADS = NW * 64, ITEMS = RT * 256, IT = (ITEMS + THREADS - 1) / THREADS;
    __builtin_amdgcn_s_setprio(3);                       // see lstm_step_fused_kernel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u0 = blockIdx.x * 16, r0 = blockIdx.y * (RT * 16);
    const int m = lane & 15, kk = lane >> 4;
    const long K = 4L * U;
    const int kq = (4 * U) / NW, kbeg = wave * kq;              // this wave's share of K = 4U
    const int nch = kq / 16;                                    // 16 k values per chunk
    const float* bp = U_rec + (long)(u0 + m) * K + kbeg + 4 * kk;
    const float* ap[RT];
    f32x4_t acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        ap[rt] = dz_next + (long)min(r0 + 16 * rt + m, B - 1) * K + kbeg + 4 * kk;
        acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    constexpr int PF = (RT == 1) ? 16 : 8;
    f4_t a[PF][RT], b[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int c = min(p, nch - 1);
        b[p] = *reinterpret_cast<const f4_t*>(bp + 16 * c);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[p][rt] = *reinterpret_cast<const f4_t*>(ap[rt] + 16 * c);
    }
    // the gate phase's operands: issued now, consumed after the K loop
    float gz[IT][4], gcp[IT], gdh[IT], gdc[IT];
    bool gmk[IT];
#pragma unroll
    for (int q = 0; q < IT; ++q) {
        const int e = tid + q * THREADS, row = e >> 4, uu = e & 15;
        const int brow = min(r0 + row, B - 1), u = u0 + uu;
        const long idx = (long)brow * U + u;
        const float* z = z_t + (long)brow * K + u;
#pragma unroll
        for (int g = 0; g < 4; ++g) gz[q][g] = z[(long)g * U];
        gcp[q] = c_prev ? c_prev[idx] : 0.f;
        gdh[q] = dh_io[idx] + (dh_out_t ? dh_out_t[idx] : 0.f);
        gdc[q] = dc_io[idx];
        gmk[q] = mask_t ? (mask_t[brow] != 0) : true;
    }
    for (int c0 = 0; c0 < nch; c0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            f4_t x[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) x[rt] = a[p][rt];
            const f4_t y = b[p];
            if (c0 + PF < nch) {                                // block-uniform: the next ring of chunks (clamped at the end)
                const int cn = min(c0 + p + PF, nch - 1);
                b[p] = *reinterpret_cast<const f4_t*>(bp + 16 * cn);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[p][rt] = *reinterpret_cast<const f4_t*>(ap[rt] + 16 * cn);
            }
            if (c0 + p < nch) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].x, y.x, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].y, y.y, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].z, y.z, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].w, y.w, acc[rt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][16 * rt + 4 * kk + r][m] = acc[rt][r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < IT; ++q) {
        const int e = tid + q * THREADS, row = e >> 4, uu = e & 15;
        const int brow = r0 + row;
        if (e >= ITEMS || brow >= B) continue;
        const int u = u0 + uu;
        const long idx = (long)brow * U + u;
        float rec = 0.f;
        if (rec_masks) {
            const long gs = (long)B * U;
#pragma unroll
            for (int w = 0; w < NW; ++w) rec += part[w][row][uu] * rec_masks[(w * 4 / NW) * gs + idx];
        } else {
#pragma unroll
            for (int w = 0; w < NW; ++w) rec += part[w][row][uu];
        }
        const float zi = gz[q][0], zf = gz[q][1], zc = gz[q][2], zo = gz[q][3];
        const float i = hard_sigmoid(zi), f = hard_sigmoid(zf), g = tanhf(zc), o = hard_sigmoid(zo);
        const float cp = gcp[q];
        const float tc = tanhf(f * cp + i * g);
        const float dh = gdh[q] + rec;
        const float dc = gdc[q];
        float* dz = dz_t + (long)brow * K;
        if (gmk[q]) {
            const float dcn = dc + dh * o * (1.f - tc * tc);
            dz[u] = dcn * g * hard_sigmoid_grad(zi);
            dz[U + u] = dcn * cp * hard_sigmoid_grad(zf);
            dz[2 * U + u] = dcn * i * (1.f - g * g);
            dz[3 * U + u] = dh * tc * hard_sigmoid_grad(zo);
            dh_io[idx] = 0.f;
            dc_io[idx] = dcn * f;
        } else {
            dz[u] = 0.f; dz[U + u] = 0.f; dz[2 * U + u] = 0.f; dz[3 * U + u] = 0.f;
            dh_io[idx] = dh;
            dc_io[idx] = dc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// recurrent_dropout (Keras LSTM(recurrent_dropout=0.2), text_generation_model.py:141-142): in the training phase
// h_{t-1} enters each gate through its own inverted-dropout mask, drawn once per call and fixed over the timesteps:
//   z_g = x W_g + (h_{t-1} * m_g) U_g + b_g,   g in {i, f, c, o},  m_g in {0, 1/(1-rate)}^[B,U]
// Four masked copies of h and four U x U products per step replace the single h * U_rec product; this path is taken only
// when the caller passes masks (training with dropout on), the fused step kernel otherwise.
// ------------------------------------------------------------------------------------------------
// out[g][r][u] = x[r][u] * masks[g][r % B][u]   (rows = B for one step, (T-1)*B for the weight-gradient operand)
__global__ __launch_bounds__(256) void lstm_mask_rows_kernel(const float* __restrict__ x, const float* __restrict__ masks, float* __restrict__ out,
                                                             long rows, int B, int U) {
    const long n = rows * U, per = (long)B * U;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const long r = idx / U;
        const long mo = (r % B) * U + (idx - r * U);
        const float v = x[idx];
#pragma unroll
        for (int g = 0; g < 4; ++g) out[g * n + idx] = v * masks[g * per + mo];
    }
}

// dh[b][u] += sum_g masks[g][b][u] * tmp[g][b][u]
__global__ __launch_bounds__(256) void lstm_masked_acc_kernel(float* __restrict__ dh, const float* __restrict__ masks, const float* __restrict__ tmp, long n) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        float s = dh[idx];
#pragma unroll
        for (int g = 0; g < 4; ++g) s += masks[g * n + idx] * tmp[g * n + idx];
        dh[idx] = s;
    }
}

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static dc_gemm_desc hU_desc(int B, int U, const float* h_prev, const float* U_rec, float* z_t) {
    dc_gemm_desc g{};
    g.M = B; g.N = 4 * U; g.K = U;
    g.A = h_prev; g.lda = U; g.a_trans = 0;
    g.B = U_rec; g.ldb = 4 * U; g.b_trans = 0;
    g.C = z_t; g.ldc = 4 * U;
    g.accumulate = 1;
    return g;
}

static dc_gemm_desc dzUt_desc(int B, int U, const float* dz_t, const float* U_rec, float* dh) {
    dc_gemm_desc g{};
    g.M = B; g.N = U; g.K = 4 * U;
    g.A = dz_t; g.lda = 4 * U; g.a_trans = 0;
    g.B = U_rec; g.ldb = 4 * U; g.b_trans = 1;
    g.C = dh; g.ldc = U;
    g.accumulate = 1;
    return g;
}

static dc_gemm_desc dU_desc(int B, int T, int U, const float* h_seq, const float* dz, float* dU, int accumulate) {
    dc_gemm_desc g{};
    g.M = U; g.N = 4 * U; g.K = (T - 1) * B;
    g.A = h_seq; g.lda = U; g.a_trans = 1;            // rows t*B+b, t = 0..T-2  == h_{t-1} for steps 1..T-1
    g.B = dz + (long)B * 4 * U; g.ldb = 4 * U; g.b_trans = 0;
    g.C = dU; g.ldc = 4 * U;
    g.accumulate = accumulate;
    return g;
}

}  // namespace dcap

using namespace dcap;


extern "C" size_t dc_lstm_seq_workspace_bytes(int B, int T, int U) {
    if (B <= 0 || T <= 0 || U <= 0) return 0;
    dc_gemm_desc a = hU_desc(B, U, nullptr, nullptr, nullptr);
    dc_gemm_desc b = dzUt_desc(B, U, nullptr, nullptr, nullptr);
    size_t g = std::max(dc_gemm_workspace_bytes(&a), dc_gemm_workspace_bytes(&b));
    if (T > 1) {
        dc_gemm_desc c = dU_desc(B, T, U, nullptr, nullptr, nullptr, 0);
        g = std::max(g, dc_gemm_workspace_bytes(&c));
    }
    const size_t bwd = align_up(g) + 2 * align_up((size_t)B * U * sizeof(float));
    const size_t fwd = align_up(g) + ((U & 31) == 0 ? align_up((size_t)4 * U * U * sizeof(float)) : 0);     // + the repacked U_rec
    // recurrent-dropout path: four masked copies of h (forward), of dz * U^T and of h_seq (backward); its per-gate GEMMs
    // ([B,U] x [U,U] and [U,(T-1)B] x [(T-1)B,U]) choose their own split-K slabs
    {
        dc_gemm_desc a{};
        a.M = B; a.N = U; a.K = U; a.lda = U; a.ldb = 4 * U; a.ldc = 4 * U;
        g = std::max(g, dc_gemm_workspace_bytes(&a));
        a.b_trans = 1; a.lda = 4 * U; a.ldc = U;
        g = std::max(g, dc_gemm_workspace_bytes(&a));
        if (T > 1) {
            dc_gemm_desc c{};
            c.M = U; c.N = U; c.K = (T - 1) * B; c.lda = U; c.a_trans = 1; c.ldb = 4 * U; c.ldc = 4 * U;
            g = std::max(g, dc_gemm_workspace_bytes(&c));
        }
    }
    const size_t drop = align_up(g) + 2 * align_up((size_t)B * U * sizeof(float)) + align_up((size_t)4 * B * U * sizeof(float)) +
                        align_up((size_t)4 * (size_t)std::max(T - 1, 1) * B * U * sizeof(float)) + 1024;
    // fused masked forward steps: the transposed U_rec (the masks are applied to the A fragments in registers: no copies of h)
    const size_t drop_fused = align_up(g) + align_up((size_t)4 * U * U * sizeof(float)) + 1024;
    return std::max(std::max(std::max(fwd, bwd), drop), drop_fused);
}

extern "C" int dc_lstm_seq_fwd_f32(const dc_lstm_fwd_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(d && d->z && d->U_rec && d->h_seq && d->c_seq, DC_EINVAL, "dc_lstm_seq_fwd: null pointer");
    DC_REQUIRE(d->B > 0 && d->T > 0 && d->U > 0 && (d->U & 3) == 0, DC_EINVAL, "dc_lstm_seq_fwd: bad B/T/U (U %% 4 == 0)");
    DC_REQUIRE(workspace_bytes >= dc_lstm_seq_workspace_bytes(d->B, d->T, d->U) && (workspace || workspace_bytes == 0),
               DC_EWORKSPACE, "dc_lstm_seq_fwd: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = d->B, U = d->U, n = B * U, blocks = (n + 255) / 256;
    const bool fused = (U & 31) == 0 && d->T > 1;   // (else: per-step GEMM + gate kernel, with masks a mask kernel + four per-gate GEMMs)
    int frt = ((U / 8) * ((B + 31) / 32) <= 2 * kNumCU) ? 1 : 2;      // 32- or 64-row blocks (measured again in round 5 at 200 x 512: 12.3 vs 13.2 us)
    int fnw = (frt == 2 && (U & 63) == 0) ? 8 : 4;                     // measured: 4 waves at 32 rows, 8 at 64
    float* Upk = nullptr;
    float* hm = nullptr;                            // dropout: [4][B][U] masked copies of h_{t-1}, at the END of the workspace
    void* gws = workspace;
    size_t gws_bytes = workspace_bytes;
    size_t tail = 0;                                // bytes taken from the END of the workspace
    if (d->rec_masks && !fused) {                   // (the fused masked step masks its A fragments in registers: no copies)
        const size_t hm_bytes = align_up((size_t)4 * n * sizeof(float));
        char* end = static_cast<char*>(workspace) + workspace_bytes / 256 * 256;
        hm = reinterpret_cast<float*>(end - hm_bytes);
        tail = hm_bytes + (workspace_bytes - workspace_bytes / 256 * 256);
        gws_bytes = workspace_bytes - tail;
    }
    if (fused) {                                   // line-contiguous copy of the recurrent weights at the END of the workspace (below the masked copies)
        const size_t pack_bytes = align_up((size_t)4 * U * U * sizeof(float));
        Upk = reinterpret_cast<float*>(static_cast<char*>(workspace) + (workspace_bytes - tail - pack_bytes) / 256 * 256);
        const long total = (long)4 * U * U;
        if (d->rec_masks) hipLaunchKernelGGL(lstm_pack_urec_t_kernel, dim3(4 * U / 32, U / 32), dim3(256), 0, s, d->U_rec, Upk, U);       // [4U][U]: k contiguous
        else hipLaunchKernelGGL(lstm_pack_urec_kernel, dim3((int)std::min<long>((total + 255) / 256, (long)kNumCU * 8)), dim3(256), 0, s, d->U_rec, Upk, U);
        int rc = check_launch("lstm_pack_urec_kernel");
        if (rc) return rc;
    }
    for (int t = 0; t < d->T; ++t) {
        float* z_t = d->z + (long)t * B * 4 * U;
        const float* hp = t ? d->h_seq + (long)(t - 1) * n : nullptr;
        const float* cp = t ? d->c_seq + (long)(t - 1) * n : nullptr;
        if (t && fused) {
            const uint8_t* mk = d->mask ? d->mask + (long)t * B : nullptr;
            float* h_t = d->h_seq + (long)t * n;
            float* c_t = d->c_seq + (long)t * n;
            const dim3 grid(U / 8, (B + 32 * frt - 1) / (32 * frt));
            if (d->rec_masks) {                        // recurrent dropout: one wave per gate, masks applied to the A fragments (lstm_step_masked_kernel)
                const bool two = (U / 16) * ((B + 15) / 16) > kNumCU;              // 32-row blocks once 16-row blocks no longer fit the chip at once (measured: B = 200)
                const dim3 gridm(U / 16, (B + (two ? 31 : 15)) / (two ? 32 : 16));
                if (two) hipLaunchKernelGGL((lstm_step_masked_kernel<2>), gridm, dim3(512), 0, s, z_t, Upk, hp, cp, mk, h_t, c_t, B, U, d->rec_masks);
                else hipLaunchKernelGGL((lstm_step_masked_kernel<1>), gridm, dim3(512), 0, s, z_t, Upk, hp, cp, mk, h_t, c_t, B, U, d->rec_masks);
                int rc = check_launch("lstm_step_masked_kernel");
                if (rc) return rc;
                continue;
            }
#define LAUNCH_FWD_STEP(RT_, NW_) hipLaunchKernelGGL((lstm_step_fused_kernel<RT_, NW_>), grid, dim3(NW_ * 64), 0, s, z_t, Upk, hp, cp, mk, h_t, c_t, B, U)
            if (frt == 1 && fnw == 8) LAUNCH_FWD_STEP(1, 8);
            else if (frt == 1) LAUNCH_FWD_STEP(1, 4);
            else if (fnw == 8) LAUNCH_FWD_STEP(2, 8);
            else LAUNCH_FWD_STEP(2, 4);
#undef LAUNCH_FWD_STEP
            int rc = check_launch("lstm_step_fused_kernel");
            if (rc) return rc;
            continue;
        }
        if (t && d->rec_masks) {
            hipLaunchKernelGGL(lstm_mask_rows_kernel, dim3(std::min(blocks, kNumCU * 8)), dim3(256), 0, s, hp, d->rec_masks, hm, (long)B, B, U);
            int rc = check_launch("lstm_mask_rows_kernel");
            if (rc) return rc;
            for (int gate = 0; gate < 4; ++gate) {           // z_t[:, gate] += (h * m_gate) U_gate
                dc_gemm_desc g{};
                g.M = B; g.N = U; g.K = U;
                g.A = hm + (size_t)gate * n; g.lda = U;
                g.B = d->U_rec + (size_t)gate * U; g.ldb = 4 * U;
                g.C = z_t + (size_t)gate * U; g.ldc = 4 * U;
                g.accumulate = 1;
                rc = dc_gemm_f32(&g, gws, gws_bytes, stream);
                if (rc) return rc;
            }
        } else if (t) {
            dc_gemm_desc g = hU_desc(B, U, hp, d->U_rec, z_t);
            int rc = dc_gemm_f32(&g, workspace, workspace_bytes, stream);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(lstm_gate_fwd_kernel, dim3(blocks), dim3(256), 0, s, z_t, hp, cp, d->mask ? d->mask + (long)t * B : nullptr,
                           d->h_seq + (long)t * n, d->c_seq + (long)t * n, B, U);
        int rc = check_launch("lstm_gate_fwd_kernel");
        if (rc) return rc;
    }
    return DC_OK;
}

extern "C" int dc_lstm_seq_bwd_f32(const dc_lstm_bwd_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(d && d->z && d->U_rec && d->h_seq && d->c_seq && d->dz, DC_EINVAL, "dc_lstm_seq_bwd: null pointer");
    DC_REQUIRE(d->B > 0 && d->T > 0 && d->U > 0 && (d->U & 3) == 0, DC_EINVAL, "dc_lstm_seq_bwd: bad B/T/U (U %% 4 == 0)");
    DC_REQUIRE(workspace && workspace_bytes >= dc_lstm_seq_workspace_bytes(d->B, d->T, d->U), DC_EWORKSPACE,
               "dc_lstm_seq_bwd: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = d->B, U = d->U, T = d->T, n = B * U, blocks = (n + 255) / 256;
    const size_t state_bytes = align_up((size_t)n * sizeof(float));
    char* wsp = static_cast<char*>(workspace);
    float* dh = reinterpret_cast<float*>(wsp);
    float* dc = reinterpret_cast<float*>(wsp + state_bytes);
    void* gws = wsp + 2 * state_bytes;
    size_t gws_bytes = workspace_bytes - 2 * state_bytes;
    float* tmp4 = nullptr;                          // dropout: [4][B][U] per-gate dz_g * U_g^T, then [4][(T-1)B][U] masked h_seq
    float* hm_seq = nullptr;
    if (d->rec_masks) {
        const size_t t4 = align_up((size_t)4 * n * sizeof(float)), hs = align_up((size_t)4 * (size_t)std::max(T - 1, 1) * n * sizeof(float));
        char* end = wsp + workspace_bytes / 256 * 256;
        hm_seq = reinterpret_cast<float*>(end - hs);
        tmp4 = reinterpret_cast<float*>(end - hs - t4);
        gws_bytes = (size_t)(reinterpret_cast<char*>(tmp4) - static_cast<char*>(gws));
    }
    int zrc = zero_fill_async(wsp, 2 * state_bytes, s);
    if (zrc) return zrc;
    const bool fused = (U & 15) == 0;
    int rt_rows = ((U / 16) * ((B + 15) / 16) <= kNumCU) ? 16 : 32;      // measured: 200 x 512 is 1.2x faster with 32-row blocks
    int nw = (U & 31) == 0 ? 8 : 4;
    for (int t = T - 1; t >= 0; --t) {
        const float* z_t = d->z + (long)t * B * 4 * U;
        float* dz_t = d->dz + (long)t * B * 4 * U;
        const float* cp = t ? d->c_seq + (long)(t - 1) * n : nullptr;
        if (fused && t < T - 1) {                               // dh_t = carry + dz_{t+1} U_rec^T + dh_seq[t], gates, in one launch
            const uint8_t* mk = d->mask ? d->mask + (long)t * B : nullptr;
            const float* dho = d->dh_seq ? d->dh_seq + (long)t * n : nullptr;
            const float* dz_next = dz_t + (long)B * 4 * U;
            const dim3 grid(U / 16, (B + rt_rows - 1) / rt_rows);
#define LAUNCH_BWD_STEP(RT_, NW_) hipLaunchKernelGGL((lstm_bwd_step_fused_kernel<RT_, NW_>), grid, dim3(NW_ * 64), 0, s, z_t, cp, mk, dho, dz_next, d->U_rec, dh, dc, dz_t, B, U, d->rec_masks)
            if (rt_rows == 16 && nw == 8) LAUNCH_BWD_STEP(1, 8);
            else if (rt_rows == 16) LAUNCH_BWD_STEP(1, 4);
            else if (nw == 8) LAUNCH_BWD_STEP(2, 8);
            else LAUNCH_BWD_STEP(2, 4);
#undef LAUNCH_BWD_STEP
            int rc = check_launch("lstm_bwd_step_fused_kernel");
            if (rc) return rc;
            continue;
        }
        hipLaunchKernelGGL(lstm_gate_bwd_kernel, dim3(blocks), dim3(256), 0, s, z_t, cp, d->mask ? d->mask + (long)t * B : nullptr,
                           d->dh_seq ? d->dh_seq + (long)t * n : nullptr, (t == T - 1) ? d->dh_last : nullptr, dh, dc, dz_t, B, U);
        int rc = check_launch("lstm_gate_bwd_kernel");
        if (rc) return rc;
        if (t && d->rec_masks && !fused) {                      // dh_{t-1} += sum_g m_g * (dz_g U_g^T)   (fused: the next iteration's kernel does it)
            for (int gate = 0; gate < 4; ++gate) {
                dc_gemm_desc g{};
                g.M = B; g.N = U; g.K = U;
                g.A = dz_t + (size_t)gate * U; g.lda = 4 * U;
                g.B = d->U_rec + (size_t)gate * U; g.ldb = 4 * U; g.b_trans = 1;
                g.C = tmp4 + (size_t)gate * n; g.ldc = U;
                rc = dc_gemm_f32(&g, gws, gws_bytes, stream);
                if (rc) return rc;
            }
            hipLaunchKernelGGL(lstm_masked_acc_kernel, dim3(std::min(blocks, kNumCU * 8)), dim3(256), 0, s, dh, d->rec_masks, tmp4, (long)n);
            rc = check_launch("lstm_masked_acc_kernel");
            if (rc) return rc;
        } else if (t && !fused) {
            dc_gemm_desc g = dzUt_desc(B, U, dz_t, d->U_rec, dh);
            rc = dc_gemm_f32(&g, gws, gws_bytes, stream);
            if (rc) return rc;
        }
    }
    if (!d->dU_rec) return DC_OK;                               // the caller forms dU_rec itself (e.g. on the bf16 pipe from its bf16 copies of h_seq and dz)
    if (T > 1 && d->rec_masks) {                                // dU_g = sum_t (h_{t-1} * m_g)^T dz_{t,g}
        const long rows = (long)(T - 1) * B;
        hipLaunchKernelGGL(lstm_mask_rows_kernel, dim3((int)std::min<long>((rows * U + 255) / 256, (long)kNumCU * 8)), dim3(256), 0, s, d->h_seq,
                           d->rec_masks, hm_seq, rows, B, U);
        int rc = check_launch("lstm_mask_rows_kernel");
        if (rc) return rc;
        for (int gate = 0; gate < 4; ++gate) {
            dc_gemm_desc g{};
            g.M = U; g.N = U; g.K = (int)rows;
            g.A = hm_seq + (size_t)gate * rows * U; g.lda = U; g.a_trans = 1;
            g.B = d->dz + (size_t)B * 4 * U + (size_t)gate * U; g.ldb = 4 * U;
            g.C = d->dU_rec + (size_t)gate * U; g.ldc = 4 * U;
            g.accumulate = d->accumulate_dU;
            rc = dc_gemm_f32(&g, gws, gws_bytes, stream);
            if (rc) return rc;
        }
        return DC_OK;
    }
    if (T > 1) {
        dc_gemm_desc g = dU_desc(B, T, U, d->h_seq, d->dz, d->dU_rec, d->accumulate_dU);
        return dc_gemm_f32(&g, gws, gws_bytes, stream);
    }
    if (!d->accumulate_dU) {
        zrc = zero_fill_async(d->dU_rec, (size_t)U * 4 * U * sizeof(float), s);
        if (zrc) return zrc;
    }
    return DC_OK;
}
