// conv_bf16.hip -- conv2d forward (and, on rotated weights, the data gradient) with bf16 STORAGE: bf16 NHWC activations and bf16
// packed weights in HBM, fp32 accumulation, fp32 and / or bf16 output.  BASELINE configs[4] ("bf16").
//
// The implicit GEMM  y[pixel][cout] = sum_k im2col(x)[pixel][k] * w[cout][k],  k = (tap, ci),  runs on the bf16 GEMM main loop
// (bgemm_core.h: 128 x 128 x 64 tiles, LDS-DMA staging, v_mfma_f32_32x32x16_bf16).  A K-tile is 64 channels of ONE tap
// (Cin % 64 == 0), so the A-operand tile is, for each of the 128 output pixels of the block, a contiguous 128-byte run of the
// shifted input pixel: each lane's 16-byte LDS-DMA piece is addressed as  pixel base + tap offset  and taps that fall into the
// padding load hardware zeros (the lane's offset is replaced by an out-of-range one).  No im2col buffer, no staging
// registers, no conversion: the split-bf16 loop this replaces (igemm_bf16s.h with one product) loaded fp32, rounded on the
// way to LDS and synchronised every 32 k -- 250 TFLOP/s on the joint model's layers.
#include "bgemm64_core.h"
#include <algorithm>
#include <cstdlib>

namespace dcap {

struct BConvA {
    const unsigned short* x;       // bf16 [N, H, W, Cin]
    int H, W, Cin, Ho, Wo, stride, pad_t, pad_l, kw, M;      // M = N*Ho*Wo output pixels
    unsigned bytes;
};

struct BLoadConvA {
    static constexpr bool KC = true;
    __amdgpu_buffer_rsrc_t rsrc;
    BConvA c;
    unsigned base[B_NP];           // byte offset of (image n, row 0, col 0, this lane's channel chunk)
    int iy0[B_NP], ix0[B_NP];      // input row / column of tap (0, 0) for this lane's output pixel
    int kpt;                       // K-tiles per tap = Cin / 64
    __device__ __forceinline__ void init(const BConvA& cc, int m0, int lane, int wave) {
        c = cc;
        kpt = c.Cin / BKB;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(c.x), 0, (int)c.bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < B_NP; ++j) {
            const int pc = wave * B_NP + j;
            const int r = 8 * pc + (lane >> 3);                       // tile row (output pixel) of this lane's chunk
            const int ch = (lane & 7) ^ ((r >> 1) & 7);               // source chunk that lands in LDS chunk (lane & 7)
            const int p = min(m0 + r, c.M - 1);                       // pixels past the end feed rows that are never stored
            const int n = p / (c.Ho * c.Wo), rem = p - n * (c.Ho * c.Wo);
            const int oy = rem / c.Wo, ox = rem - oy * c.Wo;
            iy0[j] = oy * c.stride - c.pad_t;
            ix0[j] = ox * c.stride - c.pad_l;
            base[j] = (unsigned)(((long)n * c.H * c.W * c.Cin + 8 * ch) * 2);
        }
    }
    __device__ __forceinline__ void issue(char* img, int k0, int kend, int wave) const {
        const int t = k0 / BKB;                                        // block-uniform: K-tile -> (tap, channel chunk)
        const int tap = t / kpt, c0 = (t - tap * kpt) * BKB;
        const int ky = tap / c.kw, kx = tap - ky * c.kw;
        const bool live = k0 < kend;
#pragma unroll
        for (int j = 0; j < B_NP; ++j) {
            const int iy = iy0[j] + ky, ix = ix0[j] + kx;
            const bool in = live && (unsigned)iy < (unsigned)c.H && (unsigned)ix < (unsigned)c.W;
            const unsigned off = base[j] + (unsigned)(((iy * c.W + ix) * c.Cin + c0) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DC_LDS void*)(img + (wave * B_NP + j) * 1024), 16, (int)(in ? off : kOobOffset), 0, 0, 0);
        }
    }
};

__global__ __launch_bounds__(256, 2) void bconv_kernel(BConvA a, BOperand b, Epilogue ep, int M, int N, int K, int klen,
                                                       float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    char* smem = reinterpret_cast<char*>(smem_f);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_n = (N + BT - 1) / BT;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * BT, n0 = (lid % tiles_n) * BT;
    const int kbeg = blockIdx.z * klen, kend = min(K, kbeg + klen);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    BLoadConvA la;
    BLoadOp<true> lb;
    la.init(a, m0, lane, wave);
    lb.init(b, n0, lane, wave);
    f32x16 acc[2][2];
    bgemm_mainloop_t(la, lb, smem, kbeg, kend, acc, wm, wn);
    store_tile<BT, BT>(acc, smem_f, ep, partial, M, N, m0, n0, wm, wn);
}

// The same A operand for the 256 x 256 tile (bgemm256_core.h): two 128-row sub-images per K-tile, two 1-KiB pieces per wave each.
struct BLoadConvA256 {
    static constexpr bool KC = true;
    __amdgpu_buffer_rsrc_t rsrc;
    BConvA c;
    unsigned base[2][2];           // [half][piece]: byte offset of (image n, row 0, col 0, this lane's channel chunk)
    int iy0[2][2], ix0[2][2];      // input row / column of tap (0, 0) for this lane's output pixel
    int kpt;                       // K-tiles per tap = Cin / 64
    __device__ __forceinline__ void init(const BConvA& cc, int m0, int lane, int wave) {
        c = cc;
        kpt = c.Cin / BKB;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(c.x), 0, (int)c.bytes, 0x00020000);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int rp = 8 * (2 * wave + jj) + (lane >> 3);         // sub-image row of this lane's chunk
                const int ch = (lane & 7) ^ ((rp >> 1) & 7);
                const int p = min(m0 + b256::tile_index<true>(u, rp), c.M - 1);   // pixels past the end feed rows that are never stored
                const int n = p / (c.Ho * c.Wo), rem = p - n * (c.Ho * c.Wo);
                const int oy = rem / c.Wo, ox = rem - oy * c.Wo;
                iy0[u][jj] = oy * c.stride - c.pad_t;
                ix0[u][jj] = ox * c.stride - c.pad_l;
                base[u][jj] = (unsigned)(((long)n * c.H * c.W * c.Cin + 8 * ch) * 2);
            }
    }
    __device__ __forceinline__ void issue(int u, char* sub, int k0, int kend, int wave) const {
        const int t = k0 / BKB;                                        // block-uniform: K-tile -> (tap, channel chunk)
        const int tap = t / kpt, c0 = (t - tap * kpt) * BKB;
        const int ky = tap / c.kw, kx = tap - ky * c.kw;
        const bool live = k0 < kend;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int iy = iy0[u][jj] + ky, ix = ix0[u][jj] + kx;
            const bool in = live && (unsigned)iy < (unsigned)c.H && (unsigned)ix < (unsigned)c.W;
            const unsigned off = base[u][jj] + (unsigned)(((iy * c.W + ix) * c.Cin + c0) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DC_LDS void*)(sub + (2 * wave + jj) * 1024), 16, (int)(in ? off : kOobOffset), 0, 0, 0);
        }
    }
};

__global__ __launch_bounds__(b256::NTHREADS, 2) void bconv256_kernel(BConvA a, BOperand b, Epilogue ep, int M, int N, int K, int klen, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_m = (M + b256::BM - 1) / b256::BM, tiles_n = (N + b256::BN - 1) / b256::BN;
    int tm, tn;
    b256::tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * b256::BM, n0 = tn * b256::BN;
    const int kbeg = blockIdx.z * klen, kend = min(K, kbeg + klen);
    BLoadConvA256 la;
    b256::Load<true, false> lb;
    la.init(a, m0, lane, wave);
    lb.init(b, n0, lane, wave);
    b256::f32x4 acc[8][4];
    b256::mainloop(la, lb, reinterpret_cast<char*>(smem_f), kbeg, kend, acc);
    b256::store_tile(acc, ep, partial, M, N, m0, n0);
}

// ... and for the 64 x 64 tile (bgemm64_core.h): one 64-row image per K-tile, two pieces per wave.
struct BLoadConvA64 {
    static constexpr bool KC = true;
    __amdgpu_buffer_rsrc_t rsrc;
    BConvA c;
    unsigned base[2];
    int iy0[2], ix0[2];
    int ky, kx, c0;                // (tap, channel chunk) of the NEXT K-tile: the main loop issues K-tiles in order, 64 channels at a time
    __device__ __forceinline__ void init(const BConvA& cc, int m0, int lane, int wave) {
        c = cc;
        ky = kx = c0 = 0;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(c.x), 0, (int)c.bytes, 0x00020000);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int rp = 8 * (2 * wave + jj) + (lane >> 3);
            const int ch = (lane & 7) ^ ((rp >> 1) & 7);
            const int p = min(m0 + rp, c.M - 1);
            const int n = p / (c.Ho * c.Wo), rem = p - n * (c.Ho * c.Wo);
            const int oy = rem / c.Wo, ox = rem - oy * c.Wo;
            iy0[jj] = oy * c.stride - c.pad_t;
            ix0[jj] = ox * c.stride - c.pad_l;
            base[jj] = (unsigned)(((long)n * c.H * c.W * c.Cin + 8 * ch) * 2);
        }
    }
    // K-tiles arrive in order starting at k = 0 (bgemm64_core.h: no split-K), so the tap walk is an add and two compares
    __device__ __forceinline__ void issue(char* img, int k0, int kend, int wave) {
        const bool live = k0 < kend;
        const int tapoff = ((ky * c.W + kx) * c.Cin + c0) * 2;             // block-uniform
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int iy = iy0[jj] + ky, ix = ix0[jj] + kx;
            const bool in = live && (unsigned)iy < (unsigned)c.H && (unsigned)ix < (unsigned)c.W;
            const unsigned off = base[jj] + (unsigned)((iy0[jj] * c.W + ix0[jj]) * c.Cin * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DC_LDS void*)(img + (2 * wave + jj) * 1024), 16, (int)(in ? off + (unsigned)tapoff : kOobOffset), 0, 0, 0);
        }
        c0 += BKB;
        if (c0 >= c.Cin) {
            c0 = 0;
            if (++kx >= c.kw) { kx = 0; ++ky; }
        }
    }
};

__global__ __launch_bounds__(b64::NTHREADS, 2) void bconv64_kernel(BConvA a, BOperand b, Epilogue ep, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_n = (N + b64::BN - 1) / b64::BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * b64::BM, n0 = (lid % tiles_n) * b64::BN;
    BLoadConvA64 la;
    b64::Load lb;
    la.init(a, m0, lane, wave);
    lb.init(b, n0, lane, wave);
    b64::f32x4 acc[2][2];
    b64::mainloop(la, lb, reinterpret_cast<char*>(smem_f), 0, K, acc);
    b64::store_tile(acc, ep, M, N, m0, n0);
}

static int conv_bf16_validate(const dc_conv_bf16_desc* d) {
    DC_REQUIRE(d && d->x && d->w && (d->y || d->y_bf16), DC_EINVAL, "dc_conv2d_bf16: x, w and at least one of y / y_bf16 must be non-null");
    DC_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->kh >= 1 && d->kw >= 1 && d->stride >= 1 && d->Cout > 0, DC_EINVAL,
               "dc_conv2d_bf16: bad shape");
    DC_REQUIRE(d->Cin % BKB == 0, DC_EINVAL, "dc_conv2d_bf16: Cin %% 64 == 0 required (a K-tile is 64 channels of one tap), got %d", d->Cin);
    DC_REQUIRE(d->res_mode >= 0 && d->res_mode <= 2 && (d->res_mode == 0) == (d->residual == nullptr), DC_EINVAL, "dc_conv2d_bf16: residual / res_mode mismatch");
    DC_REQUIRE(d->res_mode != 2 || ((d->Ho & 1) == 0 && (d->Wo & 1) == 0), DC_EINVAL, "dc_conv2d_bf16: res_mode 2 needs even Ho, Wo");
    DC_REQUIRE(aligned16(d->x) && aligned16(d->w) && (!d->y || aligned16(d->y)) && (!d->y_bf16 || aligned16(d->y_bf16)), DC_EALIGN,
               "dc_conv2d_bf16: x, w, y, y_bf16 must be 16-byte aligned");
    DC_REQUIRE((size_t)d->N * d->H * d->W * d->Cin * 2 < (size_t)0x7FFFFFF0u && (size_t)d->Cout * d->kh * d->kw * d->Cin * 2 < (size_t)0x7FFFFFF0u, DC_EINVAL,
               "dc_conv2d_bf16: x and w must span < 2 GiB");
    return DC_OK;
}

// split-K for the convolution: two blocks per CU on the chip, >= 4 K-tiles per slice
static BSplit bconv_split(int M, int N, int K, int user_split) {
    constexpr int target = 2 * kNumCU;
    if (user_split > 0) return bgemm_split(M, N, K, user_split);
    const int tiles = ((M + BT - 1) / BT) * ((N + BT - 1) / BT);
    const int ktiles = (K + BKB - 1) / BKB;
    int s = 1;
    if (tiles < target && ktiles >= 8) {
        s = (target + tiles - 1) / tiles;
        s = std::min(s, std::min(ktiles / 4, 32));
        s = std::max(s, 1);
    }
    const int klen = ((ktiles + s - 1) / s) * BKB;
    return BSplit{(K + klen - 1) / klen, klen};
}

}  // namespace dcap

using namespace dcap;

// Which tile runs this layer: 256 (the P2 / P3-level FPN and RPN layers and their data gradients), 64 (the one-image trunk layers:
// whole K loop in one block, no split-K slabs) or 128, by the cost model of bgemm256_core.h / bgemm64_core.h.
// dc_conv_bf16_desc.tile = 64 | 128 | 256 forces a choice where the shape allows it (tests, benches).
static int bconv_tile(const dc_conv_bf16_desc* d, int M, int N, int K) {
    const bool vec4 = (d->Cout & 3) == 0 && (!d->residual || aligned16(d->residual)) && (!d->scale || aligned16(d->scale)) && (!d->shift || aligned16(d->shift));
    const int forced = d->tile;
    if (!vec4 || M < 4 || N < 4) return 128;
    if (forced == 128) return 128;
    if (forced == 64 && d->split_k <= 1) return 64;
    if (forced == 256) return ((long)((M + 255) / 256) * ((N + 255) / 256) * 65536 <= 4L * M * N) ? 256 : 128;     // (not for slivers of a tile)
    const bool can256 = (double)((M + 255) / 256) * ((N + 255) / 256) * 65536.0 <= 1.25 * (double)M * N;
    const double c256 = can256 ? b256::tile_cost_us(M, N, K, 256, b256::split(M, N, K, d->split_k)) : 1e30;
    const double c128 = b256::tile_cost_us(M, N, K, 128, bconv_split(M, N, K, d->split_k));
    const double c64 = d->split_k > 1 ? 1e30 : b64::cost_us(M, N, K);           // (a caller that asks for slabs gets a split-K tile)
    return (c256 <= c128 && c256 <= c64) ? 256 : (c64 < c128 ? 64 : 128);
}
static BSplit bconv_split_for(int tile, int M, int N, int K, int user_split) {
    if (tile == 256) return b256::split(M, N, K, user_split);
    if (tile == 64) return BSplit{1, ((K + BKB - 1) / BKB) * BKB};
    return bconv_split(M, N, K, user_split);
}

extern "C" size_t dc_conv2d_bf16_workspace_bytes(const dc_conv_bf16_desc* d) {
    if (!d || conv_bf16_validate(d)) return 0;
    const int M = d->N * d->Ho * d->Wo, N = d->Cout, K = d->kh * d->kw * d->Cin;
    const BSplit sp = bconv_split_for(bconv_tile(d, M, N, K), M, N, K, d->split_k);
    return sp.split > 1 ? (size_t)sp.split * M * N * sizeof(float) : 0;
}

extern "C" int dc_conv2d_bf16_tile(const dc_conv_bf16_desc* d, int* split_k) {
    if (!d || conv_bf16_validate(d)) return 0;
    const int M = d->N * d->Ho * d->Wo, N = d->Cout, K = d->kh * d->kw * d->Cin;
    const int tile = bconv_tile(d, M, N, K);
    if (split_k) *split_k = bconv_split_for(tile, M, N, K, d->split_k).split;
    return tile;
}

extern "C" int dc_conv2d_bf16(const dc_conv_bf16_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = conv_bf16_validate(d);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int M = d->N * d->Ho * d->Wo, N = d->Cout, K = d->kh * d->kw * d->Cin;
    const int tile = bconv_tile(d, M, N, K);
    const bool big = tile == 256;
    const BSplit sp = bconv_split_for(tile, M, N, K, d->split_k);
    float* partial = nullptr;
    if (sp.split > 1) {
        const size_t need = (size_t)sp.split * M * N * sizeof(float);
        DC_REQUIRE(workspace != nullptr && workspace_bytes >= need, DC_EWORKSPACE, "dc_conv2d_bf16 split-K needs %zu workspace bytes, got %zu", need,
                   workspace_bytes);
        partial = static_cast<float*>(workspace);
    }
    Epilogue ep{d->y, d->Cout, d->scale, d->shift, d->residual, d->Cout, d->res_mode, d->Ho, d->Wo, d->relu, 0, 0};
    ep.vec4 = (d->Cout & 3) == 0 && (!d->residual || aligned16(d->residual)) && (!d->scale || aligned16(d->scale)) && (!d->shift || aligned16(d->shift));
    ep.Cb = d->y_bf16;
    ep.ldcb = d->Cout;
    BConvA a{d->x, d->H, d->W, d->Cin, d->Ho, d->Wo, d->stride, d->pad_t, d->pad_l, d->kw, M, (unsigned)((size_t)d->N * d->H * d->W * d->Cin * 2)};
    BOperand b{d->w, K, N, nullptr, (unsigned)((size_t)N * K * 2)};
    if (tile == 64) {
        DC_ENSURE_DYN_LDS(&bconv64_kernel, 160 * 1024);
        const int tiles = ((M + b64::BM - 1) / b64::BM) * ((N + b64::BN - 1) / b64::BN);
        hipLaunchKernelGGL(bconv64_kernel, dim3(tiles), dim3(b64::NTHREADS), b64::LDS_BYTES, s, a, b, ep, M, N, K);
        return check_launch("bconv64_kernel");
    }
    if (big) {
        DC_ENSURE_DYN_LDS(&bconv256_kernel, 160 * 1024);
        const int tiles = ((M + b256::BM - 1) / b256::BM) * ((N + b256::BN - 1) / b256::BN);
        hipLaunchKernelGGL(bconv256_kernel, dim3(tiles, 1, sp.split), dim3(b256::NTHREADS), b256::LDS_BYTES, s, a, b, ep, M, N, K, sp.klen, partial);
        rc = check_launch("bconv256_kernel");
    } else {
        DC_ENSURE_DYN_LDS(&bconv_kernel, 160 * 1024);
        const int tiles = ((M + BT - 1) / BT) * ((N + BT - 1) / BT);
        hipLaunchKernelGGL(bconv_kernel, dim3(tiles, 1, sp.split), dim3(256), bgemm_lds_bytes(), s, a, b, ep, M, N, K, sp.klen, partial);
        rc = check_launch("bconv_kernel");
    }
    if (rc || sp.split <= 1) return rc;
    const long total = (long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(splitk_reduce_blocks(total)), dim3(256), 0, s, partial, sp.split, M, N, ep);
    return check_launch("splitk_reduce_kernel");
}
