// conv.hip -- dc_conv2d_nhwc_f32: NHWC convolution forward as an implicit GEMM on the fp32 MFMA main
// loop (igemm_core.h): M = N*Ho*Wo output pixels, N = Cout, K = kh*kw*Cin; the A operand is the
// im2col view gathered straight from the activation tensor into LDS tiles (never materialised),
// B is the packed weight [Cout][K].  Frozen BN + bias + residual/upsample-add + ReLU ride in the
// epilogue.  Plus the two bandwidth kernels of the stem: mold_image->RGBX and maxpool 3x3/s2 SAME.
#include "igemm_core.h"
#include <cstdio>
#include <algorithm>

namespace dcap {

static int conv_validate(const dc_conv_desc* d, bool& stem) {
    DC_REQUIRE(d != nullptr, DC_EINVAL, "dc_conv2d: null descriptor");
    DC_REQUIRE(d->x && d->w && d->y, DC_EINVAL, "dc_conv2d: x, w, y must be non-null");
    DC_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cout > 0 && d->Ho > 0 && d->Wo > 0, DC_EINVAL, "dc_conv2d: bad shape");
    stem = (d->Cin == 4);
    if (stem) {
        DC_REQUIRE(d->kh == 7 && d->kw == 7 && d->stride == 2 && d->pad_t == 3 && d->pad_l == 3, DC_EINVAL,
                   "dc_conv2d: Cin==4 is the RGBX stem path (7x7, stride 2, pad 3) only");
    } else {
        DC_REQUIRE(d->Cin % 32 == 0, DC_EINVAL, "dc_conv2d: Cin must be a multiple of 32 (got %d)", d->Cin);
        DC_REQUIRE(d->kh >= 1 && d->kw >= 1 && d->stride >= 1, DC_EINVAL, "dc_conv2d: bad kernel/stride");
    }
    DC_REQUIRE((d->Ho - 1) * d->stride - d->pad_t + d->kh - 1 < d->H + d->kh && (d->Wo - 1) * d->stride - d->pad_l < d->W,
               DC_EINVAL, "dc_conv2d: output size inconsistent with input/stride/pad");
    DC_REQUIRE(aligned16(d->x) && aligned16(d->w), DC_EALIGN, "dc_conv2d: x and w must be 16-byte aligned");
    DC_REQUIRE((size_t)d->N * d->H * d->W * d->Cin * sizeof(float) < (size_t)0x80000000u &&
                   (size_t)d->Cout * d->kh * d->kw * (stem ? 32 / 7.0 * 7 : d->Cin) * sizeof(float) < (size_t)0xFFFFFFFFu,
               DC_EINVAL, "dc_conv2d: x must be < 2 GiB and w < 4 GiB (32-bit offsets)");
    DC_REQUIRE(d->kh <= 8 && d->kw <= 8, DC_EINVAL, "dc_conv2d: kernel size up to 8x8");
    DC_REQUIRE(d->res_mode >= 0 && d->res_mode <= 2 && (d->res_mode == 0 || d->residual), DC_EINVAL,
               "dc_conv2d: res_mode/residual mismatch");
    DC_REQUIRE(d->res_mode != 2 || ((d->Ho & 1) == 0 && (d->Wo & 1) == 0), DC_EINVAL,
               "dc_conv2d: upsample-add needs even output size");
    return DC_OK;
}

// Tile choice of the split-bf16 main loop: its K-tile is ~3x shorter than the fp32 one, so barriers and the split cost
// weigh more on small tiles -- always the widest tile the output allows, and split-K to put two blocks on every CU.
static TileChoice conv_tile_bs(int M, int N, int K, int user_split) {
    auto nb = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    TileChoice t{64, 64, 1};
    if (M >= 128 && N >= 128) t = {128, 128, 1};
    else if (M >= 128 && N >= 64) t = {128, 64, 1};
    if (t.bn == 128 && nb(128, 128) < 2 * kNumCU && nb(128, 64) >= 2 * kNumCU) t = {128, 64, 1};   // fills the chip without slabs
    const int ktiles = (K + BK - 1) / BK;
    const int blocks = ((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn);
    if (user_split > 0) {
        t.split = user_split;
    } else if (blocks < 2 * kNumCU && ktiles >= 8) {
        int s = (2 * kNumCU + blocks - 1) / blocks;
        s = std::min(s, std::min(ktiles / 4, 16));
        t.split = std::max(s, 1);
    }
    return t;
}

static inline bool conv_is_pointwise(const dc_conv_desc* d) {
    return d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_t == 0 && d->pad_l == 0 && d->Ho == d->H && d->Wo == d->W;
}

static inline TileChoice conv_tile(const dc_conv_desc* d, int M, int N, int K) {
    if (d->math != DC_MATH_F32) return conv_tile_bs(M, N, K, d->split_k);
    TileChoice t = choose_tile(M, N, K, d->split_k);
    // Measured on the encoder's layer shapes (tools/conv_bench.py): with 128-row tiles, K >= 128 and N >= 128 the producer /
    // consumer kernel on 128x64 tiles beats the single-role 128x128 and 128x64 kernels (res4_2c 48.5 -> 42.5 us, fpn_p3
    // 293.6 -> 276.8, res3_2b 88.1 -> 82.6): twice the blocks of a 128x128 grid, so a short-K layer no longer runs as one
    // lock-step round, and no wave waits on its own loads (profiles/r02_conv_bench.txt has the round-1 rule beside it).
    if (d->Cin != 4 && t.bm == 128 && t.split == 1 && K >= 128 && N >= 128) {
        t.bn = 64;
        t.pc = true;
    }
    return t;
}

static inline void conv_dims(const dc_conv_desc* d, bool stem, int& M, int& N, int& K) {
    M = d->N * d->Ho * d->Wo;
    N = d->Cout;
    K = stem ? 7 * 32 : d->kh * d->kw * d->Cin;
}

using WeightKC = DenseKCT<true>;   // packed weights: K = kh*kw*Cin is a multiple of 32, rows 16-byte aligned

// producer / consumer kernel or single-role kernel for this tile?  64x64 always, 128x64 where the tile rule asked for it.
static bool conv_uses_pc(const TileChoice& t) {
    if (t.bm == 128 && t.bn == 128) return false;
    if (t.bm == 128 && t.bn == 64) return t.pc;
    return true;
}

static Epilogue conv_epilogue(const dc_conv_desc* d) {
    Epilogue ep{d->y, d->Cout, d->scale, d->shift, d->residual, d->Cout, d->res_mode, d->Ho, d->Wo, d->relu, 0, 0};
    ep.vec4 = (d->Cout & 3) == 0 && aligned16(d->y) && (!d->residual || aligned16(d->residual)) && (!d->scale || aligned16(d->scale)) &&
              (!d->shift || aligned16(d->shift));
    return ep;
}

template <class AL, class BL>
static int conv_dispatch(const AL& al, const BL& bl, const Epilogue& ep, int M, int N, int K, const TileChoice& t, void* ws,
                         size_t wsb, hipStream_t s) {
    const bool pc = conv_uses_pc(t);
    if (t.bm == 128 && t.bn == 128) {
        if (pc) return launch_igemm<128, 128, AL, BL, true>(al, bl, ep, M, N, K, t.split, ws, wsb, s);
        return launch_igemm<128, 128, AL, BL>(al, bl, ep, M, N, K, t.split, ws, wsb, s);
    }
    if (t.bm == 128 && t.bn == 64) {
        if (pc) return launch_igemm<128, 64, AL, BL, true>(al, bl, ep, M, N, K, t.split, ws, wsb, s);
        return launch_igemm<128, 64, AL, BL>(al, bl, ep, M, N, K, t.split, ws, wsb, s);
    }
    if (pc) return launch_igemm<64, 64, AL, BL, true>(al, bl, ep, M, N, K, t.split, ws, wsb, s);
    return launch_igemm<64, 64, AL, BL>(al, bl, ep, M, N, K, t.split, ws, wsb, s);
}

// ---- maxpool 3x3 / stride 2 / TF SAME (pad only where the window leaves the image; padded cells never win)
__global__ void maxpool3x3s2_kernel(const float4* __restrict__ x, float4* __restrict__ y, int N, int H, int W, int C4, int Ho, int Wo,
                                    int pt, int pl) {
    const long total = (long)N * Ho * Wo * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        long p = idx / C4;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - pt + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - pl + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const float4 v = x[(((long)n * H + iy) * W + ix) * C4 + c];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        y[idx] = m;
    }
}

// ---- its backward (training the stem: layers = "all"): gather per input pixel.  The gradient of a window goes to the FIRST
// element (row-major) that equals the window's maximum, as TF's MaxPoolGrad routes it; a pixel sits in up to four windows.
__global__ void maxpool3x3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                                        int N, int H, int W, int C, int Ho, int Wo, int pt, int pl) {
    const long total = (long)N * H * W * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        long p = idx / C;
        const int ix = (int)(p % W);
        p /= W;
        const int iy = (int)(p % H);
        const int n = (int)(p / H);
        const float v = x[idx];
        float g = 0.f;
        for (int oy = max(0, (iy + pt - 1) / 2); oy <= min(Ho - 1, (iy + pt) / 2); ++oy) {           // windows with oy*2 - pt <= iy <= oy*2 - pt + 2
            if (iy < oy * 2 - pt || iy > oy * 2 - pt + 2) continue;
            for (int ox = max(0, (ix + pl - 1) / 2); ox <= min(Wo - 1, (ix + pl) / 2); ++ox) {
                if (ix < ox * 2 - pl || ix > ox * 2 - pl + 2) continue;
                const long o = (((long)n * Ho + oy) * Wo + ox) * C + c;
                if (y[o] != v) continue;
                bool first = true;                                   // is there an equal element earlier in the window?
                for (int ky = 0; ky < 3 && first; ++ky) {
                    const int jy = oy * 2 - pt + ky;
                    if ((unsigned)jy >= (unsigned)H) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int jx = ox * 2 - pl + kx;
                        if ((unsigned)jx >= (unsigned)W) continue;
                        if (jy == iy && jx == ix) { ky = 3; break; }
                        if (x[(((long)n * H + jy) * W + jx) * C + c] == v) { first = false; break; }
                    }
                }
                if (first) g += dy[o];
            }
        }
        dx[idx] = g;
    }
}

// ---- maxpool 2x2 / stride 2 / valid (VGG16's block pools; H and W even)
__global__ void maxpool2x2s2_kernel(const float4* __restrict__ x, float4* __restrict__ y, int N, int H, int W, int C4) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)N * Ho * Wo * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        long p = idx / C4;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const float4* r0 = x + (((long)n * H + 2 * oy) * W + 2 * ox) * C4 + c;
        const float4* r1 = r0 + (long)W * C4;
        const float4 a = r0[0], b = r0[C4], d = r1[0], e = r1[C4];
        y[idx] = make_float4(fmaxf(fmaxf(a.x, b.x), fmaxf(d.x, e.x)), fmaxf(fmaxf(a.y, b.y), fmaxf(d.y, e.y)),
                             fmaxf(fmaxf(a.z, b.z), fmaxf(d.z, e.z)), fmaxf(fmaxf(a.w, b.w), fmaxf(d.w, e.w)));
    }
}

// ---- uint8 RGB -> float RGBX minus MEAN_PIXEL
__global__ void mold_rgbx_kernel(const uint8_t* __restrict__ img, float4* __restrict__ out, long npix, float mr, float mg, float mb) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const uint8_t* p = img + 3 * i;
        out[i] = make_float4((float)p[0] - mr, (float)p[1] - mg, (float)p[2] - mb, 0.f);
    }
}

// ---- the same with the pixel zero-padded to CP channels (CP % 4 == 0): one thread per 16-byte channel quad
__global__ void mold_padded_kernel(const uint8_t* __restrict__ img, float4* __restrict__ out, long npix, int cq, float mr, float mg, float mb) {
    const long total = npix * cq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / cq;
        const int q = (int)(i - pix * cq);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q == 0) {
            const uint8_t* p = img + 3 * pix;
            v = make_float4((float)p[0] - mr, (float)p[1] - mg, (float)p[2] - mb, 0.f);
        }
        out[i] = v;
    }
}

}  // namespace dcap

using namespace dcap;

extern "C" size_t dc_conv2d_workspace_bytes(const dc_conv_desc* d) {
    bool stem;
    if (conv_validate(d, stem)) return 0;
    int M, N, K;
    conv_dims(d, stem, M, N, K);
    if (!stem && conv_winograd_supported(d)) return 0;
    const TileChoice t = conv_tile(d, M, N, K);
    return t.split > 1 ? (size_t)t.split * M * N * sizeof(float) : 0;
}

extern "C" int dc_conv2d_tile_config(const dc_conv_desc* d, int* bm, int* bn, int* split_k) {
    bool stem;
    int rc = conv_validate(d, stem);
    if (rc) return rc;
    int M, N, K;
    conv_dims(d, stem, M, N, K);
    const TileChoice t = conv_tile(d, M, N, K);
    if (bm) *bm = t.bm;
    if (bn) *bn = t.bn;
    if (split_k) *split_k = t.split;
    return DC_OK;
}

extern "C" int dc_conv2d_kernel_name(const dc_conv_desc* d, char* buf, size_t buf_bytes) {
    bool stem;
    int rc = conv_validate(d, stem);
    if (rc) return rc;
    DC_REQUIRE(buf && buf_bytes >= 96, DC_EINVAL, "dc_conv2d_kernel_name: need a buffer of >= 96 bytes");
    int M, N, K;
    conv_dims(d, stem, M, N, K);
    const TileChoice t = conv_tile(d, M, N, K);
    if (d->math != DC_MATH_F32) {
        snprintf(buf, buf_bytes, "igemm_bs_kernel<%d, %d>", t.bm, t.bn);
        return DC_OK;
    }
    if (!stem && conv_winograd_supported(d)) {
        snprintf(buf, buf_bytes, "wino%d%s_kernel", conv_winograd_tiles(d), conv_winograd_split_bf16(d) ? "b" : "");
        return DC_OK;
    }
    const bool pw = !stem && conv_is_pointwise(d);
    if (pw && t.split == 1 && conv_pw_stream_supported(d, conv_epilogue(d))) {
        snprintf(buf, buf_bytes, "pwconv_stream_kernel<%d, %d>", d->Cin, d->res_mode);
        return DC_OK;
    }
    snprintf(buf, buf_bytes, "%s<%d, %d, dcap::%s, dcap::DenseKCT<true> >", conv_uses_pc(t) ? "igemm_pc_kernel" : "igemm_kernel", t.bm, t.bn,
             stem ? "StemKC" : (pw ? "DenseKCT<true>" : "Im2colKCT<false>"));
    return DC_OK;
}

extern "C" int dc_conv2d_is_pointwise(const dc_conv_desc* d) { return d && d->Cin != 4 && d->math == DC_MATH_F32 && conv_is_pointwise(d) ? 1 : 0; }

extern "C" int dc_conv2d_nhwc_f32(const dc_conv_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    bool stem;
    int rc = conv_validate(d, stem);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int M, N, K;
    conv_dims(d, stem, M, N, K);
    const TileChoice t = conv_tile(d, M, N, K);
    const Epilogue ep = conv_epilogue(d);
    DC_REQUIRE(d->math == DC_MATH_F32 || d->math == DC_MATH_BF16X3 || d->math == DC_MATH_BF16X2 || d->math == DC_MATH_BF16, DC_EINVAL,
               "dc_conv2d: unknown math mode %d", d->math);
    if (d->math != DC_MATH_F32) return conv2d_bf16x3(d, stem, ep, M, N, K, t.bm, t.bn, t.split, workspace, workspace_bytes, s);
    if (!stem && conv_winograd_supported(d)) return conv2d_winograd(d, s);
    WeightKC bl{d->w, K, N, nullptr};
    if (stem) {
        StemKC al{d->x, d->H, d->W, d->Ho, d->Wo, M, (unsigned)((size_t)d->N * d->H * d->W * 4 * sizeof(float))};
        return conv_dispatch(al, bl, ep, M, N, K, t, workspace, workspace_bytes, s);
    }
    if (conv_is_pointwise(d)) {
        // 1x1 / stride 1 / no padding: the im2col matrix IS the activation tensor [pixels][Cin] -- the dense K-contiguous loader
        // (no per-row pixel decode, no tap masks at block start: on the short-K layers, K = 64..256, that set-up was a
        // visible share of a block's life)
        if (t.split == 1 && conv_pw_stream_supported(d, ep)) return conv2d_pointwise_stream(d, ep, M, N, s);
        DenseKCT<true> al{d->x, d->Cin, M, nullptr};
        return conv_dispatch(al, bl, ep, M, N, K, t, workspace, workspace_bytes, s);
    }
    Im2colKC al{d->x, d->H, d->W, d->Cin, d->Ho, d->Wo, d->stride, d->pad_t, d->pad_l, d->kw, d->kh * d->kw, M,
                (unsigned)((size_t)d->N * d->H * d->W * d->Cin * sizeof(float))};
    return conv_dispatch(al, bl, ep, M, N, K, t, workspace, workspace_bytes, s);
}

static int wgrad_validate(const dc_conv_desc* d) {
    DC_REQUIRE(d && d->x && d->w && d->y, DC_EINVAL, "dc_conv2d_wgrad: x, dy (y) and dw (w) must be non-null");
    DC_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->kh >= 1 && d->kw >= 1 && d->kh <= 8 && d->kw <= 8 &&
                   d->stride >= 1,
               DC_EINVAL, "dc_conv2d_wgrad: bad shape");
    DC_REQUIRE(d->Cin % 64 == 0 && d->Cout % 4 == 0, DC_EINVAL, "dc_conv2d_wgrad: needs Cin %% 64 == 0 and Cout %% 4 == 0 (pad the head)");
    DC_REQUIRE(aligned16(d->x) && aligned16(d->y) && aligned16(d->w), DC_EALIGN, "dc_conv2d_wgrad: pointers must be 16-byte aligned");
    DC_REQUIRE((size_t)d->N * d->H * d->W * d->Cin * sizeof(float) < (size_t)0x80000000u &&
                   (size_t)d->N * d->Ho * d->Wo * d->Cout * sizeof(float) < (size_t)0xFFFFFFF0u,
               DC_EINVAL, "dc_conv2d_wgrad: x must be < 2 GiB and dy < 4 GiB");
    return DC_OK;
}

static TileChoice wgrad_tile(const dc_conv_desc* d) {
    const int M = d->Cout, N = d->kh * d->kw * d->Cin, K = d->N * d->Ho * d->Wo;
    // a column tile must stay inside one (ky,kx) tap: 128-wide tiles need Cin % 128 == 0; the range-checked A loader
    // (pixel count not a multiple of 32) exists for 64x64 only
    const bool big = d->Cin % 128 == 0 && d->Cout >= 128 && K % 32 == 0;
    return choose_tile(M, N, K, d->split_k, big);
}

extern "C" size_t dc_conv2d_wgrad_workspace_bytes(const dc_conv_desc* d) {
    if (wgrad_validate(d)) return 0;
    const TileChoice t = wgrad_tile(d);
    return t.split > 1 ? (size_t)t.split * d->Cout * d->kh * d->kw * d->Cin * sizeof(float) : 0;
}

extern "C" int dc_conv2d_wgrad_f32(const dc_conv_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = wgrad_validate(d);
    if (rc) return rc;
    const int M = d->Cout, N = d->kh * d->kw * d->Cin, K = d->N * d->Ho * d->Wo;
    const TileChoice t = wgrad_tile(d);
    Epilogue ep{const_cast<float*>(d->w), N, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, d->accumulate, 1};
    Im2colMC bl{d->x, d->H, d->W, d->Cin, d->Ho, d->Wo, d->stride, d->pad_t, d->pad_l, d->kw, K,
                (unsigned)((size_t)d->N * d->H * d->W * d->Cin * sizeof(float))};
    if (K % 32 == 0) {
        DenseMCT<true> al{d->y, d->Cout, M, nullptr};               // A^T: dy is [pixels][Cout]
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (t.bm == 128 && t.bn == 128)
            return launch_igemm<128, 128, DenseMCT<true>, Im2colMC>(al, bl, ep, M, N, K, t.split, workspace, workspace_bytes, s);
        if (t.bm == 128 && t.bn == 64)
            return launch_igemm<128, 64, DenseMCT<true>, Im2colMC>(al, bl, ep, M, N, K, t.split, workspace, workspace_bytes, s);
        return launch_igemm<64, 64, DenseMCT<true>, Im2colMC, true>(al, bl, ep, M, N, K, t.split, workspace, workspace_bytes, s);
    }
    DenseMCT<false> al{d->y, d->Cout, M, nullptr};                  // pixel count not a multiple of the K-tile (tiny pyramid levels)
    return launch_igemm<64, 64, DenseMCT<false>, Im2colMC>(al, bl, ep, M, N, K, t.split, workspace, workspace_bytes,
                                                          static_cast<hipStream_t>(stream));
}

// packed forward weights [Cout][kh*kw*Cin] -> data-gradient weights [Cin][kh*kw*Cout], taps rotated by 180 degrees
__global__ void dgrad_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int kh, int kw, int Cin) {
    const long total = (long)Cout * kh * kw * Cin;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % Cout);
        long r = idx / Cout;
        const int tap = (int)(r % (kh * kw));
        const int ci = (int)(r / (kh * kw));
        const int ky = tap / kw, kx = tap - ky * kw;
        out[idx] = w[((long)co * kh * kw + (kh - 1 - ky) * kw + (kw - 1 - kx)) * Cin + ci];
    }
}

__global__ void scatter2_add_kernel(const float4* __restrict__ coarse, float4* __restrict__ fine, int N, int Hc, int Wc, int C4) {
    const long total = (long)N * Hc * Wc * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        long p = idx / C4;
        const int x = (int)(p % Wc);
        p /= Wc;
        const int y = (int)(p % Hc), n = (int)(p / Hc);
        const long o = (((long)n * 2 * Hc + 2 * y) * 2 * Wc + 2 * x) * C4 + c;
        float4 f = fine[o];
        const float4 q = coarse[idx];
        f.x += q.x; f.y += q.y; f.z += q.z; f.w += q.w;
        fine[o] = f;
    }
}

extern "C" int dc_conv_weight_dgrad_pack_f32(const float* w, float* out, int Cout, int kh, int kw, int Cin, void* stream) {
    DC_REQUIRE(w && out && Cout > 0 && kh > 0 && kw > 0 && Cin > 0, DC_EINVAL, "dc_conv_weight_dgrad_pack: bad arguments");
    const long total = (long)Cout * kh * kw * Cin;
    const int blocks = (int)std::min<long>((total + 255) / 256, (long)kNumCU * 8);
    hipLaunchKernelGGL(dgrad_pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w, out, Cout, kh, kw, Cin);
    return check_launch("dgrad_pack_kernel");
}

extern "C" int dc_scatter2_add_f32(const float* coarse, float* fine, int N, int Hc, int Wc, int C, void* stream) {
    DC_REQUIRE(coarse && fine && N > 0 && Hc > 0 && Wc > 0 && C > 0 && (C & 3) == 0, DC_EINVAL, "dc_scatter2_add: bad arguments");
    DC_REQUIRE(aligned16(coarse) && aligned16(fine), DC_EALIGN, "dc_scatter2_add: pointers must be 16-byte aligned");
    const long total = (long)N * Hc * Wc * (C / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, (long)kNumCU * 8);
    hipLaunchKernelGGL(scatter2_add_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(coarse), reinterpret_cast<float4*>(fine), N, Hc, Wc, C / 4);
    return check_launch("scatter2_add_kernel");
}

__global__ void downsample2x_sum_kernel(const float4* __restrict__ fine, float4* __restrict__ out, int N, int Ho, int Wo, int C4, int acc,
                                        unsigned short* __restrict__ out_bf16) {
    const long total = (long)N * Ho * Wo * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        long p = idx / C4;
        const int x = (int)(p % Wo);
        p /= Wo;
        const int y = (int)(p % Ho), n = (int)(p / Ho);
        const long base = (((long)n * 2 * Ho + 2 * y) * 2 * Wo + 2 * x) * C4 + c;
        const float4 a = fine[base], b = fine[base + C4], e = fine[base + (long)2 * Wo * C4], f = fine[base + (long)2 * Wo * C4 + C4];
        float4 o = make_float4((a.x + b.x) + (e.x + f.x), (a.y + b.y) + (e.y + f.y), (a.z + b.z) + (e.z + f.z), (a.w + b.w) + (e.w + f.w));
        if (acc) { const float4 q = out[idx]; o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
        out[idx] = o;
        if (out_bf16) {                                    // the bf16 copy the lateral weight gradient reads (instead of a cast pass)
            typedef unsigned short us4 __attribute__((ext_vector_type(4)));
            auto bits = [](float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); };
            reinterpret_cast<us4*>(out_bf16)[idx] = us4{bits(o.x), bits(o.y), bits(o.z), bits(o.w)};
        }
    }
}

extern "C" int dc_downsample2x_sum_dual_f32(const float* fine, float* out, uint16_t* out_bf16, int N, int Ho, int Wo, int C, int accumulate,
                                            void* stream) {
    DC_REQUIRE(fine && out && N > 0 && Ho > 0 && Wo > 0 && C > 0 && (C & 3) == 0, DC_EINVAL, "dc_downsample2x_sum: bad arguments");
    DC_REQUIRE(aligned16(fine) && aligned16(out) && (reinterpret_cast<uintptr_t>(out_bf16) & 7u) == 0, DC_EALIGN,
               "dc_downsample2x_sum: pointers must be 16-byte aligned (out_bf16: 8)");
    const long total = (long)N * Ho * Wo * (C / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, (long)kNumCU * 8);
    hipLaunchKernelGGL(downsample2x_sum_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(fine), reinterpret_cast<float4*>(out), N, Ho, Wo, C / 4, accumulate, out_bf16);
    return check_launch("downsample2x_sum_kernel");
}

extern "C" int dc_downsample2x_sum_f32(const float* fine, float* out, int N, int Ho, int Wo, int C, int accumulate, void* stream) {
    return dc_downsample2x_sum_dual_f32(fine, out, nullptr, N, Ho, Wo, C, accumulate, stream);
}

extern "C" int dc_maxpool3x3s2_same_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    DC_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, DC_EINVAL, "dc_maxpool3x3s2: bad arguments");
    DC_REQUIRE(aligned16(x) && aligned16(y), DC_EALIGN, "dc_maxpool3x3s2: pointers must be 16-byte aligned");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int pad_h = std::max((Ho - 1) * 2 + 3 - H, 0), pad_w = std::max((Wo - 1) * 2 + 3 - W, 0);
    const long total = (long)N * Ho * Wo * (C / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, (long)kNumCU * 8);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), N, H, W, C / 4, Ho, Wo, pad_h / 2, pad_w / 2);
    return check_launch("maxpool3x3s2_kernel");
}

extern "C" int dc_maxpool2x2s2_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    DC_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && (H & 1) == 0 && (W & 1) == 0, DC_EINVAL,
               "dc_maxpool2x2s2: bad arguments (C %% 4 == 0, even H and W)");
    DC_REQUIRE(aligned16(x) && aligned16(y), DC_EALIGN, "dc_maxpool2x2s2: pointers must be 16-byte aligned");
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, (long)kNumCU * 16);
    hipLaunchKernelGGL(maxpool2x2s2_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), N, H, W, C / 4);
    return check_launch("maxpool2x2s2_kernel");
}

extern "C" int dc_mold_image_rgbx_f32(const uint8_t* img, float* out, int N, int H, int W, float mean_r, float mean_g, float mean_b,
                                      void* stream) {
    DC_REQUIRE(img && out && N > 0 && H > 0 && W > 0, DC_EINVAL, "dc_mold_image: bad arguments");
    DC_REQUIRE(aligned16(out), DC_EALIGN, "dc_mold_image: out must be 16-byte aligned");
    const long npix = (long)N * H * W;
    const int blocks = (int)std::min<long>((npix + 255) / 256, (long)kNumCU * 8);
    hipLaunchKernelGGL(mold_rgbx_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), img,
                       reinterpret_cast<float4*>(out), npix, mean_r, mean_g, mean_b);
    return check_launch("mold_rgbx_kernel");
}

extern "C" int dc_mold_image_padded_f32(const uint8_t* img, float* out, int N, int H, int W, int channels, float mean_r, float mean_g, float mean_b,
                                        void* stream) {
    DC_REQUIRE(img && out && N > 0 && H > 0 && W > 0 && channels >= 4 && (channels & 3) == 0, DC_EINVAL, "dc_mold_image_padded: bad arguments (channels % 4 == 0)");
    DC_REQUIRE(aligned16(out), DC_EALIGN, "dc_mold_image_padded: out must be 16-byte aligned");
    const long npix = (long)N * H * W, total = npix * (channels / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, (long)kNumCU * 8);
    hipLaunchKernelGGL(mold_padded_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), img, reinterpret_cast<float4*>(out), npix,
                       channels / 4, mean_r, mean_g, mean_b);
    return check_launch("mold_padded_kernel");
}

extern "C" int dc_maxpool3x3s2_same_bwd_f32(const float* x, const float* y, const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    DC_REQUIRE(x && y && dy && dx && N > 0 && H > 0 && W > 0 && C > 0, DC_EINVAL, "dc_maxpool3x3s2_same_bwd: bad arguments");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int pad_h = std::max((Ho - 1) * 2 + 3 - H, 0), pad_w = std::max((Wo - 1) * 2 + 3 - W, 0);
    const long total = (long)N * H * W * C;
    const int blocks = (int)std::min<long>((total + 255) / 256, (long)kNumCU * 16);
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, dy, dx, N, H, W, C, Ho, Wo, pad_h / 2, pad_w / 2);
    return check_launch("maxpool3x3s2_bwd_kernel");
}
