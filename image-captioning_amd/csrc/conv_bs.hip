// conv_bs.hip -- dc_conv2d_nhwc_f32 with math == DC_MATH_BF16X3: the implicit-GEMM convolution on the bf16 matrix pipe
// (operands split into three bf16 pieces on the fly, six products, fp32 accumulate; igemm_bf16s.h).
#include "igemm_bf16s.h"
#include <algorithm>

namespace dcap {

using WeightKCb = DenseKCT<true>;

// Single-role waves, two blocks per CU.  (A producer / consumer version on pre-split weight planes was ~2 % ahead on the large 3x3
// layers only and behind elsewhere -- profiles/r02_presplit_experiment.txt; removed in round 5.)
template <class AL, class BL>
static int dispatch_bs(const AL& al, const BL& bl, const Epilogue& ep, int M, int N, int K, int bm, int bn, int split, void* ws,
                       size_t wsb, hipStream_t s, int pieces) {
    if (pieces == 1) {
        if (bm == 128 && bn == 128) return launch_igemm_bs<128, 128, AL, BL, 1>(al, bl, ep, M, N, K, split, ws, wsb, s);
        if (bm == 128 && bn == 64) return launch_igemm_bs<128, 64, AL, BL, 1>(al, bl, ep, M, N, K, split, ws, wsb, s);
        return launch_igemm_bs<64, 64, AL, BL, 1>(al, bl, ep, M, N, K, split, ws, wsb, s);
    }
    if (pieces == 2) {
        if (bm == 128 && bn == 128) return launch_igemm_bs<128, 128, AL, BL, 2>(al, bl, ep, M, N, K, split, ws, wsb, s);
        if (bm == 128 && bn == 64) return launch_igemm_bs<128, 64, AL, BL, 2>(al, bl, ep, M, N, K, split, ws, wsb, s);
        return launch_igemm_bs<64, 64, AL, BL, 2>(al, bl, ep, M, N, K, split, ws, wsb, s);
    }
    if (bm == 128 && bn == 128) return launch_igemm_bs<128, 128, AL, BL>(al, bl, ep, M, N, K, split, ws, wsb, s);
    if (bm == 128 && bn == 64) return launch_igemm_bs<128, 64, AL, BL>(al, bl, ep, M, N, K, split, ws, wsb, s);
    return launch_igemm_bs<64, 64, AL, BL>(al, bl, ep, M, N, K, split, ws, wsb, s);
}

__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ x, unsigned short* __restrict__ out, size_t n) {
    const size_t pairs = (n + 1) / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (size_t)gridDim.x * 256) {
        const float a = x[2 * i], b = (2 * i + 1 < n) ? x[2 * i + 1] : 0.f;
        unsigned p0, p1, p2;
        split_pair(a, b, p0, p1, p2);
        out[2 * i] = (unsigned short)p0; out[n + 2 * i] = (unsigned short)p1; out[2 * n + 2 * i] = (unsigned short)p2;
        if (2 * i + 1 < n) {
            out[2 * i + 1] = (unsigned short)(p0 >> 16); out[n + 2 * i + 1] = (unsigned short)(p1 >> 16); out[2 * n + 2 * i + 1] = (unsigned short)(p2 >> 16);
        }
    }
}

int conv2d_bf16x3(const dc_conv_desc* d, bool stem, const Epilogue& ep, int M, int N, int K, int bm, int bn, int split, void* workspace,
                  size_t workspace_bytes, hipStream_t s) {
    const int pieces = d->math == DC_MATH_BF16 ? 1 : (d->math == DC_MATH_BF16X2 ? 2 : 3);
    if (stem) {
        WeightKCb bl{d->w, K, N, nullptr};
        StemKC al{d->x, d->H, d->W, d->Ho, d->Wo, M, (unsigned)((size_t)d->N * d->H * d->W * 4 * sizeof(float))};
        return dispatch_bs(al, bl, ep, M, N, K, bm, bn, split, workspace, workspace_bytes, s, pieces);
    }
    ConvWeightKC bl{d->w, K, N, d->kh * d->kw, d->Cin};
    Im2colKCcm al{d->x, d->H, d->W, d->Cin, d->Ho, d->Wo, d->stride, d->pad_t, d->pad_l, d->kw, d->kh * d->kw, M,
                (unsigned)((size_t)d->N * d->H * d->W * d->Cin * sizeof(float))};
    return dispatch_bs(al, bl, ep, M, N, K, bm, bn, split, workspace, workspace_bytes, s, pieces);
}

}  // namespace dcap

extern "C" int dc_split_bf16x3_f32(const float* x, uint16_t* out, size_t n, void* stream) {
    using namespace dcap;
    DC_REQUIRE(x && out && n > 0, DC_EINVAL, "dc_split_bf16x3: bad arguments");
    const int blocks = (int)std::min<size_t>((n / 2 + 255) / 256 + 1, (size_t)kNumCU * 8);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, n);
    return check_launch("split_bf16x3_kernel");
}
