// conv_bs.hip -- dc_conv2d_nhwc_f32 with math == DC_MATH_BF16X3: the implicit-GEMM convolution on the bf16 matrix pipe
// (operands split into three bf16 pieces on the fly, six products, fp32 accumulate; igemm_bf16s.h).
#include "igemm_bf16s.h"

namespace dcap {

using WeightKCb = DenseKCT<true>;

template <class AL, class BL>
static int dispatch_bs(const AL& al, const BL& bl, const Epilogue& ep, int M, int N, int K, int bm, int bn, int split, void* ws,
                       size_t wsb, hipStream_t s) {
    if (bm == 128 && bn == 128) return launch_igemm_bs<128, 128, AL, BL>(al, bl, ep, M, N, K, split, ws, wsb, s);
    if (bm == 128 && bn == 64) return launch_igemm_bs<128, 64, AL, BL>(al, bl, ep, M, N, K, split, ws, wsb, s);
    return launch_igemm_bs<64, 64, AL, BL>(al, bl, ep, M, N, K, split, ws, wsb, s);
}

int conv2d_bf16x3(const dc_conv_desc* d, bool stem, const Epilogue& ep, int M, int N, int K, int bm, int bn, int split, void* workspace,
                  size_t workspace_bytes, hipStream_t s) {
    if (stem) {
        WeightKCb bl{d->w, K, N, nullptr};
        StemKC al{d->x, d->H, d->W, d->Ho, d->Wo, M, (unsigned)((size_t)d->N * d->H * d->W * 4 * sizeof(float))};
        return dispatch_bs(al, bl, ep, M, N, K, bm, bn, split, workspace, workspace_bytes, s);
    }
    ConvWeightKC bl{d->w, K, N, d->kh * d->kw, d->Cin};
    Im2colKC al{d->x, d->H, d->W, d->Cin, d->Ho, d->Wo, d->stride, d->pad_t, d->pad_l, d->kw, d->kh * d->kw, M,
                (unsigned)((size_t)d->N * d->H * d->W * d->Cin * sizeof(float))};
    return dispatch_bs(al, bl, ep, M, N, K, bm, bn, split, workspace, workspace_bytes, s);
}

}  // namespace dcap
