// gemm_tn.hip -- the A-transposed (wgrad: A^T * dY) fast instantiations of dc_gemm_f32, compiled apart
// from gemm.hip so the two translation units build in parallel.
#include "igemm_core.h"

namespace dcap {

template <class AL, class BL>
static int dispatch(const AL& al, const BL& bl, const Epilogue& ep, const dc_gemm_desc* d, const TileChoice& t, void* ws, size_t wsb,
                    hipStream_t s) {
    if (t.bm == 128 && t.bn == 128) return launch_igemm<128, 128, AL, BL>(al, bl, ep, d->M, d->N, d->K, t.split, ws, wsb, s);
    if (t.bm == 128 && t.bn == 64) return launch_igemm<128, 64, AL, BL>(al, bl, ep, d->M, d->N, d->K, t.split, ws, wsb, s);
    return launch_igemm<64, 64, AL, BL, true>(al, bl, ep, d->M, d->N, d->K, t.split, ws, wsb, s);      // producer / consumer waves
}

int gemm_fast_tn(const dc_gemm_desc* d, const Epilogue& ep, const TileChoice& t, void* ws, size_t wsb, hipStream_t s) {
    if (!d->b_trans)
        return dispatch(DenseMCT<true>{d->A, d->lda, d->M, d->a_gather}, DenseMCT<true>{d->B, d->ldb, d->N, nullptr}, ep, d, t, ws, wsb, s);
    return dispatch(DenseMCT<true>{d->A, d->lda, d->M, d->a_gather}, DenseKCT<true>{d->B, d->ldb, d->N, nullptr}, ep, d, t, ws, wsb, s);
}

}  // namespace dcap
