// gemm.hip -- dc_gemm_f32: dense fp32 GEMM on the MFMA main loop of igemm_core.h, with the
// split-K slab reducer and the library's error plumbing.
#include "igemm_core.h"
#include <string.h>

namespace dcap {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int splits, int M, int N, Epilogue ep) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)M * N;
    if (idx >= total) return;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += partial[(long)s * total + idx];   // fixed order: reproducible
    const int row = (int)(idx / N), col = (int)(idx - (long)row * N);
    ep.C[(long)row * ep.ldc + col] = ep.apply(v, row, col);
}

template <class AL, class BL>
static int gemm_dispatch(const AL& al, const BL& bl, const Epilogue& ep, const dc_gemm_desc* d, const TileChoice& t, void* ws,
                         size_t wsb, hipStream_t s) {
    if (t.bm == 128 && t.bn == 128) return launch_igemm<128, 128, AL, BL>(al, bl, ep, d->M, d->N, d->K, t.split, ws, wsb, s);
    if (t.bm == 128 && t.bn == 64) return launch_igemm<128, 64, AL, BL>(al, bl, ep, d->M, d->N, d->K, t.split, ws, wsb, s);
    return launch_igemm<64, 64, AL, BL>(al, bl, ep, d->M, d->N, d->K, t.split, ws, wsb, s);
}

static int gemm_validate(const dc_gemm_desc* d) {
    DC_REQUIRE(d != nullptr, DC_EINVAL, "dc_gemm_f32: null descriptor");
    DC_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, DC_EINVAL, "dc_gemm_f32: M,N,K must be positive (got %d,%d,%d)", d->M, d->N,
               d->K);
    DC_REQUIRE(d->A && d->B && d->C, DC_EINVAL, "dc_gemm_f32: A, B and C must be non-null");
    DC_REQUIRE(d->lda >= (d->a_trans ? d->M : d->K) && d->ldb >= (d->b_trans ? d->K : d->N) && d->ldc >= d->N, DC_EINVAL,
               "dc_gemm_f32: leading dimension smaller than the row length");
    DC_REQUIRE((d->lda & 3) == 0 && (d->ldb & 3) == 0 && aligned16(d->A) && aligned16(d->B), DC_EALIGN,
               "dc_gemm_f32: A/B must be 16-byte aligned with lda, ldb multiples of 4");
    DC_REQUIRE(!d->residual || d->ldr >= d->N, DC_EINVAL, "dc_gemm_f32: ldr smaller than N");
    return DC_OK;
}

}  // namespace dcap

using namespace dcap;

extern "C" int dc_version(void) { return 1; }
extern "C" const char* dc_last_error(void) { return g_err; }

extern "C" size_t dc_gemm_workspace_bytes(const dc_gemm_desc* d) {
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
    const TileChoice t = choose_tile(d->M, d->N, d->K, d->split_k);
    return t.split > 1 ? (size_t)t.split * d->M * d->N * sizeof(float) : 0;
}

extern "C" int dc_gemm_f32(const dc_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = gemm_validate(d);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const TileChoice t = choose_tile(d->M, d->N, d->K, d->split_k);
    Epilogue ep{d->C, d->ldc, d->scale, d->shift, d->residual, d->ldr, d->residual ? (d->res_rows > 0 ? 3 : 1) : 0, d->res_rows, 0, d->relu, d->accumulate};
    if (!d->a_trans && !d->b_trans) {
        return gemm_dispatch(DenseKC{d->A, d->lda, d->M, d->a_gather}, DenseMC{d->B, d->ldb, d->N, nullptr}, ep, d, t, workspace,
                             workspace_bytes, s);
    } else if (!d->a_trans && d->b_trans) {
        return gemm_dispatch(DenseKC{d->A, d->lda, d->M, d->a_gather}, DenseKC{d->B, d->ldb, d->N, nullptr}, ep, d, t,
                             workspace, workspace_bytes, s);
    } else if (d->a_trans && !d->b_trans) {
        return gemm_dispatch(DenseMC{d->A, d->lda, d->M, d->a_gather}, DenseMC{d->B, d->ldb, d->N, nullptr}, ep, d, t, workspace, workspace_bytes, s);
    }
    return gemm_dispatch(DenseMC{d->A, d->lda, d->M, d->a_gather}, DenseKC{d->B, d->ldb, d->N, nullptr}, ep, d, t, workspace, workspace_bytes,
                         s);
}
