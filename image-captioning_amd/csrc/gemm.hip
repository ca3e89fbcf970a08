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

// Slab reducer + epilogue.  N % 4 == 0 (every conv / GEMM of the path): a thread owns four consecutive columns of one row --
// 16-byte slab loads and output stores (the scalar first version spent 8 us per call on 1-8 MB of slabs; the split-bf16
// convolutions of the joint model call it ~150 times per step).  Slabs are added in a fixed order: reproducible.
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int splits, int M, int N, Epilogue ep) {
    const long total = (long)M * N;
    if ((N & 3) == 0 && (reinterpret_cast<uintptr_t>(partial) & 15u) == 0) {
        const long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
        if (idx >= total) return;
        float4 v = *reinterpret_cast<const float4*>(partial + idx);
        for (int s = 1; s < splits; ++s) {
            const float4 q = *reinterpret_cast<const float4*>(partial + (long)s * total + idx);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        const int row = (int)(idx / N), col = (int)(idx - (long)row * N);
        ep.put(row, col, ep.apply(v.x, row, col));
        ep.put(row, col + 1, ep.apply(v.y, row, col + 1));
        ep.put(row, col + 2, ep.apply(v.z, row, col + 2));
        ep.put(row, col + 3, ep.apply(v.w, row, col + 3));
        return;
    }
    for (int e = 0; e < 4; ++e) {
        const long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4 + e;
        if (idx >= total) return;
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += partial[(long)s * total + idx];
        const int row = (int)(idx / N), col = (int)(idx - (long)row * N);
        ep.put(row, col, ep.apply(v, row, col));
    }
}

template <class AL, class BL>
int gemm_dispatch(const AL& al, const BL& bl, const Epilogue& ep, const dc_gemm_desc* d, const TileChoice& t, void* ws,
                  size_t wsb, hipStream_t s) {
    if (t.bm == 128 && t.bn == 128) return launch_igemm<128, 128, AL, BL>(al, bl, ep, d->M, d->N, d->K, t.split, ws, wsb, s);
    if (t.bm == 128 && t.bn == 64) return launch_igemm<128, 64, AL, BL>(al, bl, ep, d->M, d->N, d->K, t.split, ws, wsb, s);
    return launch_igemm<64, 64, AL, BL, true>(al, bl, ep, d->M, d->N, d->K, t.split, ws, wsb, s);      // producer / consumer waves
}

// ragged shapes (K % 32 != 0, or a K-major operand whose row length is not a multiple of 4): guarded
// loaders, 64x64 tiles only
template <class AL, class BL>
static int gemm_ragged(const AL& al, const BL& bl, const Epilogue& ep, const dc_gemm_desc* d, const TileChoice& t, void* ws,
                       size_t wsb, hipStream_t s) {
    return launch_igemm<64, 64, AL, BL>(al, bl, ep, d->M, d->N, d->K, t.split, ws, wsb, s);
}

// fast TN / TT instantiations live in gemm_tn.hip (parallel compilation)
int gemm_fast_tn(const dc_gemm_desc* d, const Epilogue& ep, const TileChoice& t, void* ws, size_t wsb, hipStream_t s);

static bool gemm_is_fast(const dc_gemm_desc* d) {
    // the fast loaders address with 32-bit byte offsets: operands (and a gathered table) must span < 4 GiB
    const size_t a_span = (size_t)(d->a_trans ? d->K : d->M) * d->lda * sizeof(float);
    const size_t b_span = (size_t)(d->b_trans ? d->N : d->K) * d->ldb * sizeof(float);
    const size_t lim = (size_t)0xFFFFFFF0u;
    return (d->K & 31) == 0 && (!d->a_trans || ((d->M & 3) == 0 && d->M >= 4)) && (d->b_trans || ((d->N & 3) == 0 && d->N >= 4)) &&
           (d->a_gather || a_span < lim) && b_span < lim;
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


extern "C" int dc_version(void) { return DC_ABI_VERSION; }
extern "C" const char* dc_last_error(void) { return g_err; }

// K not a multiple of the 32-deep K-tile (vocabulary 50 000, T*B = 3000 caption rows, 300-d embeddings): the bulk runs on
// the unchecked fast loaders and the last K % 32 columns are accumulated by one launch of the range-checked kernel.
// Only for additive epilogues (bias / residual / accumulate), which commute with the split.
static bool gemm_split_tail(const dc_gemm_desc* d, dc_gemm_desc* bulk, dc_gemm_desc* tail) {
    if ((d->K & 31) == 0 || d->K < 128 || d->relu || d->scale) return false;
    const int K0 = d->K & ~31;
    *bulk = *d;
    bulk->K = K0;
    if (!gemm_is_fast(bulk)) return false;
    *tail = *d;
    tail->K = d->K - K0;
    if (d->a_gather && d->a_trans) tail->a_gather = d->a_gather + K0;
    else tail->A = d->A + (d->a_trans ? (size_t)K0 * d->lda : (size_t)K0);
    tail->B = d->B + (d->b_trans ? (size_t)K0 : (size_t)K0 * d->ldb);
    tail->scale = tail->shift = tail->residual = nullptr;
    tail->res_rows = 0;
    tail->accumulate = 1;
    tail->split_k = 1;
    return true;
}

extern "C" size_t dc_gemm_workspace_bytes(const dc_gemm_desc* d) {
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
    dc_gemm_desc bulk, tail;
    if (gemm_split_tail(d, &bulk, &tail)) d = &bulk;
    const TileChoice t = choose_tile(d->M, d->N, d->K, d->split_k, gemm_is_fast(d));
    return t.split > 1 ? (size_t)t.split * d->M * d->N * sizeof(float) : 0;
}

static int gemm_run(const dc_gemm_desc* d, void* workspace, size_t workspace_bytes, hipStream_t s);

extern "C" int dc_gemm_f32(const dc_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = gemm_validate(d);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dc_gemm_desc bulk, tail;
    if (gemm_split_tail(d, &bulk, &tail)) {
        rc = gemm_run(&bulk, workspace, workspace_bytes, s);
        return rc ? rc : gemm_run(&tail, workspace, workspace_bytes, s);
    }
    return gemm_run(d, workspace, workspace_bytes, s);
}

static int gemm_run(const dc_gemm_desc* d, void* workspace, size_t workspace_bytes, hipStream_t s) {
    const bool fast = gemm_is_fast(d);
    const TileChoice t = choose_tile(d->M, d->N, d->K, d->split_k, fast);
    Epilogue ep{d->C, d->ldc, d->scale, d->shift, d->residual, d->ldr, d->residual ? (d->res_rows > 0 ? 3 : 1) : 0, d->res_rows, 0, d->relu, d->accumulate, 0};
    ep.vec4 = (d->N & 3) == 0 && (d->ldc & 3) == 0 && aligned16(d->C) && (!d->residual || ((d->ldr & 3) == 0 && aligned16(d->residual))) &&
              (!d->scale || aligned16(d->scale)) && (!d->shift || aligned16(d->shift));
    void* ws = workspace;
    const size_t wsb = workspace_bytes;
    if (fast) {
        if (d->a_trans) return gemm_fast_tn(d, ep, t, ws, wsb, s);
        if (!d->b_trans)
            return gemm_dispatch(DenseKCT<true>{d->A, d->lda, d->M, d->a_gather}, DenseMCT<true>{d->B, d->ldb, d->N, nullptr}, ep, d, t, ws, wsb, s);
        return gemm_dispatch(DenseKCT<true>{d->A, d->lda, d->M, d->a_gather}, DenseKCT<true>{d->B, d->ldb, d->N, nullptr}, ep, d, t, ws, wsb, s);
    }
    if (!d->a_trans && !d->b_trans)
        return gemm_ragged(DenseKC{d->A, d->lda, d->M, d->a_gather}, DenseMC{d->B, d->ldb, d->N, nullptr}, ep, d, t, ws, wsb, s);
    if (!d->a_trans && d->b_trans)
        return gemm_ragged(DenseKC{d->A, d->lda, d->M, d->a_gather}, DenseKC{d->B, d->ldb, d->N, nullptr}, ep, d, t, ws, wsb, s);
    if (d->a_trans && !d->b_trans)
        return gemm_ragged(DenseMC{d->A, d->lda, d->M, d->a_gather}, DenseMC{d->B, d->ldb, d->N, nullptr}, ep, d, t, ws, wsb, s);
    return gemm_ragged(DenseMC{d->A, d->lda, d->M, d->a_gather}, DenseKC{d->B, d->ldb, d->N, nullptr}, ep, d, t, ws, wsb, s);
}
