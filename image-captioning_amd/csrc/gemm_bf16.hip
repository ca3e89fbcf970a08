// gemm_bf16.hip -- dc_gemm_bf16: dense GEMM with bf16 operands in memory and fp32 accumulation on
// v_mfma_f32_32x32x16_bf16 (bgemm_core.h), all three layouts of a training step (NN / NT / TN), the fused epilogue of
// dc_gemm_f32 plus an optional bf16 copy of the output; and the fp32 -> bf16 cast that feeds it.
#include "bgemm256_core.h"
#include <algorithm>
#include <cstdlib>

namespace dcap {

// dW tile [128 cout][128 (tap, ci)] over the pixel range of this split-K slice
__global__ __launch_bounds__(256, 2) void bgemm_wgrad_kernel(BOperand dy, BIm2col xc, Epilogue ep, int M, int N, int K, int klen,
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
    BLoadOp<false> la;
    BLoadIm2col lb;
    la.init(dy, m0, lane, wave);
    lb.init(xc, n0, kbeg, lane, wave);
    f32x16 acc[2][2];
    bgemm_mainloop_t(la, lb, smem, kbeg, kend, acc, wm, wn);
    store_tile<BT, BT>(acc, smem_f, ep, partial, M, N, m0, n0, wm, wn);
}

// The same weight gradient on the 256 x 256 tile (bgemm256_core.h): the K-major im2col operand as two 128-column sub-images.
// A 256-column tile may span several taps (Cin = 128: two), so the tap offset is per lane (fixed over the K loop: a lane's columns
// never change); the pixel walk advances once per K-tile, after the second half has been issued (the main loop always issues
// half 0, then half 1 of a K-tile).
struct BLoadIm2col256 {
    static constexpr bool KC = false;
    __amdgpu_buffer_rsrc_t rsrc;
    BIm2col c;
    int n[2], oy[2], ox[2];        // [piece]: output pixel of this lane's K row in the NEXT K-tile to issue
    int dy[2][2], dx[2][2];        // [half][piece]: tap offset of this lane's 8-column chunk
    unsigned cio[2][2];            // byte offset of its first channel inside the pixel
    int krow[2];
    __device__ __forceinline__ void init(const BIm2col& cc, int col0, int kbeg, int lane, int wave) {
        c = cc;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(c.x), 0, (int)c.bytes, 0x00020000);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int k = 4 * (2 * wave + jj) + (lane >> 4);               // K row inside the tile
            krow[jj] = k;
            const int p = kbeg + k;
            const int nn = p / (c.Ho * c.Wo), rem = p - nn * (c.Ho * c.Wo);
            n[jj] = nn;
            oy[jj] = rem / c.Wo;
            ox[jj] = rem - oy[jj] * c.Wo;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ch = (lane & 15) ^ (((k & 3) << 2) | ((k >> 2) & 3));
                const int col = min(col0 + b256::tile_index<false>(u, 8 * ch), c.ncols - 8);   // columns past the edge: clamped, never stored
                const int tap = col / c.Cin, ci = col - tap * c.Cin;
                const int ky = tap / c.kw, kx = tap - ky * c.kw;
                dy[u][jj] = ky - c.pad_t;
                dx[u][jj] = kx - c.pad_l;
                cio[u][jj] = (unsigned)(ci * 2);
            }
        }
    }
    __device__ __forceinline__ void issue(int u, char* sub, int k0, int kend, int wave) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int iy = oy[jj] * c.stride + dy[u][jj], ix = ox[jj] * c.stride + dx[u][jj];
            const bool in = (unsigned)iy < (unsigned)c.H && (unsigned)ix < (unsigned)c.W && k0 + krow[jj] < min(c.P, kend);
            const unsigned off = (unsigned)((((long)n[jj] * c.H + iy) * c.W + ix) * c.Cin * 2) + cio[u][jj];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DC_LDS void*)(sub + (2 * wave + jj) * 1024), 16, (int)(in ? off : kOobOffset), 0, 0, 0);
            if (u == 1) {                                                       // both halves of this K-tile are out: 64 pixels on
                ox[jj] += b256::BK;
                while (ox[jj] >= c.Wo) { ox[jj] -= c.Wo; ++oy[jj]; }
                while (oy[jj] >= c.Ho) { oy[jj] -= c.Ho; ++n[jj]; }
            }
        }
    }
};

__global__ __launch_bounds__(b256::NTHREADS, 2) void bgemm256_wgrad_kernel(BOperand dy, BIm2col xc, Epilogue ep, int M, int N, int K, int klen, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_m = (M + b256::BM - 1) / b256::BM, tiles_n = (N + b256::BN - 1) / b256::BN;
    int tm, tn;
    b256::tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * b256::BM, n0 = tn * b256::BN;
    const int kbeg = blockIdx.z * klen, kend = min(K, kbeg + klen);
    b256::Load<false, true> la;
    BLoadIm2col256 lb;
    la.init(dy, m0, lane, wave);
    lb.init(xc, n0, kbeg, lane, wave);
    b256::f32x4 acc[8][4];
    b256::mainloop(la, lb, reinterpret_cast<char*>(smem_f), kbeg, kend, acc);
    b256::store_tile(acc, ep, partial, M, N, m0, n0);
}

static int bgemm_validate(const dc_gemm_bf16_desc* d) {
    DC_REQUIRE(d != nullptr, DC_EINVAL, "dc_gemm_bf16: null descriptor");
    DC_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, DC_EINVAL, "dc_gemm_bf16: M,N,K must be positive (got %d,%d,%d)", d->M, d->N, d->K);
    DC_REQUIRE(d->A && d->B && (d->C || d->Cb), DC_EINVAL, "dc_gemm_bf16: A, B and one of C / Cb must be non-null");
    DC_REQUIRE(d->lda >= (d->a_trans ? d->M : d->K) && d->ldb >= (d->b_trans ? d->K : d->N), DC_EINVAL,
               "dc_gemm_bf16: leading dimension smaller than the row length");
    DC_REQUIRE((!d->C || d->ldc >= d->N) && (!d->Cb || d->ldcb >= d->N), DC_EINVAL, "dc_gemm_bf16: ldc / ldcb smaller than N");
    DC_REQUIRE((d->K & 7) == 0 && (d->lda & 7) == 0 && (d->ldb & 7) == 0 && aligned16(d->A) && aligned16(d->B), DC_EALIGN,
               "dc_gemm_bf16: K, lda, ldb must be multiples of 8 (16-byte chunks) and A, B 16-byte aligned");
    DC_REQUIRE((!d->a_trans || ((d->M & 7) == 0)) && (d->b_trans || ((d->N & 7) == 0)), DC_EALIGN,
               "dc_gemm_bf16: a K-major operand needs its row length (M for A^T, N for B) to be a multiple of 8");
    DC_REQUIRE(!d->accumulate || d->C, DC_EINVAL, "dc_gemm_bf16: accumulate needs the fp32 output C");
    DC_REQUIRE(!d->residual || d->ldr >= d->N, DC_EINVAL, "dc_gemm_bf16: ldr smaller than N");
    const size_t lim = (size_t)0x7FFFFFF0u;              // 32-bit buffer offsets, with the out-of-range marker above every operand
    const size_t a_span = (size_t)(d->a_trans ? d->K : d->M) * d->lda * 2, b_span = (size_t)(d->b_trans ? d->N : d->K) * d->ldb * 2;
    DC_REQUIRE((d->a_gather || a_span < lim) && b_span < lim, DC_EINVAL, "dc_gemm_bf16: operands must span < 2 GiB");
    DC_REQUIRE(!d->a_gather || d->a_gather_rows > 0, DC_EINVAL, "dc_gemm_bf16: a_gather needs a_gather_rows (rows of the gathered table)");
    DC_REQUIRE(!d->a_gather || (size_t)d->a_gather_rows * d->lda * 2 < lim, DC_EINVAL, "dc_gemm_bf16: gathered table must span < 2 GiB");
    return DC_OK;
}

static Epilogue bgemm_epilogue(const dc_gemm_bf16_desc* d) {
    Epilogue ep{d->C, d->ldc, d->scale, d->shift, d->residual, d->ldr, d->residual ? (d->res_rows > 0 ? 3 : 1) : 0, d->res_rows, 0, d->relu,
                d->accumulate, 0};
    ep.Cb = d->Cb;
    ep.ldcb = d->ldcb;
    ep.vec4 = (d->N & 3) == 0 && (!d->C || ((d->ldc & 3) == 0 && aligned16(d->C))) && (!d->Cb || ((d->ldcb & 3) == 0 && (reinterpret_cast<uintptr_t>(d->Cb) & 7u) == 0)) &&
              (!d->residual || ((d->ldr & 3) == 0 && aligned16(d->residual))) && (!d->scale || aligned16(d->scale)) && (!d->shift || aligned16(d->shift));
    return ep;
}

static void bgemm_operands(const dc_gemm_bf16_desc* d, BOperand& a, BOperand& b) {
    const size_t a_rows = d->a_gather ? (size_t)d->a_gather_rows : (size_t)(d->a_trans ? d->K : d->M);
    a = BOperand{d->A, d->lda, d->M, d->a_gather, (unsigned)(a_rows * d->lda * 2)};
    b = BOperand{d->B, d->ldb, d->N, nullptr, (unsigned)((size_t)(d->b_trans ? d->N : d->K) * d->ldb * 2)};
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ out, size_t n) {
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        reinterpret_cast<us4*>(out)[i] = us4{Epilogue::bf16_bits(v.x), Epilogue::bf16_bits(v.y), Epilogue::bf16_bits(v.z), Epilogue::bf16_bits(v.w)};
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = Epilogue::bf16_bits(x[i]);
}

// bf16 -> fp32 (exact): the gradient bucket after a bf16 all-reduce, back where the clip norm and the optimizer read it
__global__ __launch_bounds__(256) void uncast_bf16_kernel(const unsigned short* __restrict__ x, float* __restrict__ out, size_t n) {
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    auto f = [](unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); };
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const us4 v = reinterpret_cast<const us4*>(x)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(f(v[0]), f(v[1]), f(v[2]), f(v[3]));
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = f(x[i]);
}

// rows x cols fp32 (row stride ld_in) -> bf16 (row stride ld_out), columns cols..cols_out-1 zero-filled (K padding)
__global__ __launch_bounds__(256) void cast_bf16_2d_kernel(const float* __restrict__ x, long ld_in, unsigned short* __restrict__ out, long ld_out,
                                                           int rows, int cols, int cols_out) {
    const long total = (long)rows * cols_out;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int r = (int)(idx / cols_out), c = (int)(idx - (long)r * cols_out);
        out[(long)r * ld_out + c] = c < cols ? Epilogue::bf16_bits(x[(long)r * ld_in + c]) : (unsigned short)0;
    }
}

}  // namespace dcap

using namespace dcap;

// the 256-square kernel (bgemm256_core.h) where its grid fills the chip; the K-major gather (embedding-side weight gradient) stays
// on the 128-square loop
static bool use_b256(const dc_gemm_bf16_desc* d) { return !(d->a_gather && d->a_trans) && b256::prefer(d->M, d->N, d->K, d->split_k, bgemm_epilogue(d).vec4 != 0); }

extern "C" size_t dc_gemm_bf16_workspace_bytes(const dc_gemm_bf16_desc* d) {
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
    const BSplit sp = use_b256(d) ? b256::split(d->M, d->N, d->K, d->split_k) : bgemm_split(d->M, d->N, d->K, d->split_k);
    return sp.split > 1 ? (size_t)sp.split * d->M * d->N * sizeof(float) : 0;
}

extern "C" int dc_gemm_bf16_tile(const dc_gemm_bf16_desc* d, int* split_k) {
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
    const bool big = use_b256(d);
    if (split_k) *split_k = (big ? b256::split(d->M, d->N, d->K, d->split_k) : bgemm_split(d->M, d->N, d->K, d->split_k)).split;
    return big ? 256 : 128;
}

extern "C" int dc_gemm_bf16(const dc_gemm_bf16_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = bgemm_validate(d);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Epilogue ep = bgemm_epilogue(d);
    BOperand a, b;
    bgemm_operands(d, a, b);
    if (use_b256(d)) {
        if (!d->a_trans && !d->b_trans) return b256::launch<true, false>(a, b, ep, d->M, d->N, d->K, d->split_k, workspace, workspace_bytes, s);
        if (!d->a_trans && d->b_trans) return b256::launch<true, true>(a, b, ep, d->M, d->N, d->K, d->split_k, workspace, workspace_bytes, s);
        if (d->a_trans && !d->b_trans) return b256::launch<false, false>(a, b, ep, d->M, d->N, d->K, d->split_k, workspace, workspace_bytes, s);
        return b256::launch<false, true>(a, b, ep, d->M, d->N, d->K, d->split_k, workspace, workspace_bytes, s);
    }
    if (!d->a_trans && !d->b_trans) return launch_bgemm<true, false>(a, b, ep, d->M, d->N, d->K, d->split_k, workspace, workspace_bytes, s);
    if (!d->a_trans && d->b_trans) return launch_bgemm<true, true>(a, b, ep, d->M, d->N, d->K, d->split_k, workspace, workspace_bytes, s);
    if (d->a_trans && !d->b_trans) return launch_bgemm<false, false>(a, b, ep, d->M, d->N, d->K, d->split_k, workspace, workspace_bytes, s);
    return launch_bgemm<false, true>(a, b, ep, d->M, d->N, d->K, d->split_k, workspace, workspace_bytes, s);
}

static int wgrad_bf16_validate(const dc_conv_wgrad_bf16_desc* d) {
    DC_REQUIRE(d && d->x && d->dy && d->dw, DC_EINVAL, "dc_conv2d_wgrad_bf16: x, dy and dw must be non-null");
    DC_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->kh >= 1 && d->kw >= 1 && d->stride >= 1, DC_EINVAL,
               "dc_conv2d_wgrad_bf16: bad shape");
    DC_REQUIRE(d->Cin % 128 == 0 && d->Cout % 8 == 0, DC_EINVAL,
               "dc_conv2d_wgrad_bf16: needs Cin %% 128 == 0 (a 128-column tile inside one tap) and Cout %% 8 == 0");
    DC_REQUIRE(aligned16(d->x) && aligned16(d->dy) && aligned16(d->dw), DC_EALIGN, "dc_conv2d_wgrad_bf16: pointers must be 16-byte aligned");
    DC_REQUIRE((size_t)d->N * d->H * d->W * d->Cin * 2 < (size_t)0x7FFFFFF0u && (size_t)d->N * d->Ho * d->Wo * d->Cout * 2 < (size_t)0x7FFFFFF0u,
               DC_EINVAL, "dc_conv2d_wgrad_bf16: x and dy must span < 2 GiB");
    return DC_OK;
}

static bool wgrad_big(const dc_conv_wgrad_bf16_desc* d, int M, int N, int K) { return (d->Cout & 7) == 0 && b256::prefer(M, N, K, d->split_k, true); }

extern "C" size_t dc_conv2d_wgrad_bf16_workspace_bytes(const dc_conv_wgrad_bf16_desc* d) {
    if (!d || wgrad_bf16_validate(d)) return 0;
    const int M = d->Cout, N = d->kh * d->kw * d->Cin, K = d->N * d->Ho * d->Wo;
    const BSplit sp = wgrad_big(d, M, N, K) ? b256::split(M, N, K, d->split_k) : bgemm_split(M, N, K, d->split_k);
    return sp.split > 1 ? (size_t)sp.split * M * N * sizeof(float) : 0;
}

extern "C" int dc_conv2d_wgrad_bf16_tile(const dc_conv_wgrad_bf16_desc* d, int* split_k) {
    if (!d || wgrad_bf16_validate(d)) return 0;
    const int M = d->Cout, N = d->kh * d->kw * d->Cin, K = d->N * d->Ho * d->Wo;
    const bool big = wgrad_big(d, M, N, K);
    if (split_k) *split_k = (big ? b256::split(M, N, K, d->split_k) : bgemm_split(M, N, K, d->split_k)).split;
    return big ? 256 : 128;
}

extern "C" int dc_conv2d_wgrad_bf16(const dc_conv_wgrad_bf16_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = wgrad_bf16_validate(d);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int M = d->Cout, N = d->kh * d->kw * d->Cin, K = d->N * d->Ho * d->Wo;
    const bool big = wgrad_big(d, M, N, K);
    const BSplit sp = big ? b256::split(M, N, K, d->split_k) : bgemm_split(M, N, K, d->split_k);
    float* partial = nullptr;
    if (sp.split > 1) {
        const size_t need = (size_t)sp.split * M * N * sizeof(float);
        DC_REQUIRE(workspace != nullptr && workspace_bytes >= need, DC_EWORKSPACE, "dc_conv2d_wgrad_bf16 split-K needs %zu workspace bytes, got %zu",
                   need, workspace_bytes);
        partial = static_cast<float*>(workspace);
    }
    Epilogue ep{d->dw, N, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, d->accumulate, 1};
    BOperand dy{d->dy, d->Cout, M, nullptr, (unsigned)((size_t)K * d->Cout * 2)};
    BIm2col xc{d->x, d->H, d->W, d->Cin, d->Ho, d->Wo, d->stride, d->pad_t, d->pad_l, d->kw, K, (unsigned)((size_t)d->N * d->H * d->W * d->Cin * 2), N};
    if (big) {
        ep.vec4 = 1;                                           // dw rows are k*k*Cin floats (Cin % 128 == 0), 16-byte aligned base
        DC_ENSURE_DYN_LDS(&bgemm256_wgrad_kernel, 160 * 1024);
        const int tiles = ((M + b256::BM - 1) / b256::BM) * ((N + b256::BN - 1) / b256::BN);
        hipLaunchKernelGGL(bgemm256_wgrad_kernel, dim3(tiles, 1, sp.split), dim3(b256::NTHREADS), b256::LDS_BYTES, s, dy, xc, ep, M, N, K, sp.klen, partial);
        rc = check_launch("bgemm256_wgrad_kernel");
    } else {
        DC_ENSURE_DYN_LDS(&bgemm_wgrad_kernel, 160 * 1024);
        const int tiles = ((M + BT - 1) / BT) * ((N + BT - 1) / BT);
        hipLaunchKernelGGL(bgemm_wgrad_kernel, dim3(tiles, 1, sp.split), dim3(256), bgemm_lds_bytes(), s, dy, xc, ep, M, N, K, sp.klen, partial);
        rc = check_launch("bgemm_wgrad_kernel");
    }
    if (rc || sp.split <= 1) return rc;
    const long total = (long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(splitk_reduce_blocks(total)), dim3(256), 0, s, partial, sp.split, M, N, ep);
    return check_launch("splitk_reduce_kernel");
}

extern "C" int dc_cast_f32_bf16(const float* x, uint16_t* out, size_t n, void* stream) {
    DC_REQUIRE(x && out && n > 0, DC_EINVAL, "dc_cast_f32_bf16: bad arguments");
    DC_REQUIRE(aligned16(x) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0, DC_EALIGN, "dc_cast_f32_bf16: x 16-byte, out 8-byte aligned");
    const int blocks = (int)std::min<size_t>((n / 4 + 255) / 256 + 1, (size_t)kNumCU * 8);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, n);
    return check_launch("cast_bf16_kernel");
}

extern "C" int dc_cast_bf16_f32(const uint16_t* x, float* out, size_t n, void* stream) {
    DC_REQUIRE(x && out && n > 0, DC_EINVAL, "dc_cast_bf16_f32: bad arguments");
    DC_REQUIRE(aligned16(out) && (reinterpret_cast<uintptr_t>(x) & 7u) == 0, DC_EALIGN, "dc_cast_bf16_f32: out 16-byte, x 8-byte aligned");
    const int blocks = (int)std::min<size_t>((n / 4 + 255) / 256 + 1, (size_t)kNumCU * 8);
    hipLaunchKernelGGL(uncast_bf16_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, n);
    return check_launch("uncast_bf16_kernel");
}

extern "C" int dc_cast_f32_bf16_2d(const float* x, int ld_in, uint16_t* out, int ld_out, int rows, int cols, int cols_out, void* stream) {
    DC_REQUIRE(x && out && rows > 0 && cols > 0 && cols_out >= cols && ld_in >= cols && ld_out >= cols_out, DC_EINVAL, "dc_cast_f32_bf16_2d: bad arguments");
    const long total = (long)rows * cols_out;
    const int blocks = (int)std::min<long>((total + 255) / 256, (long)kNumCU * 8);
    hipLaunchKernelGGL(cast_bf16_2d_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long)ld_in, out, (long)ld_out, rows, cols, cols_out);
    return check_launch("cast_bf16_2d_kernel");
}
