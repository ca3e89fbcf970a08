// igemm_bf16s.h -- the implicit-GEMM main loop on the bf16 matrix pipe with fp32-grade accuracy.
//
// gfx950 has no fp32 (or xf32) path through the matrix cores: v_mfma_f32_32x32x2_f32 runs at 157 TFLOP/s on the fp32
// lanes, v_mfma_f32_32x32x16_bf16 at 2.5 PFLOP/s.  Every fp32 operand element is therefore split, on its way from
// registers to LDS, into three bf16 pieces by round-to-nearest:   x = p0 + p1 + p2 + e,  |e| <= 2^-25 |x|
// (8+8+8 significant bits and the sign of each remainder; each subtraction is exact in fp32), and a product is the six
// piece products of order <= 2^-16:
//       a*b ~= a0*b0 + (a0*b1 + a1*b0) + (a0*b2 + a1*b1 + a2*b0)        dropped terms <= 3 * 2^-24 |a*b|
// accumulated in fp32 inside the MFMA, smallest terms first.  The result differs from the exact-fp32 kernel at the level
// of fp32 summation-order noise (tests: same tolerances as the f32 path), at 6 x 32 = 192 matrix-pipe cycles per 16-deep
// K-step of a 32x32 block instead of 512 fp32-lane cycles; VALU work (the split) now overlaps the MFMAs.
//
// Structure: same loaders (fp32 global -> registers, unconditional, range-checked), same XCD remap, split-K slabs and
// LDS-transposed epilogue as igemm_core.h.  LDS holds ONE K-tile as three bf16 planes per operand, rows of 32 k = 64 B
// padded to 80 B (conflict-free ds_read_b128 fragments: lane (i,h) reads k = 16s + 8h .. +7 of row i, which is exactly
// the 32x32x16 operand layout -- no K permutation).  A block is single-buffered (61 KB for 128x128, so two blocks share
// a CU and one block's split+store phase runs under the other's MFMAs); two K-tiles of staging registers keep every
// global load in flight for two MFMA phases before it is drained.
// Both operands must be K-contiguous (KC loaders): conv forward / data gradient and NT GEMMs.
// DCAP_EXP_{NOSPLIT,NOMFMA,NOFRAG,NOLOAD} are ablation switches for tools/build_variant.sh (never defined in the product build):
// they knock out one phase of the loop so that its share of the K-tile time can be read off a timing difference.
#pragma once
#include "igemm_core.h"

namespace dcap {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int LDB = BK + 8;      // bf16 elements per LDS row: 64 B of data + 16 B pad = 80 B

// x = p0 + p1 + p2 (+ <= 2^-25 |x|), pieces rounded to nearest even (v_cvt_pk_bf16_f32); the remainders are exact.
// Written on packed pairs: one conversion yields both bf16 pieces, and their fp32 values come back with a shift and a mask
// (the compiler's own lowering of the vector form converts every element twice: 30 instead of 22 VALU per four floats).
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = cvt_pk_bf16(a, b);
    a -= __uint_as_float(p0 << 16);
    b -= __uint_as_float(p0 & 0xffff0000u);
    p1 = cvt_pk_bf16(a, b);
    a -= __uint_as_float(p1 << 16);
    b -= __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16(a, b);
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// registers (thread = row tid>>3 (+32 i), k quad tid&7) -> three bf16 planes [plane][BT rows][LDB]
template <int BT, int NR, int NP = 3>
__device__ __forceinline__ void store_split_kc(__bf16* S, const f4 (&r)[NR], int tid) {
    static_assert(NR == BT / 32, "one f4 per 32 staged rows");
    const int q = tid & 7, rr = tid >> 3;
#pragma unroll
    for (int i = 0; i < BT / 32; ++i) {
        __bf16* d = S + (rr + 32 * i) * LDB + 4 * q;
        if constexpr (NP == 1) {                  // plain bf16 operands (DC_MATH_BF16): one rounding, no remainders
            *reinterpret_cast<u32x2*>(d) = u32x2{cvt_pk_bf16(r[i][0], r[i][1]), cvt_pk_bf16(r[i][2], r[i][3])};
            continue;
        }
        unsigned a0, a1, a2, b0, b1, b2;
        split_pair(r[i][0], r[i][1], a0, a1, a2);
        split_pair(r[i][2], r[i][3], b0, b1, b2);
        *reinterpret_cast<u32x2*>(d) = u32x2{a0, b0};
        *reinterpret_cast<u32x2*>(d + BT * LDB) = u32x2{a1, b1};
        if constexpr (NP == 3) *reinterpret_cast<u32x2*>(d + 2 * BT * LDB) = u32x2{a2, b2};      // NP == 2: the third piece is dead code
    }
}

template <int BM, int BN, int NP = 3>
constexpr size_t igemm_bs_lds_bytes() {
    constexpr size_t stage = (size_t)NP * (BM + BN) * LDB * sizeof(__bf16);
    constexpr size_t cimage = (size_t)BM * (BN + 4) * sizeof(float);
    return stage > cimage ? stage : cimage;
}

// NP = 3: six products (fp32-grade).  NP = 2: pieces p0, p1 only (x = p0 + p1 to 2^-17 |x|) and the three products
// a0*b0 + a0*b1 + a1*b0 -- a 16-bit-mantissa product (2^-16 relative, 32x finer than TF32) at half the MFMAs.
template <int BM, int BN, class AL, class BL, int NP = 3>
__global__ __launch_bounds__(256, 2) void igemm_bs_kernel(AL al, BL bl, Epilogue ep, int M, int N, int K, int klen,
                                                       float* __restrict__ partial) {
    static_assert(AL::KC && BL::KC, "split-bf16 main loop: both operands K-contiguous");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = BM / 64, TN = BN / 64;
    __bf16* As = reinterpret_cast<__bf16*>(smem);
    __bf16* Bs = As + NP * BM * LDB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
    const int kbeg = blockIdx.z * klen;
    const int kend = min(K, kbeg + klen);
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    const int li = lane & 31, lh = lane >> 5;

    typename AL::template State<BM> sa;
    typename BL::template State<BN> sb;
    al.template init<BM>(sa, m0, tid);
    bl.template init<BN>(sb, n0, tid);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // two K-tiles of staging registers: a tile's global loads are issued two MFMA phases before they are drained to LDS
    // (one phase, ~0.6 us, is shorter than an L2/HBM round trip under load)
    f4 ra[2][BM / 32], rb[2][BN / 32];
    const int nkt = (kend - kbeg + BK - 1) / BK;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k0 = kbeg + j * BK;
        al.template load<BM>(sa, ra[j], al.kclamp(k0, kend), kend, tid);
        bl.template load<BN>(sb, rb[j], bl.kclamp(k0, kend), kend, tid);
    }

    const __bf16* afrag = As + (wm + li) * LDB + 8 * lh;
    const __bf16* bfrag = Bs + (wn + li) * LDB + 8 * lh;

    auto phase = [&](int kt, f4 (&qa)[BM / 32], f4 (&qb)[BN / 32]) {
        // registers (tile kt) -> bf16 planes in LDS; every wave is past the previous tile's fragment reads (barrier below)
        store_split_kc<BM, BM / 32, NP>(As, qa, tid);
        store_split_kc<BN, BN / 32, NP>(Bs, qb, tid);
        __syncthreads();
        // refill the drained registers with tile kt+2: unconditional (clamped / range-checked)
        {
            const int k0 = kbeg + (kt + 2) * BK;
            al.template load<BM>(sa, qa, al.kclamp(k0, kend), kend, tid);
            bl.template load<BN>(sb, qb, bl.kclamp(k0, kend), kend, tid);
        }
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 a[TM][NP], b[TN][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) a[tm][p] = *reinterpret_cast<const bf16x8*>(afrag + (p * BM + tm * 32) * LDB + 16 * s);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) b[tn][p] = *reinterpret_cast<const bf16x8*>(bfrag + (p * BN + tn * 32) * LDB + 16 * s);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    f32x16 c = acc[tm][tn];
                    if constexpr (NP == 3) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][2], b[tn][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][2], c, 0, 0, 0);
                    }
                    if constexpr (NP >= 2) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][1], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][0], c, 0, 0, 0);
                    acc[tm][tn] = c;
                }
        }
        __syncthreads();                 // fragment reads done: the LDS image may be overwritten
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        phase(kt, ra[0], rb[0]);
        if (kt + 1 < nkt) phase(kt + 1, ra[1], rb[1]);
    }
    store_tile<BM, BN>(acc, smem, ep, partial, M, N, m0, n0, wm, wn);
}

template <int BM, int BN, class AL, class BL, int NP = 3>
int launch_igemm_bs(const AL& al, const BL& bl, const Epilogue& ep, int M, int N, int K, int split_k, void* workspace,
                    size_t workspace_bytes, hipStream_t stream) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int ktiles = (K + BK - 1) / BK;
    if (split_k < 1) split_k = 1;
    if (split_k > ktiles) split_k = ktiles;
    const int klen = ((ktiles + split_k - 1) / split_k) * BK;
    split_k = (K + klen - 1) / klen;
    float* partial = nullptr;
    if (split_k > 1) {
        const size_t need = (size_t)split_k * M * N * sizeof(float);
        DC_REQUIRE(workspace != nullptr && workspace_bytes >= need, DC_EWORKSPACE,
                   "igemm split-K needs %zu workspace bytes, got %zu", need, workspace_bytes);
        partial = static_cast<float*>(workspace);
    }
    constexpr size_t lds = igemm_bs_lds_bytes<BM, BN, NP>();
    DC_ENSURE_DYN_LDS((&igemm_bs_kernel<BM, BN, AL, BL, NP>), 160 * 1024);
    dim3 grid(tiles, 1, split_k);
    hipLaunchKernelGGL((igemm_bs_kernel<BM, BN, AL, BL, NP>), grid, dim3(256), lds, stream, al, bl, ep, M, N, K, klen, partial);
    int rc = check_launch("igemm_bs_kernel");
    if (rc) return rc;
    if (split_k > 1) {
        const long total = (long)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(splitk_reduce_blocks(total)), dim3(256), 0, stream, partial, split_k, M, N, ep);
        rc = check_launch("splitk_reduce_kernel");
    }
    return rc;
}

}  // namespace dcap
