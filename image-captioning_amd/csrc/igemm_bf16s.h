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
#ifdef DCAP_EXP_NOSPLIT
    p0 = p1 = p2 = cvt_pk_bf16(a, b);
    return;
#endif
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

// Conv weights split on the host into three bf16 planes [3][Cout][taps*Cin] (frozen encoder weights: the split is paid
// once at load time).  Walked chunk-major like ConvWeightKC; a thread moves 16-byte pieces (8 k of one row of one plane)
// straight from global memory to the LDS image -- no VALU.
struct SplitWeightKC {
    static constexpr bool KC = true;
    static constexpr bool PRESPLIT = true;
    __device__ __forceinline__ int kclamp(int k0, int) const { return k0; }
    const unsigned short* p;      // [3][rows][ld] bf16 bit patterns
    long ld;
    int rows, taps, Cin;
    template <int BT>
    struct State {
        unsigned goff[3 * BT / 64];     // byte offset of this thread's pieces at K-tile 0
        unsigned soff[3 * BT / 64];     // byte offset inside the stage's B image
        int tap, koff;
    };
    template <int BT>
    __device__ __forceinline__ void init(State<BT>& s, int row0, int tid) const {
        s.tap = -1;
#pragma unroll
        for (int j = 0; j < 3 * BT / 64; ++j) {
            const int c = tid + 256 * j, plane = c / (4 * BT), row = (c % (4 * BT)) >> 2, part = c & 3;
            s.goff[j] = (unsigned)((((long)plane * rows + min(row0 + row, rows - 1)) * ld + 8 * part) * 2);
            s.soff[j] = (unsigned)(((plane * BT + row) * LDB + 8 * part) * 2);
        }
    }
    template <int BT>
    __device__ __forceinline__ void load(State<BT>& s, f4 (&r)[3 * BT / 64], int k0, int, int) const {
        if (s.tap < 0) {
            const int t = k0 >> 5, chunk = t / taps;
            s.tap = t - chunk * taps;
            s.koff = s.tap * Cin + chunk * 32;
        }
        const char* kb = reinterpret_cast<const char*>(p + min(s.koff, (int)ld - BK));
#pragma unroll
        for (int j = 0; j < 3 * BT / 64; ++j) r[j] = *reinterpret_cast<const f4*>(kb + s.goff[j]);
        if (++s.tap == taps) { s.tap = 0; s.koff += 32 - (taps - 1) * Cin; } else { s.koff += Cin; }
    }
    template <int BT>
    __device__ __forceinline__ void store_planes(const State<BT>& s, __bf16* S, const f4 (&r)[3 * BT / 64]) const {
#pragma unroll
        for (int j = 0; j < 3 * BT / 64; ++j) *reinterpret_cast<f4*>(reinterpret_cast<char*>(S) + s.soff[j]) = r[j];
    }
};

template <class L>
struct is_presplit { static constexpr bool value = false; };
template <>
struct is_presplit<SplitWeightKC> { static constexpr bool value = true; };
template <class L, int BT>
constexpr int stage_regs() { return is_presplit<L>::value ? 3 * BT / 64 : BT / 32; }

template <int BT, class L, class ST, int NR>
__device__ __forceinline__ void stage_store(const L& l, const ST& st, __bf16* S, f4 (&r)[NR], int tid) {
    if constexpr (is_presplit<L>::value) l.template store_planes<BT>(st, S, r);
    else store_split_kc<BT>(S, r, tid);
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
#ifndef DCAP_EXP_NOLOAD
        {
            const int k0 = kbeg + (kt + 2) * BK;
            al.template load<BM>(sa, qa, al.kclamp(k0, kend), kend, tid);
            bl.template load<BN>(sb, qb, bl.kclamp(k0, kend), kend, tid);
        }
#endif
#ifndef DCAP_EXP_NOFRAG
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
#ifdef DCAP_EXP_NOMFMA
                    c[0] += (float)a[tm][0][0] + (float)a[tm][1][1] + (float)a[tm][2][2] + (float)b[tn][0][3] + (float)b[tn][1][4] + (float)b[tn][2][5];
                    acc[tm][tn] = c;
                    continue;
#endif
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
#endif
        __syncthreads();                 // fragment reads done: the LDS image may be overwritten
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        phase(kt, ra[0], rb[0]);
        if (kt + 1 < nkt) phase(kt + 1, ra[1], rb[1]);
    }
    store_tile<BM, BN>(acc, smem, ep, partial, M, N, m0, n0, wm, wn);
}

// ------------------------------------------------------------------------------------------------
// v2: producer / consumer waves.  A block is 8 waves: waves 0-3 (one per SIMD) only read fragments and issue MFMAs;
// waves 4-7 (one per SIMD) only move data: global loads two tiles ahead, the bf16 split, LDS writes.  LDS is
// double-buffered and both roles meet at ONE barrier per K-tile, so a SIMD always has an MFMA-bound and a VALU/LDS-bound
// wave to pick from: the split, the LDS writes (the slow LDS direction, ~70 B/clk/CU) and the loads run in the shadow of
// the MFMAs instead of in series with them (v1: load+split+write 1300 + fragment reads 600 + MFMA 1536 cycles per K-tile,
// nearly additive even with two blocks per CU, which settle into lock-step).
// ------------------------------------------------------------------------------------------------
template <int BM, int BN>
constexpr size_t igemm_bs2_lds_bytes() {
    constexpr size_t stages = (size_t)2 * 3 * (BM + BN) * LDB * sizeof(__bf16);
    constexpr size_t cimage = (size_t)BM * (BN + 4) * sizeof(float);
    return stages > cimage ? stages : cimage;
}

// NCW consumer waves (4: one per SIMD; 8: two per SIMD, so one wave's fragment-read / barrier latency hides under the
// other's MFMAs) + 4 producer waves.  A consumer wave owns TM x TN 32x32 accumulator blocks.
template <int BM, int BN>
struct bs2_shape {
    static constexpr int NCW = (BM * BN > 64 * 64) ? 8 : 4;
    static constexpr int BLOCKS = (BM / 32) * (BN / 32);                 // 32x32 blocks in the tile
    static constexpr int PER = BLOCKS / NCW;                             // per consumer wave
    static constexpr int TN = 1, TM = PER;                               // a wave's blocks are stacked along M
    static constexpr int WC = BN / 32, WR = BM / (32 * TM);              // wave grid
    static constexpr int THREADS = (NCW + 4) * 64;
};

template <int BM, int BN, class AL, class BL>
__global__ __launch_bounds__((bs2_shape<BM, BN>::THREADS), 1) void igemm_bs2_kernel(AL al, BL bl, Epilogue ep, int M, int N, int K, int klen,
                                                                                 float* __restrict__ partial) {
    static_assert(AL::KC && BL::KC, "split-bf16 main loop: both operands K-contiguous");
    using SH = bs2_shape<BM, BN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = SH::TM, TN = SH::TN, NCW = SH::NCW;
    constexpr int STAGE = 3 * (BM + BN) * LDB;                 // bf16 elements per stage
    __bf16* S0 = reinterpret_cast<__bf16*>(smem);

    const int tid = threadIdx.x, wave = tid >> 6;
    const int tiles_n = (N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
    const int kbeg = blockIdx.z * klen;
    const int kend = min(K, kbeg + klen);
    const int nkt = (kend - kbeg + BK - 1) / BK;

    if (wave >= NCW) {
        // ------------------------------------------------------------------ producer waves
        const int ptid = tid - NCW * 64;
        typename AL::template State<BM> sa;
        typename BL::template State<BN> sb;
        al.template init<BM>(sa, m0, ptid);
        bl.template init<BN>(sb, n0, ptid);
        // PF K-tiles of staging registers in flight: 4 producer waves must cover an L2 / Infinity-Cache round trip
        // (~1.4 us under load) with their own loads only
        constexpr int PF = 3;
        constexpr int NRA = stage_regs<AL, BM>(), NRB = stage_regs<BL, BN>();
        f4 ra[PF][NRA], rb[PF][NRB];
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int k0 = kbeg + j * BK;
            al.template load<BM>(sa, ra[j], al.kclamp(k0, kend), kend, ptid);
            bl.template load<BN>(sb, rb[j], bl.kclamp(k0, kend), kend, ptid);
        }
        auto fill = [&](int t, f4 (&qa)[NRA], f4 (&qb)[NRB]) {     // tile t: registers -> stage t&1, refill with tile t+PF
            if (t < nkt) {
                __bf16* st = S0 + (t & 1) * STAGE;
                stage_store<BM>(al, sa, st, qa, ptid);
                stage_store<BN>(bl, sb, st + 3 * BM * LDB, qb, ptid);
            }
#ifndef DCAP_EXP_NOLOAD
            const int k0 = kbeg + (t + PF) * BK;
            al.template load<BM>(sa, qa, al.kclamp(k0, kend), kend, ptid);
            bl.template load<BN>(sb, qb, bl.kclamp(k0, kend), kend, ptid);
#endif
        };
        fill(0, ra[0], rb[0]);
        __syncthreads();                                     // stage 0 ready
        // tile t+1 is staged while the consumers work on tile t; register set = tile % PF, so the loop is unrolled PF-fold
        for (int kt = 0; kt < nkt; kt += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (kt + u < nkt) {
                    fill(kt + u + 1, ra[(u + 1) % PF], rb[(u + 1) % PF]);
                    __syncthreads();
                }
            }
        }
        __syncthreads();                                     // the epilogue's barrier (store_tile)
        return;
    }
    // ---------------------------------------------------------------------- consumer waves
    const int lane = tid & 63;
    const int wm = (wave / SH::WC) * (32 * TM), wn = (wave % SH::WC) * (32 * TN);
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
    const int aoff = (wm + li) * LDB + 8 * lh, boff = 3 * BM * LDB + (wn + li) * LDB + 8 * lh;
    __syncthreads();                                         // stage 0 ready
    for (int kt = 0; kt < nkt; ++kt) {
        const __bf16* afrag = S0 + (kt & 1) * STAGE + aoff;
        const __bf16* bfrag = S0 + (kt & 1) * STAGE + boff;
#ifndef DCAP_EXP_NOFRAG
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 a[TM][3], b[TN][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
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
#ifdef DCAP_EXP_NOMFMA
                    c[0] += (float)a[tm][0][0] + (float)a[tm][1][1] + (float)a[tm][2][2] + (float)b[tn][0][3] + (float)b[tn][1][4] + (float)b[tn][2][5];
                    acc[tm][tn] = c;
                    continue;
#endif
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][2], b[tn][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][0], c, 0, 0, 0);
                    acc[tm][tn] = c;
                }
        }
#endif
        __syncthreads();                                     // tile kt consumed; tile kt+1 staged
    }
    store_tile<BM, BN, TM, TN, NCW * 64>(acc, smem, ep, partial, M, N, m0, n0, wm, wn);
}

template <int BM, int BN, class AL, class BL>
int launch_igemm_bs2(const AL& al, const BL& bl, const Epilogue& ep, int M, int N, int K, int split_k, void* workspace,
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
    constexpr size_t lds = igemm_bs2_lds_bytes<BM, BN>();
    DC_ENSURE_DYN_LDS((&igemm_bs2_kernel<BM, BN, AL, BL>), 160 * 1024);
    dim3 grid(tiles, 1, split_k);
    hipLaunchKernelGGL((igemm_bs2_kernel<BM, BN, AL, BL>), grid, dim3(bs2_shape<BM, BN>::THREADS), lds, stream, al, bl, ep, M, N, K, klen, partial);
    int rc = check_launch("igemm_bs2_kernel");
    if (rc) return rc;
    if (split_k > 1) {
        const long total = (long)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(splitk_reduce_blocks(total)), dim3(256), 0, stream, partial, split_k, M, N, ep);
        rc = check_launch("splitk_reduce_kernel");
    }
    return rc;
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
