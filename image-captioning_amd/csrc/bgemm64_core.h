// bgemm64_core.h -- the SMALL-tile bf16 MFMA main loop (round 3): 64 x 64 x 64 block tile, 256 threads = 4 waves (2 x 2), each wave
// a 32 x 32 sub-tile on v_mfma_f32_16x16x32_bf16, a FOUR-stage LDS ring filled three K-tiles ahead by LDS-DMA, one barrier per
// K-tile, two blocks per CU.
//
// For the layers whose output is too small for the large tiles to cover the chip: at one image per GPU (BASELINE configs[4]) a
// stage-4 convolution is 4096 pixels x 256 channels = 64 tiles of 128 x 128 on 256 CUs; the 128-tile loop therefore splits K
// eight ways, runs two or four K-tiles per block behind a full DMA latency each, writes fp32 slabs and needs a second launch
// to sum them (round 2: 123 slab reductions per joint step).  Here the same layer is 256 tiles with the WHOLE K loop in one
// block and no slabs; what bounds it is the L2 -> LDS fill of a CU (16 KiB per K-tile, measured 66-73 GB/s per CU from L2:
// MI355X_MICROARCH.md, indexed rows), so the loop's only job is to keep that stream busy: three K-tiles (48 KiB) in flight behind
// a counted s_waitcnt vmcnt(8), the DMA of tile t + 3 issued right after the barrier that retires tile t - 1.
//
// Operands: both K-contiguous ("KC": rows of 64 k = 128 bytes, chunk c of row r at c ^ ((r >> 1) & 7), the image format of
// bgemm_core.h / bgemm256_core.h).  The MFMA takes the B-side fragment as its A operand (transposed 16 x 16 tiles), so a
// lane owns four consecutive columns of an output row and the epilogue is 16-byte accesses straight from the accumulators.
#pragma once
#include "bgemm256_core.h"

namespace dcap {
namespace b64 {

constexpr int BM = 64, BN = 64, BK = 64, NTHREADS = 256, NS = 4;
constexpr int IMG = 64 * BK * 2;           // one operand image: 64 rows x 64 k of bf16 = 8 KiB
constexpr int STAGE = 2 * IMG;             // A, B
constexpr int LDS_BYTES = NS * STAGE;      // 64 KiB

typedef b256::f32x4 f32x4;

// A dense K-contiguous operand: two 1-KiB pieces per wave and K-tile (wave w fills rows 16 w .. 16 w + 15).
struct Load {
    static constexpr bool KC = true;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[2];
    int kloc[2];
    __device__ __forceinline__ void init(const BOperand& o, int origin, int lane, int wave) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(o.p), 0, (int)o.bytes, 0x00020000);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int rp = 8 * (2 * wave + jj) + (lane >> 3);
            const int c = (lane & 7) ^ ((rp >> 1) & 7);
            const int row = min(origin + rp, o.extent - 1);
            const long src = o.gather ? (long)o.gather[row] : (long)row;
            voff[jj] = (unsigned)((src * o.ld + 8 * c) * 2);
            kloc[jj] = 8 * c;
        }
    }
    __device__ __forceinline__ void issue(char* img, int k0, int kend, int wave) const {
        const bool tail = k0 + BK > kend;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const unsigned v = (tail && k0 + kloc[jj] >= kend) ? kOobOffset : voff[jj];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DC_LDS void*)(img + (2 * wave + jj) * 1024), 16, (int)v, k0 * 2, 0, 0);
        }
    }
};

// acc[tm][tn]: element j of lane l is C[m0 + 32 wm + 16 tm + (l & 15)][n0 + 32 wn + 16 tn + 4 (l >> 4) + j].
// LA / LB: loaders with issue(char* image, int k0, int kend, int wave) issuing exactly two pieces each.
template <class LA, class LB>
__device__ __forceinline__ void mainloop(LA& la, LB& lb, char* smem, int kbeg, int kend, f32x4 (&acc)[2][2]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned lds0 = (unsigned)(size_t)(DC_LDS char*)smem;
    b256::FragAddr<true, 2> fa, fb;
    fa.init(lds0, 32 * wm, lane);
    fb.init(lds0 + IMG, 32 * wn, lane);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkt = (kend - kbeg + BK - 1) / BK;
    // prologue: K-tiles 0, 1, 2 into stages 0, 1, 2 (tiles past the end are issued out of range: hardware zeros, uniform counts)
#pragma unroll
    for (int j = 0; j < NS - 1; ++j) {
        la.issue(smem + j * STAGE, kbeg + j * BK, kend, wave);
        lb.issue(smem + j * STAGE + IMG, kbeg + j * BK, kend, wave);
    }
    auto step = [&](int kt, auto st_c) {
        constexpr int ST = decltype(st_c)::value;                          // stage of K-tile kt
        constexpr int NX = (ST + NS - 1) % NS;                             // stage that tile kt + 3 goes into (= tile kt - 1's)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                   // this wave's pieces of tile kt have landed (kt + 1, kt + 2 in flight)
        __builtin_amdgcn_s_barrier();                                      // every wave's have; every wave is done reading tile kt - 1
        la.issue(smem + NX * STAGE, kbeg + (kt + NS - 1) * BK, kend, wave);
        lb.issue(smem + NX * STAGE + IMG, kbeg + (kt + NS - 1) * BK, kend, wave);
        b256::FragReg<true> A[2][2], B[2][2];
        b256::frag_read<true, 2, 0>(fa, (unsigned)(ST * STAGE), A);
        b256::frag_read<true, 2, 0>(fb, (unsigned)(ST * STAGE), B);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        b256::frag_touch(A);
        b256::frag_touch(B);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[tn][s].get(), A[tm][s].get(), acc[tm][tn], 0, 0, 0);
    };
    for (int kt = 0; kt < nkt; kt += NS) {
        step(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < nkt) step(kt + 1, std::integral_constant<int, 1>{});
        if (kt + 2 < nkt) step(kt + 2, std::integral_constant<int, 2>{});
        if (kt + 3 < nkt) step(kt + 3, std::integral_constant<int, 3>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the (empty) tiles past the end
}

__device__ __forceinline__ void store_tile(f32x4 (&acc)[2][2], const Epilogue& ep, int M, int N, int m0, int n0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rbase = m0 + 32 * (wave >> 1) + (lane & 15), cbase = n0 + 32 * (wave & 1) + 4 * (lane >> 4);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int row = rbase + 16 * tm, rowc = min(row, M - 1);
        const float* rr = ep.res_row(rowc);
        float* crow = ep.C + (long)rowc * ep.ldc;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = cbase + 16 * tn, colc = min(col, N - 4);
            const bool live = row < M && col < N;
            f32x4 v = acc[tm][tn];
            if (ep.scale) v *= *reinterpret_cast<const f32x4*>(ep.scale + colc);
            if (ep.shift) v += *reinterpret_cast<const f32x4*>(ep.shift + colc);
            if (rr) v += *reinterpret_cast<const f32x4*>(rr + colc);
            if (ep.relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            if (ep.accumulate) v += *reinterpret_cast<const f32x4*>(crow + colc);
            if (live && ep.C) *reinterpret_cast<f32x4*>(crow + col) = v;
            if (live && ep.Cb) {
                typedef unsigned short us4 __attribute__((ext_vector_type(4)));
                *reinterpret_cast<us4*>(ep.Cb + (long)row * ep.ldcb + col) =
                    us4{Epilogue::bf16_bits(v[0]), Epilogue::bf16_bits(v[1]), Epilogue::bf16_bits(v[2]), Epilogue::bf16_bits(v[3])};
            }
        }
    }
}

// cost model entry (see b256::tile_cost_us): no split-K; a block streams 16 KiB per K-tile from L2
inline double cost_us(int M, int N, int K) {
    const long blocks = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const double rounds = (double)((blocks + 2 * kNumCU - 1) / (2 * kNumCU));
    const double kt = (double)((K + BK - 1) / BK);
    return rounds * (kt * (blocks > kNumCU ? 0.47 : 0.33) + 3.0);           // measured: profiles/r03_bconv_bench.txt
}

}  // namespace b64
}  // namespace dcap
