// bgemm256_core.h -- the LARGE-tile bf16 MFMA main loop (round 3): 256 x 256 x 64 block tile, 512 threads = 8 waves as
// 2 (M) x 4 (N), each wave a 128 x 64 sub-tile on v_mfma_f32_16x16x32_bf16 (fp32 accumulate), ONE block per CU.
// Replaces the 128 x 128 loop of bgemm_core.h wherever the problem fills the chip with 256-square tiles: dc_gemm_bf16,
// the fused vocabulary softmax / cross-entropy passes (vocab_ce.hip) and the bf16 convolutions of BASELINE configs[4].
//
// Why (MI355X_MICROARCH.md, LDS): a 64 x 64 wave tile reads one 1-KiB fragment per MFMA, 128 B/clk/CU of fragment reads plus
// the LDS-DMA's own writes against the 256 B/clk/CU the LDS array delivers -- the 128-tile loop sat at 0.9-1.0 PFLOP/s.  A
// 128 x 64 wave tile needs 0.375 fragment reads per 16x16x32 MFMA: 96 B/clk/CU of reads + 32 B/clk/CU of DMA writes.
//
// Schedule (cdna_hip_programming.md section 5, the 256-square 8-phase structure, re-derived for this layout):
//  * a K-tile (64 deep) is four PHASES = the four 64 x 32 quadrants of the wave tile, 16 MFMAs each.  A phase is
//      { fragment reads of the operand half that is new in this phase | LDS-DMA of one quarter of the NEXT K-tile |
//        counted s_waitcnt vmcnt(4) } s_barrier { lgkmcnt(0); 16 MFMAs } s_barrier
//    and the two wave groups (waves 0-3 = rows 0-127, waves 4-7 = rows 128-255; wave w and w + 4 share a SIMD) run ONE
//    barrier apart (`if (group == 1) s_barrier` in the prologue): while one group issues its 16 MFMAs the other issues its
//    reads and DMA, so each SIMD's matrix pipe always has a wave with operands in registers.
//  * the LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) is never drained in the loop: a quarter tile is
//    issued per phase (2 pieces per wave), waited for with vmcnt(4) two phases later, and read one phase after that wait
//    (two barriers later: with the groups one barrier apart that is what orders every wave's DMA before every wave's read).
//  * operands: each 256-row operand tile is TWO 128-row sub-images, one per half that a phase consumes (A: rows
//    {0-63, 128-191} / {64-127, 192-255} = the two 64-row halves of both wave groups; B: the two 32-column halves of the four
//    wave columns).  A sub-image has exactly the format of bgemm_core.h's images -- KC (K contiguous in memory: 128-byte rows,
//    chunk c of row r at c ^ ((r >> 1) & 7), one ds_read_b128 per 16 x 32 fragment) or MC (K-major in memory: 256-byte
//    rows of 128 columns, chunk swizzle c ^ (((k & 3) << 2) | ((k >> 2) & 3)), two ds_read_b64_tr_b16 per fragment) -- both
//    conflict-free for the 16x16x32 operand maps (tools/micro/lds_banks_b256.py).  2 stages x 4 sub-images x 16 KiB = 128 KiB.
//  * the MFMA takes the B-side fragment as its A operand and the A-side fragment as its B operand, i.e. it computes the
//    transposed 16 x 16 tile: a lane then holds FOUR CONSECUTIVE COLUMNS of one output row (C/D map: col = lane & 15 -> row
//    of C, row = 4 (lane >> 4) + j -> column of C), so the epilogue's scale / shift / residual / store are 16-byte accesses
//    straight from the accumulators -- no LDS transpose -- and a row reduction (vocab_ce) is 16 in-lane adds + 2 shuffles.
#pragma once
#include "bgemm_core.h"

namespace dcap {
namespace b256 {

constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 512;
constexpr int SUB = 128 * BK * 2;          // one sub-image: 128 rows (or columns) x 64 k of bf16 = 16 KiB
constexpr int STAGE = 4 * SUB;             // A half 0, A half 1, B half 0, B half 1
constexpr int LDS_BYTES = 2 * STAGE;       // 128 KiB
constexpr int OFF_A = 0, OFF_B = 2 * SUB;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// sub-image row r' (0..127) of half u  ->  row (column) of the 256-wide tile
template <bool IS_A>
__device__ __forceinline__ int tile_index(int u, int rp) {
    if constexpr (IS_A) return (rp >> 6) * 128 + u * 64 + (rp & 63);     // wave group g = r' >> 6 owns rows g*128 .. g*128+127
    else return (rp >> 5) * 64 + u * 32 + (rp & 31);                     // wave column c = r' >> 5 owns columns c*64 .. c*64+63
}

// A dense operand (BOperand of bgemm_core.h): the LDS-DMA of its two sub-images, two 1-KiB pieces per wave and sub-image.
template <bool KC_, bool IS_A>
struct Load {
    static constexpr bool KC = KC_;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[2][2];       // [half][piece]: per-lane byte offset of the 16-byte chunk at K-tile 0
    int kloc[2][2];            // k of the chunk inside the tile (KC) / k row inside the tile (MC)
    long ld;
    __device__ __forceinline__ void init(const BOperand& o, int origin, int lane, int wave) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(o.p), 0, (int)o.bytes, 0x00020000);
        ld = o.ld;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int pc = 2 * wave + jj;                                  // piece 0..15 of the sub-image
                if constexpr (KC) {
                    const int rp = 8 * pc + (lane >> 3);                       // sub-image row of this lane's chunk
                    const int c = (lane & 7) ^ ((rp >> 1) & 7);                // source chunk that lands in LDS chunk (lane & 7)
                    const int row = min(origin + tile_index<IS_A>(u, rp), o.extent - 1);   // past the edge: feeds nothing that is stored
                    const long src = o.gather ? (long)o.gather[row] : (long)row;
                    voff[u][jj] = (unsigned)((src * o.ld + 8 * c) * 2);
                    kloc[u][jj] = 8 * c;
                } else {
                    const int k = 4 * pc + (lane >> 4);                        // K row inside the tile
                    const int c = (lane & 15) ^ (((k & 3) << 2) | ((k >> 2) & 3));
                    const int col = min(origin + tile_index<IS_A>(u, 8 * c), o.extent - 8);  // clamped, never stored
                    voff[u][jj] = (unsigned)(((long)k * o.ld + col) * 2);
                    kloc[u][jj] = k;
                }
            }
    }
    // LDS-DMA of half u of the K-tile starting at k0 into the sub-image at `sub` (wave-uniform LDS address)
    __device__ __forceinline__ void issue(int u, char* sub, int k0, int kend, int wave) const {
        const bool tail = k0 + BK > kend;                                      // block-uniform; also covers "no such tile" (k0 >= kend)
        const int soff = KC ? k0 * 2 : (int)((long)k0 * ld * 2);               // < 2 GiB (host-checked span)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const unsigned v = (tail && k0 + kloc[u][jj] >= kend) ? kOobOffset : voff[u][jj];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DC_LDS void*)(sub + (2 * wave + jj) * 1024), 16, (int)v, soff, 0, 0);
        }
    }
};

// One fragment (16 rows x 32 k) in registers: KC one 128-bit read, MC two transposing 64-bit reads.
template <bool KC> struct FragReg;
template <> struct FragReg<true> {
    bh8 v;
    __device__ __forceinline__ bh8 get() const { return v; }
};
template <> struct FragReg<false> {
    sh4 lo, hi;
    __device__ __forceinline__ bh8 get() const { return __builtin_bit_cast(bh8, sh8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}); }
};

// Per-lane LDS byte addresses of a wave's fragments inside sub-image 0 of its operand in stage 0; R0 = the wave's first row
// inside a sub-image (A: group * 64, B: wave column * 32; both multiples of 16, which the swizzle keys below rely on).
template <bool KC, int NT>
struct FragAddr {
    unsigned off[KC ? 2 : 2 * NT];
    __device__ __forceinline__ void init(unsigned sub0, int R0, int lane) {
        if constexpr (KC) {
            // tile t, k-step s: row r = R0 + 16 t + i, chunk 4 s + q  ->  byte r * 128 + (((4 s + q) ^ key) << 4), key = (r >> 1) & 7
            // = (i >> 1) & 7 for every t; s flips bit 6, t adds 2048
            const int i = lane & 15, q = lane >> 4;
            const unsigned base = (unsigned)((R0 + i) * 128 + ((q ^ ((i >> 1) & 7)) << 4));
            off[0] = sub0 + base;
            off[1] = sub0 + (base ^ 64u);
        } else {
            // 16-lane group qg reads the 4 k x 16 column block at k = 32 s + 8 qg + 4 hf, columns R0 + 16 t; lane 4 a + b of the
            // group addresses row a, columns 4 b .. 4 b + 3 of the block; s adds 8192
            const int qg = lane >> 4, a = (lane & 15) >> 2, b = lane & 3;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int k = 8 * qg + 4 * hf + a;
                    const int ch = (R0 + 16 * t) / 8 + (b >> 1);
                    const int key = ((k & 3) << 2) | ((k >> 2) & 3);
                    off[2 * t + hf] = sub0 + (unsigned)(256 * k + 16 * (ch ^ key) + 8 * (b & 1));
                }
        }
    }
};

// issue the reads of all NT tiles x 2 k-steps of one sub-image; OFF = its byte offset from sub-image 0 of stage 0 (compile time)
template <bool KC, int NT, int OFF>
__device__ __forceinline__ void frag_read(const FragAddr<KC, NT>& f, unsigned stage_add, FragReg<KC> (&r)[NT][2]) {
    if constexpr (KC) {
        const unsigned a0 = f.off[0] + stage_add, a1 = f.off[1] + stage_add;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[t][0].v) : "v"(a0), "n"(OFF + t * 2048));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[t][1].v) : "v"(a1), "n"(OFF + t * 2048));
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const unsigned alo = f.off[2 * t] + stage_add, ahi = f.off[2 * t + 1] + stage_add;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r[t][s].lo) : "v"(alo), "n"(OFF + s * 8192));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r[t][s].hi) : "v"(ahi), "n"(OFF + s * 8192));
            }
        }
    }
}

// Ties the fragment registers to the s_waitcnt lgkmcnt(0) in front of it: an MFMA that consumes them cannot be scheduled
// above this statement (hipcc moves register-only instructions across a bare asm wait: cdna_hip_programming.md 5.4 rule 18).
template <int NT>
__device__ __forceinline__ void frag_touch(FragReg<true> (&r)[NT][2]) {
    if constexpr (NT == 4)
        asm volatile("" : "+v"(r[0][0].v), "+v"(r[0][1].v), "+v"(r[1][0].v), "+v"(r[1][1].v), "+v"(r[2][0].v), "+v"(r[2][1].v), "+v"(r[3][0].v), "+v"(r[3][1].v));
    else
        asm volatile("" : "+v"(r[0][0].v), "+v"(r[0][1].v), "+v"(r[1][0].v), "+v"(r[1][1].v));
}
template <int NT>
__device__ __forceinline__ void frag_touch(FragReg<false> (&r)[NT][2]) {
#pragma unroll
    for (int t = 0; t < NT; t += 2)
        asm volatile("" : "+v"(r[t][0].lo), "+v"(r[t][0].hi), "+v"(r[t][1].lo), "+v"(r[t][1].hi), "+v"(r[t + 1][0].lo), "+v"(r[t + 1][0].hi), "+v"(r[t + 1][1].lo),
                     "+v"(r[t + 1][1].hi));
}

#define DC_B256_WAIT_VM4() asm volatile("s_waitcnt vmcnt(4)" ::: "memory")
#define DC_B256_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// The main loop.  acc[mt][nt]: mt = 4 * (A half) + tile, nt = 2 * (B half) + tile; element j of lane l is
//   C[m0 + 128 group + 64 (mt >> 2) + 16 (mt & 3) + (l & 15)][n0 + 64 wcol + 32 (nt >> 1) + 16 (nt & 1) + 4 (l >> 4) + j].
// LA / LB: loaders with  static constexpr bool KC  and  issue(int half, char* sub_image, int k0, int kend, int wave).
template <class LA, class LB>
__device__ __forceinline__ void mainloop(LA& la, LB& lb, char* smem, int kbeg, int kend, f32x4 (&acc)[8][4]) {
    constexpr bool AKC = LA::KC, BKC = LB::KC;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // provably uniform: LDS-DMA destinations live in M0
    const int group = wave >> 2, wcol = wave & 3;
    const unsigned lds0 = (unsigned)(size_t)(DC_LDS char*)smem;
    FragAddr<AKC, 4> fa;
    FragAddr<BKC, 2> fb;
    fa.init(lds0 + OFF_A, group * 64, lane);
    fb.init(lds0 + OFF_B, wcol * 32, lane);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkt = (kend - kbeg + BK - 1) / BK;

    FragReg<AKC> A[4][2];               // the A half of the current phase pair
    FragReg<BKC> B0[2][2][2];           // B half 0, two register sets: the next K-tile's is read while this one's is still in use
    FragReg<BKC> B1[2][2];              // B half 1

    // quadrant (ah, bh): 16 MFMAs; B-side fragment as the MFMA's A operand (transposed 16 x 16 tiles: see the header)
    auto mma = [&](auto ah_c, auto bh_c, FragReg<BKC> (&Bq)[2][2]) {
        constexpr int AH = decltype(ah_c)::value, BH = decltype(bh_c)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[4 * AH + t][2 * BH + tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bq[tn][s].get(), A[t][s].get(), acc[4 * AH + t][2 * BH + tn], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---- prologue: K-tile 0 into stage 0 in the order the phases consume it (B half 0, A half 0, B half 1, A half 1)
    lb.issue(0, smem + OFF_B, kbeg, kend, wave);
    la.issue(0, smem + OFF_A, kbeg, kend, wave);
    lb.issue(1, smem + OFF_B + SUB, kbeg, kend, wave);
    la.issue(1, smem + OFF_A + SUB, kbeg, kend, wave);
    DC_B256_WAIT_VM4();                                  // this wave's pieces of B half 0 and A half 0 have landed
    __builtin_amdgcn_s_barrier();                        // ... and every wave's
    if (group == 1) __builtin_amdgcn_s_barrier();        // the stagger: group 1 runs one barrier behind group 0 from here on
    frag_read<BKC, 2, OFF_B - OFF_B>(fb, 0u, B0[0]);

    // One K-tile out of stage S (0 / 1); the next tile goes into the other stage, one quarter per phase.
    auto tile = [&](int kt, auto s_c) {
        constexpr int S = decltype(s_c)::value;
        constexpr unsigned CUR = S * STAGE;
        char* const nxt = smem + (1 - S) * STAGE;
        const int k1 = kbeg + (kt + 1) * BK;
        // ---- phase 1: quadrant (A half 0, B half 0)
        frag_read<AKC, 4, 0>(fa, CUR, A);
        lb.issue(0, nxt + OFF_B, k1, kend, wave);
        DC_B256_WAIT_VM4();                              // B half 1 of this tile (read in phase 2)
        __builtin_amdgcn_s_barrier();
        DC_B256_WAIT_LGKM0();
        frag_touch(A);
        frag_touch(B0[S]);
        mma(I0{}, I0{}, B0[S]);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: quadrant (A half 0, B half 1)
        frag_read<BKC, 2, SUB>(fb, CUR, B1);
        la.issue(0, nxt + OFF_A, k1, kend, wave);
        DC_B256_WAIT_VM4();                              // A half 1 of this tile (read in phase 3)
        __builtin_amdgcn_s_barrier();
        DC_B256_WAIT_LGKM0();
        frag_touch(B1);
        mma(I0{}, I1{}, B1);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: quadrant (A half 1, B half 1)
        frag_read<AKC, 4, SUB>(fa, CUR, A);
        lb.issue(1, nxt + OFF_B + SUB, k1, kend, wave);
        DC_B256_WAIT_VM4();                              // B half 0 of the NEXT tile (read in phase 4)
        __builtin_amdgcn_s_barrier();
        DC_B256_WAIT_LGKM0();
        frag_touch(A);
        mma(I1{}, I1{}, B1);
        __builtin_amdgcn_s_barrier();
        // ---- phase 4: quadrant (A half 1, B half 0); the next tile's B half 0 goes into the other register set
        frag_read<BKC, 2, 0>(fb, (unsigned)((1 - S) * STAGE), B0[1 - S]);
        la.issue(1, nxt + OFF_A + SUB, k1, kend, wave);
        DC_B256_WAIT_VM4();                              // A half 0 of the next tile (read in its phase 1)
        __builtin_amdgcn_s_barrier();
        mma(I1{}, I0{}, B0[S]);
        __builtin_amdgcn_s_barrier();
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        tile(kt, I0{});
        if (kt + 1 < nkt) tile(kt + 1, I1{});
    }
    if (group == 0) __builtin_amdgcn_s_barrier();        // pairs with group 1's last barrier
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the (empty) tile past the end and the last phase-4 reads
    frag_touch(B0[0]);
    frag_touch(B0[1]);
    __builtin_amdgcn_s_barrier();                        // LDS is free for the epilogue
}

// Epilogue straight from the accumulators: every (mt, nt) block is 16 rows x 16 columns, a lane owns 4 consecutive columns.
// Vector accesses only: the launchers route problems whose rows are not 16-byte addressable (N % 4 != 0, unaligned C /
// residual / scale / shift: Epilogue::vec4 == 0) to the 128-square kernel, so a lane's four columns are all inside N or all outside.
__device__ __forceinline__ void store_tile(f32x4 (&acc)[8][4], const Epilogue& ep, float* __restrict__ partial, int M, int N, int m0, int n0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int group = wave >> 2, wcol = wave & 3;
    const int rbase = m0 + 128 * group + (lane & 15), cbase = n0 + 64 * wcol + 4 * (lane >> 4);
    if (partial) {
        float* slab = partial + (long)blockIdx.z * M * N;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int row = rbase + 64 * (mt >> 2) + 16 * (mt & 3);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = cbase + 32 * (nt >> 1) + 16 * (nt & 1);
                if (row < M && col < N) *reinterpret_cast<f32x4*>(slab + (long)row * N + col) = acc[mt][nt];
            }
        }
        return;
    }
    f32x4 sc[4], sh[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = min(cbase + 32 * (nt >> 1) + 16 * (nt & 1), N - 4);
        sc[nt] = ep.scale ? *reinterpret_cast<const f32x4*>(ep.scale + col) : f32x4{1.f, 1.f, 1.f, 1.f};
        sh[nt] = ep.shift ? *reinterpret_cast<const f32x4*>(ep.shift + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int row = rbase + 64 * (mt >> 2) + 16 * (mt & 3);
        const int rowc = min(row, M - 1);
        const float* rr = ep.res_row(rowc);
        float* crow = ep.C + (long)rowc * ep.ldc;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = cbase + 32 * (nt >> 1) + 16 * (nt & 1);
            const bool live = row < M && col < N;
            const int colc = min(col, N - 4);
            f32x4 v = acc[mt][nt] * sc[nt] + sh[nt];
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

// tile id -> (tile row, tile column): ids walk down up to GM tile rows first (they share the B panel in the XCD's L2), then across
__device__ __forceinline__ void tile_coords(int lid, int tiles_m, int tiles_n, int& tm, int& tn) {
    constexpr int GM = 16;
    const int width = GM * tiles_n, g = lid / width, first = g * GM;
    const int gsz = min(tiles_m - first, GM), r = lid - g * width;
    tm = first + r % gsz;
    tn = r / gsz;
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(NTHREADS, 2) void bgemm256_kernel(BOperand a, BOperand b, Epilogue ep, int M, int N, int K, int klen, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    char* smem = reinterpret_cast<char*>(smem_f);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * klen, kend = min(K, kbeg + klen);
    Load<AKC, true> la;
    Load<BKC, false> lb;
    la.init(a, m0, lane, wave);
    lb.init(b, n0, lane, wave);
    f32x4 acc[8][4];
    mainloop(la, lb, smem, kbeg, kend, acc);
    store_tile(acc, ep, partial, M, N, m0, n0);
}

// split-K so that a small grid still covers the chip in ONE round of blocks (one block per CU: a grid of 1.1 rounds takes as long as
// one of 2); slices are multiples of the 64-deep K-tile
inline BSplit split(int M, int N, int K, int user_split) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int ktiles = (K + BK - 1) / BK;
    int s = user_split;
    if (s <= 0) {
        s = 1;
        if (tiles < (3 * kNumCU) / 4 && ktiles >= 16) {
            s = kNumCU / tiles;                           // floor: tiles * s <= 256
            if (s > ktiles / 8) s = ktiles / 8;           // >= 8 K-tiles (512 deep) per slice
            if (s > 32) s = 32;
            if (s < 1) s = 1;
        }
    }
    if (s > ktiles) s = ktiles;
    const int klen = ((ktiles + s - 1) / s) * BK;
    return BSplit{(K + klen - 1) / klen, klen};
}

// Which block tile is faster for this problem?  A small cost model fitted to measurements on MI355X (profiles/r03_bgemm_bench.txt,
// r03_bconv_bench.txt): rounds of blocks x (K-tiles per slice x time per K-tile + fixed cost per block) + the split-K slab traffic.
//   256 x 256 tile, one block per CU:    1.55 us per 64-deep K-tile (8.4 MFLOP), 4 us per block (prologue, epilogue)
//   128 x 128 tile, two blocks per CU:   1.13 us per K-tile and block (2.1 MFLOP, two co-resident), 3 us per block
//   split-K: (slices + 1) x M x N x 4 bytes at 4 TB/s + one more launch (3 us)
inline double tile_cost_us(int M, int N, int K, int tile, const BSplit& sp) {
    const long tiles = (long)((M + tile - 1) / tile) * ((N + tile - 1) / tile);
    const long blocks = tiles * sp.split, slots = tile == 256 ? kNumCU : 2 * kNumCU;
    const double rounds = (double)((blocks + slots - 1) / slots);
    const double kt = (double)(sp.klen / 64);
    double t = rounds * (tile == 256 ? kt * 1.55 + 4.0 : kt * 1.13 + 3.0);
    if (sp.split > 1) t += (sp.split + 1.0) * (double)M * N * 4.0 / 4e6 + 3.0;
    return t;
}

inline bool prefer(int M, int N, int K, int user_split, bool vec4 = true) {
    if (!vec4 || (N & 3) || M < 4 || N < 4) return false;       // the epilogue is 16-byte accesses only
    const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    if ((double)tiles * BM * BN > 1.25 * (double)M * N) return false;              // padded / useful outputs
    return tile_cost_us(M, N, K, 256, split(M, N, K, user_split)) < tile_cost_us(M, N, K, 128, bgemm_split(M, N, K, user_split));
}

template <bool AKC, bool BKC>
int launch(const BOperand& a, const BOperand& b, const Epilogue& ep, int M, int N, int K, int user_split, void* workspace, size_t workspace_bytes,
           hipStream_t stream) {
    const BSplit sp = split(M, N, K, user_split);
    float* partial = nullptr;
    if (sp.split > 1) {
        const size_t need = (size_t)sp.split * M * N * sizeof(float);
        DC_REQUIRE(workspace != nullptr && workspace_bytes >= need, DC_EWORKSPACE, "bgemm256 split-K needs %zu workspace bytes, got %zu", need, workspace_bytes);
        partial = static_cast<float*>(workspace);
    }
    DC_ENSURE_DYN_LDS((&bgemm256_kernel<AKC, BKC>), 160 * 1024);
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipLaunchKernelGGL((bgemm256_kernel<AKC, BKC>), dim3(tiles, 1, sp.split), dim3(NTHREADS), LDS_BYTES, stream, a, b, ep, M, N, K, sp.klen, partial);
    int rc = check_launch("bgemm256_kernel");
    if (rc) return rc;
    if (sp.split > 1) {
        const long total = (long)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(splitk_reduce_blocks(total)), dim3(256), 0, stream, partial, sp.split, M, N, ep);
        rc = check_launch("splitk_reduce_kernel");
    }
    return rc;
}

}  // namespace b256
}  // namespace dcap
