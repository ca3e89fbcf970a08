// bgemm_core.h -- the bf16 MFMA GEMM main loop (v_mfma_f32_32x32x16_bf16, fp32 accumulate) behind dc_gemm_bf16 and the
// bf16 legs of the fused vocabulary softmax / cross-entropy (vocab_ce.hip).  BASELINE configs[4] runs the RoI head, the
// caption decoder and the vocabulary layers with bf16 storage (fp32 master weights) on this loop.
//
// C[M,N] = A[M,K] * B[K,N]; block = 256 threads = 4 waves (2x2), block tile 128 x 128 x 64, each wave 64 x 64 as 2 x 2
// MFMA tiles of 32 x 32.  Operand tiles travel global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per
// wave-instruction, no staging registers, no address VALU per K-tile: per-lane byte offsets are fixed at kernel entry and
// the K advance rides in the instruction's SGPR offset); two LDS stages, the next tile's DMA is issued before the current
// tile's fragment reads and MFMAs, one wait + barrier per K-tile.  Two blocks per CU.
//
// Two LDS images, chosen per operand by where the contraction index lies in memory (all three layouts of a training step
// run: NN forward, NT data gradient, TN weight gradient):
//   KC  rows of the operand hold K contiguously ([rows][K]): image [128 rows][64 k] bf16 = 128-byte rows of eight 16-byte
//       chunks, chunk c of row r stored at chunk c ^ ((r >> 1) & 7): the 16-lane groups of a ds_read_b128 then cover 16
//       distinct 16-byte slots of the 256-byte bank row (conflict-free).  A lane's fragment (8 consecutive k of one row) is
//       ONE ds_read_b128.
//   MC  the operand is K-major in memory ([K][cols]: Keras [in,out] kernels, A^T and dY for weight gradients): image
//       [64 k][128 cols] = 256-byte rows of sixteen chunks, chunk c of row k stored at c ^ (((k & 3) << 2) | ((k >> 2) & 3));
//       a lane's fragment is two ds_read_b64_tr_b16 (the hardware's transposing LDS read: 4 k x 16 columns per 16 lanes).
// The LDS-DMA destination is lane-linear, so both swizzles are applied to the per-lane SOURCE address and to the read.
// Out-of-range handling is the buffer descriptor's: K tails (K % 64 != 0, K % 8 == 0 required) and split-K slice ends load
// hardware zeros (the lane's offset is replaced by an out-of-range one); rows / columns beyond M / N feed only outputs
// that are never stored.
#pragma once
#include "igemm_core.h"
#include <type_traits>

namespace dcap {

typedef __bf16 bh8 __attribute__((ext_vector_type(8)));
typedef __bf16 bh4 __attribute__((ext_vector_type(4)));
typedef short sh4 __attribute__((ext_vector_type(4)));
#define DC_LDS __attribute__((address_space(3)))

constexpr int BKB = 64;                  // K-tile depth (bf16 elements)
constexpr int BT = 128;                  // block tile edge (both M and N)
constexpr int B_IMG = BT * BKB * 2;      // bytes of one operand image (16 KiB)
constexpr int B_STAGE = 2 * B_IMG;
constexpr int B_NP = B_IMG / 1024 / 4;   // 1-KiB LDS-DMA pieces per wave and operand tile (4)

struct BOperand {
    const unsigned short* p;   // bf16 bit patterns
    long ld;                   // elements between consecutive rows in memory
    int extent;                // KC: number of rows (M or N); MC: number of columns (M or N)
    const int32_t* gather;     // optional: KC: tile row r is memory row gather[r]; MC: K row k is memory row gather[k]
    unsigned bytes;            // size of the addressed region (buffer range)
};

template <bool KC>
struct BLoad {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[B_NP];       // per-lane byte offset of the 16-byte chunk at K-tile 0
    int kloc[B_NP];            // KC: k of the chunk inside the tile; MC: k row inside the tile
    __device__ __forceinline__ void init(const BOperand& o, int origin, int lane, int wave) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(o.p), 0, (int)o.bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < B_NP; ++j) {
            const int pc = wave * B_NP + j;
            if constexpr (KC) {
                const int r = 8 * pc + (lane >> 3);                       // tile row of this lane's chunk
                const int c = (lane & 7) ^ ((r >> 1) & 7);                // source chunk that lands in LDS chunk (lane & 7)
                const int row = min(origin + r, o.extent - 1);            // rows past the edge feed nothing that is stored
                const long src = o.gather ? (long)o.gather[row] : (long)row;
                voff[j] = (unsigned)((src * o.ld + 8 * c) * 2);
                kloc[j] = 8 * c;
            } else {
                const int k = 4 * pc + (lane >> 4);                       // K row inside the tile
                const int c = (lane & 15) ^ (((k & 3) << 2) | ((k >> 2) & 3));
                const int col = min(origin + 8 * c, o.extent - 8);        // columns past the edge: clamped, never stored
                voff[j] = (unsigned)(((long)k * o.ld + col) * 2);
                if (o.gather) voff[j] = (unsigned)(col * 2);
                kloc[j] = k;
            }
        }
    }
    // issue the LDS-DMA of the K-tile starting at k0 into the image at `img` (wave-uniform LDS address)
    __device__ __forceinline__ void issue(const BOperand& o, char* img, int k0, int kend, int wave) const {
        const bool tail = k0 + BKB > kend;                                 // block-uniform
        if constexpr (KC) {
            const int soff = k0 * 2;
#pragma unroll
            for (int j = 0; j < B_NP; ++j) {
                const unsigned v = (tail && k0 + kloc[j] >= kend) ? kOobOffset : voff[j];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DC_LDS void*)(img + (wave * B_NP + j) * 1024), 16, (int)v, soff, 0, 0);
            }
        } else {
            if (o.gather) {                                                // gathered K rows (embedding-side weight gradient)
#pragma unroll
                for (int j = 0; j < B_NP; ++j) {
                    const int k = k0 + kloc[j];
                    const unsigned v = (k < kend) ? (unsigned)((long)o.gather[k] * o.ld * 2) + voff[j] : kOobOffset;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DC_LDS void*)(img + (wave * B_NP + j) * 1024), 16, (int)v, 0, 0, 0);
                }
                return;
            }
            const int soff = (int)((long)k0 * o.ld * 2);                   // < 4 GiB (host-checked span)
#pragma unroll
            for (int j = 0; j < B_NP; ++j) {
                const unsigned v = (tail && k0 + kloc[j] >= kend) ? kOobOffset : voff[j];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DC_LDS void*)(img + (wave * B_NP + j) * 1024), 16, (int)v, soff, 0, 0);
            }
        }
    }
};

// Per-lane LDS byte ADDRESSES of a wave's fragments inside the operand images of stage 0 (w0 = the wave's first row / column
// in the tile).  The fragment reads are inline asm: hipcc cannot tell a ds_read_b64_tr_b16 from the LDS-DMA in flight into
// the OTHER stage and drains the DMA (s_waitcnt vmcnt(0)) in front of the first one, which serialises load and compute; the
// asm reads carry their own counted s_waitcnt lgkmcnt below.
template <bool KC>
struct BFrag {
    static constexpr int READS = KC ? 2 : 4;       // ds instructions per k-step (two 32-row blocks)
    unsigned off[4];
    __device__ __forceinline__ void init(unsigned img_addr, int w0, int lane) {
        const int i = lane & 31, h = lane >> 5;
        if constexpr (KC) {
            // k-step s, 32-row block t: row r = w0 + 32 t + i, chunk 2 s + h  ->  byte (r * 128) + (((2 s + h) ^ key) * 16),
            // key = (r >> 1) & 7 = (i >> 1) & 7; the k-step only flips bits 5-6: off[s] = off[0] ^ (32 s); t adds 4096
            const unsigned base = (unsigned)((w0 + i) * 128 + ((h ^ ((i >> 1) & 7)) << 4));
#pragma unroll
            for (int s = 0; s < 4; ++s) off[s] = img_addr + (base ^ (unsigned)(32 * s));
        } else {
            // 16-lane group g = (lane >> 4) & 1 reads the 4 k x 16 column block at k = 16 s + 8 h + 4 hf, columns
            // w0 + 32 t + 16 g; lane 4 q + p of the group addresses row q, columns 4 p .. 4 p + 3 of the block
            const int g = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int k = 8 * h + 4 * hf + q;                     // + 16 s (adds 4096 bytes per s)
                    const int ch = (w0 + 32 * t) / 8 + 2 * g + (p >> 1);
                    const int key = ((k & 3) << 2) | ((k >> 2) & 3);
                    off[2 * t + hf] = img_addr + (unsigned)(256 * k + 16 * (ch ^ key) + 8 * (p & 1));
                }
        }
    }
};

typedef short sh8 __attribute__((ext_vector_type(8)));
struct BFragRegs {                   // one k-step's fragments of one operand: two 32-row blocks
    bh8 kc[2];
    sh4 lo[2], hi[2];
};

// issue the reads of k-step S (compile-time) from the stage at byte offset STAGE_OFF (compile-time) -- no waits
template <bool KC, int STAGE_OFF, int S>
__device__ __forceinline__ void bfrag_issue(const BFrag<KC>& f, BFragRegs& r) {
    if constexpr (KC) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r.kc[0]) : "v"(f.off[S]), "n"(STAGE_OFF));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r.kc[1]) : "v"(f.off[S]), "n"(STAGE_OFF + 4096));
    } else {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r.lo[0]) : "v"(f.off[0]), "n"(STAGE_OFF + S * 4096));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r.hi[0]) : "v"(f.off[1]), "n"(STAGE_OFF + S * 4096));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r.lo[1]) : "v"(f.off[2]), "n"(STAGE_OFF + S * 4096));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r.hi[1]) : "v"(f.off[3]), "n"(STAGE_OFF + S * 4096));
    }
}
template <bool KC>
__device__ __forceinline__ bh8 bfrag_get(const BFragRegs& r, int t) {
    if constexpr (KC) return r.kc[t];
    else return __builtin_bit_cast(bh8, sh8{r.lo[t][0], r.lo[t][1], r.lo[t][2], r.lo[t][3], r.hi[t][0], r.hi[t][1], r.hi[t][2], r.hi[t][3]});
}

constexpr size_t bgemm_lds_bytes() {
    constexpr size_t stages = 2 * (size_t)B_STAGE;
    constexpr size_t cimage = (size_t)BT * (BT + 4) * sizeof(float);     // epilogue transpose image (store_tile)
    return stages > cimage ? stages : cimage;
}

template <int N>
__device__ __forceinline__ void lgkm_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
    __builtin_amdgcn_sched_barrier(0);             // hipcc may hoist a register-only MFMA above an asm wait otherwise
}

// The main loop: accumulates A[m0.., kbeg..kend) * B[kbeg..kend), n0..] into acc (MFMA layout, wave origin wm, wn).
// LA / LB: operand loaders with  static constexpr bool KC  (which LDS image they fill) and
//   issue(char* img, int k0, int kend, int wave)  -- the LDS-DMA of one K-tile (BLoadOp below wraps BLoad + its operand;
//   BLoadIm2col gathers the K-major im2col matrix of a weight gradient).
template <class LA, class LB>
__device__ __forceinline__ void bgemm_mainloop_t(LA& la, LB& lb, char* smem, int kbeg, int kend, f32x16 (&acc)[2][2], int wm, int wn) {
    constexpr bool AKC = LA::KC, BKC = LB::KC;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // provably uniform: LDS-DMA destinations live in M0
    const unsigned lds0 = (unsigned)(size_t)(DC_LDS char*)smem;
    BFrag<AKC> fa;
    BFrag<BKC> fb;
    fa.init(lds0, wm, lane);
    fb.init(lds0 + B_IMG, wn, lane);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
    const int nkt = (kend - kbeg + BKB - 1) / BKB;
    la.issue(smem, kbeg, kend, wave);
    lb.issue(smem + B_IMG, kbeg, kend, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    constexpr int RA = BFrag<AKC>::READS, RB = BFrag<BKC>::READS, RS = RA + RB;      // ds reads per k-step
    auto mma = [&](const BFragRegs& ra, const BFragRegs& rb) {
        const bh8 a0 = bfrag_get<AKC>(ra, 0), a1 = bfrag_get<AKC>(ra, 1), b0 = bfrag_get<BKC>(rb, 0), b1 = bfrag_get<BKC>(rb, 1);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    };
    // one K-tile from the stage at CUR (byte offset, compile-time); the next tile's DMA goes to the other stage first
    auto tile = [&](int kt, auto cur_c) {
        constexpr int CUR = decltype(cur_c)::value;
        if (kt + 1 < nkt) {                                               // the other stage was released by the last barrier
            const int k0 = kbeg + (kt + 1) * BKB;
            la.issue(smem + (B_STAGE - CUR), k0, kend, wave);
            lb.issue(smem + (B_STAGE - CUR) + B_IMG, k0, kend, wave);
        }
        BFragRegs a0, b0, a1, b1;                                         // k-steps ping-pong between the two register sets
        bfrag_issue<AKC, CUR, 0>(fa, a0);
        bfrag_issue<BKC, CUR, 0>(fb, b0);
        bfrag_issue<AKC, CUR, 1>(fa, a1);
        bfrag_issue<BKC, CUR, 1>(fb, b1);
        lgkm_wait<RS>();
        mma(a0, b0);
        bfrag_issue<AKC, CUR, 2>(fa, a0);
        bfrag_issue<BKC, CUR, 2>(fb, b0);
        lgkm_wait<RS>();
        mma(a1, b1);
        bfrag_issue<AKC, CUR, 3>(fa, a1);
        bfrag_issue<BKC, CUR, 3>(fb, b1);
        lgkm_wait<RS>();
        mma(a0, b0);
        lgkm_wait<0>();
        mma(a1, b1);
        // this wave's DMA of the next tile has landed; after the barrier every wave's has, and every wave is done reading CUR
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        tile(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < nkt) tile(kt + 1, std::integral_constant<int, B_STAGE>{});
    }
}

template <bool KC_>
struct BLoadOp {                                   // a dense operand: BLoad + the operand it reads
    static constexpr bool KC = KC_;
    BLoad<KC_> l;
    BOperand o;
    __device__ __forceinline__ void init(const BOperand& op, int origin, int lane, int wave) { o = op; l.init(op, origin, lane, wave); }
    __device__ __forceinline__ void issue(char* img, int k0, int kend, int wave) const { l.issue(o, img, k0, kend, wave); }
};

template <bool AKC, bool BKC>
__device__ __forceinline__ void bgemm_mainloop(const BOperand& a, const BOperand& b, char* smem, int m0, int n0, int kbeg, int kend,
                                               f32x16 (&acc)[2][2], int wm, int wn) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    BLoadOp<AKC> la;
    BLoadOp<BKC> lb;
    la.init(a, m0, lane, wave);
    lb.init(b, n0, lane, wave);
    bgemm_mainloop_t(la, lb, smem, kbeg, kend, acc, wm, wn);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of a convolution on the bf16 pipe:  dW[cout][(tap, ci)] = sum over output pixels p of
// dy[p][cout] * x[pixel(p) + tap][ci]  =  dy^T (K-major: rows = pixels) times the im2col matrix, K-major as well: K row p is the
// run of Cin channels of ONE (shifted) input pixel, so a 128-column tile inside one tap (Cin % 128 == 0) is a contiguous 256 bytes
// of x.  The loader keeps (n, oy, ox) of each of its four K rows and walks them 64 pixels per K-tile (add + wrap, no division);
// taps that fall outside the image and pixels past the end load hardware zeros.
// ------------------------------------------------------------------------------------------------
struct BIm2col {
    const unsigned short* x;       // bf16 [N, H, W, Cin]
    int H, W, Cin, Ho, Wo, stride, pad_t, pad_l, kw, P;      // P = N*Ho*Wo output pixels (the K extent)
    unsigned bytes;
    int ncols;                     // kh * kw * Cin: columns of the im2col matrix (the 256-tile loader clamps against it)
};

struct BLoadIm2col {
    static constexpr bool KC = false;
    __amdgpu_buffer_rsrc_t rsrc;
    BIm2col c;
    int n[B_NP], oy[B_NP], ox[B_NP];
    int dy, dx;                    // tap offset of this column tile (block-uniform)
    int ci_base;                   // first channel of this column tile inside its tap
    __device__ __forceinline__ void init(const BIm2col& cc, int col0, int kbeg, int lane, int wave) {
        c = cc;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(c.x), 0, (int)c.bytes, 0x00020000);
        const int tap = col0 / c.Cin, ci0 = col0 - tap * c.Cin;
        const int ky = tap / c.kw, kx = tap - ky * c.kw;
        dy = ky - c.pad_t;
        dx = kx - c.pad_l;
        ci_base = ci0;
#pragma unroll
        for (int j = 0; j < B_NP; ++j) {
            const int k = 4 * (wave * B_NP + j) + (lane >> 4);              // K row inside the tile
            const int p = kbeg + k;
            const int nn = p / (c.Ho * c.Wo), rem = p - nn * (c.Ho * c.Wo);
            n[j] = nn;
            oy[j] = rem / c.Wo;
            ox[j] = rem - oy[j] * c.Wo;
        }
    }
    __device__ __forceinline__ void issue(char* img, int k0, int kend, int wave) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int j = 0; j < B_NP; ++j) {
            const int k = 4 * (wave * B_NP + j) + (lane >> 4);
            const int ch = (lane & 15) ^ (((k & 3) << 2) | ((k >> 2) & 3));
            const int iy = oy[j] * c.stride + dy, ix = ox[j] * c.stride + dx;
            const bool in = (unsigned)iy < (unsigned)c.H && (unsigned)ix < (unsigned)c.W && k0 + k < min(c.P, kend);
            const unsigned off = (unsigned)(((((long)n[j] * c.H + iy) * c.W + ix) * c.Cin + ci_base + 8 * ch) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (DC_LDS void*)(img + (wave * B_NP + j) * 1024), 16, (int)(in ? off : kOobOffset), 0, 0, 0);
            ox[j] += BKB;                                                   // next K-tile: 64 pixels further along the row-major walk
            while (ox[j] >= c.Wo) { ox[j] -= c.Wo; ++oy[j]; }
            while (oy[j] >= c.Ho) { oy[j] -= c.Ho; ++n[j]; }
        }
    }
};

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256, 2) void bgemm_kernel(BOperand a, BOperand b, Epilogue ep, int M, int N, int K, int klen,
                                                    float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    char* smem = reinterpret_cast<char*>(smem_f);
    const int wave = threadIdx.x >> 6;
    const int tiles_n = (N + BT - 1) / BT;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * BT, n0 = (lid % tiles_n) * BT;
    const int kbeg = blockIdx.z * klen, kend = min(K, kbeg + klen);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
    bgemm_mainloop<AKC, BKC>(a, b, smem, m0, n0, kbeg, kend, acc, wm, wn);
    store_tile<BT, BT>(acc, smem_f, ep, partial, M, N, m0, n0, wm, wn);
}

struct BSplit {
    int split, klen;
};
// split-K so that small grids still put about two blocks on every CU; slices are multiples of the 64-deep K-tile
inline BSplit bgemm_split(int M, int N, int K, int user_split) {
    const int tiles = ((M + BT - 1) / BT) * ((N + BT - 1) / BT);
    const int ktiles = (K + BKB - 1) / BKB;
    int s = user_split;
    if (s <= 0) {
        s = 1;
        if (tiles < 2 * kNumCU && ktiles >= 8) {
            s = (2 * kNumCU + tiles - 1) / tiles;
            if (s > ktiles / 4) s = ktiles / 4;           // keep >= 4 K-tiles (256 deep) per slice
            if (s > 32) s = 32;
            if (s < 1) s = 1;
        }
    }
    if (s > ktiles) s = ktiles;
    const int klen = ((ktiles + s - 1) / s) * BKB;
    return BSplit{(K + klen - 1) / klen, klen};
}

template <bool AKC, bool BKC>
int launch_bgemm(const BOperand& a, const BOperand& b, const Epilogue& ep, int M, int N, int K, int user_split, void* workspace,
                 size_t workspace_bytes, hipStream_t stream) {
    const BSplit sp = bgemm_split(M, N, K, user_split);
    float* partial = nullptr;
    if (sp.split > 1) {
        const size_t need = (size_t)sp.split * M * N * sizeof(float);
        DC_REQUIRE(workspace != nullptr && workspace_bytes >= need, DC_EWORKSPACE, "bgemm split-K needs %zu workspace bytes, got %zu", need,
                   workspace_bytes);
        partial = static_cast<float*>(workspace);
    }
    DC_ENSURE_DYN_LDS((&bgemm_kernel<AKC, BKC>), 160 * 1024);
    const int tiles = ((M + BT - 1) / BT) * ((N + BT - 1) / BT);
    hipLaunchKernelGGL((bgemm_kernel<AKC, BKC>), dim3(tiles, 1, sp.split), dim3(256), bgemm_lds_bytes(), stream, a, b, ep, M, N, K, sp.klen, partial);
    int rc = check_launch("bgemm_kernel");
    if (rc) return rc;
    if (sp.split > 1) {
        const long total = (long)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(splitk_reduce_blocks(total)), dim3(256), 0, stream, partial, sp.split, M, N, ep);
        rc = check_launch("splitk_reduce_kernel");
    }
    return rc;
}

}  // namespace dcap
