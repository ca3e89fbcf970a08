// conv_wino.hip -- 3x3 / stride 1 / 'same' convolutions with FROZEN weights in the Winograd F(2x2, 3x3) form, fp32 throughout
// (round 3).  The encoder's 3x3 layers (ResNet 2b branches, the FPN output convolutions: 525 of the 870 GFLOP of a two-image
// pass) are bound by the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: 157 TFLOP/s); the minimal-filtering form needs 16
// products per 2x2 output tile and (cin, cout) pair where the direct form needs 36: 2.25x fewer MFMAs for the same layer.
//
//   U[xi]  = (G g G^T)[xi]                 per (cin, cout), xi = 4 a + b over the 4 x 4 transform positions: packed ONCE per weight
//   V[xi]  = (B^T d B)[xi]                 per (tile, cin):  d = the 4 x 4 input patch whose top-left pixel is (2 ty - 1, 2 tx - 1)
//   M[xi]  = V[xi] (tiles x cin) . U[xi] (cin x cout)        16 independent GEMMs on the matrix pipe
//   Y      = A^T M A                       2 x 2 output pixels per tile, then scale / shift / ReLU (frozen BN + bias folded)
//
// One kernel does all of it (nothing of V or M ever reaches HBM).  Two kernels; common to both: a work item = a group of tiles x
// 32 output channels x all 16 xi; a wave owns whole xi positions (no two waves of a tile group share a weight), so U goes from L2
// STRAIGHT INTO REGISTERS -- it is packed in fragment order, 1 KiB contiguous per wave-instruction; the MFMA takes U as its A
// operand (rows = output channels), so a lane ends up with four consecutive output channels of one tile per accumulator quad;
// the output transform goes through LDS, thread (tile, channel quad) applies A^T . A and the epilogue and stores 16-byte pieces
// (128 contiguous bytes per pixel and item).  Items are ordered slice-major through the XCD remap, so an XCD's L2 keeps ONE
// slice of U (16 x Cin x 32 x 4 bytes).
//
// What bounds the form is not the matrix pipe alone: per MFMA it moves 2.25x the operand bytes of the direct one (U is 16/9 of
// the kernel and is re-read by every tile group; 4 x 4 patches overlap), a CU takes 66-73 GB/s from L2, and every vector-memory
// instruction costs the issuing wave tens of cycles (MI355X_MICROARCH.md).
//   The first version (round 3: thread (tile, channel quad) loads its 4 x 4 patch into registers, transforms it and writes a V image,
//     two blocks per CU) asked L2 for 128 KB per 1.05 MFLOP-chunk and block: 810 us on fpn_p2, where the MFMAs alone need 437; it
//     was removed in round 5 (profiles/r03_winograd_bench.txt keeps its numbers).
//   wino64_kernel / wino32_kernel = wino_body<2 / 1> (HISTORY.md section 5 has the measurements that led here): 64 / 32 tiles per
//     item, the input patch staged ONCE per 32 channels by LDS-DMA, no V image -- every wave makes its own MFMA fragments straight
//     from the patch --, persistent blocks, vector-memory instructions issued one per MFMA: 639 us.  fp32 products: 64-tile items
//     (512 threads, one block per CU) where every CU gets one, 32-tile items (256 threads, two blocks per CU) for the small layers;
//     split-bf16 products (the default plan): 32-tile items everywhere since round 6 (conv_winograd_tiles below).
#include "igemm_bf16s.h"
#include <algorithm>

namespace dcap {
namespace wino {

constexpr int NTHREADS = 256;

struct Args {
    const float* x;
    const f4* u;              // packed by wino_pack_kernel
    float* y;
    const float* scale;
    const float* shift;
    int N, H, W, Cin, Cout, relu;
    int gy, gx, groups;       // tile groups per image (rows, columns), in all
    unsigned x_bytes, u_bytes;
};

// U in fragment order, K in chunks of 16: f4 index ((((xi * NTG + ntg) * KC16 + kc) * 2 + j) * 64 + lane), lane = 32 h + i,
// component e   <->   cout = 32 ntg + i, cin = 16 kc + 8 h + 4 j + e     (MFMA step 4 j + e of a chunk contracts cin 8 h + 4 j + e)
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout) {
    const long total = (long)Cin * Cout;
    const int NTG = Cout / 32, KC = Cin / 16;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cin = (int)(idx % Cin), cout = (int)(idx / Cin);
        float g[3][3];
#pragma unroll
        for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = w[((long)cout * 9 + t) * Cin + cin];
        float gg[4][3];                                   // G g
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gg[0][c] = g[0][c];
            gg[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
            gg[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
            gg[3][c] = g[2][c];
        }
        const int ntg = cout >> 5, i = cout & 31, kc = cin >> 4, kk = cin & 15, h = kk >> 3, j = (kk & 7) >> 2, e = kk & 3;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float r[4] = {gg[a][0], 0.5f * (gg[a][0] + gg[a][1] + gg[a][2]), 0.5f * (gg[a][0] - gg[a][1] + gg[a][2]), gg[a][2]};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xi = 4 * a + b;
                u[((((long)(xi * NTG + ntg) * KC + kc) * 2 + j) * 64 + (h * 32 + i)) * 4 + e] = r[b];
            }
        }
    }
}

// The same U for the split-bf16 kernels (wino_body<.., true>): every element as three bf16 pieces (x = p0 + p1 + p2, rounded to
// nearest even, remainders exact) in the fragment order of v_mfma_f32_32x32x16_bf16: ushort index
//   (((((xi * NTG + ntg) * KC16 + kc) * 3 + piece) * 64 + lane) * 8 + jj),  lane = 32 h + i  <->  cout = 32 ntg + i, cin = 16 kc + 8 h + jj
__global__ void wino_pack_b3_kernel(const float* __restrict__ w, unsigned short* __restrict__ u, int Cin, int Cout) {
    const long total = (long)Cin * Cout;
    const int NTG = Cout / 32, KC = Cin / 16;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cin = (int)(idx % Cin), cout = (int)(idx / Cin);
        float g[3][3];
#pragma unroll
        for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = w[((long)cout * 9 + t) * Cin + cin];
        float gg[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gg[0][c] = g[0][c];
            gg[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
            gg[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
            gg[3][c] = g[2][c];
        }
        const int ntg = cout >> 5, i = cout & 31, kc = cin >> 4, kk = cin & 15, h = kk >> 3, jj = kk & 7;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float r[4] = {gg[a][0], 0.5f * (gg[a][0] + gg[a][1] + gg[a][2]), 0.5f * (gg[a][0] - gg[a][1] + gg[a][2]), gg[a][2]};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xi = 4 * a + b;
                float x = r[b];
#pragma unroll
                for (int piece = 0; piece < 3; ++piece) {
                    const unsigned short pb = __builtin_bit_cast(unsigned short, (__bf16)x);
                    x -= __uint_as_float((unsigned)pb << 16);
                    u[((((long)(xi * NTG + ntg) * KC + kc) * 3 + piece) * 64 + (h * 32 + i)) * 8 + jj] = pb;
                }
            }
        }
    }
}

// [16 xi][32 tiles][32 floats]: 128-byte rows, 16-byte chunk c of row r at c ^ ((r >> 1) & 7): the M (output transform) image.
__device__ __forceinline__ unsigned img32_addr(int xi, int tile, int chunk) {
    return (unsigned)((((xi * 32 + tile) << 3) + (chunk ^ ((tile >> 1) & 7))) << 4);
}
// the column half of B^T d B (the row half is one FMA per value in wino_body)
__device__ __forceinline__ void bt_cols(const f4 (&t)[4], f4 (&v)[4]) {
    v[0] = t[0] - t[2];
    v[1] = t[1] + t[2];
    v[2] = t[2] - t[1];
    v[3] = t[1] - t[3];
}

// ------------------------------------------------------------------------------------------------------------------------
// wino_body<HALVES>: 32 HALVES tiles (HALVES = 2: 8 x 8 tiles = 16 x 16 output pixels, 512 threads = 8 waves, one block per CU;
// described below.  HALVES = 1: 4 x 8 tiles, 256 threads, a 10 x 18 patch, two blocks per CU) x 32 output channels.
//
// No V image at all.  The block's 18 x 18-pixel input patch goes to LDS ONCE per 32 channels -- LDS-DMA, whole 128-byte lines,
// every byte of the patch fetched once per block (the register-staged transform of wino32_kernel asks for every interior pixel
// four times, and with 16-channel chunks for every line twice) -- double-buffered, the next 32 channels in flight for two chunks.
// Wave (a, th) owns the transform row a (xi = 4 a + b, b = 0..3) of the 32-tile half th and makes ITS OWN MFMA operand fragments
// straight from the patch: lane (tile, k-half) reads the two patch rows that B^T's row a combines (8 ds_read_b128 per 4 channels
// x 2), forms t = d[r1] +- d[r2] (row a of B^T d: one FMA per value) and the four column combinations v[b] -- exactly the V[xi]
// values its next 16 MFMAs take as B operands.  Nothing is transformed twice, nothing transformed is stored.
// Per 16-channel chunk and wave: 16 ds_read_b128, 64 VALU, 8 global_load_dwordx4 of U (one chunk ahead), 32 MFMAs; two waves
// per SIMD (the two tile halves of one row: the same U lines, the second finds them in L1), one barrier per 32 channels.
// Persistent: one block per CU walks the work items; the next item's first patch and U loads are requested before this item's
// output transform.  Inside a pair of chunks the vector-memory instructions go out one per MFMA (sched_group_barrier), the
// transform's VALU work four per MFMA: issued in bursts they stalled both waves of a SIMD at the same time.
// Patch image: pixel p = 128 bytes = eight 16-byte chunks, chunk c at c ^ ((p >> 1) & 7): the DMA permutes the SOURCE chunk per
// lane (its LDS destination is lane-linear), the eight tiles of a row then read eight different chunks.
// ------------------------------------------------------------------------------------------------------------------------
namespace wp {
constexpr int PW = 18, NDMA = 6;                         // patch width (8 tiles + halo) in pixels; DMA pieces per thread and pair
// Patch image: pixel (y, x) of the patch is 128 bytes at index y * 18 + (x >> 1) + 9 (x & 1) (a row's even columns first,
// so that horizontally neighbouring TILES alternate between the two halves of the 256-byte bank row), its 16-byte channel chunk c
// at position c ^ swizzle(y, x).  A ds_read_b128 is served in 16-lane groups that hold four tile rows x four consecutive tile
// columns (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, ...): per half of the bank row that is 4 rows x 2 columns two tiles
// apart, which the swizzle's bits ((y >> 1) & 3, (x >> 2) & 1) tell apart whatever patch row / column (r, c) is being read:
// conflict-free.  (First version: index y * 18 + x, swizzle (index >> 1) & 7: SQ_LDS_BANK_CONFLICT = 74 % of SQ_LDS_IDX_ACTIVE.)
__device__ __forceinline__ int patch_index(int y, int x) { return y * PW + (x >> 1) + (PW / 2) * (x & 1); }
__device__ __forceinline__ int patch_swizzle(int y, int x) { return ((x >> 2) & 1) | (((y >> 1) & 3) << 1); }
// HALVES 32-tile halves per work item: 2 = 64 tiles (8 x 8), 512 threads, one block per CU; 1 = 32 tiles (4 rows x 8), 256 threads,
// two blocks per CU -- for layers with too few tiles to give every CU a 64-tile item (stage 5, fpn_p5, one-image batches).
template <int HALVES>
struct Cfg {
    static constexpr int NT = 256 * HALVES, TGY = 4 * HALVES, TGX = 8, PH = 2 * TGY + 2, NPIX = PH * PW, SLOTS = NPIX * 8;
    static constexpr int BUF = 32768 * HALVES;           // one patch buffer (HALVES = 2: 41.5 KiB used); buffer 1 = buffer 0 ^ BUF
    static constexpr int LDS_BYTES = 2 * BUF;
    static_assert(NDMA * NT >= SLOTS && NDMA * NT * 16 <= BUF, "patch buffer; the kernel issues the pieces three per step");
};
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// eight fp32 values (channels 8 h + 0..3 in va, 8 h + 4..7 in vb) -> the three bf16x8 pieces of an MFMA operand fragment
__device__ __forceinline__ void split8(const f4& va, const f4& vb, u32x4& p0, u32x4& p1, u32x4& p2) {
    unsigned a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;
    split_pair(va[0], va[1], a0, a1, a2);
    split_pair(va[2], va[3], b0, b1, b2);
    split_pair(vb[0], vb[1], c0, c1, c2);
    split_pair(vb[2], vb[3], d0, d1, d2);
    p0 = u32x4{a0, b0, c0, d0};
    p1 = u32x4{a1, b1, c1, d1};
    p2 = u32x4{a2, b2, c2, d2};
}
__device__ __forceinline__ f32x16 mfma_b(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// B3 = true (round 5): the 16 products per tile on the BF16 matrix pipe in split arithmetic -- U pre-split once per frozen weight into
// three bf16 pieces (wino_pack_b3_kernel), the transformed patch values V split in registers, six v_mfma_f32_32x32x16_bf16 products
// per fp32 product (p0 q0 + p0 q1 + p1 q0 + p0 q2 + p1 q1 + p2 q0, small terms first, fp32 accumulation: the arithmetic of
// DC_MATH_BF16X3, igemm_bf16s.h, which holds the fp32 kernel's tolerances).  Per 16-channel chunk and wave: 24 MFMAs of 32 cycles
// instead of 32 of 64 (the fp32 matrix pipe's floor per 32-channel pair 3.41 us -> 1.28); the patch staging, the transform, the
// output transform and the persistent item walk are the fp32 kernel's.  A lane's fragment of a K = 16 MFMA is its 8 channels
// 8 h .. 8 h + 7 of the chunk: the values of the fp32 kernel's steps j = 0 and j = 1 side by side.
template <int HALVES, bool B3>
__device__ __forceinline__ void wino_body(const Args& a) {
    using namespace wp;
    typedef Cfg<HALVES> C;
    constexpr int NT64 = C::NT, SLOTS = C::SLOTS, BUF = C::BUF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave & 3, th = wave >> 2;
    const int gpi = a.gy * a.gx;
    const int NTG = a.Cout >> 5, KC = a.Cin >> 4, KP = a.Cin >> 5;
    const int total = a.groups * NTG;                      // work items (slice, tile group); this block takes blockIdx.x, + gridDim.x, ...
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);

    // ---- per work item: coordinates, patch DMA sources (slot s = n * 512 + tid of the image <-> pixel s >> 3, image chunk
    // s & 7 = source chunk ^ swizzle), U fragment base
    int nb, img, gyi, gxi;
    unsigned doff[NDMA];
    const f4* ubase;
    unsigned ubyte3;                                       // B3: byte offset of this wave row's fragments for the item (block-uniform)
    auto setup = [&](int item) {
        const int bid = xcd_remap(item, total);            // gridDim.x % 8 == 0: item % 8 is this block's XCD for every item it takes
        nb = bid / a.groups;
        const int g = bid - nb * a.groups;
        img = g / gpi;
        const int gr = g - img * gpi;
        gyi = gr / a.gx;
        gxi = gr - gyi * a.gx;
#pragma unroll
        for (int n = 0; n < NDMA; ++n) {
            const int s = n * NT64 + tid, pq = s >> 3, py = pq / PW, rem = pq - py * PW;
            const int px = rem < PW / 2 ? 2 * rem : 2 * (rem - PW / 2) + 1;          // inverse of patch_index(): even columns first
            const int iy = gyi * 2 * C::TGY - 1 + py, ix = gxi * 2 * C::TGX - 1 + px;
            const bool in = s < SLOTS && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const int c8 = (s & 7) ^ patch_swizzle(py, px);
            doff[n] = in ? (unsigned)(((((long)img * a.H + iy) * a.W + ix) * a.Cin + 4 * c8) * 4) : kOobOffset;
        }
        ubase = a.u + lane + (((long)(4 * wa) * NTG + nb) * KC << 7);
        ubyte3 = (unsigned)((4 * wa) * NTG + nb) * (unsigned)KC * 3072u;
    };
    auto issue_dma_pieces = [&](int pair, int buf, int n0, int n1) {
#pragma unroll
        for (int n = n0; n < n1; ++n)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + buf * BUF + (n * NT64 + wave * 64) * 16), 16,
                                                     (int)doff[n], pair * 128, 0, 0);
    };
    auto issue_dma = [&](int pair, int buf) { issue_dma_pieces(pair, buf, 0, NDMA); };
    const long ustep = (long)NTG * KC << 7;                // xi -> xi + 1
    auto load_u_part = [&](f4 (&dst)[4][2], int kc, int b0, int b1) {
#pragma unroll
        for (int b = b0; b < b1; ++b) {
            const f4* p = ubase + b * ustep + ((long)kc << 7);
            dst[b][0] = p[0];
            dst[b][1] = p[64];
        }
    };
    auto load_u = [&](f4 (&dst)[4][2], int kc) { load_u_part(dst, kc, 0, 4); };
    // B3: U through a buffer resource -- the lane's offset (16 lane) is a fixed VGPR, everything else (item, position, chunk, piece) one
    // block-uniform SGPR offset: no address VALU beside the loads
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4*>(a.u), 0, (int)a.u_bytes, 0x00020000);
    const unsigned ustep3 = (unsigned)NTG * KC * 3 * 1024u;                  // bytes: xi -> xi + 1
    auto load_u3 = [&](u32x4 (&dst)[3], int kc, int b) {
        const unsigned so = __builtin_amdgcn_readfirstlane(ubyte3 + (unsigned)b * ustep3 + (unsigned)kc * 3072u);
        dst[0] = __builtin_bit_cast(u32x4, buf_f4s(rsrc_u, (unsigned)lane * 16u, so));
        dst[1] = __builtin_bit_cast(u32x4, buf_f4s(rsrc_u, (unsigned)lane * 16u, so + 1024u));
        dst[2] = __builtin_bit_cast(u32x4, buf_f4s(rsrc_u, (unsigned)lane * 16u, so + 2048u));
    };

    // ---- fragment geometry: lane (tile i of this wave's half, k-half h); transform row wa combines patch rows r1, r2
    const int fi = lane & 31, fh = lane >> 5;
    const int tile = 32 * th + fi, tyl = tile >> 3, txl = tile & 7;
    const int r1 = (wa == 0) ? 0 : (wa == 2 ? 2 : 1), r2 = (wa == 0 || wa == 1) ? 2 : (wa == 2 ? 1 : 3);
    const float sgn = (wa == 1) ? 1.f : -1.f;
    // byte address of patch pixel (row rr, column c) of this lane's tile, image chunk 2 h (step 0 of a pair), in the buffer being read:
    // the chunk of step st = 2 half + j is 4 half + 2 h + j, i.e. this address ^ (j << 4) ^ (half << 6) -- the swizzle is an XOR too
    unsigned paddr[2][4];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int py = 2 * tyl + (rr ? r2 : r1), px = 2 * txl + c;
            paddr[rr][c] = (unsigned)(patch_index(py, px) * 128 + (((2 * fh) ^ patch_swizzle(py, px)) << 4));
        }
    // step st = 2 half + j of a pair: channels 16 half + 8 h + 4 j .. + 3 = image chunk 4 half + 2 h + j
    auto read_d = [&](f4 (&d)[2][4], int st) {
        const unsigned k = (unsigned)(((st & 1) << 4) ^ ((st >> 1) << 6));
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int c = 0; c < 4; ++c) d[rr][c] = *reinterpret_cast<const f4*>(smem + (paddr[rr][c] ^ k));
    };
    auto transform = [&](const f4 (&d)[2][4], f4 (&v)[4]) {
        f4 t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] = d[1][c] * sgn + d[0][c];         // row wa of B^T d
        bt_cols(t, v);
    };

    int item = blockIdx.x;
    if (item >= total) return;                             // (block-uniform; only when a tiny layer is forced onto this kernel)
    setup(item);
    issue_dma(0, 0);
    f4 ub[2][4][2];
    u32x4 u3[4][3];                                        // B3: the chunk's U fragments, refilled in place one chunk ahead
    if constexpr (B3) {
#pragma unroll
        for (int b = 0; b < 4; ++b) load_u3(u3[b], 0, b);
    } else {
        load_u(ub[0], 0);
    }
    for (;;) {
        f32x16 acc[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        auto pair = [&](int p, auto more_c) {
            constexpr bool more = decltype(more_c)::value;      // a pair p + 1 of this item exists: request its patch
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the patch (and its U fragments) have landed
            __syncthreads();                               // every wave's have; every wave is done with the other buffer
            if constexpr (B3) {
                // Eight position steps q = 4 half + b per pair.  Step q issues the six MFMAs of position b on the operands prepared during
                // step q - 1 and, BETWEEN them (one MFMA : nine VALU, sched_group_barrier -- a 32x32x16 MFMA holds the vector issue for 8 of
                // its 32 cycles, so the six dependent MFMAs of a position leave room for ~36 VALU instructions that would otherwise run
                // after them), prepares position q + 1: the column combination of t (8 VALU) and the three-piece split (44).  The second
                // half's patch rows are read during steps 1 and 2 and combined at the top of step 3; the next pair's patch goes out one
                // DMA piece per step; U fragments are refilled in place right behind their MFMAs, four steps ahead of their use.
                f4 dA[2][4], dB[2][4], t0[4], t1[4];
                u32x4 V[2][3];
                auto make_t = [&]() {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        t0[c] = dA[1][c] * sgn + dA[0][c];          // row wa of B^T d, channels 8 h + 0..3 and 8 h + 4..7
                        t1[c] = dB[1][c] * sgn + dB[0][c];
                    }
                };
                auto prep = [&](u32x4 (&dst)[3], int b) {
                    const f4 va = b == 0 ? t0[0] - t0[2] : (b == 1 ? t0[1] + t0[2] : (b == 2 ? t0[2] - t0[1] : t0[1] - t0[3]));
                    const f4 vb = b == 0 ? t1[0] - t1[2] : (b == 1 ? t1[1] + t1[2] : (b == 2 ? t1[2] - t1[1] : t1[1] - t1[3]));
                    split8(va, vb, dst[0], dst[1], dst[2]);
                };
                read_d(dA, 0);
                read_d(dB, 1);
                make_t();
                prep(V[0], 0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int half = q >> 2, b = q & 3;
                    if (q == 1) read_d(dA, 2);
                    if (q == 2) read_d(dB, 3);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (more) if (q < NDMA) issue_dma_pieces(p + 1, (p + 1) & 1, q, q + 1);
                    const u32x4 (&vq)[3] = V[q & 1];
                    acc[b] = mfma_b(u3[b][0], vq[2], acc[b]);       // small terms first
                    acc[b] = mfma_b(u3[b][1], vq[1], acc[b]);
                    acc[b] = mfma_b(u3[b][2], vq[0], acc[b]);
                    acc[b] = mfma_b(u3[b][1], vq[0], acc[b]);
                    acc[b] = mfma_b(u3[b][0], vq[1], acc[b]);
                    acc[b] = mfma_b(u3[b][0], vq[0], acc[b]);
                    if (q == 3) make_t();                           // (the first half's t is dead: position 3 was prepared during step 2)
                    if (q < 7) prep(V[(q + 1) & 1], (q + 1) & 3);
                    load_u3(u3[b], min(2 * p + half + 1, KC - 1), b);          // (the last chunk re-reads: uniform counts)
#pragma unroll
                    for (int m = 0; m < 6; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one vector-memory instruction (the DMA piece), where there is one
                        if (q == 3) __builtin_amdgcn_sched_group_barrier(0x002, 15, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int c = 0; c < 4; ++c) paddr[rr][c] ^= (unsigned)BUF;
                return;
            }
            // four steps of 16 MFMAs.  The patch reads of step st + 1 are issued at the top of step st, ahead of its first eight MFMAs;
            // the transform of step st + 1 is VALU work placed beside the last eight.  The scheduling fences keep the compiler from
            // sinking the reads to just before their use (it did: every step then opened with an exposed LDS round trip).
            f4 d[2][4], vcur[4], vnext[4];
            read_d(d, 0);
            transform(d, vcur);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int half = st >> 1, j = st & 1;
                if (st < 3) read_d(d, st + 1);
                __builtin_amdgcn_sched_barrier(0);
                // vector-memory instructions cost the wave tens of issue cycles each (an LDS-DMA piece 60 - 185: MI355X_MICROARCH.md):
                // they go out ONE per MFMA inside the first eight of a step -- the next pair's patch in steps 0 and 1 (three pieces
                // each), the U fragments of the next 16-channel chunk half per step -- instead of in a burst behind the barrier, where
                // both waves of a SIMD issued theirs at the same time with no MFMA in flight
                if constexpr (more) if (st < 2) issue_dma_pieces(p + 1, (p + 1) & 1, 3 * st, 3 * st + 3);
                load_u_part(ub[half ^ 1], min(2 * p + half + 1, KC - 1), 2 * j, 2 * j + 2);          // (the last one re-reads: uniform counts)
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ub[half][b][j][e], vcur[b][e], acc[b], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);          // one vector-memory read (where there is one left)
                }
                __builtin_amdgcn_sched_barrier(0);
                if (st < 3) transform(d, vnext);
#pragma unroll
                for (int e = 2; e < 4; ++e)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ub[half][b][j][e], vcur[b][e], acc[b], 0, 0, 0);
                if (st < 3) {                              // the transform's VALU work spread over the eight MFMAs: none of them waits for it
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (st < 3) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) vcur[b] = vnext[b];
                }
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int c = 0; c < 4; ++c) paddr[rr][c] ^= (unsigned)BUF;      // the other buffer
        };
        for (int p = 0; p < KP - 1; ++p) pair(p, std::true_type{});
        pair(KP - 1, std::false_type{});
        if (KP & 1) {                                      // every item starts in buffer 0
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int c = 0; c < 4; ++c) paddr[rr][c] ^= (unsigned)BUF;
        }

        // ---- this item is summed.  Start the next item's first patch and U loads, then finish this one beside them.
        const int onb = nb, oimg = img, ogy = gyi, ogx = gxi;
        const int next = item + (int)gridDim.x;
        // this item's epilogue operands, requested BEFORE anything of the next item: vmcnt retires in issue order, so waiting for them
        // later must not mean waiting for the next patch
        const int oq = tid & 7, ocout = onb * 32 + 4 * oq;
        f4 sc = (f4)(1.f), sh = (f4)(0.f);
        if (a.scale) sc = *reinterpret_cast<const f4*>(a.scale + ocout);
        if (a.shift) sh = *reinterpret_cast<const f4*>(a.shift + ocout);
        __syncthreads();                                   // every wave is done with both patch buffers
        if (next < total) {
            setup(next);
            issue_dma(0, 0);
            if constexpr (B3) {
#pragma unroll
                for (int b = 0; b < 4; ++b) load_u3(u3[b], 0, b);
            } else {
                load_u(ub[0], 0);
            }
        }
        // output transform.  In registers: the column half (A applied to this wave's row a): s[qx] from M[a][0..3]; through LDS (the
        // SECOND buffer: the first is being filled): the row half across the four waves of a tile half.  Plane 2 a + qx of half th.
        {
            const f32x16 s0 = acc[0] + acc[1] + acc[2], s1 = acc[1] - acc[2] - acc[3];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f4 m0 = {s0[4 * gq], s0[4 * gq + 1], s0[4 * gq + 2], s0[4 * gq + 3]};
                const f4 m1 = {s1[4 * gq], s1[4 * gq + 1], s1[4 * gq + 2], s1[4 * gq + 3]};
                *reinterpret_cast<f4*>(smem + BUF + th * 32768 + img32_addr(2 * wa, fi, 2 * gq + fh)) = m0;      // couts 8 gq + 4 h .. + 3 of tile fi
                *reinterpret_cast<f4*>(smem + BUF + th * 32768 + img32_addr(2 * wa + 1, fi, 2 * gq + fh)) = m1;
            }
        }
        __syncthreads();
        {
            const int oth = tid >> 8, ti = (tid >> 3) & 31;
            f4 sv[4][2];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int qx = 0; qx < 2; ++qx) sv[r][qx] = *reinterpret_cast<const f4*>(smem + BUF + oth * 32768 + img32_addr(2 * r + qx, ti, oq));
            const int otile = 32 * oth + ti, ty = ogy * C::TGY + (otile >> 3), tx = ogx * C::TGX + (otile & 7), cout = ocout;
#pragma unroll
            for (int qx = 0; qx < 2; ++qx) {
                const f4 o[2] = {sv[0][qx] + sv[1][qx] + sv[2][qx], sv[1][qx] - sv[2][qx] - sv[3][qx]};
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    f4 val = o[pr] * sc + sh;
                    if (a.relu) val = f4{fmaxf(val[0], 0.f), fmaxf(val[1], 0.f), fmaxf(val[2], 0.f), fmaxf(val[3], 0.f)};
                    if (2 * ty + pr < a.H && 2 * tx + qx < a.W)
                        *reinterpret_cast<f4*>(a.y + (((long)oimg * a.H + 2 * ty + pr) * a.W + 2 * tx + qx) * a.Cout + cout) = val;
                }
            }
        }
        if (next >= total) break;
        item = next;                                       // (the next pair loop opens with a barrier: the M image is read before buffer 1 is refilled)
    }
}

__global__ __launch_bounds__(512, 1) void wino64_kernel(Args a) { wino_body<2, false>(a); }
__global__ __launch_bounds__(256, 2) void wino32_kernel(Args a) { wino_body<1, false>(a); }
__global__ __launch_bounds__(512, 1) void wino64b_kernel(Args a) { wino_body<2, true>(a); }
__global__ __launch_bounds__(256, 2) void wino32b_kernel(Args a) { wino_body<1, true>(a); }

}  // namespace wino

static bool wino_shape_ok(const dc_conv_desc* d, const void* u) {
    return u != nullptr && d->math == DC_MATH_F32 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 &&
           d->Ho == d->H && d->Wo == d->W && d->Cin % 32 == 0 && d->Cout % 32 == 0 && d->res_mode == 0 && d->split_k <= 1 && aligned16(d->y) &&
           aligned16(u) && (!d->scale || aligned16(d->scale)) && (!d->shift || aligned16(d->shift)) &&
           (size_t)d->N * d->H * d->W * d->Cin * sizeof(float) < (1ull << 31);      // one buffer resource, 32-bit offsets: larger inputs take the direct kernels
}

// w_wino_b3 (the split-bf16 products) wins when both are given
static bool wino_b3(const dc_conv_desc* d) { return wino_shape_ok(d, d->w_wino_b3); }
bool conv_winograd_supported(const dc_conv_desc* d) { return wino_b3(d) || wino_shape_ok(d, d->w_wino); }
bool conv_winograd_split_bf16(const dc_conv_desc* d) { return wino_b3(d); }

// tiles per work item: 64 where every CU still gets an item, else 32.  DCAP_WINO_TILES = 32 / 64 forces one (tests).
static int wino_force() {
    static const int force = env_int("DCAP_WINO_TILES", 0);
    return force;
}

// CUs the PERSISTENT grids may occupy (a multiple of 8, one share per XCD).  A persistent block never yields its CU: with one block
// per CU on all 256 a kernel of another queue -- RCCL's all-reduce in a data-parallel run -- would wait for the whole encoder pass.
// dc_set_persistent_cus(n) / DCAP_WINO_CUS leave 256 - n CUs free for it (bench.py and ParallelModel set 248 when world > 1).
static std::atomic<int> g_persistent_cus{-1};
static int persistent_cus() {
    int v = g_persistent_cus.load(std::memory_order_relaxed);
    if (v < 0) {
        static const int env = env_int("DCAP_WINO_CUS", kNumCU);
        v = env;
    }
    v = std::max(8, std::min(v, kNumCU));
    return v / 8 * 8;
}

int conv_winograd_tiles(const dc_conv_desc* d) {
    const int force = wino_force();
    if (force == 32 || force == 64) return force;
    // Split-bf16 products (round 6): 32-tile items on every layer.  Alone on the chip the two item sizes are within 0 - 4 % of each other
    // (tools/conv_bench.py: res2_2b 51.3 -> 48.9 us, fpn_p3 138.5 -> 132.5, res4_2b 39.7 -> 39.5), but in the training pipeline two
    // 256-thread blocks per CU hide each other's fixed cost per item (7.5 us: first patch, U prefetch, output transform) and leave
    // the decoder stream's small kernels a place beside them: whole encoder pass 5.76 -> 5.41 ms, headline 10 440 -> 10 930 captions/s
    // (profiles/r06_wino_tiles.txt).  The fp32-product kernels keep the item-count rule they were measured with (rounds 3 - 4).
    if (wino_b3(d)) return 32;
    const int th = (d->H + 1) / 2, tw = (d->W + 1) / 2;
    const long items64 = (long)d->N * ((th + 7) / 8) * ((tw + 7) / 8) * (d->Cout / 32);
    return items64 >= kNumCU ? 64 : 32;
}

int conv2d_winograd(const dc_conv_desc* d, hipStream_t s) {
    wino::Args a;
    a.x = d->x;
    const bool b3 = wino_b3(d);
    a.u = b3 ? reinterpret_cast<const f4*>(d->w_wino_b3) : reinterpret_cast<const f4*>(d->w_wino);
    a.y = d->y;
    a.scale = d->scale;
    a.shift = d->shift;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.relu = d->relu;
    const int th = (d->H + 1) / 2, tw = (d->W + 1) / 2;
    const size_t x_bytes = (size_t)d->N * d->H * d->W * d->Cin * sizeof(float);
    // the kernels address the input through ONE buffer resource with 32-bit byte offsets (and a signed num_records): a P2-level input of
    // 32 or more 1024 x 1024 images does not fit -- split the batch
    DC_REQUIRE(x_bytes < (1ull << 31), DC_EINVAL, "dc_conv2d (winograd): the input tensor has %zu bytes, the kernel addresses < 2 GiB", x_bytes);
    a.x_bytes = (unsigned)x_bytes;
    a.u_bytes = (unsigned)((size_t)16 * d->Cin * d->Cout * (b3 ? 6 : 4));
    const bool big = conv_winograd_tiles(d) == 64;
    a.gy = (th + (big ? 8 : 4) - 1) / (big ? 8 : 4);
    a.gx = (tw + 7) / 8;
    a.groups = d->N * a.gy * a.gx;
    const long items = (long)a.groups * (d->Cout / 32);
    DC_REQUIRE(items < (1l << 31), DC_EINVAL, "dc_conv2d (winograd): grid too large");
    // persistent: the blocks that fit the chip walk the work items (a multiple of 8 blocks, so that an item's XCD is fixed by item % 8)
    // the CU budget (dc_set_persistent_cus) applies to the LONG launches only -- eight or more rounds of work items per block -- where a
    // grid that never yields would keep another queue's kernels (RCCL) waiting for hundreds of microseconds.  On a short launch a smaller
    // grid costs a whole extra round (256 items on 248 blocks: two rounds instead of one, measured 69 -> 102 us on the stage-4 layers) and
    // buys nothing: the launch is over in tens of microseconds.
    const long full = big ? kNumCU : 2 * kNumCU;
    const long slots = items >= 8 * full ? (big ? persistent_cus() : 2 * persistent_cus()) : full;
    const unsigned grid = items >= slots ? (unsigned)slots : (unsigned)std::max<long>(8, items / 8 * 8);
    if (b3 && big) {
        DC_ENSURE_DYN_LDS(wino::wino64b_kernel, wino::wp::Cfg<2>::LDS_BYTES);
        hipLaunchKernelGGL(wino::wino64b_kernel, dim3(grid), dim3(wino::wp::Cfg<2>::NT), wino::wp::Cfg<2>::LDS_BYTES, s, a);
    } else if (b3) {
        DC_ENSURE_DYN_LDS(wino::wino32b_kernel, wino::wp::Cfg<1>::LDS_BYTES);
        hipLaunchKernelGGL(wino::wino32b_kernel, dim3(grid), dim3(wino::wp::Cfg<1>::NT), wino::wp::Cfg<1>::LDS_BYTES, s, a);
    } else if (big) {
        DC_ENSURE_DYN_LDS(wino::wino64_kernel, wino::wp::Cfg<2>::LDS_BYTES);
        hipLaunchKernelGGL(wino::wino64_kernel, dim3(grid), dim3(wino::wp::Cfg<2>::NT), wino::wp::Cfg<2>::LDS_BYTES, s, a);
    } else {
        DC_ENSURE_DYN_LDS(wino::wino32_kernel, wino::wp::Cfg<1>::LDS_BYTES);
        hipLaunchKernelGGL(wino::wino32_kernel, dim3(grid), dim3(wino::wp::Cfg<1>::NT), wino::wp::Cfg<1>::LDS_BYTES, s, a);
    }
    return check_launch("dc_conv2d (winograd)");
}

}  // namespace dcap

using namespace dcap;

extern "C" size_t dc_conv2d_winograd_weight_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 32) return 0;
    return (size_t)16 * Cin * Cout * sizeof(float);
}

extern "C" int dc_conv2d_winograd_pack_f32(const float* w, float* u, int Cin, int Cout, void* stream) {
    DC_REQUIRE(w && u, DC_EINVAL, "dc_conv2d_winograd_pack: null pointer");
    DC_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DC_EINVAL, "dc_conv2d_winograd_pack: Cin and Cout must be multiples of 32");
    DC_REQUIRE(aligned16(u), DC_EALIGN, "dc_conv2d_winograd_pack: u must be 16-byte aligned");
    const long total = (long)Cin * Cout;
    const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(wino::wino_pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w, u, Cin, Cout);
    return check_launch("dc_conv2d_winograd_pack_f32");
}

extern "C" size_t dc_conv2d_winograd_b3_weight_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 32) return 0;
    return (size_t)16 * Cin * Cout * 3 * sizeof(uint16_t);
}

extern "C" int dc_conv2d_winograd_pack_b3(const float* w, uint16_t* u, int Cin, int Cout, void* stream) {
    DC_REQUIRE(w && u, DC_EINVAL, "dc_conv2d_winograd_pack_b3: null pointer");
    DC_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0, DC_EINVAL, "dc_conv2d_winograd_pack_b3: Cin and Cout must be multiples of 32");
    DC_REQUIRE(aligned16(u), DC_EALIGN, "dc_conv2d_winograd_pack_b3: u must be 16-byte aligned");
    const long total = (long)Cin * Cout;
    const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(wino::wino_pack_b3_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w, u, Cin, Cout);
    return check_launch("dc_conv2d_winograd_pack_b3");
}

extern "C" int dc_set_persistent_cus(int n) {
    DC_REQUIRE(n == 0 || (n >= 8 && n <= kNumCU), DC_EINVAL, "dc_set_persistent_cus: 0 (default: DCAP_WINO_CUS or all 256) or 8..256, got %d", n);
    dcap::g_persistent_cus.store(n == 0 ? -1 : n, std::memory_order_relaxed);
    return DC_OK;
}

extern "C" int dc_get_persistent_cus(void) { return dcap::persistent_cus(); }
