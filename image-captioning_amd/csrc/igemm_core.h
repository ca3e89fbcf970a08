// igemm_core.h -- the fp32 MFMA (implicit-)GEMM main loop shared by dc_gemm_f32 and dc_conv2d_nhwc_f32.
//
// C[M,N] = A[M,K] * B[K,N], fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact f32 fma chain;
// gfx950 has no xf32/TF32).  Block = 256 threads = 4 waves (2x2), block tile BM x BN x 32,
// each wave (BM/2)x(BN/2) as 32x32 MFMA tiles.  Operand tiles are staged global -> registers -> LDS
// (double-buffered, one barrier per K-tile; the next tile's global loads are issued before the
// current tile's MFMAs so HBM/L2 latency hides under the matrix pipe).
//
// Two LDS images, chosen per operand by where the contraction index lies in memory:
//   KC  rows of the operand hold K contiguously (activations [pixel][cin], packed conv weights
//       [cout][K], dY for dgrad): image [rows][36] floats; a lane fetches 4 consecutive k with ONE
//       ds_read_b128 (row stride 144 B = 9 x 16-B slots, odd => the 16-lane read groups hit 16
//       distinct slots: conflict-free) and feeds them to 4 successive MFMAs.
//   MC  the operand is K-major in memory (Keras [in,out] kernels, A^T for wgrad): image
//       [32][rows+4] floats; a lane reads one float per MFMA (32 consecutive floats per half-wave).
// MFMA j of an 8-wide K chunk contracts k = {j, 4+j} (lane half h takes k = 4h+j) on BOTH operands,
// so the permuted K order is consistent.
#pragma once
#include <math.h>
#include "dcap_internal.h"
#include <stdlib.h>

namespace dcap {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// staging registers use the native vector type: HIP's float4 struct copies lower to memcpy, which kept
// a pure load->LDS-store staging array in scratch
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 f4_zero() { return (f4)(0.f); }

constexpr int BK = 32;
constexpr int LDKC = BK + 4;

// ------------------------------------------------------------------------------------------------
// Operand loaders.  load<BT>() fills this thread's BT/32 f4 registers for the K-tile starting
// at k0 (zero beyond the operand's edge), store<BT>() writes them into the LDS image.
// ------------------------------------------------------------------------------------------------

template <int BT>
__device__ __forceinline__ void store_kc(float* S, const f4 (&r)[BT / 32], int tid) {
    const int q = tid & 7, rr = tid >> 3;
#pragma unroll
    for (int i = 0; i < BT / 32; ++i) *reinterpret_cast<f4*>(&S[(rr + 32 * i) * LDKC + 4 * q]) = r[i];
}

template <int BT>
__device__ __forceinline__ void store_mc(float* S, const f4 (&r)[BT / 32], int tid) {
    constexpr int QPR = BT / 4, RPP = 256 / QPR;
    const int q = tid % QPR, kr = tid / QPR;
#pragma unroll
    for (int i = 0; i < BT / 32; ++i) *reinterpret_cast<f4*>(&S[(kr + RPP * i) * (BT + 4) + 4 * q]) = r[i];
}

__device__ __forceinline__ f4 load4_guard(const float* a, int avail) {
    // avail = number of valid floats starting at a (>=1)
    if (avail >= 4) return *reinterpret_cast<const f4*>(a);
    f4 v = f4_zero();
    v.x = a[0];
    if (avail > 1) v.y = a[1];
    if (avail > 2) v.z = a[2];
    return v;
}

// Loaders come in two flavours.  FAST (host-checked: K % 32 == 0, 16-byte aligned rows, row length a
// multiple of 4, operand < 4 GiB) is branch-free and nearly VALU-free per K-tile: every thread keeps a
// 32-bit byte offset per staged row, computed once, and each K-tile only moves a block-uniform base
// (SGPR) -- an s_memtime profile of the first version showed ~580 cycles of per-tile address arithmetic
// per wave against 1024 cycles of MFMAs.  Rows beyond the operand's edge are CLAMPED to the last valid
// row (they only feed outputs that are never stored), so every load is an unconditional
// global_load_dwordx4.  The ragged flavour guards every access and zero-fills the K tail.

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// Raw buffer load with the hardware range check: byte offsets >= num_records return 0, which is how the
// im2col halo (TF 'SAME' zero padding) is produced -- no select instructions on the loaded data.
__device__ __forceinline__ f4 buf_f4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0));
}
// per-lane offset (VGPR, range-checked) + block-uniform offset (SGPR): no VALU between the address and the load
__device__ __forceinline__ f4 buf_f4s(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, unsigned uniform_off) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, (int)uniform_off, 0));
}
constexpr unsigned kOobOffset = 0x80000000u;     // beyond any tensor the host admits (< 2 GiB)

__device__ __forceinline__ f4 ldg_f4(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(base) + byte_off);
}

// rows x K, K contiguous; optional row gather (embedding lookup).
template <bool FAST>
struct DenseKCT {
    static constexpr bool KC = true;
    // loads past the last K-tile are issued unconditionally (their data is never stored): keep them in range
    __device__ __forceinline__ int kclamp(int k0, int kend) const { return FAST ? min(k0, kend - BK) : k0; }
    const float* p;
    long ld;
    int rows;
    const int32_t* gather;
    template <int BT>
    struct State {
        const float* base[FAST ? 1 : BT / 32];
        unsigned boff[FAST ? BT / 32 : 1];
        __amdgpu_buffer_rsrc_t rsrc;        // FAST: lane offset in a VGPR, K offset in an SGPR -> no address VALU per K-tile
    };
    template <int BT>
    __device__ __forceinline__ void init(State<BT>& s, int row0, int tid) const {
        const int rr = tid >> 3;
        if constexpr (FAST) s.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)0xFFFFFFF0u, 0x00020000);
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) {
            int row = row0 + rr + 32 * i;
            if constexpr (FAST) {
                row = min(row, rows - 1);
                const long src = gather ? (long)gather[row] : (long)row;
                s.boff[i] = (unsigned)((src * ld + 4 * (tid & 7)) * 4);
            } else if (row < rows) {
                const long src = gather ? (long)gather[row] : (long)row;
                s.base[i] = p + src * ld;
            } else {
                s.base[i] = nullptr;
            }
        }
    }
    template <int BT>
    __device__ __forceinline__ void load(State<BT>& s, f4 (&r)[BT / 32], int k0, int kend, int tid) const {
        if constexpr (FAST) {
#pragma unroll
            for (int i = 0; i < BT / 32; ++i) r[i] = buf_f4s(s.rsrc, s.boff[i], (unsigned)k0 * 4u);      // k0 block-uniform
        } else {
            const int k = k0 + 4 * (tid & 7);
#pragma unroll
            for (int i = 0; i < BT / 32; ++i) {
                r[i] = (s.base[i] != nullptr && k < kend) ? load4_guard(s.base[i] + k, kend - k) : f4_zero();
            }
        }
    }
    template <int BT>
    __device__ __forceinline__ void store(const State<BT>&, float* S, f4 (&r)[BT / 32], int tid) const {
        store_kc<BT>(S, r, tid);
    }
};
using DenseKC = DenseKCT<false>;

// Packed conv weights [Cout][taps*Cin] walked in Im2colKC's chunk-major K order: K-tile t -> tap t % taps, channels
// 32*(t / taps).  Host-checked like the FAST dense loader (Cin % 32 == 0, aligned rows, < 4 GiB).
struct ConvWeightKC {
    static constexpr bool KC = true;
    __device__ __forceinline__ int kclamp(int k0, int) const { return k0; }     // tiles past the end re-read in-range data
    const float* p;
    long ld;
    int rows, taps, Cin;
    template <int BT>
    struct State {
        unsigned boff[BT / 32];
        int tap, koff;              // block-uniform position of the NEXT K-tile (incremental, like Im2colKC)
    };
    template <int BT>
    __device__ __forceinline__ void init(State<BT>& s, int row0, int tid) const {
        const int rr = tid >> 3;
        s.tap = -1;
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) s.boff[i] = (unsigned)(((long)min(row0 + rr + 32 * i, rows - 1) * ld + 4 * (tid & 7)) * 4);
    }
    template <int BT>
    __device__ __forceinline__ void load(State<BT>& s, f4 (&r)[BT / 32], int k0, int, int) const {
        if (s.tap < 0) {                                  // first tile of this block: one division
            const int t = k0 >> 5, chunk = t / taps;
            s.tap = t - chunk * taps;
            s.koff = s.tap * Cin + chunk * 32;
        }
        const float* kb = p + min(s.koff, (int)ld - BK);   // block-uniform; clamped for the loads issued past the last tile
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) r[i] = ldg_f4(kb, s.boff[i]);
        if (++s.tap == taps) { s.tap = 0; s.koff += 32 - (taps - 1) * Cin; } else { s.koff += Cin; }
    }
    template <int BT>
    __device__ __forceinline__ void store(const State<BT>&, float* S, f4 (&r)[BT / 32], int tid) const {
        store_kc<BT>(S, r, tid);
    }
};

// K x cols, cols contiguous (K-major operand); optional gather of the K rows.
template <bool FAST>
struct DenseMCT {
    static constexpr bool KC = false;
    __device__ __forceinline__ int kclamp(int k0, int kend) const { return FAST ? min(k0, kend - BK) : k0; }
    const float* p;
    long ld;
    int cols;
    const int32_t* gather;   // optional: K row k is stored at row gather[k]
    template <int BT>
    struct State {
        int col;
        unsigned boff[BT / 32];
        __amdgpu_buffer_rsrc_t rsrc;        // FAST: see DenseKCT
    };
    template <int BT>
    __device__ __forceinline__ void init(State<BT>& s, int col0, int tid) const {
        constexpr int QPR = BT / 4, RPP = 256 / QPR;
        s.col = col0 + 4 * (tid % QPR);
        if constexpr (FAST) {
            s.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)0xFFFFFFF0u, 0x00020000);
            s.col = min(s.col, cols - 4);
#pragma unroll
            for (int i = 0; i < BT / 32; ++i) s.boff[i] = (unsigned)((((long)(tid / QPR) + RPP * i) * ld + s.col) * 4);
        }
    }
    template <int BT>
    __device__ __forceinline__ void load(State<BT>& s, f4 (&r)[BT / 32], int k0, int kend, int tid) const {
        constexpr int QPR = BT / 4, RPP = 256 / QPR;
        const int kr = tid / QPR;
        if constexpr (FAST) {
            if (gather == nullptr) {
                const unsigned soff = (unsigned)((long)k0 * ld * 4);          // block-uniform, < 4 GiB (host-checked span)
#pragma unroll
                for (int i = 0; i < BT / 32; ++i) r[i] = buf_f4s(s.rsrc, s.boff[i], soff);
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) {
            const int k = k0 + kr + RPP * i;
            if (FAST || (k < kend && s.col < cols)) {
                const long src = gather ? (long)gather[k] : (long)k;
                r[i] = FAST ? *reinterpret_cast<const f4*>(p + src * ld + s.col) : load4_guard(p + src * ld + s.col, cols - s.col);
            } else {
                r[i] = f4_zero();
            }
        }
    }
    template <int BT>
    __device__ __forceinline__ void store(const State<BT>&, float* S, f4 (&r)[BT / 32], int tid) const {
        store_mc<BT>(S, r, tid);
    }
};
using DenseMC = DenseMCT<false>;

// NHWC activations viewed as the im2col matrix [N*Ho*Wo][kh*kw*Cin] (cin fastest), Cin % 32 == 0:
// one K-tile of 32 lies inside one (ky,kx) tap, so a row's 128 B are contiguous in memory.
// CM = false: K is visited in storage order (tap-major, the packed weights read linearly by DenseKCT).
// CM = true: chunk-major -- K-tile t covers channels 32*(t / taps) .. +31 of tap t % taps (ConvWeightKC walks the packed
// weights in the same order): the kh*kw taps of one 32-channel chunk touch the same few cache lines per pixel.  Worth ~3 %
// on the split-bf16 main loop, whose K-tile is 3x shorter; on the fp32 loop the extra per-tile scalar work costs more
// than the locality gains (64x64 tiles: -4 %), so it stays on storage order.
// Per thread and staged row: the byte offset of the output-aligned pixel (always inside the image) and a
// bit mask of the taps that fall inside the image; per K-tile: one uniform tap offset.  Out-of-image
// taps re-load the aligned pixel (valid memory) and are zeroed at store() time, after the MFMAs.
template <bool CM>
struct Im2colKCT {
    static constexpr bool KC = true;
    __device__ __forceinline__ int kclamp(int k0, int) const { return k0; }     // range-checked buffer loads
    const float* x;
    int H, W, Cin, Ho, Wo, stride, pad_t, pad_l, kw, taps, M;
    unsigned x_bytes;               // size of the activation tensor (buffer range for the zero-filling loads)
    template <int BT>
    struct State {
        __amdgpu_buffer_rsrc_t rsrc;
        unsigned boff[BT / 32];     // ((n*H + oy*stride)*W + ox*stride)*Cin*4 + 16*(tid&7): the output-aligned pixel
        unsigned long long mask[BT / 32];     // bit (ky*kw + kx): that tap of this output pixel lies inside the image
        unsigned sel[BT / 32];      // storage order only: boff or the out-of-range offset, for the CURRENT tap
        int ky, kx, c;              // block-uniform position of the NEXT K-tile (incremental, no division)
        int fresh;                  // sel[] must be recomputed (first tile of the block / of a tap)
    };
    template <int BT>
    __device__ __forceinline__ void init(State<BT>& s, int row0, int tid) const {
        const int rr = tid >> 3;
        if constexpr (CM) {
            s.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)x_bytes, 0x00020000);
        } else {
            // Storage order: the tap offset and the channel offset are block-uniform and travel in the load's SGPR offset,
            // which must be non-negative: the descriptor's base is moved back by the largest negative tap offset.  Only
            // in-image taps are ever issued with an in-range lane offset (the others carry kOobOffset), so no byte in
            // front of the tensor is touched.
            const long bias = ((long)pad_t * W + pad_l) * Cin;
            s.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) - bias, 0, (int)(x_bytes + bias * 4), 0x00020000);
        }
        s.ky = -1;
        s.fresh = 1;
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) {
            const int m = min(row0 + rr + 32 * i, M - 1);          // rows past M feed nothing that is stored
            const int n = m / (Ho * Wo), rem = m - n * (Ho * Wo);
            const int oy = rem / Wo, ox = rem - oy * Wo;
            s.boff[i] = (unsigned)(((((long)n * H + oy * stride) * W + ox * stride) * Cin + 4 * (tid & 7)) * 4);
            unsigned ym = 0, xm = 0;
            for (int t = 0; t < 8; ++t) {
                ym |= ((unsigned)(oy * stride - pad_t + t) < (unsigned)H) ? (1u << t) : 0u;
                xm |= ((unsigned)(ox * stride - pad_l + t) < (unsigned)W) ? (1u << t) : 0u;
            }
            unsigned long long mk = 0;
            for (int ty = 0; ty < 8; ++ty)
                if ((ym >> ty) & 1u) mk |= (unsigned long long)(xm & ((1u << kw) - 1u)) << (ty * kw);
            s.mask[i] = mk;
        }
    }
    template <int BT>
    __device__ __forceinline__ void load(State<BT>& s, f4 (&r)[BT / 32], int k0, int kend, int tid) const {
        if (s.ky < 0) {                                   // first tile of this block (split-K start): one division
            const int kt = k0 >> 5;
            int tap;
            if constexpr (CM) {
                const int chunk = kt / taps;
                tap = kt - chunk * taps;
                s.c = chunk * 32;
            } else {
                const int cin_tiles = Cin >> 5;
                tap = kt / cin_tiles;
                s.c = (kt - tap * cin_tiles) * 32;
            }
            s.ky = tap / kw;
            s.kx = tap - s.ky * kw;
        }
        const int ky = s.ky, kx = s.kx;
        const int tap = ky * kw + kx;                                                  // block-uniform
        if constexpr (CM) {
            const int ubytes = (((ky - pad_t) * W + (kx - pad_l)) * Cin + s.c) * 4;  // block-uniform, may be negative
#pragma unroll
            for (int i = 0; i < BT / 32; ++i) {
                const bool in = tap < taps && ((s.mask[i] >> (tap & 63)) & 1ull) != 0;
                r[i] = buf_f4(s.rsrc, in ? s.boff[i] + (unsigned)ubytes : kOobOffset); // out of image -> hardware zero
            }
        } else {
            // f32 MFMAs run on the VALU lanes: a co-resident wave's address arithmetic cannot issue while another wave's
            // MFMA burst holds the SIMD, so its loads would go out only after the burst (stamps: the "issue loads" phase
            // was as long as the MFMA phase).  Per K-tile this path is VALU-free: the per-lane select happens once per tap.
            if (s.fresh) {
                s.fresh = 0;
#pragma unroll
                for (int i = 0; i < BT / 32; ++i)       // tap >= taps: a tile issued past the end of K (never stored) must not address memory
                    s.sel[i] = (tap < taps && ((s.mask[i] >> (tap & 63)) & 1ull)) ? s.boff[i] : kOobOffset;
            }
            const unsigned soff = (unsigned)(((ky * W + kx) * Cin + s.c) * 4);         // block-uniform, >= 0 (SGPR)
#pragma unroll
            for (int i = 0; i < BT / 32; ++i) r[i] = buf_f4s(s.rsrc, s.sel[i], soff);
        }
        if constexpr (CM) {
            if (++s.kx == kw) {                           // next tap of this channel chunk; then the next chunk
                s.kx = 0;
                if ((s.ky + 1) * kw == taps) { s.ky = 0; s.c += 32; } else { ++s.ky; }
            }
        } else {
            s.c += 32;
            if (s.c == Cin) {
                s.c = 0;
                s.fresh = 1;
                if (++s.kx == kw) { s.kx = 0; ++s.ky; }
            }
        }
    }
    template <int BT>
    __device__ __forceinline__ void store(const State<BT>&, float* S, f4 (&r)[BT / 32], int tid) const {
        store_kc<BT>(S, r, tid);
    }
};

using Im2colKC = Im2colKCT<false>;
using Im2colKCcm = Im2colKCT<true>;

// wgrad's B operand: the im2col matrix K-major.  K rows = output pixels (all images), columns
// n = tap*Cin + ci; a column tile of BT lies inside one tap (Cin % BT == 0), so a K row is a contiguous run of
// channels of ONE (shifted) input pixel.  Each thread walks its pixels incrementally (32 pixels per K-tile:
// an add and a wrap test instead of divisions); out-of-image taps and pixels past the end read hardware zeros.
struct Im2colMC {
    static constexpr bool KC = false;
    __device__ __forceinline__ int kclamp(int k0, int) const { return k0; }
    const float* x;
    int H, W, Cin, Ho, Wo, stride, pad_t, pad_l, kw, P;      // P = N*Ho*Wo pixels (the K extent)
    unsigned x_bytes;
    template <int BT>
    struct State {
        __amdgpu_buffer_rsrc_t rsrc;
        int oy[BT / 32], ox[BT / 32], n[BT / 32];
        int dy, dx;            // tap offset of this column tile (block-uniform)
        unsigned coff;         // (ci0 + 4*q) * 4 bytes
        int started;
    };
    template <int BT>
    __device__ __forceinline__ void init(State<BT>& s, int col0, int tid) const {
        constexpr int QPR = BT / 4;
        s.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)x_bytes, 0x00020000);
        const int tap = col0 / Cin, ci0 = col0 - tap * Cin;
        const int ky = tap / kw, kx = tap - ky * kw;
        s.dy = ky - pad_t;
        s.dx = kx - pad_l;
        s.coff = (unsigned)((ci0 + 4 * (tid % QPR)) * 4);
        s.started = 0;
    }
    template <int BT>
    __device__ __forceinline__ void load(State<BT>& s, f4 (&r)[BT / 32], int k0, int kend, int tid) const {
        constexpr int QPR = BT / 4, RPP = 256 / QPR;
        if (!s.started) {                               // first K-tile of this block: decode once
            s.started = 1;
#pragma unroll
            for (int i = 0; i < BT / 32; ++i) {
                const int p = k0 + tid / QPR + RPP * i;
                const int n = p / (Ho * Wo), rem = p - n * (Ho * Wo);
                s.n[i] = n;
                s.oy[i] = rem / Wo;
                s.ox[i] = rem - s.oy[i] * Wo;
            }
        }
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) {
            const int iy = s.oy[i] * stride + s.dy, ix = s.ox[i] * stride + s.dx;
            const bool in = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && s.n[i] * (Ho * Wo) + s.oy[i] * Wo + s.ox[i] < min(P, kend);
            const unsigned off = (unsigned)((((long)s.n[i] * H + iy) * W + ix) * Cin * 4) + s.coff;
            r[i] = buf_f4(s.rsrc, in ? off : kOobOffset);
            s.ox[i] += BK;                              // next K-tile: 32 pixels further along the row-major walk
            while (s.ox[i] >= Wo) { s.ox[i] -= Wo; ++s.oy[i]; }
            while (s.oy[i] >= Ho) { s.oy[i] -= Ho; ++s.n[i]; }
        }
    }
    template <int BT>
    __device__ __forceinline__ void store(const State<BT>&, float* S, f4 (&r)[BT / 32], int tid) const {
        store_mc<BT>(S, r, tid);
    }
};

// The 7x7/stride-2 stem on an RGBX image [N,H,W,4]: K-tile ky = one kernel row, float4 q = tap kx
// (kx == 7 is a zero-weight pad), so K = 7*8*4 = 224 and every load is one aligned pixel.
struct StemKC {
    static constexpr bool KC = true;
    __device__ __forceinline__ int kclamp(int k0, int) const { return k0; }
    const float* x;
    int H, W, Ho, Wo, M;
    unsigned x_bytes;
    template <int BT>
    struct State {
        __amdgpu_buffer_rsrc_t rsrc;
        unsigned boff[BT / 32];     // byte offset of pixel (oy*2, ox*2): always inside the image
        unsigned mask[BT / 32];     // bits 0-6: row oy*2-3+ky in range; bit 8: this lane's column in range
    };
    template <int BT>
    __device__ __forceinline__ void init(State<BT>& s, int row0, int tid) const {
        const int rr = tid >> 3, q = tid & 7;
        s.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)x_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) {
            const int m = min(row0 + rr + 32 * i, M - 1);
            const int n = m / (Ho * Wo), rem = m - n * (Ho * Wo);
            const int oy = rem / Wo, ox = rem - oy * Wo;
            s.boff[i] = (unsigned)((((long)n * H + oy * 2) * W + ox * 2) * 16);
            unsigned mk = ((unsigned)(ox * 2 - 3 + q) < (unsigned)W) ? (1u << 8) : 0u;
            for (int t = 0; t < 7; ++t) mk |= ((unsigned)(oy * 2 - 3 + t) < (unsigned)H) ? (1u << t) : 0u;
            s.mask[i] = mk;
        }
    }
    template <int BT>
    __device__ __forceinline__ void load(State<BT>& s, f4 (&r)[BT / 32], int k0, int kend, int tid) const {
        const int ky = k0 >> 5;
        const int ubytes = ((ky - 3) * W + ((tid & 7) - 3)) * 16;
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) {
            const bool in = ((s.mask[i] >> ky) & (s.mask[i] >> 8) & 1u) != 0;
            r[i] = buf_f4(s.rsrc, in ? s.boff[i] + (unsigned)ubytes : kOobOffset);
        }
    }
    template <int BT>
    __device__ __forceinline__ void store(const State<BT>&, float* S, f4 (&r)[BT / 32], int tid) const {
        store_kc<BT>(S, r, tid);
    }
};

// ------------------------------------------------------------------------------------------------
// Epilogue: v = acc*scale[n] + shift[n] (+ residual) (relu) ; C = v or C += v.
// ------------------------------------------------------------------------------------------------
struct Epilogue {
    float* C;
    long ldc;
    const float* scale;
    const float* shift;
    const float* res;
    long ldr;
    int res_mode;   // 0 none, 1 same rows, 2 rows are NHWC pixels and res is the 2x coarser map,
                    // 3 residual row = row % Ho (a per-RoI term broadcast over timesteps)
    int Ho, Wo;
    int relu;
    int accumulate;
    int vec4;       // host-checked: C/res rows 16-byte aligned and N % 4 == 0 => 16-B epilogue accesses
    unsigned short* Cb = nullptr;   // optional bf16 copy of the finished output (the next bf16 GEMM's operand); C may then be null
    long ldcb = 0;
    // residual row pointer for output row `row` (nullptr when there is no residual)
    __device__ __forceinline__ const float* res_row(int row) const {
        if (res_mode == 1) return res + (long)row * ldr;
        if (res_mode == 2) {
            const int n = row / (Ho * Wo), rem = row - n * (Ho * Wo);
            const int y = rem / Wo, xx = rem - y * Wo;
            return res + (((long)n * (Ho >> 1) + (y >> 1)) * (Wo >> 1) + (xx >> 1)) * ldr;
        }
        if (res_mode == 3) return res + (long)(row % Ho) * ldr;
        return nullptr;
    }
    __device__ __forceinline__ float finish(float v, float sc, float sh, const float* rr, float* crow, int col) const {
        v = v * sc + sh;
        if (rr) v += rr[col];
        if (relu) v = fmaxf(v, 0.f);
        if (accumulate) v += crow[col];
        return v;
    }
    __device__ __forceinline__ float apply(float v, int row, int col) const {
        if (scale) v *= scale[col];
        if (shift) v += shift[col];
        if (res_mode == 1) {
            v += res[(long)row * ldr + col];
        } else if (res_mode == 2) {
            const int n = row / (Ho * Wo), rem = row - n * (Ho * Wo);
            const int y = rem / Wo, xx = rem - y * Wo;
            const long rr = ((long)n * (Ho >> 1) + (y >> 1)) * (Wo >> 1) + (xx >> 1);
            v += res[rr * ldr + col];
        } else if (res_mode == 3) {
            v += res[(long)(row % Ho) * ldr + col];
        }
        if (relu) v = fmaxf(v, 0.f);
        if (accumulate) v += C[(long)row * ldc + col];
        return v;
    }
    // round-to-nearest-even bf16 bits of a finite or non-finite float (the plain cast: v_cvt_pk_bf16_f32, NaN stays NaN)
    static __device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
    __device__ __forceinline__ void put(int row, int col, float v) const {
        if (C) C[(long)row * ldc + col] = v;
        if (Cb) Cb[(long)row * ldcb + col] = bf16_bits(v);
    }
};

// ------------------------------------------------------------------------------------------------
// One K-tile of MFMAs from the LDS images.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, bool AKC, bool BKC, int TM, int TN>
__device__ __forceinline__ void load_frags(const float* __restrict__ As, const float* __restrict__ Bs, float (&a)[TM][4],
                                           float (&b)[TN][4], int c, int wm, int wn, int i, int h) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        if constexpr (AKC) {
            const float4 t = *reinterpret_cast<const float4*>(&As[(wm + tm * 32 + i) * LDKC + 8 * c + 4 * h]);
            a[tm][0] = t.x; a[tm][1] = t.y; a[tm][2] = t.z; a[tm][3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[tm][j] = As[(8 * c + 4 * h + j) * (BM + 4) + wm + tm * 32 + i];
        }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        if constexpr (BKC) {
            const float4 t = *reinterpret_cast<const float4*>(&Bs[(wn + tn * 32 + i) * LDKC + 8 * c + 4 * h]);
            b[tn][0] = t.x; b[tn][1] = t.y; b[tn][2] = t.z; b[tn][3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) b[tn][j] = Bs[(8 * c + 4 * h + j) * (BN + 4) + wn + tn * 32 + i];
        }
    }
}

// One K-tile: ALL fragment reads (4 chunks of 8 k) are issued first, then the 16*TM*TN MFMAs run
// back to back with nothing to wait for; the next tile's global loads (issued before this call) and
// the LDS latency of this burst are covered by the co-resident block's MFMAs.
template <int BM, int BN, bool AKC, bool BKC>
__device__ __forceinline__ void mma_tile(const float* __restrict__ As, const float* __restrict__ Bs,
                                         f32x16 (&acc)[BM / 64][BN / 64], int wm, int wn, int lane) {
    constexpr int TM = BM / 64, TN = BN / 64;
    const int i = lane & 31, h = lane >> 5;
    float a[4][TM][4], b[4][TN][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) load_frags<BM, BN, AKC, BKC, TM, TN>(As, Bs, a[c], b[c], c, wm, wn, i, h);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][tm][j], b[c][tn][j], acc[tm][tn], 0, 0, 0);
}

template <int BT, bool KC>
constexpr int lds_floats() {
    return KC ? BT * LDKC : BK * (BT + 4);
}

// K-tiles per barrier: the small tile runs two 32-deep K-tiles between barriers (half the
// synchronisation points per MFMA); the larger tiles would no longer fit two blocks per CU.
template <int BM, int BN>
constexpr int tiles_per_sync() { return (BM * BN <= 64 * 64) ? 2 : 1; }

template <int BM, int BN, class AL, class BL>
constexpr size_t igemm_lds_bytes() {
    constexpr size_t stages = 2 * tiles_per_sync<BM, BN>() * (size_t)(lds_floats<BM, AL::KC>() + lds_floats<BN, BL::KC>()) * sizeof(float);
    constexpr size_t cimage = (size_t)BM * (BN + 4) * sizeof(float);     // epilogue transpose image
    return stages > cimage ? stages : cimage;
}

// ------------------------------------------------------------------------------------------------
// Tile epilogue shared by the f32 and the split-bf16 main loops.
// ------------------------------------------------------------------------------------------------
// TM x TN = 32x32 accumulator blocks per wave (wave origin wm, wn inside the tile); NT = threads taking part (waves 0..NT/64-1).
template <int BM, int BN, int TM = BM / 64, int TN = BN / 64, int NT = 256>
__device__ __forceinline__ void store_tile(f32x16 (&acc)[TM][TN], float* smem, const Epilogue& ep, float* __restrict__ partial,
                                           int M, int N, int m0, int n0, int wm, int wn) {
    const int tid = threadIdx.x, lane = tid & 63;
    // ---- epilogue: transpose the accumulators through LDS so every lane owns 4 consecutive columns:
    // 16-byte residual loads / output stores in full 256-512 B row segments instead of 4-byte accesses
    // (the store tail of a wide 1x1 conv is issue-bound, not bandwidth-bound, with the raw MFMA layout).
    // Residual / accumulate operands are fetched a group of rows ahead of the stores: C and res may alias as far as the
    // compiler knows, so a load written after a store is never hoisted above it, and a short-K 1x1 conv (K = 256:
    // 8 K-tiles) otherwise spends as long waiting on 16 dependent round trips as it spent on its MFMAs.
    constexpr int LDC = BN + 4;
    constexpr int CPR = BN / 4, RPP = NT / CPR;          // float4 columns per row, rows per pass
    constexpr int PASSES = BM / RPP, G = PASSES < 8 ? PASSES : 8;
    const int c4 = tid % CPR, rp = tid / CPR;
    const int col = n0 + 4 * c4;
    const bool full = !partial && ep.vec4 && col + 3 < N;
    const bool pre_res = full && ep.res_mode != 0, pre_acc = full && ep.accumulate;
    f4 qres[G], qacc[G];
    auto prefetch = [&](int p0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int row = min(m0 + (p0 + g) * RPP + rp, M - 1);      // clamped: rows past M are never stored
            if (pre_res) qres[g] = *reinterpret_cast<const f4*>(ep.res_row(row) + col);
            if (pre_acc) qacc[g] = *reinterpret_cast<const f4*>(ep.C + (long)row * ep.ldc + col);
        }
    };
    prefetch(0);                                         // in flight while the tile goes through LDS
    float* Cs = smem;                       // all waves are past the loop's final barrier: LDS is free
    {
        const int i = lane & 31, h = lane >> 5;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Cs[(wm + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDC + wn + tn * 32 + i] = acc[tm][tn][r];
    }
    __syncthreads();
    if (col >= N) return;
    if (partial) {
        float* prow = partial + (long)blockIdx.z * M * N;
#pragma unroll 4
        for (int p = 0; p < PASSES; ++p) {
            const int lr = p * RPP + rp, row = m0 + lr;
            if (row >= M) break;
            const float4 v = *reinterpret_cast<const float4*>(&Cs[lr * LDC + 4 * c4]);
            float* o = prow + (long)row * N + col;
            if ((N & 3) == 0) {
                *reinterpret_cast<float4*>(o) = v;
            } else {
                o[0] = v.x;
                if (col + 1 < N) o[1] = v.y;
                if (col + 2 < N) o[2] = v.z;
                if (col + 3 < N) o[3] = v.w;
            }
        }
        return;
    }
    if (full) {
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ep.scale) sc = *reinterpret_cast<const float4*>(ep.scale + col);
        if (ep.shift) sh = *reinterpret_cast<const float4*>(ep.shift + col);
#pragma unroll
        for (int p0 = 0; p0 < PASSES; p0 += G) {
            if (p0 > 0) prefetch(p0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int lr = (p0 + g) * RPP + rp, row = m0 + lr;
                float4 v = *reinterpret_cast<const float4*>(&Cs[lr * LDC + 4 * c4]);
                v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                if (pre_res) { v.x += qres[g][0]; v.y += qres[g][1]; v.z += qres[g][2]; v.w += qres[g][3]; }
                if (ep.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (pre_acc) { v.x += qacc[g][0]; v.y += qacc[g][1]; v.z += qacc[g][2]; v.w += qacc[g][3]; }
                if (row < M) {
                    if (ep.C) *reinterpret_cast<float4*>(ep.C + (long)row * ep.ldc + col) = v;
                    if (ep.Cb) {
                        typedef unsigned short us4 __attribute__((ext_vector_type(4)));
                        *reinterpret_cast<us4*>(ep.Cb + (long)row * ep.ldcb + col) =
                            us4{Epilogue::bf16_bits(v.x), Epilogue::bf16_bits(v.y), Epilogue::bf16_bits(v.z), Epilogue::bf16_bits(v.w)};
                    }
                }
            }
        }
        return;
    }
#pragma unroll 4
    for (int p = 0; p < PASSES; ++p) {
        const int lr = p * RPP + rp, row = m0 + lr;
        if (row >= M) break;
        const float4 v = *reinterpret_cast<const float4*>(&Cs[lr * LDC + 4 * c4]);
        float* crow = ep.C + (long)row * ep.ldc;
        const float* rr = ep.res_row(row);
#define DC_TAIL(j, e)                                                                                       \
    if (col + j < N)                                                                                        \
        ep.put(row, col + j, ep.finish(e, ep.scale ? ep.scale[col + j] : 1.f, ep.shift ? ep.shift[col + j] : 0.f, rr, crow, col + j));
        DC_TAIL(0, v.x) DC_TAIL(1, v.y) DC_TAIL(2, v.z) DC_TAIL(3, v.w)
#undef DC_TAIL
    }
}

// ------------------------------------------------------------------------------------------------
// The kernel.  grid.x = tiles_m*tiles_n (XCD-remapped so consecutive tiles along N, which share the
// A panel, run on one XCD), grid.z = split-K slices.  With split-K the raw partial sums go to the
// slab  partial[z][M][N]  and splitk_reduce_kernel applies the epilogue.
// ------------------------------------------------------------------------------------------------
// The main loop as a device function (shared with the fused vocabulary softmax / cross-entropy kernels of vocab_ce.hip):
// accumulates A[m0.., kbeg..kend) * B[kbeg..kend), n0..] into acc.
template <int BM, int BN, class AL, class BL>
__device__ __forceinline__ void igemm_mainloop(const AL& al, const BL& bl, float* smem, int m0, int n0, int kbeg, int kend,
                                               f32x16 (&acc)[BM / 64][BN / 64], int wm, int wn) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_FL = lds_floats<BM, AL::KC>(), B_FL = lds_floats<BN, BL::KC>(), STAGE = A_FL + B_FL;
    const int tid = threadIdx.x, lane = tid & 63;
    typename AL::template State<BM> sa;
    typename BL::template State<BN> sb;
    al.template init<BM>(sa, m0, tid);
    bl.template init<BN>(sb, n0, tid);

#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    constexpr int KT = tiles_per_sync<BM, BN>();
    f4 ra[KT][BM / 32], rb[KT][BN / 32];
    const int nkt = (kend - kbeg + BK - 1) / BK;
#pragma unroll
    for (int j = 0; j < KT; ++j)
        if (j < nkt) {
            al.template load<BM>(sa, ra[j], kbeg + j * BK, kend, tid);
            bl.template load<BN>(sb, rb[j], kbeg + j * BK, kend, tid);
        }
#pragma unroll
    for (int j = 0; j < KT; ++j)
        if (j < nkt) {
            al.template store<BM>(sa, smem + j * STAGE, ra[j], tid);
            bl.template store<BN>(sb, smem + j * STAGE + A_FL, rb[j], tid);
        }
    __syncthreads();
    // f32 MFMAs execute on the SIMD's fp32 lanes: VALU instructions do NOT overlap them (tools/micro/
    // coissue.hip: every extra VALU adds its full ~4 cycles per 64-cycle MFMA), so the loop keeps the
    // per-tile VALU count minimal (32-bit offsets, uniform bases, hardware zero-fill) and only memory
    // latency is left to hide behind the co-resident block's burst.
    // One pipeline step on compile-time LDS stages: fragment and store addresses become a constant VGPR plus an immediate
    // offset (a run-time stage index costs an address add per ds instruction -- VALU, i.e. MFMA time on this pipe).
    auto step = [&](int kt, float* cur, float* nxt) {
        // unconditional (a conditional load merges with the old registers and the compiler then waits for
        // the data right here): tiles past the end are clamped / range-checked and never stored
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const int k0 = kbeg + (kt + KT + j) * BK;
            al.template load<BM>(sa, ra[j], al.kclamp(k0, kend), kend, tid);
            bl.template load<BN>(sb, rb[j], bl.kclamp(k0, kend), kend, tid);
        }
#pragma unroll
        for (int j = 0; j < KT; ++j)
            if (kt + j < nkt) mma_tile<BM, BN, AL::KC, BL::KC>(cur + j * STAGE, cur + j * STAGE + A_FL, acc, wm, wn, lane);
#pragma unroll
        for (int j = 0; j < KT; ++j)
            if (kt + KT + j < nkt) {
                al.template store<BM>(sa, nxt + j * STAGE, ra[j], tid);
                bl.template store<BN>(sb, nxt + j * STAGE + A_FL, rb[j], tid);
            }
        __syncthreads();
    };
    float* const stage0 = smem;
    float* const stage1 = smem + KT * STAGE;
    for (int kt = 0; kt < nkt; kt += 2 * KT) {
        step(kt, stage0, stage1);
        if (kt + KT < nkt) step(kt + KT, stage1, stage0);
    }

}

template <int BM, int BN, class AL, class BL>
__global__ __launch_bounds__(256, 2) void igemm_kernel(AL al, BL bl, Epilogue ep, int M, int N, int K, int klen,
                                                    float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6;
    const int tiles_n = (N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
    const int kbeg = blockIdx.z * klen;
    const int kend = min(K, kbeg + klen);
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    f32x16 acc[BM / 64][BN / 64];
    igemm_mainloop<BM, BN, AL, BL>(al, bl, smem, m0, n0, kbeg, kend, acc, wm, wn);
    store_tile<BM, BN>(acc, smem, ep, partial, M, N, m0, n0, wm, wn);
}

// ------------------------------------------------------------------------------------------------
// Producer / consumer variant for the 64x64 tile (the layers whose grids put only two blocks on a CU: ResNet stage 4/5 at
// two images per GPU).  With the single-role kernel a wave that waits for its global loads cannot issue MFMAs, and at two
// waves per SIMD that wait shows (stamps: 20-45 % of a K-tile in s_waitcnt vmcnt on these layers; a second register set
// did not help).  Here waves 0-3 (one per SIMD) only read fragments and issue MFMAs, waves 4-7 only move data.  The data
// movers need NO VALU per K-tile (buffer loads with SGPR offsets, ds_write with immediate offsets), so they do not take
// fp32-lane cycles from the MFMAs; they keep four K-tiles of loads in flight and fill a four-stage LDS ring three tiles
// ahead of the consumers.  One barrier per K-tile.  LDS 4 x 18 KB per block: two blocks per CU.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN>
constexpr int pc_stages() { return (BM * BN <= 64 * 64) ? 4 : (BM * BN >= 128 * 128 ? 3 : 2); }      // ring depth = register sets

template <int BM, int BN, class AL, class BL>
constexpr size_t igemm_pc_lds_bytes() {
    constexpr size_t ring = (size_t)pc_stages<BM, BN>() * (lds_floats<BM, AL::KC>() + lds_floats<BN, BL::KC>()) * sizeof(float);
    constexpr size_t cimage = (size_t)BM * (BN + 4) * sizeof(float);
    return ring > cimage ? ring : cimage;
}

template <int BM, int BN, class AL, class BL>
__global__ __launch_bounds__(512, (BM * BN >= 128 * 128) ? 1 : 2) void igemm_pc_kernel(AL al, BL bl, Epilogue ep, int M, int N, int K, int klen,
                                                       float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = BM / 64, TN = BN / 64, NS = pc_stages<BM, BN>();
    constexpr int A_FL = lds_floats<BM, AL::KC>(), B_FL = lds_floats<BN, BL::KC>(), STAGE = A_FL + B_FL;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int tiles_n = (N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
    const int kbeg = blockIdx.z * klen;
    const int kend = min(K, kbeg + klen);
    const int nkt = (kend - kbeg + BK - 1) / BK;

    if (wave >= 4) {
        // ------------------------------------------------------------------ data movers
        const int ptid = tid - 256;
        typename AL::template State<BM> sa;
        typename BL::template State<BN> sb;
        al.template init<BM>(sa, m0, ptid);
        bl.template init<BN>(sb, n0, ptid);
        f4 ra[NS][BM / 32], rb[NS][BN / 32];              // register set j % NS holds tile j between its load and its LDS write
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int k0 = kbeg + j * BK;
            al.template load<BM>(sa, ra[j], al.kclamp(k0, kend), kend, ptid);
            bl.template load<BN>(sb, rb[j], bl.kclamp(k0, kend), kend, ptid);
        }
        // tile t: registers -> stage t % NS (compile-time `st`), then refill the set with tile t + NS
        auto fill = [&](int t, float* st, f4 (&qa)[BM / 32], f4 (&qb)[BN / 32]) {
            if (t < nkt) {
                al.template store<BM>(sa, st, qa, ptid);
                bl.template store<BN>(sb, st + A_FL, qb, ptid);
            }
            const int k0 = kbeg + (t + NS) * BK;
            al.template load<BM>(sa, qa, al.kclamp(k0, kend), kend, ptid);
            bl.template load<BN>(sb, qb, bl.kclamp(k0, kend), kend, ptid);
        };
#pragma unroll
        for (int j = 0; j < NS - 1; ++j) fill(j, smem + j * STAGE, ra[j], rb[j]);          // tiles 0 .. NS-2 staged up front
        __syncthreads();
        for (int t0 = 0; t0 < nkt; t0 += NS) {
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                if (t0 + u < nkt) {                          // consumers are on tile t0+u; stage (u+NS-1) % NS was released at the last barrier
                    fill(t0 + u + NS - 1, smem + ((u + NS - 1) % NS) * STAGE, ra[(u + NS - 1) % NS], rb[(u + NS - 1) % NS]);
                    __syncthreads();
                }
            }
        }
        __syncthreads();                                     // the epilogue's barrier (store_tile)
        return;
    }
    // ---------------------------------------------------------------------- MFMA waves
    const int lane = tid & 63;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
    __syncthreads();
    for (int t0 = 0; t0 < nkt; t0 += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            if (t0 + u < nkt) {
                mma_tile<BM, BN, AL::KC, BL::KC>(smem + u * STAGE, smem + u * STAGE + A_FL, acc, wm, wn, lane);
                __syncthreads();
            }
        }
    }
    store_tile<BM, BN>(acc, smem, ep, partial, M, N, m0, n0, wm, wn);
}

__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int splits, int M, int N, Epilogue ep);
inline int splitk_reduce_blocks(long total) { return (int)((total + 1023) / 1024); }      // four elements per thread

// conv forward on the split-bf16 main loop (conv_bs.hip)
int conv2d_bf16x3(const dc_conv_desc* d, bool stem, const Epilogue& ep, int M, int N, int K, int bm, int bn, int split, void* workspace,
                  size_t workspace_bytes, hipStream_t s);

// short-K pointwise convolutions on the streaming kernel (conv_pw.hip)
bool conv_pw_stream_supported(const dc_conv_desc* d, const Epilogue& ep);
int conv2d_pointwise_stream(const dc_conv_desc* d, const Epilogue& ep, int M, int N, hipStream_t s);
// conv_wino.hip: 3x3 / stride 1 / pad 1 layers with pre-transformed frozen weights in the Winograd F(2x2, 3x3) form
bool conv_winograd_supported(const dc_conv_desc* d);
bool conv_winograd_split_bf16(const dc_conv_desc* d);
int conv_winograd_tiles(const dc_conv_desc* d);
int conv2d_winograd(const dc_conv_desc* d, hipStream_t s);

// Host-side launch helper.  PC = true selects the producer / consumer kernel (64x64 tiles).
template <int BM, int BN, class AL, class BL, bool PC = false>
int launch_igemm(const AL& al, const BL& bl, const Epilogue& ep, int M, int N, int K, int split_k, void* workspace,
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
    constexpr size_t lds = PC ? igemm_pc_lds_bytes<BM, BN, AL, BL>() : igemm_lds_bytes<BM, BN, AL, BL>();
    dim3 grid(tiles, 1, split_k);
    if constexpr (PC) {
        DC_ENSURE_DYN_LDS((&igemm_pc_kernel<BM, BN, AL, BL>), 160 * 1024);
        hipLaunchKernelGGL((igemm_pc_kernel<BM, BN, AL, BL>), grid, dim3(512), lds, stream, al, bl, ep, M, N, K, klen, partial);
    } else {
        DC_ENSURE_DYN_LDS((&igemm_kernel<BM, BN, AL, BL>), 160 * 1024);
        hipLaunchKernelGGL((igemm_kernel<BM, BN, AL, BL>), grid, dim3(256), lds, stream, al, bl, ep, M, N, K, klen, partial);
    }
    int rc = check_launch("igemm_kernel");
    if (rc) return rc;
    if (split_k > 1) {
        const long total = (long)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(splitk_reduce_blocks(total)), dim3(256), 0, stream, partial, split_k, M, N, ep);
        rc = check_launch("splitk_reduce_kernel");
    }
    return rc;
}

// Shared tile/split heuristic: the largest tile that still gives every CU about two blocks; if even
// 64x64 tiles cannot fill the chip, split K (deterministic slab reduction).
struct TileChoice {
    int bm, bn, split;
    bool pc = false;       // 128x64 only: run the producer / consumer kernel (64x64 tiles always do)
};
// Tile / split-K choice.  Grids that fill the chip twice over keep the largest tile that does so; smaller ones run 64x64
// tiles with split-K (the rule the encoder's layer table was tuned with).  Tall-K problems that under-fill the chip
// (K >= 8192: weight gradients over pixels, the vocabulary GEMMs) are priced with a small model: waves of blocks over the
// CUs (2 resident blocks per CU -> half-wave granularity) x K-tiles per slice x time per K-tile, where bigger tiles run
// the MFMA pipe more efficiently (measured: ~0.70 / 0.58 / 0.45 of peak for 128x128 / 128x64 / 64x64), plus the split-K
// slab traffic.
inline TileChoice choose_tile(int M, int N, int K, int user_split, bool allow_128 = true) {
    auto nb = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    TileChoice t{64, 64, 1};
    const int ktiles = (K + BK - 1) / BK;
    bool priced = false;
    if (!allow_128) t = {64, 64, 1};
    else if (N >= 128 && nb(128, 128) >= 2 * kNumCU) t = {128, 128, 1};
    else if (nb(128, 64) >= 2 * kNumCU) t = {128, 64, 1};
    else if (user_split <= 0 && N >= 64 && ktiles >= 256) priced = true;
    if (priced) {
        const int cand[3][2] = {{128, 128}, {128, 64}, {64, 64}};
        const double eff[3] = {0.70, 0.58, 0.45};
        const double cu_flops = 157.3e12 / kNumCU;
        double best = 1e30;
        for (int c = 0; c < 3; ++c) {
            const int bm = cand[c][0], bn = cand[c][1];
            if (bn > 64 && N < 128) continue;
            const double blocks = nb(bm, bn);
            const double t_tile = 2.0 * bm * bn * BK / (cu_flops * eff[c]);
            for (int sp = 1; sp <= 32; sp = sp < 4 ? sp + 1 : sp + sp / 2) {
                if (sp > 1 && ktiles / sp < 4) break;               // keep >= 4 K-tiles (128 deep) per slice
                const double w = blocks * sp / kNumCU;
                const double waves = w <= 1.0 ? 1.0 : ceil(w * 2.0) / 2.0;
                double cost = waves * ((ktiles + sp - 1) / sp) * t_tile + 4e-6;
                if (sp > 1) cost += (double)M * N * 4.0 * (sp + 1) / 4e12 + 3e-6;
                if (cost < best) { best = cost; t = {bm, bn, sp}; }
            }
        }
        return t;
    }
    if (user_split > 0) {
        t.split = user_split;
    } else {
        const int blocks = nb(t.bm, t.bn);
        if (blocks < kNumCU && ktiles >= 8) {
            int s = (2 * kNumCU + blocks - 1) / blocks;
            if (s > ktiles / 4) s = ktiles / 4;      // keep >= 4 K-tiles (128 deep) per slice
            if (s > 32) s = 32;
            if (s < 1) s = 1;
            t.split = s;
        }
    }
    return t;
}

}  // namespace dcap
