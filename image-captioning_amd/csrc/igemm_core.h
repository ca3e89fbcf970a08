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
#include "dcap_internal.h"

namespace dcap {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDKC = BK + 4;

// ------------------------------------------------------------------------------------------------
// Operand loaders.  load<BT>() fills this thread's BT/32 float4 registers for the K-tile starting
// at k0 (zero beyond the operand's edge), store<BT>() writes them into the LDS image.
// ------------------------------------------------------------------------------------------------

template <int BT>
__device__ __forceinline__ void store_kc(float* S, const float4 (&r)[BT / 32], int tid) {
    const int q = tid & 7, rr = tid >> 3;
#pragma unroll
    for (int i = 0; i < BT / 32; ++i) *reinterpret_cast<float4*>(&S[(rr + 32 * i) * LDKC + 4 * q]) = r[i];
}

template <int BT>
__device__ __forceinline__ void store_mc(float* S, const float4 (&r)[BT / 32], int tid) {
    constexpr int QPR = BT / 4, RPP = 256 / QPR;
    const int q = tid % QPR, kr = tid / QPR;
#pragma unroll
    for (int i = 0; i < BT / 32; ++i) *reinterpret_cast<float4*>(&S[(kr + RPP * i) * (BT + 4) + 4 * q]) = r[i];
}

__device__ __forceinline__ float4 load4_guard(const float* a, int avail) {
    // avail = number of valid floats starting at a (>=1)
    if (avail >= 4) return *reinterpret_cast<const float4*>(a);
    float4 v = make_float4(a[0], 0.f, 0.f, 0.f);
    if (avail > 1) v.y = a[1];
    if (avail > 2) v.z = a[2];
    return v;
}

// rows x K, K contiguous; optional row gather (embedding lookup).
struct DenseKC {
    static constexpr bool KC = true;
    const float* p;
    long ld;
    int rows;
    const int32_t* gather;
    template <int BT>
    struct State {
        const float* base[BT / 32];
    };
    template <int BT>
    __device__ __forceinline__ void init(State<BT>& s, int row0, int tid) const {
        const int rr = tid >> 3;
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) {
            const int row = row0 + rr + 32 * i;
            if (row < rows) {
                const long src = gather ? (long)gather[row] : (long)row;
                s.base[i] = p + src * ld;
            } else {
                s.base[i] = nullptr;
            }
        }
    }
    template <int BT>
    __device__ __forceinline__ void load(const State<BT>& s, float4 (&r)[BT / 32], int k0, int kend, int tid) const {
        const int k = k0 + 4 * (tid & 7);
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) {
            r[i] = (s.base[i] != nullptr && k < kend) ? load4_guard(s.base[i] + k, kend - k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    template <int BT>
    __device__ __forceinline__ void store(float* S, const float4 (&r)[BT / 32], int tid) const {
        store_kc<BT>(S, r, tid);
    }
};

// K x cols, cols contiguous (K-major operand).
struct DenseMC {
    static constexpr bool KC = false;
    const float* p;
    long ld;
    int cols;
    const int32_t* gather;   // optional: K row k is stored at row gather[k]
    template <int BT>
    struct State {
        int col;
    };
    template <int BT>
    __device__ __forceinline__ void init(State<BT>& s, int col0, int tid) const {
        s.col = col0 + 4 * (tid % (BT / 4));
    }
    template <int BT>
    __device__ __forceinline__ void load(const State<BT>& s, float4 (&r)[BT / 32], int k0, int kend, int tid) const {
        constexpr int QPR = BT / 4, RPP = 256 / QPR;
        const int kr = tid / QPR;
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) {
            const int k = k0 + kr + RPP * i;
            if (k < kend && s.col < cols) {
                const long src = gather ? (long)gather[k] : (long)k;
                r[i] = load4_guard(p + src * ld + s.col, cols - s.col);
            } else {
                r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    template <int BT>
    __device__ __forceinline__ void store(float* S, const float4 (&r)[BT / 32], int tid) const {
        store_mc<BT>(S, r, tid);
    }
};

// NHWC activations viewed as the im2col matrix [N*Ho*Wo][kh*kw*Cin] (cin fastest), Cin % 32 == 0:
// one K-tile of 32 lies inside one (ky,kx) tap, so a row's 128 B are contiguous in memory.
struct Im2colKC {
    static constexpr bool KC = true;
    const float* x;
    int H, W, Cin, Ho, Wo, stride, pad_t, pad_l, kw, cin_tiles, M;
    template <int BT>
    struct State {
        int iy0[BT / 32], ix0[BT / 32];
        long nb[BT / 32];
    };
    template <int BT>
    __device__ __forceinline__ void init(State<BT>& s, int row0, int tid) const {
        const int rr = tid >> 3;
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) {
            const int m = row0 + rr + 32 * i;
            if (m < M) {
                const int n = m / (Ho * Wo), rem = m - n * (Ho * Wo);
                const int oy = rem / Wo, ox = rem - oy * Wo;
                s.iy0[i] = oy * stride - pad_t;
                s.ix0[i] = ox * stride - pad_l;
                s.nb[i] = (long)n * H;
            } else {
                s.iy0[i] = -(1 << 28);
                s.ix0[i] = 0;
                s.nb[i] = 0;
            }
        }
    }
    template <int BT>
    __device__ __forceinline__ void load(const State<BT>& s, float4 (&r)[BT / 32], int k0, int kend, int tid) const {
        const int kt = k0 >> 5;                         // block-uniform
        const int tap = kt / cin_tiles;
        const int c = (kt - tap * cin_tiles) * 32 + 4 * (tid & 7);
        const int ky = tap / kw, kx = tap - ky * kw;
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) {
            const int iy = s.iy0[i] + ky, ix = s.ix0[i] + kx;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            r[i] = ok ? *reinterpret_cast<const float4*>(x + ((s.nb[i] + iy) * W + ix) * Cin + c)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    template <int BT>
    __device__ __forceinline__ void store(float* S, const float4 (&r)[BT / 32], int tid) const {
        store_kc<BT>(S, r, tid);
    }
};

// The 7x7/stride-2 stem on an RGBX image [N,H,W,4]: K-tile ky = one kernel row, float4 q = tap kx
// (kx == 7 is a zero-weight pad), so K = 7*8*4 = 224 and every load is one aligned pixel.
struct StemKC {
    static constexpr bool KC = true;
    const float* x;
    int H, W, Ho, Wo, M;
    template <int BT>
    struct State {
        int iy0[BT / 32], ix0[BT / 32];
        long nb[BT / 32];
    };
    template <int BT>
    __device__ __forceinline__ void init(State<BT>& s, int row0, int tid) const {
        const int rr = tid >> 3;
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) {
            const int m = row0 + rr + 32 * i;
            if (m < M) {
                const int n = m / (Ho * Wo), rem = m - n * (Ho * Wo);
                const int oy = rem / Wo, ox = rem - oy * Wo;
                s.iy0[i] = oy * 2 - 3;
                s.ix0[i] = ox * 2 - 3 + (tid & 7);
                s.nb[i] = (long)n * H;
            } else {
                s.iy0[i] = -(1 << 28);
                s.ix0[i] = 0;
                s.nb[i] = 0;
            }
        }
    }
    template <int BT>
    __device__ __forceinline__ void load(const State<BT>& s, float4 (&r)[BT / 32], int k0, int kend, int tid) const {
        const int ky = k0 >> 5;
#pragma unroll
        for (int i = 0; i < BT / 32; ++i) {
            const int iy = s.iy0[i] + ky, ix = s.ix0[i];
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            r[i] = ok ? *reinterpret_cast<const float4*>(x + ((s.nb[i] + iy) * W + ix) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    template <int BT>
    __device__ __forceinline__ void store(float* S, const float4 (&r)[BT / 32], int tid) const {
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
};

// ------------------------------------------------------------------------------------------------
// One K-tile of MFMAs from the LDS images.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, bool AKC, bool BKC>
__device__ __forceinline__ void mma_tile(const float* __restrict__ As, const float* __restrict__ Bs,
                                         f32x16 (&acc)[BM / 64][BN / 64], int wm, int wn, int lane) {
    constexpr int TM = BM / 64, TN = BN / 64;
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float a[TM][4], b[TN][4];
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
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
    }
}

template <int BT, bool KC>
constexpr int lds_floats() {
    return KC ? BT * LDKC : BK * (BT + 4);
}

template <int BM, int BN, class AL, class BL>
constexpr size_t igemm_lds_bytes() {
    return 2 * (size_t)(lds_floats<BM, AL::KC>() + lds_floats<BN, BL::KC>()) * sizeof(float);
}

// ------------------------------------------------------------------------------------------------
// The kernel.  grid.x = tiles_m*tiles_n (XCD-remapped so consecutive tiles along N, which share the
// A panel, run on one XCD), grid.z = split-K slices.  With split-K the raw partial sums go to the
// slab  partial[z][M][N]  and splitk_reduce_kernel applies the epilogue.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, class AL, class BL>
__global__ __launch_bounds__(256) void igemm_kernel(AL al, BL bl, Epilogue ep, int M, int N, int K, int klen,
                                                    float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_FL = lds_floats<BM, AL::KC>(), B_FL = lds_floats<BN, BL::KC>(), STAGE = A_FL + B_FL;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
    const int kbeg = blockIdx.z * klen;
    const int kend = min(K, kbeg + klen);
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);

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

    float4 ra[BM / 32], rb[BN / 32];
    const int nkt = (kend - kbeg + BK - 1) / BK;
    if (nkt > 0) {
        al.template load<BM>(sa, ra, kbeg, kend, tid);
        bl.template load<BN>(sb, rb, kbeg, kend, tid);
        al.template store<BM>(smem, ra, tid);
        bl.template store<BN>(smem + A_FL, rb, tid);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        float* cur = smem + (kt & 1) * STAGE;
        float* nxt = smem + ((kt & 1) ^ 1) * STAGE;
        const bool more = kt + 1 < nkt;
        if (more) {
            al.template load<BM>(sa, ra, kbeg + (kt + 1) * BK, kend, tid);
            bl.template load<BN>(sb, rb, kbeg + (kt + 1) * BK, kend, tid);
        }
        mma_tile<BM, BN, AL::KC, BL::KC>(cur, cur + A_FL, acc, wm, wn, lane);
        if (more) {
            al.template store<BM>(nxt, ra, tid);
            bl.template store<BN>(nxt + A_FL, rb, tid);
        }
        __syncthreads();
    }

    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = n0 + wn + tn * 32 + i;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= M) continue;
                if (partial) {
                    partial[((long)blockIdx.z * M + row) * N + col] = acc[tm][tn][r];
                } else {
                    ep.C[(long)row * ep.ldc + col] = ep.apply(acc[tm][tn][r], row, col);
                }
            }
        }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int splits, int M, int N, Epilogue ep);

// Host-side launch helper (defined in igemm_launch.hip).
template <int BM, int BN, class AL, class BL>
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
    constexpr size_t lds = igemm_lds_bytes<BM, BN, AL, BL>();
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, AL, BL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid(tiles, 1, split_k);
    hipLaunchKernelGGL((igemm_kernel<BM, BN, AL, BL>), grid, dim3(256), lds, stream, al, bl, ep, M, N, K, klen, partial);
    int rc = check_launch("igemm_kernel");
    if (rc) return rc;
    if (split_k > 1) {
        const long total = (long)M * N;
        const int blocks = (int)((total + 255) / 256);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, partial, split_k, M, N, ep);
        rc = check_launch("splitk_reduce_kernel");
    }
    return rc;
}

// Shared tile/split heuristic: the largest tile that still gives every CU about two blocks; if even
// 64x64 tiles cannot fill the chip, split K (deterministic slab reduction).
struct TileChoice {
    int bm, bn, split;
};
inline TileChoice choose_tile(int M, int N, int K, int user_split, bool allow_128 = true) {
    auto nb = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    TileChoice t{64, 64, 1};
    if (allow_128 && N >= 128 && nb(128, 128) >= 2 * kNumCU) t = {128, 128, 1};
    else if (allow_128 && nb(128, 64) >= 2 * kNumCU) t = {128, 64, 1};
    if (user_split > 0) {
        t.split = user_split;
    } else {
        const int blocks = nb(t.bm, t.bn);
        const int ktiles = (K + BK - 1) / BK;
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
