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
// One kernel does all of it (nothing of V or M ever reaches HBM).  Block = 256 threads = 4 waves, 32 tiles (4 x 8: 8 x 16 output
// pixels) x NB = 32 NT output channels, K-chunks of 32 input channels:
//   * input transform: thread (tile, channel quad) loads its 4 x 4 patch as 16 dwordx4 (out-of-image pixels carry an out-of-range
//     buffer offset: hardware zeros = TF 'SAME' padding), 32 adds per channel, 16 ds_write_b128 into V[16][32 tiles][32 k]
//     (64 KiB, 128-B rows, 16-B chunk c of row r at c ^ ((r >> 1) & 7): conflict-free for these writes and for the reads below);
//     the loads of chunk k + 1 are issued before the MFMAs of chunk k;
//   * MFMA phase: wave w owns xi = 4 w .. 4 w + 3 (no two waves share a weight): per xi 4 ds_read_b128 of V and 4 NT
//     global_load_dwordx4 of U straight into registers -- U is packed in fragment order, 1 KiB contiguous per wave-instruction,
//     and prefetched one xi ahead -- feed 16 NT MFMAs.  The MFMA takes U as its A operand (rows = output channels), so a lane
//     ends up with four consecutive output channels of one tile per accumulator quad;
//   * output transform: the 16 M[xi] tiles go through the same LDS image, thread (tile, channel quad) gathers its 16 values,
//     applies A^T . A and the epilogue, and stores 16-byte pieces (128 contiguous bytes per pixel and block).
// Two blocks per CU (64 KiB of LDS, <= 256 VGPRs each): one block's input transform runs beside the other's MFMAs.
// Grid: block -> (output-channel slice, tile group), slice-major through the XCD remap, so an XCD's L2 keeps ONE slice of U
// (16 x Cin x NB x 4 bytes) and streams the activations once.
#include "igemm_core.h"
#include <algorithm>

namespace dcap {
namespace wino {

constexpr int TILES = 32, TGY = 4, TGX = 8, KCH = 32, NTHREADS = 256;
constexpr int LDS_BYTES = 16 * TILES * KCH * 4;          // 64 KiB

struct Args {
    const float* x;
    const f4* u;              // packed by wino_pack_kernel
    float* y;
    const float* scale;
    const float* shift;
    int N, H, W, Cin, Cout, relu;
    int gy, gx, groups;       // tile groups per image (rows, columns), in all
    unsigned x_bytes;
};

__device__ __forceinline__ unsigned img_addr(int xi, int tile, int chunk) {
    return (unsigned)((((xi * TILES + tile) << 3) + (chunk ^ ((tile >> 1) & 7))) << 4);
}

// U in fragment order: f4 index ((((xi * NTG + ntg) * KC + kc) * 4 + j) * 64 + lane), lane = 32 h + i, component e
//   <->  cout = 32 ntg + i, cin = 32 kc + 16 h + 4 j + e        (MFMA step 4 j + e contracts cin 16 h + 4 j + e of the chunk)
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout) {
    const long total = (long)Cin * Cout;
    const int NTG = Cout / 32, KC = Cin / 32;
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
        const int ntg = cout >> 5, i = cout & 31, kc = cin >> 5, kk = cin & 31, h = kk >> 4, j = (kk & 15) >> 2, e = kk & 3;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float r[4] = {gg[a][0], 0.5f * (gg[a][0] + gg[a][1] + gg[a][2]), 0.5f * (gg[a][0] - gg[a][1] + gg[a][2]), gg[a][2]};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xi = 4 * a + b;
                u[((((long)(xi * NTG + ntg) * KC + kc) * 4 + j) * 64 + (h * 32 + i)) * 4 + e] = r[b];
            }
        }
    }
}

template <int NT>
__global__ __launch_bounds__(NTHREADS, NT == 1 ? 2 : 1) void wino_conv_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nb = bid / a.groups, g = bid - nb * a.groups;
    const int gpi = a.gy * a.gx;
    const int img = g / gpi, gr = g - img * gpi;
    const int gyi = gr / a.gx, gxi = gr - gyi * a.gx;
    const int NTG = a.Cout >> 5, KC = a.Cin >> 5;

    // ---- this thread's tile and channel quad (input and output transforms)
    const int ti = tid >> 3, q = tid & 7;
    const int ty = gyi * TGY + (ti >> 3), tx = gxi * TGX + (ti & 7);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    unsigned off[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int iy = 2 * ty - 1 + r, ix = 2 * tx - 1 + c;
            const bool in = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            off[r][c] = in ? (unsigned)(((((long)img * a.H + iy) * a.W + ix) * a.Cin + 4 * q) * 4) : kOobOffset;
        }
    f4 raw[4][4];
    auto load_raw = [&](int kc) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) raw[r][c] = buf_f4s(rsrc, off[r][c], (unsigned)(kc * KCH * 4));
    };
#ifdef WINO_EXP_NORAW
    auto load_raw_loop = [&](int) {};
#else
    auto load_raw_loop = load_raw;
#endif

    // ---- this wave's xi range and fragment addresses
    const int fi = lane & 31, fh = lane >> 5;
    const f4* ubase = a.u + lane;
    auto load_u = [&](f4 (&dst)[NT][4], int kc, int xl) {
        const int xi = 4 * wave + xl;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const f4* p = ubase + (((long)(xi * NTG + nb * NT + n) * KC + kc) << 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[n][j] = p[j * 64];
        }
    };

    f32x16 acc[4][NT];
#pragma unroll
    for (int xl = 0; xl < 4; ++xl)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[xl][n][r] = 0.f;

    f4 ub[2][NT][4];
    load_raw(0);
    load_u(ub[0], 0, 0);
#ifdef WINO_EXP_NOU
    load_u(ub[1], 0, 1);
#endif

    for (int kc = 0; kc < KC; ++kc) {
        // input transform B^T d B of this thread's patch, four channels at a time
        f4 v[4][4];
        {
            f4 t[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0][c] = raw[0][c] - raw[2][c];
                t[1][c] = raw[1][c] + raw[2][c];
                t[2][c] = raw[2][c] - raw[1][c];
                t[3][c] = raw[1][c] - raw[3][c];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r][0] = t[r][0] - t[r][2];
                v[r][1] = t[r][1] + t[r][2];
                v[r][2] = t[r][2] - t[r][1];
                v[r][3] = t[r][1] - t[r][3];
            }
        }
        __syncthreads();                                   // every wave is done reading the previous chunk's V
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<f4*>(smem + img_addr(4 * r + c, ti, q)) = v[r][c];
        load_raw_loop(min(kc + 1, KC - 1));                // in flight during the MFMA phase (the last one re-reads: uniform counts)
        __syncthreads();
#pragma unroll
        for (int xl = 0; xl < 4; ++xl) {
            const int cur = xl & 1;
#ifndef WINO_EXP_NOU
            if (xl < 3) load_u(ub[cur ^ 1], kc, xl + 1);
            else load_u(ub[cur ^ 1], min(kc + 1, KC - 1), 0);
#endif
            f4 vb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) vb[j] = *reinterpret_cast<const f4*>(smem + img_addr(4 * wave + xl, fi, 4 * fh + j));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#ifdef WINO_EXP_NOMFMA
                        acc[xl][n][(4 * j + e) & 15] += ub[cur][n][j][e] * vb[j][e];
#else
                        acc[xl][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ub[cur][n][j][e], vb[j][e], acc[xl][n], 0, 0, 0);
#endif
        }
    }

    // ---- output transform and epilogue, one 32-channel slice at a time through the LDS image
    const bool row_in[2] = {2 * ty < a.H, 2 * ty + 1 < a.H}, col_in[2] = {2 * tx < a.W, 2 * tx + 1 < a.W};
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        __syncthreads();
#pragma unroll
        for (int xl = 0; xl < 4; ++xl)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f4 m = {acc[xl][n][4 * gq], acc[xl][n][4 * gq + 1], acc[xl][n][4 * gq + 2], acc[xl][n][4 * gq + 3]};
                *reinterpret_cast<f4*>(smem + img_addr(4 * wave + xl, fi, 2 * gq + fh)) = m;      // couts 8 gq + 4 h .. + 3 of tile fi
            }
        __syncthreads();
        f4 m[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) m[r][c] = *reinterpret_cast<const f4*>(smem + img_addr(4 * r + c, ti, q));
        const int cout = (nb * NT + n) * 32 + 4 * q;
        f4 sc = (f4)(1.f), sh = (f4)(0.f);
        if (a.scale) sc = *reinterpret_cast<const f4*>(a.scale + cout);
        if (a.shift) sh = *reinterpret_cast<const f4*>(a.shift + cout);
        f4 s[2][4];                                        // A^T M
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s[0][c] = m[0][c] + m[1][c] + m[2][c];
            s[1][c] = m[1][c] - m[2][c] - m[3][c];
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f4 o[2] = {s[p][0] + s[p][1] + s[p][2], s[p][1] - s[p][2] - s[p][3]};
#pragma unroll
            for (int qx = 0; qx < 2; ++qx) {
                f4 val = o[qx] * sc + sh;
                if (a.relu) val = f4{fmaxf(val[0], 0.f), fmaxf(val[1], 0.f), fmaxf(val[2], 0.f), fmaxf(val[3], 0.f)};
                if (row_in[p] && col_in[qx])
                    *reinterpret_cast<f4*>(a.y + (((long)img * a.H + 2 * ty + p) * a.W + 2 * tx + qx) * a.Cout + cout) = val;
            }
        }
    }
}

}  // namespace wino

bool conv_winograd_supported(const dc_conv_desc* d) {
    return d->w_wino != nullptr && d->math == DC_MATH_F32 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 &&
           d->Ho == d->H && d->Wo == d->W && d->Cin % 32 == 0 && d->Cout % 32 == 0 && d->res_mode == 0 && d->split_k <= 1 && aligned16(d->y) &&
           aligned16(d->w_wino) && (!d->scale || aligned16(d->scale)) && (!d->shift || aligned16(d->shift));
}

int conv_winograd_slices(const dc_conv_desc* d) {
    // output channels per block: 64 where the grid still covers the chip twice over (half the V transforms and V reads per MFMA)
    static const int force = env_int("DCAP_WINO_NT", 0);
    if (force == 1 || force == 2) return (d->Cout % (32 * force) == 0) ? force : 1;
    return 1;
}

int conv2d_winograd(const dc_conv_desc* d, hipStream_t s) {
    wino::Args a;
    a.x = d->x;
    a.u = reinterpret_cast<const f4*>(d->w_wino);
    a.y = d->y;
    a.scale = d->scale;
    a.shift = d->shift;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.relu = d->relu;
    const int th = (d->H + 1) / 2, tw = (d->W + 1) / 2;
    a.gy = (th + wino::TGY - 1) / wino::TGY;
    a.gx = (tw + wino::TGX - 1) / wino::TGX;
    a.groups = d->N * a.gy * a.gx;
    a.x_bytes = (unsigned)((size_t)d->N * d->H * d->W * d->Cin * sizeof(float));
    const int nt = conv_winograd_slices(d);
    const long blocks = (long)a.groups * (d->Cout / (32 * nt));
    DC_REQUIRE(blocks < (1l << 31), DC_EINVAL, "dc_conv2d (winograd): grid too large");
    if (nt == 2) {
        DC_ENSURE_DYN_LDS(wino::wino_conv_kernel<2>, wino::LDS_BYTES);
        hipLaunchKernelGGL(wino::wino_conv_kernel<2>, dim3((unsigned)blocks), dim3(wino::NTHREADS), wino::LDS_BYTES, s, a);
    } else {
        DC_ENSURE_DYN_LDS(wino::wino_conv_kernel<1>, wino::LDS_BYTES);
        hipLaunchKernelGGL(wino::wino_conv_kernel<1>, dim3((unsigned)blocks), dim3(wino::NTHREADS), wino::LDS_BYTES, s, a);
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
