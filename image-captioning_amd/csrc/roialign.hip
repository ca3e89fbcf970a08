// roialign.hip -- PyramidROIAlign forward: FPN level routing + tf.image.crop_and_resize (bilinear,
// one sample per bin, extrapolation 0) as one HBM-bound gather kernel.
// One wave per output bin: the 64 lanes read the four corner pixels' channel vectors as float4
// (C = 256 -> one coalesced 1 KiB row per corner) and write the bin's 1 KiB row.
// Sample coordinates follow TF's float32 kernel operation by operation (mul, div, mul, add -- no fma
// contraction) so the in-range test and floor/ceil decisions match the reference's CPU path.
#include "dcap_internal.h"
#include <math.h>
#include <algorithm>

// Every float operation of this file is rounded on its own, as the TF float32 kernels it mirrors round theirs: the decisions taken on the
// results (a sample inside the map or not, floor/ceil, IoU > threshold, sort order) are discontinuous, and a multiply fused into the
// following add moves them.  HIP's __fmul_rn / __fadd_rn do not prevent that: they are plain operators in a header parsed before this
// pragma, so their operations stay fusable after inlining -- the helpers below are compiled under it.
#pragma clang fp contract(off)
namespace dcap {
__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }      // correctly rounded: hipcc's default for fp32 division
}

namespace dcap {

__device__ __forceinline__ int roi_level(float y1, float x1, float y2, float x2, float image_area) {
    const float h = sub_rn(y2, y1), w = sub_rn(x2, x1);
    const float ratio = div_rn(__fsqrt_rn(mul_rn(h, w)), div_rn(224.0f, __fsqrt_rn(image_area)));
    const float lvl = div_rn(logf(ratio), logf(2.0f));
    if (!(lvl > -100.f)) return 2;          // log(0) = -inf, NaN: TF's int cast underflows, the clamp gives 2
    const int r = (int)rintf(lvl);          // round half to even, like tf.round
    return min(5, max(2, 4 + r));
}

__global__ __launch_bounds__(256) void roi_align_kernel(dc_roialign_desc d) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int bins = d.pool * d.pool;
    const int total = d.B * d.R * bins;
    if (wave >= total) return;
    const int box = wave / bins, bin = wave - box * bins;
    const int py = bin / d.pool, px = bin - py * d.pool;
    const float4 bx = reinterpret_cast<const float4*>(d.boxes)[box];       // y1,x1,y2,x2
    const int lvl = roi_level(bx.x, bx.y, bx.z, bx.w, d.image_area);
    if (d.levels_out && bin == 0 && lane == 0) d.levels_out[box] = lvl;
    const int li = lvl - 2;
    const int H = d.Hs[li], W = d.Ws[li], C4 = d.C >> 2;
    const float* fm = d.maps[li] + (long)(box / d.R) * H * W * d.C;
    float4* out = reinterpret_cast<float4*>(d.out) + (long)wave * C4;

    const float hs = (d.pool > 1) ? div_rn(mul_rn(sub_rn(bx.z, bx.x), (float)(H - 1)), (float)(d.pool - 1)) : 0.f;
    const float ws = (d.pool > 1) ? div_rn(mul_rn(sub_rn(bx.w, bx.y), (float)(W - 1)), (float)(d.pool - 1)) : 0.f;
    const float in_y = (d.pool > 1) ? add_rn(mul_rn(bx.x, (float)(H - 1)), mul_rn((float)py, hs))
                                    : mul_rn(mul_rn(0.5f, add_rn(bx.x, bx.z)), (float)(H - 1));
    const float in_x = (d.pool > 1) ? add_rn(mul_rn(bx.y, (float)(W - 1)), mul_rn((float)px, ws))
                                    : mul_rn(mul_rn(0.5f, add_rn(bx.y, bx.w)), (float)(W - 1));
    const bool ok = (in_y >= 0.f) && (in_y <= (float)(H - 1)) && (in_x >= 0.f) && (in_x <= (float)(W - 1));
    if (!ok) {
        for (int c = lane; c < C4; c += 64) out[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int top = (int)floorf(in_y), bot = (int)ceilf(in_y), left = (int)floorf(in_x), right = (int)ceilf(in_x);
    const float ly = in_y - (float)top, lx = in_x - (float)left;
    const float4* tl = reinterpret_cast<const float4*>(fm + ((long)top * W + left) * d.C);
    const float4* tr = reinterpret_cast<const float4*>(fm + ((long)top * W + right) * d.C);
    const float4* bl = reinterpret_cast<const float4*>(fm + ((long)bot * W + left) * d.C);
    const float4* br = reinterpret_cast<const float4*>(fm + ((long)bot * W + right) * d.C);
    for (int c = lane; c < C4; c += 64) {
        const float4 a = tl[c], b = tr[c], e = bl[c], f = br[c];
        float4 o;
#define DC_LERP(k)                                  \
    {                                               \
        const float t = a.k + (b.k - a.k) * lx;     \
        const float u = e.k + (f.k - e.k) * lx;     \
        o.k = t + (u - t) * ly;                     \
    }
        DC_LERP(x) DC_LERP(y) DC_LERP(z) DC_LERP(w)
        out[c] = o;
    }
}

// Sample coordinates of bin row / column `p` of a box on a map of extent n (TF's operation order, as in the forward kernel).
__device__ __forceinline__ float roi_sample(float lo, float hi, int p, int n, int pool) {
    if (pool > 1) {
        const float step = div_rn(mul_rn(sub_rn(hi, lo), (float)(n - 1)), (float)(pool - 1));
        return add_rn(mul_rn(lo, (float)(n - 1)), mul_rn((float)p, step));
    }
    return mul_rn(mul_rn(0.5f, add_rn(lo, hi)), (float)(n - 1));
}

// backward, DETERMINISTIC (round 3): gather per destination pixel instead of an atomic scatter per bin.  One wave per pixel of
// a pyramid level: the lanes test 64 RoIs at a time (routed to this level? does its sampling grid come within one pixel of
// this one?), the hits are then visited in ascending (RoI, bin row, bin column) order -- the same order on every run and on
// every rank -- and each contributing bin adds  weight x its 256-channel gradient row  (four channels per lane) to a register
// accumulator that is added to the map with ONE plain read-modify-write.  weight = the bilinear corner weight the forward used:
//   wy = [y == floor(in_y)] (1 - ly) + [y == ceil(in_y)] ly,  wx likewise  (both terms when in_y is integral: 1).
// No float atomics: the joint model's FPN / RPN gradients become bit-reproducible (they were the only ones that were not).
constexpr int RA_MAX_POOL = 16;

__global__ __launch_bounds__(256) void roi_align_bwd_gather_kernel(dc_roialign_desc d, int li, long wave0) {
    const int lane = threadIdx.x & 63;
    const long wv = wave0 + (((long)blockIdx.x * 256 + threadIdx.x) >> 6);
    const int H = d.Hs[li], W = d.Ws[li];
    const long per_img = (long)H * W;
    if (wv >= (long)d.B * per_img) return;
    const int img = (int)(wv / per_img);
    const int pix = (int)(wv - (long)img * per_img);
    const int y = pix / W, x = pix - y * W;
    const float4* boxes = reinterpret_cast<const float4*>(d.boxes) + (long)img * d.R;
    const int bins = d.pool * d.pool;
    float4 acc[4];                                   // channels 4 lane .. 4 lane + 3 (+ 256 k): C <= 1024
    const int C4 = d.C >> 2, nchunk = (C4 + 63) >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    bool any = false;
    for (int b0 = 0; b0 < d.R; b0 += 64) {
        const int bi = b0 + lane;
        bool hit = false;
        if (bi < d.R) {
            const float4 bx = boxes[bi];
            if (roi_level(bx.x, bx.y, bx.z, bx.w, d.image_area) - 2 == li) {
                // the sampling grid spans [in(0), in(pool - 1)] (either order for a flipped box): can it touch row y / column x?
                const float ya = roi_sample(bx.x, bx.z, 0, H, d.pool), yb = roi_sample(bx.x, bx.z, d.pool - 1, H, d.pool);
                const float xa = roi_sample(bx.y, bx.w, 0, W, d.pool), xb = roi_sample(bx.y, bx.w, d.pool - 1, W, d.pool);
                hit = fminf(ya, yb) < (float)y + 1.f && fmaxf(ya, yb) > (float)y - 1.f && fminf(xa, xb) < (float)x + 1.f && fmaxf(xa, xb) > (float)x - 1.f;
            }
        }
        unsigned long long m = __ballot(hit);
        while (m) {                                  // wave-uniform: ascending RoI index
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int box = b0 + j;
            const float4 bx = boxes[box];            // same address in every lane: one broadcast load
            float wys[RA_MAX_POOL], wxs[RA_MAX_POOL];
#pragma unroll 1
            for (int p = 0; p < d.pool; ++p) {
                const float iy = roi_sample(bx.x, bx.z, p, H, d.pool), ix = roi_sample(bx.y, bx.w, p, W, d.pool);
                float wy = 0.f, wx = 0.f;
                if (iy >= 0.f && iy <= (float)(H - 1)) {
                    const float ly = iy - floorf(iy);
                    if ((int)floorf(iy) == y) wy += 1.f - ly;
                    if ((int)ceilf(iy) == y) wy += ly;
                } else {
                    wy = -1.f;                       // this bin row is outside the map: the forward wrote zeros, no gradient
                }
                if (ix >= 0.f && ix <= (float)(W - 1)) {
                    const float lx = ix - floorf(ix);
                    if ((int)floorf(ix) == x) wx += 1.f - lx;
                    if ((int)ceilf(ix) == x) wx += lx;
                } else {
                    wx = -1.f;
                }
                wys[p] = wy;
                wxs[p] = wx;
            }
            const float* go = d.out + ((long)(img * d.R + box) * bins) * d.C;
#pragma unroll 1
            for (int py = 0; py < d.pool; ++py) {
                if (!(wys[py] > 0.f)) continue;
#pragma unroll 1
                for (int px = 0; px < d.pool; ++px) {
                    if (!(wxs[px] > 0.f)) continue;
                    const float wgt = wys[py] * wxs[px];
                    const float4* row = reinterpret_cast<const float4*>(go + (long)(py * d.pool + px) * d.C);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < nchunk && lane + 64 * k < C4) {
                            const float4 g = row[lane + 64 * k];
                            acc[k].x += g.x * wgt; acc[k].y += g.y * wgt; acc[k].z += g.z * wgt; acc[k].w += g.w * wgt;
                        }
                    any = true;
                }
            }
        }
    }
    if (!any) return;
    float4* dst = reinterpret_cast<float4*>(const_cast<float*>(d.maps[li]) + ((long)img * per_img + pix) * d.C);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < nchunk && lane + 64 * k < C4) {
            float4 v = dst[lane + 64 * k];
            v.x += acc[k].x; v.y += acc[k].y; v.z += acc[k].z; v.w += acc[k].w;
            dst[lane + 64 * k] = v;
        }
}

}  // namespace dcap

using namespace dcap;

extern "C" int dc_roi_align_pyramid_bwd_f32(const dc_roialign_desc* d, void* stream) {
    DC_REQUIRE(d && d->boxes && d->out, DC_EINVAL, "dc_roi_align_pyramid_bwd: null pointer");
    DC_REQUIRE(d->B > 0 && d->R > 0 && d->pool > 0 && d->pool <= RA_MAX_POOL && d->C > 0 && (d->C & 3) == 0 && d->C <= 1024, DC_EINVAL,
               "dc_roi_align_pyramid_bwd: bad B/R/pool/C (pool <= 16, C a multiple of 4 and <= 1024)");
    for (int l = 0; l < 4; ++l) {
        DC_REQUIRE(d->maps[l] && d->Hs[l] > 0 && d->Ws[l] > 0, DC_EINVAL, "dc_roi_align_pyramid_bwd: bad map %d", l);
        DC_REQUIRE(aligned16(d->maps[l]), DC_EALIGN, "dc_roi_align_pyramid_bwd: gradient map %d not 16-byte aligned", l);
    }
    DC_REQUIRE(aligned16(d->boxes) && aligned16(d->out), DC_EALIGN, "dc_roi_align_pyramid_bwd: boxes / dout not 16-byte aligned");
    for (int l = 0; l < 4; ++l) {
        const long waves = (long)d->B * d->Hs[l] * d->Ws[l];
        for (long w0 = 0; w0 < waves; w0 += 4L * 0x40000000) {       // (grids stay below 2^31 blocks)
            const long n = std::min(waves - w0, 4L * 0x40000000);
            hipLaunchKernelGGL(roi_align_bwd_gather_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), *d, l, w0);
        }
        int rc = check_launch("roi_align_bwd_gather_kernel");
        if (rc) return rc;
    }
    return DC_OK;
}

extern "C" int dc_roi_align_pyramid_f32(const dc_roialign_desc* d, void* stream) {
    DC_REQUIRE(d && d->boxes && d->out, DC_EINVAL, "dc_roi_align_pyramid: null pointer");
    DC_REQUIRE(d->B > 0 && d->R > 0 && d->pool > 0 && d->C > 0 && (d->C & 3) == 0, DC_EINVAL, "dc_roi_align_pyramid: bad B/R/pool/C");
    for (int l = 0; l < 4; ++l) {
        DC_REQUIRE(d->maps[l] && d->Hs[l] > 0 && d->Ws[l] > 0, DC_EINVAL, "dc_roi_align_pyramid: bad feature map %d", l);
        DC_REQUIRE(aligned16(d->maps[l]), DC_EALIGN, "dc_roi_align_pyramid: feature map %d not 16-byte aligned", l);
    }
    DC_REQUIRE(aligned16(d->boxes) && aligned16(d->out), DC_EALIGN, "dc_roi_align_pyramid: boxes/out not 16-byte aligned");
    const long waves = (long)d->B * d->R * d->pool * d->pool;
    const int blocks = (int)((waves + 3) / 4);
    hipLaunchKernelGGL(roi_align_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *d);
    return check_launch("roi_align_kernel");
}
