// roialign.hip -- PyramidROIAlign forward: FPN level routing + tf.image.crop_and_resize (bilinear,
// one sample per bin, extrapolation 0) as one HBM-bound gather kernel.
// One wave per output bin: the 64 lanes read the four corner pixels' channel vectors as float4
// (C = 256 -> one coalesced 1 KiB row per corner) and write the bin's 1 KiB row.
// Sample coordinates follow TF's float32 kernel operation by operation (mul, div, mul, add -- no fma
// contraction) so the in-range test and floor/ceil decisions match the reference's CPU path.
#include "dcap_internal.h"
#include <math.h>

namespace dcap {

__device__ __forceinline__ int roi_level(float y1, float x1, float y2, float x2, float image_area) {
    const float h = __fsub_rn(y2, y1), w = __fsub_rn(x2, x1);
    const float ratio = __fdiv_rn(__fsqrt_rn(__fmul_rn(h, w)), __fdiv_rn(224.0f, __fsqrt_rn(image_area)));
    const float lvl = __fdiv_rn(logf(ratio), logf(2.0f));
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

    const float hs = (d.pool > 1) ? __fdiv_rn(__fmul_rn(__fsub_rn(bx.z, bx.x), (float)(H - 1)), (float)(d.pool - 1)) : 0.f;
    const float ws = (d.pool > 1) ? __fdiv_rn(__fmul_rn(__fsub_rn(bx.w, bx.y), (float)(W - 1)), (float)(d.pool - 1)) : 0.f;
    const float in_y = (d.pool > 1) ? __fadd_rn(__fmul_rn(bx.x, (float)(H - 1)), __fmul_rn((float)py, hs))
                                    : __fmul_rn(__fmul_rn(0.5f, __fadd_rn(bx.x, bx.z)), (float)(H - 1));
    const float in_x = (d.pool > 1) ? __fadd_rn(__fmul_rn(bx.y, (float)(W - 1)), __fmul_rn((float)px, ws))
                                    : __fmul_rn(__fmul_rn(0.5f, __fadd_rn(bx.y, bx.w)), (float)(W - 1));
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

// backward: one wave per output bin scatters its gradient row into the four corner pixels of the routed level
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(dc_roialign_desc d) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int bins = d.pool * d.pool;
    if (wave >= d.B * d.R * bins) return;
    const int box = wave / bins, bin = wave - box * bins;
    const int py = bin / d.pool, px = bin - py * d.pool;
    const float4 bx = reinterpret_cast<const float4*>(d.boxes)[box];
    const int li = roi_level(bx.x, bx.y, bx.z, bx.w, d.image_area) - 2;
    const int H = d.Hs[li], W = d.Ws[li];
    float* gm = const_cast<float*>(d.maps[li]) + (long)(box / d.R) * H * W * d.C;
    const float* go = d.out + (long)wave * d.C;
    const float hs = (d.pool > 1) ? __fdiv_rn(__fmul_rn(__fsub_rn(bx.z, bx.x), (float)(H - 1)), (float)(d.pool - 1)) : 0.f;
    const float ws = (d.pool > 1) ? __fdiv_rn(__fmul_rn(__fsub_rn(bx.w, bx.y), (float)(W - 1)), (float)(d.pool - 1)) : 0.f;
    const float in_y = (d.pool > 1) ? __fadd_rn(__fmul_rn(bx.x, (float)(H - 1)), __fmul_rn((float)py, hs))
                                    : __fmul_rn(__fmul_rn(0.5f, __fadd_rn(bx.x, bx.z)), (float)(H - 1));
    const float in_x = (d.pool > 1) ? __fadd_rn(__fmul_rn(bx.y, (float)(W - 1)), __fmul_rn((float)px, ws))
                                    : __fmul_rn(__fmul_rn(0.5f, __fadd_rn(bx.y, bx.w)), (float)(W - 1));
    if (!((in_y >= 0.f) && (in_y <= (float)(H - 1)) && (in_x >= 0.f) && (in_x <= (float)(W - 1)))) return;
    const int top = (int)floorf(in_y), bot = (int)ceilf(in_y), left = (int)floorf(in_x), right = (int)ceilf(in_x);
    const float ly = in_y - (float)top, lx = in_x - (float)left;
    float* tl = gm + ((long)top * W + left) * d.C;
    float* tr = gm + ((long)top * W + right) * d.C;
    float* bl = gm + ((long)bot * W + left) * d.C;
    float* br = gm + ((long)bot * W + right) * d.C;
    for (int c = lane; c < d.C; c += 64) {
        const float g = go[c];
        atomicAdd(tl + c, g * (1.f - ly) * (1.f - lx));
        atomicAdd(tr + c, g * (1.f - ly) * lx);
        atomicAdd(bl + c, g * ly * (1.f - lx));
        atomicAdd(br + c, g * ly * lx);
    }
}

}  // namespace dcap

using namespace dcap;

extern "C" int dc_roi_align_pyramid_bwd_f32(const dc_roialign_desc* d, void* stream) {
    DC_REQUIRE(d && d->boxes && d->out, DC_EINVAL, "dc_roi_align_pyramid_bwd: null pointer");
    DC_REQUIRE(d->B > 0 && d->R > 0 && d->pool > 0 && d->C > 0, DC_EINVAL, "dc_roi_align_pyramid_bwd: bad B/R/pool/C");
    for (int l = 0; l < 4; ++l) DC_REQUIRE(d->maps[l] && d->Hs[l] > 0 && d->Ws[l] > 0, DC_EINVAL, "dc_roi_align_pyramid_bwd: bad map %d", l);
    DC_REQUIRE(aligned16(d->boxes), DC_EALIGN, "dc_roi_align_pyramid_bwd: boxes not 16-byte aligned");
    const long waves = (long)d->B * d->R * d->pool * d->pool;
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3((int)((waves + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), *d);
    return check_launch("roi_align_bwd_kernel");
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
