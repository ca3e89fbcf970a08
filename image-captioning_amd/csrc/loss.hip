// loss.hip -- vocabulary softmax + Keras categorical cross-entropy (forward + d/dlogits), greedy
// argmax, column sums (bias gradients), reductions and the fused AMSGrad update.
// All of these are HBM-bandwidth kernels: 16-byte accesses, wavefront (64-lane) shuffles for the
// row reductions, one pass over the data wherever the maths allows.
#include <type_traits>
#include "dcap_internal.h"
#include <algorithm>
#include <math.h>

namespace dcap {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <bool IS_MAX>
__device__ __forceinline__ float block_reduce(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = IS_MAX ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = IS_MAX ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

// One 256-thread block per row.  p = softmax(z); loss = -log(clip(p_t, 1e-7, 1-1e-7));
// dz = grad_scale * (p - onehot) when p_t lies inside the clip range, else 0
// (K.categorical_crossentropy's renormalisation p/sum(p) is the identity on a softmax row up to
// fp32 rounding and is not repeated here).
__global__ __launch_bounds__(256) void softmax_ce_kernel(dc_softmax_ce_desc d) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* z = d.logits + (long)row * d.ld;
    const int V = d.V, V4 = ((d.ld & 3) == 0) ? (V >> 2) : 0;
    const float4* z4 = reinterpret_cast<const float4*>(z);
    float mx = -INFINITY;
    for (int i = tid; i < V4; i += 256) {
        const float4 v = z4[i];
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    for (int i = 4 * V4 + tid; i < V; i += 256) mx = fmaxf(mx, z[i]);
    mx = block_reduce<true>(mx, red);
    float sum = 0.f;
    for (int i = tid; i < V4; i += 256) {
        const float4 v = z4[i];
        sum += expf(v.x - mx) + expf(v.y - mx) + expf(v.z - mx) + expf(v.w - mx);
    }
    for (int i = 4 * V4 + tid; i < V; i += 256) sum += expf(z[i] - mx);
    sum = block_reduce<false>(sum, red);
    const float inv = 1.f / sum;
    const int t = d.targets ? d.targets[row] : -1;
    float pt = 1.f;
    if (t >= 0 && t < V) pt = expf(z[t] - mx) * inv;
    // dlogits may alias logits (dcap.h): every thread has read z[t] above; no wave may start overwriting the row before
    // the slowest one has (the stores below are in place)
    __syncthreads();
    const bool live = (pt >= 1e-7f) && (pt <= 1.f - 1e-7f);
    const float rw = d.row_weights ? d.row_weights[row] : 1.f;
    if (d.keras_sparse) {
        // q = clip(p); S = sum q; U = sum over unclipped of p_k^2-free term: we need  c = sum_k g_k p_k  with
        // g_k = u_k (1/S - [k==t]/q_t)
        float sq = 0.f, sup = 0.f;
        for (int i = tid; i < V; i += 256) {
            const float p = expf(z[i] - mx) * inv;
            const bool u = (p >= 1e-7f) && (p <= 1.f - 1e-7f);
            sq += fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
            sup += u ? p : 0.f;
        }
        const float S = block_reduce<false>(sq, red);
        const float UP = block_reduce<false>(sup, red);
        const float qt = fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f);
        if (tid == 0 && d.loss_rows) d.loss_rows[row] = rw * (-logf(qt) + logf(S));
        if (!d.probs && !d.dlogits) return;
        const float invS = 1.f / S;
        const float c = UP * invS - (live ? pt / qt : 0.f);
        float* P = d.probs ? d.probs + (long)row * d.ld : nullptr;
        float* G = d.dlogits ? d.dlogits + (long)row * d.ld : nullptr;
        const float gsr = d.grad_scale * rw;
        for (int i = tid; i < V; i += 256) {
            const float p = expf(z[i] - mx) * inv;
            if (P) P[i] = p;
            if (G) {
                const bool u = (p >= 1e-7f) && (p <= 1.f - 1e-7f);
                float g = u ? invS : 0.f;
                if (i == t && live) g -= 1.f / qt;
                G[i] = gsr * p * (g - c);
            }
        }
        return;
    }
    if (tid == 0 && d.loss_rows) d.loss_rows[row] = rw * -logf(fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f));
    if (!d.probs && !d.dlogits) return;
    const float gs = live ? d.grad_scale * rw : 0.f;
    float* P = d.probs ? d.probs + (long)row * d.ld : nullptr;
    float* G = d.dlogits ? d.dlogits + (long)row * d.ld : nullptr;
    for (int i = tid; i < V4; i += 256) {
        const float4 v = z4[i];
        float4 p = make_float4(expf(v.x - mx) * inv, expf(v.y - mx) * inv, expf(v.z - mx) * inv, expf(v.w - mx) * inv);
        if (P) reinterpret_cast<float4*>(P)[i] = p;
        if (G) {
            const int c = 4 * i;
            float4 g = make_float4(gs * (p.x - (c == t ? 1.f : 0.f)), gs * (p.y - (c + 1 == t ? 1.f : 0.f)),
                                   gs * (p.z - (c + 2 == t ? 1.f : 0.f)), gs * (p.w - (c + 3 == t ? 1.f : 0.f)));
            reinterpret_cast<float4*>(G)[i] = g;
        }
    }
    for (int i = 4 * V4 + tid; i < V; i += 256) {
        const float p = expf(z[i] - mx) * inv;
        if (P) P[i] = p;
        if (G) G[i] = gs * (p - (i == t ? 1.f : 0.f));
    }
}

// argmax with the lowest index winning ties (tf.argmax / np.argmax).
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int V, int ld, int32_t* __restrict__ out) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* r = x + (long)row * ld;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
        const float v = r[i];
        if (v > best || (v == best && i < idx)) { best = v; idx = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        out[row] = (idx == 0x7fffffff) ? 0 : idx;      // all-NaN row: index 0 like np.argmax
    }
}

// out[n] (+)= sum_m x[m][n].  A block is CL column lanes x (256/CL) row lanes (CL = power of two <= 64, chosen from N so
// that narrow matrices such as the 20-channel RPN head still use every lane), VEC floats per lane; gridDim.y row chunks.
// One chunk: written straight to out.  Several chunks (tall activations / gradients): every chunk writes its partial row
// into the caller's workspace [chunks][N] and colsum_finish_kernel adds them in chunk order -- the result does not depend
// on scheduling (no atomics), which keeps training runs bit-reproducible.
template <int VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int M, int N, int ld, float* __restrict__ out,
                                                     int accumulate, int cl_log2, int rows_per_chunk, float* __restrict__ partial) {
    __shared__ float red[256 * VEC];
    const int CL = 1 << cl_log2, RL = 256 >> cl_log2;
    const int cl = threadIdx.x & (CL - 1), rl = threadIdx.x >> cl_log2;
    const int c = (blockIdx.x * CL + cl) * VEC;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float s[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[v] = 0.f;
    if (c < N) {
#pragma unroll 4
        for (int m = r0 + rl; m < r1; m += RL) {
            if constexpr (VEC == 4) {
                const float4 q = *reinterpret_cast<const float4*>(x + (long)m * ld + c);
                s[0] += q.x; s[1] += q.y; s[2] += q.z; s[3] += q.w;
            } else {
                s[0] += x[(long)m * ld + c];
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) red[(rl * CL + cl) * VEC + v] = s[v];
    __syncthreads();
    for (int h = RL >> 1; h > 0; h >>= 1) {
        if (rl < h) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) red[(rl * CL + cl) * VEC + v] += red[((rl + h) * CL + cl) * VEC + v];
        }
        __syncthreads();
    }
    if (rl == 0 && c < N) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            if (gridDim.y > 1) partial[(long)blockIdx.y * N + c + v] = red[cl * VEC + v];
            else out[c + v] = accumulate ? out[c + v] + red[cl * VEC + v] : red[cl * VEC + v];
        }
    }
}

// partial [chunks][N] -> out[N]: a block is 16 columns x 16 chunk lanes (lane q adds chunks q, q+16, ... in order), then a fixed
// LDS tree over the 16 lanes: a column's result never depends on scheduling, and the chain of dependent loads is chunks/16 long.
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, int chunks, int N, float* __restrict__ out,
                                                            int accumulate) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < N)
        for (int k = q; k < chunks; k += 16) s += partial[(long)k * N + c];
    red[q][cl] = s;
    __syncthreads();
    for (int h = 8; h > 0; h >>= 1) {
        if (q < h) red[q][cl] += red[q + h][cl];
        __syncthreads();
    }
    if (q == 0 && c < N) out[c] = accumulate ? out[c] + red[0][cl] : red[0][cl];
}

// one wave per row, float4 per lane
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int ld_src, const int32_t* __restrict__ idx,
                                                          float* __restrict__ out, int ld_out, int n_rows, int w4) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const int s = idx[row];
    float4* o = reinterpret_cast<float4*>(out + (long)row * ld_out);
    if (s < 0) {
        for (int c = lane; c < w4; c += 64) o[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const float4* p = reinterpret_cast<const float4*>(src + (long)s * ld_src);
        for (int c = lane; c < w4; c += 64) o[c] = p[c];
    }
}

__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(dc_bn_relu_desc d) {
    const long total = (long)d.M * d.N;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int r = (int)(idx / d.N), c = (int)(idx - (long)r * d.N);
        const float n = (d.acc[(long)r * d.ld + c] + d.bias[c] - d.mean[c]) / sqrtf(d.var[c] + d.eps);
        d.y[(long)r * d.ld + c] = fmaxf(d.gamma[c] * n + d.beta[c], 0.f);
    }
}

// block = 64 columns x 4 row lanes; one pass over the rows: writes dacc and reduces dgamma/dbeta/dbias
__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(dc_bn_relu_desc d) {
    __shared__ float red[3][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    float sg = 0.f, sb = 0.f, sc = 0.f;
    if (c < d.N) {
        const float inv = 1.f / sqrtf(d.var[c] + d.eps), g = d.gamma[c], be = d.beta[c], off = d.bias[c] - d.mean[c];
        for (int r = rl; r < d.M; r += 4) {
            const float n = (d.acc[(long)r * d.ld + c] + off) * inv;
            const float dz = (g * n + be > 0.f) ? d.dy[(long)r * d.ld + c] : 0.f;
            const float da = dz * g * inv;
            d.dacc[(long)r * d.ld + c] = da;
            sg += dz * n; sb += dz; sc += da;
        }
    }
    red[0][rl][threadIdx.x & 63] = sg; red[1][rl][threadIdx.x & 63] = sb; red[2][rl][threadIdx.x & 63] = sc;
    __syncthreads();
    if (rl == 0 && c < d.N) {
        const int l = threadIdx.x;
        d.dgamma[c] = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
        d.dbeta[c] = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
        d.dbias[c] = (red[2][0][l] + red[2][1][l]) + (red[2][2][l] + red[2][3][l]);
    }
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out,
                                                       int M, int N, int ld) {
    const long total = (long)M * N;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int r = (int)(idx / N), c = (int)(idx - (long)r * N);
        const long o = (long)r * ld + c;
        out[o] = y[o] > 0.f ? dy[o] : 0.f;
    }
}

// contiguous rows (ld == N, N % 4 == 0): 16-byte accesses, and optionally the bf16 copy the next bf16 GEMM / convolution reads --
// written here instead of by a cast pass that would read the fp32 result again (round 5)
__global__ __launch_bounds__(256) void relu_bwd_vec_kernel(const float4* __restrict__ dy, const float4* __restrict__ y, float4* __restrict__ out,
                                                           unsigned short* __restrict__ out_bf16, long n4) {
    typedef unsigned short us4 __attribute__((ext_vector_type(4)));
    auto bits = [](float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); };
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 a = dy[i], b = y[i];
        const float4 o = make_float4(b.x > 0.f ? a.x : 0.f, b.y > 0.f ? a.y : 0.f, b.z > 0.f ? a.z : 0.f, b.w > 0.f ? a.w : 0.f);
        if (out) out[i] = o;
        if (out_bf16) reinterpret_cast<us4*>(out_bf16)[i] = us4{bits(o.x), bits(o.y), bits(o.z), bits(o.w)};
    }
}

__global__ __launch_bounds__(256) void fold_time_kernel(const float* __restrict__ x, int T, int B, int N, int ld, float* __restrict__ out,
                                                        int ld_out) {
    const long total = (long)B * N;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int b = (int)(idx / N), c = (int)(idx - (long)b * N);
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += x[((long)t * B + b) * ld + c];
        out[(long)b * ld_out + c] = s;
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float a, const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] += a * x[i];
}

// Global-norm clipping must give every data-parallel rank the SAME scale from the same (all-reduced) gradient, and a
// single-GPU run the same bits twice: block partials go to the caller's workspace in block order and ONE block adds them
// in a fixed tree -- no atomics, no dependence on scheduling.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const size_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = x4[i];
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += x[i] * x[i];
    s = block_reduce<false>(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ out, int accumulate) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += partial[i];
    s = block_reduce<false>(s, red);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s : s;
}

// L2 weight regulariser of the joint model over the flat parameter bucket: coef[i] = WEIGHT_DECAY / size(tensor of i)
// for regularised tensors, 0 elsewhere (BN gamma/beta, padding).  grad = grad * mask + 2*coef*w (mask: the 0/1 trainable subset of
// set_trainable(), NULL = everything trains) ; loss = sum coef*w^2, block partials in block order through the caller's workspace
// and one block adding them in a fixed tree (bit-reproducible, like dc_sumsq).
__global__ __launch_bounds__(256) void l2_reg_kernel(const float* __restrict__ w, const float* __restrict__ coef, const float* __restrict__ mask,
                                                     float* __restrict__ g, size_t n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float c = coef[i], x = w[i];
        if (g) g[i] = (mask ? g[i] * mask[i] : g[i]) + 2.f * c * x;
        s += c * x * x;
    }
    if (partial) {
        s = block_reduce<false>(s, red);
        if (threadIdx.x == 0) partial[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 256) s += x[i];
    s = block_reduce<false>(s, red);
    if (threadIdx.x == 0) out[0] = s / (float)n;
}

// Segment table of a parameter bucket in LDS (dc_reg_segments): start[] ascending, nseg <= kMaxRegSegs.
constexpr int kMaxRegSegs = 1024;
struct RegLds {
    int start[kMaxRegSegs + 1];
    float coef[kMaxRegSegs], mask[kMaxRegSegs];
};
__device__ __forceinline__ void reg_load(const dc_reg_segments& r, RegLds& t) {
    for (int i = threadIdx.x; i <= r.nseg; i += blockDim.x) t.start[i] = r.start[i];
    for (int i = threadIdx.x; i < r.nseg; i += blockDim.x) {
        t.coef[i] = r.coef[i];
        t.mask[i] = r.mask ? r.mask[i] : 1.f;
    }
    __syncthreads();
}
// segment of element e: the last s with start[s] <= e
__device__ __forceinline__ int reg_find(const RegLds& t, int nseg, int e) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.start[mid] <= e) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// (coef, mask) of the four elements 4 i .. 4 i + 3.  A thread's vectors ascend (grid-stride walk: contiguous chunks per block measured
// SLOWER on HBM, 0.65 against 0.52 - 0.62 ms for the 77 M-parameter bucket), and inside the bucket's large segments -- the vocabulary
// layer is two thirds of it -- they stay in the segment of the previous one: the cursor keeps that segment's constants and its end in
// registers (no LDS access on the fast path); a vector that leaves the segment searches again and looks its elements up one by one
// (segments may end inside a vector: the fused RPN head's bias).
struct RegCursor {
    int seg = -1, end = 0;                                 // current segment and the element index where it ends
    float c = 0.f, k = 1.f;
};
__device__ __forceinline__ void reg_vec4(const RegLds& t, int nseg, size_t i, float (&c)[4], float (&k)[4], RegCursor& cur) {
    const int e0 = (int)(i << 2);
    if (cur.seg >= 0 && e0 + 3 < cur.end) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[j] = cur.c; k[j] = cur.k; }
        return;
    }
    int sgm = reg_find(t, nseg, e0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        while (sgm + 1 < nseg && t.start[sgm + 1] <= e0 + j) ++sgm;
        c[j] = t.coef[sgm];
        k[j] = t.mask[sgm];
    }
    cur.seg = sgm;
    cur.end = t.start[sgm + 1];
    cur.c = t.coef[sgm];
    cur.k = t.mask[sgm];
}

// loss partial = sum coef w^2, norm partial = sum (g mask + 2 coef w)^2: read-only, block partials in block order
__global__ __launch_bounds__(256) void reg_sumsq_kernel(const float* __restrict__ w, const float* __restrict__ g, dc_reg_segments r, size_t n,
                                                        float* __restrict__ part_loss, float* __restrict__ part_norm) {
    __shared__ RegLds t;
    __shared__ float red[4];
    reg_load(r, t);
    float sl = 0.f, sn = 0.f;
    const size_t n4 = n >> 2;
    RegCursor cur;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 wv = reinterpret_cast<const float4*>(w)[i], gv = reinterpret_cast<const float4*>(g)[i];
        float c[4], k[4];
        reg_vec4(t, r.nseg, i, c, k, cur);
        const float ww[4] = {wv.x, wv.y, wv.z, wv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = gg[j] * k[j] + 2.f * c[j] * ww[j];
            sl += c[j] * ww[j] * ww[j];
            sn += gr * gr;
        }
    }
    for (size_t e = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int sgm = reg_find(t, r.nseg, (int)e);
        const float gr = g[e] * t.mask[sgm] + 2.f * t.coef[sgm] * w[e];
        sl += t.coef[sgm] * w[e] * w[e];
        sn += gr * gr;
    }
    sl = block_reduce<false>(sl, red);
    __syncthreads();
    sn = block_reduce<false>(sn, red);
    if (threadIdx.x == 0) {
        part_loss[blockIdx.x] = sl;
        part_norm[blockIdx.x] = sn;
    }
}

template <bool REG>
__global__ __launch_bounds__(256) void amsgrad_kernel(dc_amsgrad_desc d, dc_reg_segments r) {
    __shared__ typename std::conditional<REG, RegLds, int>::type t;      // the 12 KB table only in the instantiation that reads it
    if constexpr (REG) reg_load(r, t);
    float gscale = d.grad_scale;
    if (d.gnorm_sq && d.clipnorm > 0.f) {
        const float norm = sqrtf(d.gnorm_sq[0]) * fabsf(d.grad_scale);
        if (norm >= d.clipnorm) gscale *= d.clipnorm / norm;
    }
    const float b1 = d.beta1, b2 = d.beta2;
    if (d.lr_t_dev) d.lr_t = d.lr_t_dev[0];
    const size_t n4 = d.n >> 2;
    RegCursor cur;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 g = reinterpret_cast<const float4*>(d.g)[i];
        float4 m = reinterpret_cast<float4*>(d.m)[i], v = reinterpret_cast<float4*>(d.v)[i];
        float4 vh = reinterpret_cast<float4*>(d.vhat)[i], p = reinterpret_cast<float4*>(d.p)[i];
        if constexpr (REG) {                                // the regularised, masked gradient (what dc_l2_reg_f32 would have written)
            float c[4], k[4];
            reg_vec4(t, r.nseg, i, c, k, cur);
            g.x = g.x * k[0] + 2.f * c[0] * p.x;
            g.y = g.y * k[1] + 2.f * c[1] * p.y;
            g.z = g.z * k[2] + 2.f * c[2] * p.z;
            g.w = g.w * k[3] + 2.f * c[3] * p.w;
        }
#define DC_AMS(c)                                              \
    {                                                          \
        const float gg = g.c * gscale;                         \
        m.c = b1 * m.c + (1.f - b1) * gg;                      \
        v.c = b2 * v.c + (1.f - b2) * gg * gg;                 \
        vh.c = fmaxf(vh.c, v.c);                               \
        p.c -= d.lr_t * m.c / (sqrtf(vh.c) + d.eps);           \
    }
        DC_AMS(x) DC_AMS(y) DC_AMS(z) DC_AMS(w)
        reinterpret_cast<float4*>(d.m)[i] = m;
        reinterpret_cast<float4*>(d.v)[i] = v;
        reinterpret_cast<float4*>(d.vhat)[i] = vh;
        reinterpret_cast<float4*>(d.p)[i] = p;
        if (d.p_bf16 && 4 * i < d.n_bf16) {
            typedef unsigned short us4 __attribute__((ext_vector_type(4)));
            auto bits = [](float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); };
            reinterpret_cast<us4*>(d.p_bf16)[i] = us4{bits(p.x), bits(p.y), bits(p.z), bits(p.w)};
        }
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (size_t)gridDim.x * 256) {
        float g0 = d.g[i];
        if constexpr (REG) {
            const int sgm = reg_find(t, r.nseg, (int)i);
            g0 = g0 * t.mask[sgm] + 2.f * t.coef[sgm] * d.p[i];
        }
        const float gg = g0 * gscale;
        const float m = b1 * d.m[i] + (1.f - b1) * gg;
        const float v = b2 * d.v[i] + (1.f - b2) * gg * gg;
        const float vh = fmaxf(d.vhat[i], v);
        d.m[i] = m; d.v[i] = v; d.vhat[i] = vh;
        d.p[i] -= d.lr_t * m / (sqrtf(vh) + d.eps);
    }
}

}  // namespace dcap

using namespace dcap;

extern "C" int dc_softmax_ce_f32(const dc_softmax_ce_desc* d, void* stream) {
    DC_REQUIRE(d && d->logits && d->M > 0 && d->V > 0 && d->ld >= d->V, DC_EINVAL, "dc_softmax_ce: bad arguments");
    DC_REQUIRE((d->ld & 3) != 0 || (aligned16(d->logits) && (!d->probs || aligned16(d->probs)) && (!d->dlogits || aligned16(d->dlogits))),
               DC_EALIGN, "dc_softmax_ce: rows must be 16-byte aligned when ld %% 4 == 0");
    DC_REQUIRE(d->targets || (!d->loss_rows && !d->dlogits), DC_EINVAL, "dc_softmax_ce: loss/gradient need targets");
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(d->M), dim3(256), 0, static_cast<hipStream_t>(stream), *d);
    return check_launch("softmax_ce_kernel");
}

extern "C" int dc_argmax_rows_f32(const float* x, int M, int V, int ld, int32_t* out, void* stream) {
    DC_REQUIRE(x && out && M > 0 && V > 0 && ld >= V, DC_EINVAL, "dc_argmax_rows: bad arguments");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(M), dim3(256), 0, static_cast<hipStream_t>(stream), x, V, ld, out);
    return check_launch("argmax_rows_kernel");
}

extern "C" int dc_gather_rows_f32(const float* src, int ld_src, const int32_t* idx, float* out, int ld_out, int n_rows, int width,
                                  void* stream) {
    DC_REQUIRE(src && idx && out && n_rows > 0 && width > 0, DC_EINVAL, "dc_gather_rows: bad arguments");
    DC_REQUIRE((width & 3) == 0 && (ld_src & 3) == 0 && (ld_out & 3) == 0 && aligned16(src) && aligned16(out), DC_EALIGN,
               "dc_gather_rows: width/ld must be multiples of 4 and pointers 16-byte aligned");
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), src, ld_src, idx, out,
                       ld_out, n_rows, width / 4);
    return check_launch("gather_rows_kernel");
}

static int bn_relu_check(const dc_bn_relu_desc* d, bool bwd) {
    DC_REQUIRE(d && d->acc && d->bias && d->gamma && d->beta && d->mean && d->var && d->M > 0 && d->N > 0 && d->ld >= d->N, DC_EINVAL,
               "dc_bn_relu: bad arguments");
    DC_REQUIRE(bwd ? (d->dy && d->dacc && d->dgamma && d->dbeta && d->dbias) : (d->y != nullptr), DC_EINVAL,
               "dc_bn_relu: missing output pointers");
    return DC_OK;
}

extern "C" int dc_bn_relu_fwd_f32(const dc_bn_relu_desc* d, void* stream) {
    int rc = bn_relu_check(d, false);
    if (rc) return rc;
    const int blocks = (int)std::min<long>(((long)d->M * d->N + 255) / 256, (long)kNumCU * 8);
    hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *d);
    return check_launch("bn_relu_fwd_kernel");
}

extern "C" int dc_bn_relu_bwd_f32(const dc_bn_relu_desc* d, void* stream) {
    int rc = bn_relu_check(d, true);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3((d->N + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), *d);
    return check_launch("bn_relu_bwd_kernel");
}

extern "C" int dc_relu_bwd_f32(const float* dy, const float* y, float* out, int M, int N, int ld, void* stream) {
    DC_REQUIRE(dy && y && out && M > 0 && N > 0 && ld >= N, DC_EINVAL, "dc_relu_bwd: bad arguments");
    const int blocks = (int)std::min<long>(((long)M * N + 255) / 256, (long)kNumCU * 8);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dy, y, out, M, N, ld);
    return check_launch("relu_bwd_kernel");
}

extern "C" int dc_relu_bwd_dual_f32(const float* dy, const float* y, float* out, uint16_t* out_bf16, size_t n, void* stream) {
    DC_REQUIRE(dy && y && (out || out_bf16) && n > 0 && (n & 3) == 0, DC_EINVAL, "dc_relu_bwd_dual: bad arguments (n %% 4 == 0)");
    DC_REQUIRE(aligned16(dy) && aligned16(y) && (!out || aligned16(out)) && (!out_bf16 || (reinterpret_cast<uintptr_t>(out_bf16) & 7u) == 0), DC_EALIGN,
               "dc_relu_bwd_dual: dy, y, out must be 16-byte aligned, out_bf16 8-byte aligned");
    const long n4 = (long)(n >> 2);
    const int blocks = (int)std::min<long>((n4 + 255) / 256, (long)kNumCU * 8);
    hipLaunchKernelGGL(relu_bwd_vec_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const float4*>(dy),
                       reinterpret_cast<const float4*>(y), reinterpret_cast<float4*>(out), out_bf16, n4);
    return check_launch("relu_bwd_vec_kernel");
}

extern "C" int dc_fold_time_f32(const float* x, int T, int B, int N, int ld, float* out, int ld_out, void* stream) {
    DC_REQUIRE(x && out && T > 0 && B > 0 && N > 0 && ld >= N && ld_out >= N, DC_EINVAL, "dc_fold_time: bad arguments");
    const int blocks = (int)std::min<long>(((long)B * N + 255) / 256, (long)kNumCU * 8);
    hipLaunchKernelGGL(fold_time_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, T, B, N, ld, out, ld_out);
    return check_launch("fold_time_kernel");
}

struct ColsumPlan { bool vec; int cl_log2, col_tiles, chunks, rows_per_chunk; };
static ColsumPlan colsum_plan(const float* x, int M, int N, int ld) {
    ColsumPlan p;
    p.vec = (N & 3) == 0 && (ld & 3) == 0 && aligned16(x);
    const int lanes = p.vec ? N / 4 : N;
    p.cl_log2 = 0;
    while ((1 << p.cl_log2) < lanes && p.cl_log2 < 6) ++p.cl_log2;
    const int CL = 1 << p.cl_log2, RL = 256 >> p.cl_log2;
    p.col_tiles = (lanes + CL - 1) / CL;
    // enough blocks to cover the chip a few times over, each summing at least 4 rows per row lane
    p.chunks = std::max(1, std::min(std::min((kNumCU * 4) / p.col_tiles, M / (RL * 4)), 128));   // <= 128 partial rows to combine
    p.rows_per_chunk = (M + p.chunks - 1) / p.chunks;
    p.chunks = (M + p.rows_per_chunk - 1) / p.rows_per_chunk;
    return p;
}

extern "C" size_t dc_colsum_workspace_bytes(int M, int N, int ld) {
    if (M <= 0 || N <= 0) return 0;
    const ColsumPlan p = colsum_plan(nullptr, M, N, ld);
    return p.chunks > 1 ? (size_t)p.chunks * N * sizeof(float) : 0;
}

extern "C" int dc_colsum_f32(const float* x, int M, int N, int ld, float* out, int accumulate, void* workspace, size_t workspace_bytes,
                             void* stream) {
    DC_REQUIRE(x && out && M > 0 && N > 0 && ld >= N, DC_EINVAL, "dc_colsum: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ColsumPlan p = colsum_plan(x, M, N, ld);
    if (p.chunks > 1 && (!workspace || workspace_bytes < (size_t)p.chunks * N * sizeof(float))) {
        p.chunks = 1;                                     // no scratch: one block column walks all rows (slow, still exact)
        p.rows_per_chunk = M;
    }
    float* partial = static_cast<float*>(workspace);
    if (p.vec)
        hipLaunchKernelGGL(colsum_kernel<4>, dim3(p.col_tiles, p.chunks), dim3(256), 0, s, x, M, N, ld, out, accumulate, p.cl_log2,
                           p.rows_per_chunk, partial);
    else
        hipLaunchKernelGGL(colsum_kernel<1>, dim3(p.col_tiles, p.chunks), dim3(256), 0, s, x, M, N, ld, out, accumulate, p.cl_log2,
                           p.rows_per_chunk, partial);
    int rc = check_launch("colsum_kernel");
    if (rc || p.chunks == 1) return rc;
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 15) / 16), dim3(256), 0, s, partial, p.chunks, N, out, accumulate);
    return check_launch("colsum_finish_kernel");
}

// blocks per CU of the bucket-wide HBM-bound passes: every wave slot (round 4 measured 8 / 4 / 2 / 1 = 8.61 / 8.69 / 8.88 / 9.25 ms per joint step)
static constexpr int opt_blocks_per_cu() { return 8; }
static int sumsq_blocks(size_t n) { return (int)std::min<size_t>((n + 4095) / 4096, (size_t)kNumCU * std::min(4, opt_blocks_per_cu())); }

extern "C" size_t dc_sumsq_workspace_bytes(size_t n) { return n ? (size_t)sumsq_blocks(n) * sizeof(float) : 0; }

extern "C" int dc_sumsq_f32(const float* x, size_t n, float* out, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    DC_REQUIRE(x && out && n > 0, DC_EINVAL, "dc_sumsq: bad arguments");
    DC_REQUIRE(aligned16(x), DC_EALIGN, "dc_sumsq: x must be 16-byte aligned");
    const int blocks = sumsq_blocks(n);
    DC_REQUIRE(workspace && workspace_bytes >= (size_t)blocks * sizeof(float), DC_EWORKSPACE, "dc_sumsq: needs %zu workspace bytes",
               (size_t)blocks * sizeof(float));
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, s, x, n, partial);
    int rc = check_launch("sumsq_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, s, partial, blocks, out, accumulate);
    return check_launch("sumsq_finish_kernel");
}

static int l2_reg_blocks(size_t n) { return (int)std::min<size_t>((n + 1023) / 1024, (size_t)kNumCU * std::min(4, opt_blocks_per_cu())); }

extern "C" size_t dc_l2_reg_workspace_bytes(size_t n) { return n ? (size_t)l2_reg_blocks(n) * sizeof(float) : 0; }

extern "C" int dc_l2_reg_f32(const float* w, const float* coef, const float* mask, float* grad, size_t n, float* loss, void* workspace,
                             size_t workspace_bytes, void* stream) {
    DC_REQUIRE(w && coef && n > 0 && (grad || loss), DC_EINVAL, "dc_l2_reg: bad arguments");
    DC_REQUIRE(!mask || grad, DC_EINVAL, "dc_l2_reg: a mask without a gradient");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int blocks = l2_reg_blocks(n);
    DC_REQUIRE(!loss || (workspace && workspace_bytes >= (size_t)blocks * sizeof(float)), DC_EWORKSPACE, "dc_l2_reg: the loss needs %zu workspace bytes",
               (size_t)blocks * sizeof(float));
    float* partial = loss ? static_cast<float*>(workspace) : nullptr;
    hipLaunchKernelGGL(l2_reg_kernel, dim3(blocks), dim3(256), 0, s, w, coef, mask, grad, n, partial);
    int rc = check_launch("l2_reg_kernel");
    if (rc || !loss) return rc;
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, s, partial, blocks, loss, 0);
    return check_launch("l2_reg_kernel (finish)");
}

static int reg_check(const dc_reg_segments* r, size_t n, const char* who) {
    DC_REQUIRE(r->start && r->coef && r->nseg >= 1 && r->nseg <= kMaxRegSegs, DC_EINVAL, "%s: the segment table needs 1..%d segments, got %d", who, kMaxRegSegs,
               r->nseg);
    DC_REQUIRE(n < ((size_t)1 << 31), DC_EINVAL, "%s: segment offsets are 32-bit: the bucket must hold < 2^31 elements", who);
    return DC_OK;
}
static int reg_sumsq_blocks(size_t n) { return (int)std::min<size_t>((n / 4 + 255) / 256 + 1, (size_t)kNumCU * opt_blocks_per_cu()); }

extern "C" size_t dc_reg_sumsq_workspace_bytes(size_t n) { return n ? (size_t)2 * reg_sumsq_blocks(n) * sizeof(float) : 0; }

extern "C" int dc_reg_sumsq_f32(const float* w, const float* g, const dc_reg_segments* reg, size_t n, float* loss, float* gnorm_sq, void* workspace,
                                size_t workspace_bytes, void* stream) {
    DC_REQUIRE(w && g && reg && n > 0 && (loss || gnorm_sq), DC_EINVAL, "dc_reg_sumsq: bad arguments");
    DC_REQUIRE(aligned16(w) && aligned16(g), DC_EALIGN, "dc_reg_sumsq: w and g must be 16-byte aligned");
    int rc = reg_check(reg, n, "dc_reg_sumsq");
    if (rc) return rc;
    const int blocks = reg_sumsq_blocks(n);
    DC_REQUIRE(workspace && workspace_bytes >= (size_t)2 * blocks * sizeof(float), DC_EWORKSPACE, "dc_reg_sumsq: needs %zu workspace bytes",
               (size_t)2 * blocks * sizeof(float));
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(reg_sumsq_kernel, dim3(blocks), dim3(256), 0, s, w, g, *reg, n, part, part + blocks);
    rc = check_launch("reg_sumsq_kernel");
    if (rc) return rc;
    if (loss) {
        hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, s, part, blocks, loss, 0);
        rc = check_launch("reg_sumsq_kernel (loss)");
        if (rc) return rc;
    }
    if (gnorm_sq) {
        hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, s, part + blocks, blocks, gnorm_sq, 0);
        rc = check_launch("reg_sumsq_kernel (norm)");
    }
    return rc;
}

extern "C" int dc_zero_fill(void* p, size_t bytes, void* stream) {
    DC_REQUIRE(p && bytes > 0, DC_EINVAL, "dc_zero_fill: bad arguments");
    return zero_fill_async(p, bytes, static_cast<hipStream_t>(stream));
}

extern "C" int dc_axpy_f32(float a, const float* x, float* y, size_t n, void* stream) {
    DC_REQUIRE(x && y && n > 0, DC_EINVAL, "dc_axpy: bad arguments");
    const int blocks = (int)std::min<size_t>((n + 255) / 256, (size_t)kNumCU * 8);
    hipLaunchKernelGGL(axpy_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a, x, y, n);
    return check_launch("axpy_kernel");
}

// Keras K.dropout(ones, rate) (tf.nn.dropout: keep with probability 1 - rate, kept entries scaled by 1 / (1 - rate)) from a
// counter-based generator: element i of stream (seed, offset) depends on nothing else, so a mask is reproducible and
// independent of the launch geometry.  Philox-2x32-10 keyed by the seed, counter = (i, offset).
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ out, size_t n, float rate, float keep_scale, unsigned seed,
                                                           unsigned offset, const unsigned* __restrict__ offset_dev) {
    if (offset_dev) offset += offset_dev[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned r = philox2x32((unsigned)i, offset ^ (unsigned)(i >> 32), seed);
        const float u = (float)(r >> 8) * (1.0f / 16777216.0f);                 // uniform [0, 1) on 24 bits
        out[i] = u >= rate ? keep_scale : 0.f;
    }
}

extern "C" int dc_dropout_mask_f32(float* out, size_t n, float rate, uint32_t seed, uint32_t offset, const uint32_t* offset_dev, void* stream) {
    DC_REQUIRE(out && n > 0 && rate >= 0.f && rate < 1.f, DC_EINVAL, "dc_dropout_mask: needs out, n > 0 and 0 <= rate < 1");
    const int blocks = (int)std::min<size_t>((n + 255) / 256, (size_t)kNumCU * 8);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), out, n, rate, 1.0f / (1.0f - rate), seed, offset, offset_dev);
    return check_launch("dropout_mask_kernel");
}

// ---- training ResNet stages (dense_img_cap/dense_model.py:1829-1845, layers = "3+" | "4+" | "5+" | "all"): BatchNorm layers
// run with training=False (frozen statistics, :51-61) while gamma / beta -- and the convolution in front -- are trained.
// The forward keeps the fused form  y = act(scale * conv(x) + shift [+ residual])  with
//   scale = gamma / sqrt(var + eps),  shift = beta + (bias - mean) * scale
// refreshed from the trained parameters by bn_fold before every pass; the backward recovers the normalised activation from the
// stored BN output  n = (bn_out - beta) / gamma  (bn_out = y where the ReLU passed, = out - shortcut for the block's last
// convolution; only needed where the upstream gradient is non-zero, which is exactly where it is recoverable).
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ bias,
                                                      const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                      float* __restrict__ scale, float* __restrict__ shift, int n) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const float sc = gamma[c] / sqrtf(var[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] + (bias[c] - mean[c]) * sc;
}

// dacc = dz * scale (gradient w.r.t. the convolution's output);  dzn = dz * n,  n = (a - b - beta) / gamma  (b may be null)
__global__ __launch_bounds__(256) void bn_bwd_kernel(const float4* __restrict__ dz, const float4* __restrict__ a, const float4* __restrict__ b,
                                                     const float4* __restrict__ gamma, const float4* __restrict__ beta, const float4* __restrict__ scale,
                                                     float4* __restrict__ dacc, float4* __restrict__ dzn, long rows, int c4) {
    const long total = rows * c4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % c4);
        const float4 g = dz[i], va = a[i], ga = gamma[c], be = beta[c], sc = scale[c];
        float4 vb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b) vb = b[i];
        dacc[i] = make_float4(g.x * sc.x, g.y * sc.y, g.z * sc.z, g.w * sc.w);
        // where dz == 0 the recovered n is meaningless: the product is defined as 0 there.  A DEAD channel (gamma == 0, as pretrained
        // ResNet BatchNorm layers contain) has bn_out == beta everywhere, n is not recoverable from it and 0 / 0 would put a NaN into
        // dgamma and from there into the AMSGrad state of the whole bucket: such a channel gets dgamma = 0 (it stays dead; dbeta and
        // the data gradient, which is scale * dz = 0, are exact)
        auto prod = [](float dzv, float av, float bv, float bev, float gav) {
            return (dzv != 0.f && fabsf(gav) > 1e-20f) ? dzv * ((av - bv - bev) / gav) : 0.f;
        };
        dzn[i] = make_float4(prod(g.x, va.x, vb.x, be.x, ga.x), prod(g.y, va.y, vb.y, be.y, ga.y), prod(g.z, va.z, vb.z, be.z, ga.z),
                             prod(g.w, va.w, vb.w, be.w, ga.w));
    }
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = a[i] * b[i];
}

extern "C" int dc_bn_fold_f32(const float* gamma, const float* beta, const float* bias, const float* mean, const float* var, float eps, float* scale,
                              float* shift, int n, void* stream) {
    DC_REQUIRE(gamma && beta && bias && mean && var && scale && shift && n > 0, DC_EINVAL, "dc_bn_fold: bad arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), gamma, beta, bias, mean, var, eps, scale, shift, n);
    return check_launch("bn_fold_kernel");
}

extern "C" int dc_bn_bwd_f32(const float* dz, const float* a, const float* b, const float* gamma, const float* beta, const float* scale, float* dacc,
                             float* dzn, long rows, int channels, void* stream) {
    DC_REQUIRE(dz && a && gamma && beta && scale && dacc && dzn && rows > 0 && channels > 0 && (channels & 3) == 0, DC_EINVAL,
               "dc_bn_bwd: bad arguments (channels %% 4 == 0)");
    DC_REQUIRE(aligned16(dz) && aligned16(a) && (!b || aligned16(b)) && aligned16(gamma) && aligned16(beta) && aligned16(scale) && aligned16(dacc) && aligned16(dzn),
               DC_EALIGN, "dc_bn_bwd: pointers must be 16-byte aligned");
    const long total = rows * (channels / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, (long)kNumCU * 8);
    hipLaunchKernelGGL(bn_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const float4*>(dz),
                       reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), reinterpret_cast<const float4*>(gamma),
                       reinterpret_cast<const float4*>(beta), reinterpret_cast<const float4*>(scale), reinterpret_cast<float4*>(dacc),
                       reinterpret_cast<float4*>(dzn), rows, channels / 4);
    return check_launch("bn_bwd_kernel");
}

extern "C" int dc_mul_f32(const float* a, const float* b, float* out, size_t n, void* stream) {
    DC_REQUIRE(a && b && out && n > 0, DC_EINVAL, "dc_mul: bad arguments");
    const int blocks = (int)std::min<size_t>((n + 255) / 256, (size_t)kNumCU * 8);
    hipLaunchKernelGGL(mul_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, out, n);
    return check_launch("mul_kernel");
}

extern "C" int dc_mean_f32(const float* x, size_t n, float* out, void* stream) {
    DC_REQUIRE(x && out && n > 0, DC_EINVAL, "dc_mean: bad arguments");
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), x, n, out);
    return check_launch("mean_kernel");
}

extern "C" int dc_amsgrad_step_f32(const dc_amsgrad_desc* d, void* stream) {
    DC_REQUIRE(d && d->p && d->g && d->m && d->v && d->vhat && d->n > 0, DC_EINVAL, "dc_amsgrad_step: bad arguments");
    DC_REQUIRE(aligned16(d->p) && aligned16(d->g) && aligned16(d->m) && aligned16(d->v) && aligned16(d->vhat), DC_EALIGN,
               "dc_amsgrad_step: buffers must be 16-byte aligned");
    DC_REQUIRE(!d->p_bf16 || ((d->n_bf16 & 3) == 0 && d->n_bf16 <= (d->n & ~(size_t)3) && (reinterpret_cast<uintptr_t>(d->p_bf16) & 7u) == 0), DC_EINVAL,
               "dc_amsgrad_step: the bf16 shadow must be 8-byte aligned and cover a multiple of 4 elements inside the vectorised part of p");
    const int blocks = (int)std::min<size_t>((d->n / 4 + 255) / 256 + 1, (size_t)kNumCU * opt_blocks_per_cu());
    if (d->reg) {
        int rc = reg_check(d->reg, d->n, "dc_amsgrad_step");
        if (rc) return rc;
        hipLaunchKernelGGL(amsgrad_kernel<true>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *d, *d->reg);
    } else {
        hipLaunchKernelGGL(amsgrad_kernel<false>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *d, dc_reg_segments{});
    }
    return check_launch("amsgrad_kernel");
}
