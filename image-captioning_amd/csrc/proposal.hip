// proposal.hip -- RPN scoring + ProposalLayer on the device: fg-score softmax, tf.nn.top_k as a three-pass radix select of
// the k-th score + an index-ordered compaction of exactly k candidates + a rank sort of those k from LDS (stable: score
// descending, lower anchor index first among ties), box decode / clip / normalise in TF's float32 operation order, and
// tf.image.non_max_suppression as a 64-bit suppression-mask kernel followed by a single-workgroup greedy
// scan.  All index decisions (sort order, ties, IoU > threshold) are float32 like the TF graph.
#include "dcap_internal.h"
#include <string.h>
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

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// scores[b][a] = softmax(class logits)[fg]; deltas[b][a][4] = bbox * std
__global__ __launch_bounds__(256) void rpn_score_kernel(dc_proposal_desc d, int level, int level_off, float* __restrict__ scores,
                                                        float* __restrict__ deltas, int* __restrict__ iota) {
    const int H = d.Hs[level], W = d.Ws[level], A = d.anchors_per_loc;
    const int per_img = H * W * A;
    const long total = (long)d.B * per_img;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int b = (int)(idx / per_img), r = (int)(idx - (long)b * per_img);
        const int cell = r / A, a = r - cell * A;
        const int hstride = d.head_stride > 0 ? d.head_stride : A * 6;
        const float* h = d.heads[level] + ((long)b * H * W + cell) * hstride;
        const float l0 = h[a * 2], l1 = h[a * 2 + 1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const long o = (long)b * d.A_total + level_off + r;
        scores[o] = e1 / (e0 + e1);
        const float* bb = h + A * 2 + a * 4;
        float4 dl = make_float4(mul_rn(bb[0], d.std_dev[0]), mul_rn(bb[1], d.std_dev[1]), mul_rn(bb[2], d.std_dev[2]),
                                mul_rn(bb[3], d.std_dev[3]));
        reinterpret_cast<float4*>(deltas)[o] = dl;
        if (b == 0) iota[level_off + r] = level_off + r;
    }
}

// candidate i of image b: anchor order[i]; apply_box_deltas_graph + clip_boxes_graph + normalise (float32,
// one rounding per TF op: no fma contraction)
__global__ __launch_bounds__(256) void decode_kernel(dc_proposal_desc d, const int* __restrict__ order, const float* __restrict__ deltas,
                                                     float4* __restrict__ boxes, int k) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= d.B * k) return;
    const int b = idx / k, i = idx - b * k;
    const int a = order[(long)b * d.A_total + i];
    const float4 an = reinterpret_cast<const float4*>(d.anchors)[a];
    const float4 dl = reinterpret_cast<const float4*>(deltas)[(long)b * d.A_total + a];
    float h = sub_rn(an.z, an.x), w = sub_rn(an.w, an.y);
    float cy = add_rn(an.x, mul_rn(0.5f, h)), cx = add_rn(an.y, mul_rn(0.5f, w));
    cy = add_rn(cy, mul_rn(dl.x, h));
    cx = add_rn(cx, mul_rn(dl.y, w));
    h = mul_rn(h, expf(dl.z));
    w = mul_rn(w, expf(dl.w));
    float y1 = sub_rn(cy, mul_rn(0.5f, h)), x1 = sub_rn(cx, mul_rn(0.5f, w));
    float y2 = add_rn(y1, h), x2 = add_rn(x1, w);
    y1 = fmaxf(fminf(y1, d.image_h), 0.f); x1 = fmaxf(fminf(x1, d.image_w), 0.f);
    y2 = fmaxf(fminf(y2, d.image_h), 0.f); x2 = fmaxf(fminf(x2, d.image_w), 0.f);
    boxes[idx] = make_float4(div_rn(y1, d.image_h), div_rn(x1, d.image_w), div_rn(y2, d.image_h), div_rn(x2, d.image_w));
}

__device__ __forceinline__ bool iou_exceeds(const float4 a, const float4 b, float thr) {
    const float ay0 = fminf(a.x, a.z), ay1 = fmaxf(a.x, a.z), ax0 = fminf(a.y, a.w), ax1 = fmaxf(a.y, a.w);
    const float by0 = fminf(b.x, b.z), by1 = fmaxf(b.x, b.z), bx0 = fminf(b.y, b.w), bx1 = fmaxf(b.y, b.w);
    const float area_a = mul_rn(sub_rn(ay1, ay0), sub_rn(ax1, ax0));
    const float area_b = mul_rn(sub_rn(by1, by0), sub_rn(bx1, bx0));
    if (area_a <= 0.f || area_b <= 0.f) return false;
    const float ih = fmaxf(sub_rn(fminf(ay1, by1), fmaxf(ay0, by0)), 0.f);
    const float iw = fmaxf(sub_rn(fminf(ax1, bx1), fmaxf(ax0, bx0)), 0.f);
    const float inter = mul_rn(ih, iw);
    return div_rn(inter, sub_rn(add_rn(area_a, area_b), inter)) > thr;
}

// mask[b][i][w] bit j: candidate 64*w + j (ranked after i) overlaps candidate i by more than the threshold
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes, unsigned long long* __restrict__ mask, int k, int words,
                                                      float thr) {
    __shared__ float4 cb[64];
    const int b = blockIdx.z, rowb = blockIdx.y, colb = blockIdx.x;
    if (colb < rowb) return;
    const float4* bx = boxes + (long)b * k;
    const int cj = colb * 64 + threadIdx.x;
    cb[threadIdx.x] = cj < k ? bx[cj] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int i = rowb * 64 + threadIdx.x;
    if (i >= k) return;
    const float4 me = bx[i];
    unsigned long long bits = 0;
    const int ncol = min(64, k - colb * 64);
    for (int j = (rowb == colb) ? threadIdx.x + 1 : 0; j < ncol; ++j)
        if (iou_exceeds(me, cb[j], thr)) bits |= 1ull << j;
    mask[((long)b * k + i) * words + colb] = bits;
}

// one 64-thread workgroup per image walks the candidates in rank order
__global__ __launch_bounds__(64) void nms_scan_kernel(const float4* __restrict__ boxes, const unsigned long long* __restrict__ mask, int k,
                                                      int words, int count, float4* __restrict__ proposals, int* __restrict__ keep_out) {
    extern __shared__ unsigned long long removed[];
    const int b = blockIdx.x, lane = threadIdx.x;
    for (int w = lane; w < words; w += 64) removed[w] = 0;
    __syncthreads();
    int kept = 0;
    for (int i = 0; i < k && kept < count; ++i) {
        const bool dead = (removed[i >> 6] >> (i & 63)) & 1ull;      // uniform
        if (!dead) {
            if (lane == 0) {
                proposals[(long)b * count + kept] = boxes[(long)b * k + i];
                if (keep_out) keep_out[(long)b * count + kept] = i;
            }
            ++kept;
            const unsigned long long* row = mask + ((long)b * k + i) * words;
            for (int w = (i >> 6) + lane; w < words; w += 64) removed[w] |= row[w];
        }
        __syncthreads();
    }
    for (int r = kept + lane; r < count; r += 64) {
        proposals[(long)b * count + r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (keep_out) keep_out[(long)b * count + r] = -1;
    }
}

// Greedy scan, one wave per image, 64 candidates per round (k <= 8192).  The removed-set lives in registers (lane l owns
// words l and l+64).  A round loads each candidate's suppression bits WITHIN the round (one word per lane), resolves the
// round serially on scalar values (ctz over the alive bits + readlane: no memory in the dependent chain), then ORs the
// kept candidates' full rows into the removed-set eight rows at a time (independent loads in flight).  The first version
// walked the candidates one by one with a barrier and a dependent global row load per kept box: 1.4 ms for 6000 -> 2000.
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

constexpr int NMS_ROWS = 24;  // kept rows whose suppression masks one wave fetches together
constexpr int NMS_G = 8;       // chunks (of 64 candidates) resolved between two memory round trips
constexpr int NMS_WAVES = 4;

// OR over the 64 lanes of a wave, result uniform.  DPP row operations (quad swaps, row rotates, the two row broadcasts) instead of
// ds_bpermute butterflies: ~12 short VALU instructions per 32-bit half, no LDS-latency round trips in the scan's dependent chain.
__device__ __forceinline__ unsigned wave_or32(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);       // quad_perm [1,0,3,2]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);       // quad_perm [2,3,0,1]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);      // row_ror:4
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);      // row_ror:8  -> every lane: its row's OR
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1, 3
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
    return ((unsigned long long)wave_or32((unsigned)(v >> 32)) << 32) | wave_or32((unsigned)v);
}

// Round 5.  The scan used to pay ONE dependent global round trip per 64-candidate chunk (the kept candidates' suppression rows had to
// be OR-ed into the removed-set before the next chunk could start): 94 chunks x ~3 us = 0.28 ms for 6000 -> 2000 on ONE wave, on the
// joint step's critical chain.  Now:
//   * a GROUP of NMS_G chunks is resolved by wave 0 without touching memory: besides its own word (the bits inside its chunk),
//     candidate 64 c + lane brings the words c + 1 .. group end of its row -- a state-independent triangle of G (G + 1) / 2 words per
//     lane, prefetched one group ahead -- and after a chunk is resolved the kept candidates' triangle words are OR-reduced over the
//     wave into the group's pending removed-words (seven butterfly reductions per chunk at most, no memory in the chain);
//   * the full rows of the group's kept candidates are OR-ed into the removed-set ONCE PER GROUP by all four waves (chunk i of the
//     group by wave i % 4, NMS_ROWS rows in flight per wave); every wave keeps its own partial removed-set in registers for the whole
//     scan, and only the G words the next group starts from are combined through LDS.
// 12 round trips instead of 94, four waves' worth of loads in flight.  Same greedy order, same result bit for bit.
__global__ __launch_bounds__(NMS_WAVES * 64) void nms_scan_wave_kernel(const float4* __restrict__ boxes, const unsigned long long* __restrict__ mask,
                                                                    int k, int words, int count, float4* __restrict__ proposals,
                                                                    int* __restrict__ keep_out) {
    constexpr int G = NMS_G;
    __shared__ unsigned long long s_keep[G];
    __shared__ unsigned long long s_rem[NMS_WAVES][G];
    __shared__ int s_done;
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long* M = mask + (long)b * k * words;
    unsigned long long rem0 = 0, rem1 = 0;                 // THIS wave's rows OR-ed so far: lane l owns words l and l + 64
    int kept = 0;                                          // (wave 0)
    // tri[i][d]: word (c + d) of row (64 c + lane), c = g0 + i, d = 0 .. G - 1 - i  (d = 0: the bits inside the chunk); wave 0 only
    unsigned long long tri[G][G], nxt[G][G];
    auto load_tri = [&](unsigned long long (&t)[G][G], int g0) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int c = g0 + i, row = c * 64 + lane;
            const bool in = c < words && row < k;
            const unsigned long long* r = M + (long)(in ? row : 0) * words;
#pragma unroll
            for (int d = 0; d < G - i; ++d) t[i][d] = (in && c + d < words) ? r[c + d] : 0ull;
        }
    };
    if (wave == 0) load_tri(tri, 0);
    unsigned long long grem[G];                            // (wave 0) removed-words of the group's chunks: complete when the chunk is reached
#pragma unroll
    for (int i = 0; i < G; ++i) grem[i] = 0ull;
    for (int g0 = 0; g0 < words; g0 += G) {
        if (wave == 0) {
            load_tri(nxt, g0 + G);                         // state-independent: in flight while this group resolves
            unsigned long long keepm[G];
#pragma unroll
            for (int i = 0; i < G; ++i) keepm[i] = 0ull;
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int c = g0 + i;
                if (c < words && kept < count) {           // uniform
                    const int left = k - c * 64;
                    unsigned long long alive = ~grem[i];
                    if (left < 64) alive &= (1ull << left) - 1ull;
                    unsigned long long keepmask = 0;
                    int room = count - kept;
                    while (alive != 0 && room > 0) {       // uniform: one iteration per kept candidate, no memory in the chain
                        const int j = __builtin_ctzll(alive);
                        keepmask |= 1ull << j;
                        --room;
                        alive &= ~(1ull << j);
                        alive &= ~readlane64(tri[i][0], j);
                    }
                    const bool mine = (keepmask >> lane) & 1ull;
                    if (mine) {
                        const int slot = kept + __builtin_popcountll(keepmask & ((1ull << lane) - 1ull));
                        proposals[(long)b * count + slot] = boxes[(long)b * k + c * 64 + lane];
                        if (keep_out) keep_out[(long)b * count + slot] = c * 64 + lane;
                    }
                    kept += __builtin_popcountll(keepmask);
                    keepm[i] = keepmask;
#pragma unroll
                    for (int d = 1; d < G - i; ++d) grem[i + d] |= wave_or64(mine ? tri[i][d] : 0ull);
                }
            }
            if (lane < G) {
                unsigned long long v = 0;
#pragma unroll
                for (int i = 0; i < G; ++i) v = lane == i ? keepm[i] : v;
                s_keep[lane] = v;
            }
            if (lane == 0) s_done = (kept >= count || g0 + G >= words) ? 1 : 0;
        }
        __syncthreads();                                   // the group's keep masks and the stop flag are published
        if (s_done) break;                                 // (block-uniform)
        // the group's kept rows into the removed-sets: chunk i by wave i % NMS_WAVES
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if ((i % NMS_WAVES) != wave) continue;
            unsigned long long km = s_keep[i];
            while (km != 0) {                              // wave-uniform
                unsigned long long v0[NMS_ROWS], v1[NMS_ROWS];
#pragma unroll
                for (int u = 0; u < NMS_ROWS; ++u) {
                    v0[u] = 0; v1[u] = 0;
                    if (km != 0) {
                        const int j = __builtin_ctzll(km);
                        km &= km - 1ull;
                        const unsigned long long* row = M + (long)((g0 + i) * 64 + j) * words;
                        if (lane < words) v0[u] = row[lane];
                        if (lane + 64 < words) v1[u] = row[lane + 64];
                    }
                }
#pragma unroll
                for (int u = 0; u < NMS_ROWS; ++u) { rem0 |= v0[u]; rem1 |= v1[u]; }
            }
        }
        // the words the next group starts from: word g0 + G + i is owned by lane (g0 + G + i) & 63 of every wave
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int w = g0 + G + i;
            if (lane == (w & 63)) s_rem[wave][i] = w < 64 ? rem0 : rem1;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                unsigned long long v = 0;
#pragma unroll
                for (int q = 0; q < NMS_WAVES; ++q) v |= s_rem[q][i];
                grem[i] = v;
            }
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int d = 0; d < G - i; ++d) tri[i][d] = nxt[i][d];
        }
        __syncthreads();                                   // s_keep / s_rem / s_done may be rewritten
    }
    if (wave == 0) {
        kept = __builtin_amdgcn_readfirstlane(kept);
        for (int r = kept + lane; r < count; r += 64) {
            proposals[(long)b * count + r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (keep_out) keep_out[(long)b * count + r] = -1;
        }
    }
}

__global__ void subsample2_kernel(const float4* __restrict__ x, float4* __restrict__ y, int N, int H, int W, int C4) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long total = (long)N * Ho * Wo * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        long p = idx / C4;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho), n = (int)(p / Ho);
        y[idx] = x[(((long)n * H + oy * 2) * W + ox * 2) * C4 + c];
    }
}

// RPN losses and their gradients w.r.t. the padded head outputs, one thread per selected (non-neutral) anchor.
// class: sparse softmax CE over {bg, fg}, mean over the n_sel anchors; bbox: smooth-L1 against the target rows
// of the positive anchors (in anchor order), mean over 4*n_pos elements.  ONE block walks the selection (256 anchors per image in
// the reference's configuration: RPN_TRAIN_ANCHORS_PER_IMAGE); the two loss sums are per-thread partials in anchor order combined by
// a fixed tree in LDS -- no atomics: the reported losses are bit-reproducible like the gradients.
__global__ __launch_bounds__(256) void rpn_loss_grad_kernel(dc_rpn_loss_desc d) {
    __shared__ float red[2][256];
    const int A = d.anchors_per_loc;
    if (d.counts_dev) {                                        // this image's counts from device memory (the fields hold the capacities)
        d.n_sel = min(d.n_sel, d.counts_dev[0]);
        d.n_pos = min(d.n_pos, d.counts_dev[1]);
    }
    const float invn = 1.f / (float)max(d.n_sel, 1);
    float lc = 0.f, lb = 0.f;
    for (int i = threadIdx.x; i < d.n_sel; i += 256) {
        const int l = d.sel_level[i], idx = d.sel_index[i], m = d.sel_match[i];
        const int cell = idx / A, a = idx - cell * A;
        if (l < 0 || l >= d.levels || idx < 0 || cell >= d.Hs[l] * d.Ws[l]) continue;      // a malformed selection must not write out of bounds
        const float* h = d.heads[l] + (long)cell * d.head_stride;
        float* g = d.dheads[l] + (long)cell * d.head_stride;
        const float l0 = h[a * 2], l1 = h[a * 2 + 1];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx), inv = 1.f / (e0 + e1);
        const float p0 = e0 * inv, p1 = e1 * inv;
        const int cls = (m == 1) ? 1 : 0;
        g[a * 2] = (p0 - (cls == 0 ? 1.f : 0.f)) * invn;
        g[a * 2 + 1] = (p1 - (cls == 1 ? 1.f : 0.f)) * invn;
        lc += (logf(e0 + e1) - ((cls ? l1 : l0) - mx)) * invn;      // log-sum-exp form: finite for any logits
        if (m == 1) {
            // rank of this positive among the positives = number of positives before it in sel (sel is in anchor order)
            int rank = 0;
            for (int j = 0; j < i; ++j) rank += (d.sel_match[j] == 1);
            const float* t = d.target_deltas + (long)rank * 4;
            const float* bb = h + A * 2 + a * 4;
            float* gb = g + A * 2 + a * 4;
            if (rank >= d.n_pos) continue;                         // (more positives in sel than target rows: nothing to regress to)
            const float invp = 1.f / (4.f * (float)d.n_pos);
            float ls = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float diff = t[k] - bb[k], ad = fabsf(diff);
                if (ad < 1.f) { ls += 0.5f * ad * ad; gb[k] = -diff * invp; }
                else { ls += ad - 0.5f; gb[k] = (diff > 0.f ? -1.f : 1.f) * invp; }
            }
            lb += ls * invp;
        }
    }
    red[0][threadIdx.x] = lc;
    red[1][threadIdx.x] = lb;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        d.losses[0] = red[0][0];
        d.losses[1] = red[1][0];
    }
}


// ------------------------------------------------------------------------------------------------
// tf.nn.top_k(scores, k) per image (ProposalLayer, dense_model.py:259-262): the k largest scores in descending order, the
// lower anchor index first among equal scores.  A full sort of the ~262 k anchors of a 1024^2 image is 40x more work than
// needed for k = 6000:
//   1. radix select: three histogram passes over the order-preserving 32-bit image of the scores (11 + 11 + 10 bits) find
//      the k-th largest key T and how many keys equal to T belong to the top k (`need`);
//   2. ordered compaction: every key > T plus the first `need` keys == T IN ANCHOR ORDER (per-block counts, a scan over the
//      blocks, then each block writes at its offsets) -- exactly k candidates whatever the number of ties;
//   3. the k candidates (64-bit: key, then ~index) are put in descending order by a rank sort over several workgroups (k <= 8192; round 6) or,
//      above that, stage by stage in global memory.
// ------------------------------------------------------------------------------------------------
constexpr int TK_BINS = 2048;
constexpr int TK_CHUNK = 1024;          // anchors per block in the count / compaction passes (4 per thread)
constexpr int TK_LDS_MAX = 8192;        // candidates the in-LDS sort holds (64 KB of 64-bit keys)

struct TopkState {
    unsigned hist[3][TK_BINS];
    unsigned prefix;        // the high bits of T decided so far
    unsigned remaining;     // how many of the top k lie in the bin chosen last (after pass 3: how many keys == T to take)
    unsigned pad[2];
};

__device__ __forceinline__ unsigned tk_key(float f) {      // larger float <=> larger unsigned (-0 < +0, as in a bitwise radix sort)
    const unsigned b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// PASS 0: bits 31..21, PASS 1: bits 20..10 of the keys whose bits 31..21 equal the prefix, PASS 2: bits 9..0
template <int PASS>
__global__ __launch_bounds__(256) void topk_hist_kernel(const float* __restrict__ scores, int A_total, TopkState* __restrict__ st) {
    __shared__ unsigned h[TK_BINS];
    constexpr int SHIFT = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0);
    constexpr int BINS = PASS == 2 ? 1024 : 2048;
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < BINS; i += 256) h[i] = 0;
    __syncthreads();
    const unsigned prefix = PASS == 0 ? 0u : st[b].prefix;
    const float* sc = scores + (long)b * A_total;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < A_total; i += gridDim.x * 256) {
        const unsigned key = tk_key(sc[i]);
        if (PASS == 0 || (key >> (SHIFT + (PASS == 1 ? 11 : 10))) == prefix) atomicAdd(&h[(key >> SHIFT) & (BINS - 1)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BINS; i += 256)
        if (h[i]) atomicAdd(&st[b].hist[PASS][i], h[i]);
}

// one block per image: walk the pass's histogram from the top bin down to the bin that holds the `remaining`-th largest key
template <int PASS>
__global__ __launch_bounds__(256) void topk_pick_kernel(TopkState* __restrict__ st, int k) {
    constexpr int BINS = PASS == 2 ? 1024 : 2048, PER = BINS / 256, BITS = PASS == 2 ? 10 : 11;
    __shared__ unsigned part[256];
    TopkState& s = st[blockIdx.x];
    const unsigned remaining = PASS == 0 ? (unsigned)k : s.remaining;
    const int t = threadIdx.x;
    const int top = BINS - 1 - t * PER;                 // thread t owns bins top, top-1, .., top-PER+1
    unsigned sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) sum += s.hist[PASS][top - j];
    part[t] = sum;
    __syncthreads();
    if (t == 0) {                                        // 256 partial sums: a serial walk is shorter than a scan's barriers
        unsigned cum = 0;
        int c = 0;
        while (c < 255 && cum + part[c] < remaining) cum += part[c++];
        int bin = BINS - 1 - c * PER;
        const int last = bin - PER + 1;
        while (bin > last && cum + s.hist[PASS][bin] < remaining) cum += s.hist[PASS][bin--];
        s.prefix = PASS == 0 ? (unsigned)bin : ((s.prefix << BITS) | (unsigned)bin);
        s.remaining = remaining - cum;
    }
}

// per block of TK_CHUNK consecutive anchors: how many keys > T, how many == T
__global__ __launch_bounds__(256) void topk_count_kernel(const float* __restrict__ scores, int A_total, const TopkState* __restrict__ st,
                                                         unsigned* __restrict__ cnt, int nblk) {
    __shared__ unsigned sg[4], se[4];
    const int b = blockIdx.y, blk = blockIdx.x;
    const unsigned T = st[b].prefix;
    const float* sc = scores + (long)b * A_total;
    unsigned g = 0, e = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blk * TK_CHUNK + threadIdx.x * 4 + j;
        if (i < A_total) {
            const unsigned key = tk_key(sc[i]);
            g += key > T;
            e += key == T;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { g += __shfl_down(g, o); e += __shfl_down(e, o); }
    if ((threadIdx.x & 63) == 0) { sg[threadIdx.x >> 6] = g; se[threadIdx.x >> 6] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt[((long)b * nblk + blk) * 2] = sg[0] + sg[1] + sg[2] + sg[3];
        cnt[((long)b * nblk + blk) * 2 + 1] = se[0] + se[1] + se[2] + se[3];
    }
}

// exclusive scan of the per-block counts (one block per image, blocks in anchor order)
__global__ __launch_bounds__(256) void topk_scan_kernel(unsigned* __restrict__ cnt, int nblk) {
    __shared__ unsigned pg[256], pe[256];
    unsigned* c = cnt + (long)blockIdx.x * nblk * 2;
    const int per = (nblk + 255) / 256, t = threadIdx.x;
    unsigned g = 0, e = 0;
    for (int j = 0; j < per; ++j) {
        const int i = t * per + j;
        if (i < nblk) { g += c[2 * i]; e += c[2 * i + 1]; }
    }
    pg[t] = g; pe[t] = e;
    __syncthreads();
    if (t == 0) {
        unsigned ag = 0, ae = 0;
        for (int i = 0; i < 256; ++i) { const unsigned x = pg[i], y = pe[i]; pg[i] = ag; pe[i] = ae; ag += x; ae += y; }
    }
    __syncthreads();
    g = pg[t]; e = pe[t];
    for (int j = 0; j < per; ++j) {
        const int i = t * per + j;
        if (i < nblk) { const unsigned x = c[2 * i], y = c[2 * i + 1]; c[2 * i] = g; c[2 * i + 1] = e; g += x; e += y; }
    }
}

// candidate = (key << 32) | ~index: a descending sort of it is score-descending, index-ascending
__global__ __launch_bounds__(256) void topk_compact_kernel(const float* __restrict__ scores, int A_total, const TopkState* __restrict__ st,
                                                           const unsigned* __restrict__ cnt, int nblk, int k,
                                                           unsigned long long* __restrict__ cand, long cand_stride) {
    __shared__ unsigned wg[4], we[4];
    const int b = blockIdx.y, blk = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned T = st[b].prefix, need = st[b].remaining, n_greater = (unsigned)k - need;
    const float* sc = scores + (long)b * A_total;
    unsigned key[4];
    unsigned g = 0, e = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blk * TK_CHUNK + threadIdx.x * 4 + j;
        key[j] = i < A_total ? tk_key(sc[i]) : 0u;
        const bool in = i < A_total;
        g += in && key[j] > T;
        e += in && key[j] == T;
    }
    // exclusive prefix over the block's threads (wave scan + 4 wave totals)
    unsigned ig = g, ie = e;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned a = __shfl_up(ig, o), c = __shfl_up(ie, o);
        if (lane >= o) { ig += a; ie += c; }
    }
    if (lane == 63) { wg[wave] = ig; we[wave] = ie; }
    __syncthreads();
    unsigned og = cnt[((long)b * nblk + blk) * 2] + ig - g, oe = cnt[((long)b * nblk + blk) * 2 + 1] + ie - e;
    for (int w = 0; w < wave; ++w) { og += wg[w]; oe += we[w]; }
    unsigned long long* out = cand + (long)b * cand_stride;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blk * TK_CHUNK + threadIdx.x * 4 + j;
        if (i >= A_total) break;
        const unsigned long long v = ((unsigned long long)key[j] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        if (key[j] > T) out[og++] = v;
        else if (key[j] == T) { if (oe < need) out[n_greater + oe] = v; ++oe; }
    }
}

// k <= TK_LDS_MAX, round 6: a RANK sort over several workgroups instead of the one-workgroup bitonic network (91 barrier-separated stages
// for 8192 slots: 72 us on the joint step's serial chain).  The candidates' 64-bit words are unique (key, then ~index), so a
// candidate's position in the descending order is the number of candidates above it: every workgroup stages all k words in LDS (48 KB at
// k = 6000) and ranks 32 candidates, eight lanes each (188 workgroups at k = 6000), then writes each candidate's unpacked (index, score)
// at its rank.  Same output as the network, bit for bit.  (A first version with one lane per candidate -- 3000 dependent iterations
// of a 64-bit compare per thread, 24 workgroups -- took the network's 71 us: the walk is VALU- and LDS-latency-bound per thread.)
__global__ __launch_bounds__(256) void topk_ranksort_kernel(const unsigned long long* __restrict__ cand, long cand_stride, int k,
                                                            int* __restrict__ vals, float* __restrict__ keys, long out_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int b = blockIdx.y;
    const unsigned long long* in = cand + (long)b * cand_stride;
    const int pairs = (k + 1) >> 1;
    for (int i = threadIdx.x; i < 2 * pairs; i += 256) sk[i] = i < k ? in[i] : 0ull;      // (odd k: a 0 word, below every real one)
    __syncthreads();
    // 32 candidates per workgroup, EIGHT lanes per candidate: each lane counts the candidates above `mine` in an eighth of the list,
    // four independent 16-byte reads in flight; the eight partial counts meet by three shuffles
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    const unsigned long long mine = i < k ? sk[i] : ~0ull;
    const int per = (pairs + 7) >> 3, j0 = part * per, j1 = min(pairs, j0 + per);
    int rank = 0;
    int j = j0;
    for (; j + 4 <= j1; j += 4) {
        const u64x2 a = *reinterpret_cast<const u64x2*>(&sk[2 * j]), c = *reinterpret_cast<const u64x2*>(&sk[2 * j + 2]);
        const u64x2 e = *reinterpret_cast<const u64x2*>(&sk[2 * j + 4]), g = *reinterpret_cast<const u64x2*>(&sk[2 * j + 6]);
        rank += (a[0] > mine) + (a[1] > mine) + (c[0] > mine) + (c[1] > mine) + (e[0] > mine) + (e[1] > mine) + (g[0] > mine) + (g[1] > mine);
    }
    for (; j < j1; ++j) {
        const u64x2 a = *reinterpret_cast<const u64x2*>(&sk[2 * j]);
        rank += (a[0] > mine) + (a[1] > mine);
    }
    rank += __shfl_xor(rank, 1, 64);
    rank += __shfl_xor(rank, 2, 64);
    rank += __shfl_xor(rank, 4, 64);
    if (i >= k || part != 0) return;
    vals[(long)b * out_stride + rank] = (int)(0xFFFFFFFFu - (unsigned)(mine & 0xFFFFFFFFull));
    const unsigned key = (unsigned)(mine >> 32), bits = key ^ ((key >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    keys[(long)b * out_stride + rank] = __uint_as_float(bits);
}

// k > TK_LDS_MAX: the same network, one launch per stage over the padded candidate array in global memory
__global__ __launch_bounds__(256) void topk_pad_kernel(unsigned long long* __restrict__ cand, long cand_stride, int k, int P) {
    for (int i = k + blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) cand[(long)blockIdx.y * cand_stride + i] = 0ull;
}
__global__ __launch_bounds__(256) void topk_sort_step_kernel(unsigned long long* __restrict__ cand, long cand_stride, int P, int kk, int j) {
    unsigned long long* a = cand + (long)blockIdx.y * cand_stride;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < P / 2; t += gridDim.x * 256) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
        const bool desc = (lo & kk) == 0;
        const unsigned long long x = a[lo], y = a[hi];
        if ((x < y) == desc) { a[lo] = y; a[hi] = x; }
    }
}
__global__ __launch_bounds__(256) void topk_unpack_kernel(const unsigned long long* __restrict__ cand, long cand_stride, int k,
                                                          int* __restrict__ vals, float* __restrict__ keys, long out_stride) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < k; i += gridDim.x * 256) {
        const unsigned long long v = cand[(long)b * cand_stride + i];
        vals[(long)b * out_stride + i] = (int)(0xFFFFFFFFu - (unsigned)(v & 0xFFFFFFFFull));
        const unsigned key = (unsigned)(v >> 32), bits = key ^ ((key >> 31) ? 0x80000000u : 0xFFFFFFFFu);
        keys[(long)b * out_stride + i] = __uint_as_float(bits);
    }
}

static int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// scores [B][A_total] -> vals / keys [B][A_total-strided], first k entries: the top k in tf.nn.top_k's order
static int topk_select(const float* scores, int B, int A_total, int k, TopkState* st, unsigned* cnt, unsigned long long* cand,
                       int* vals, float* keys, hipStream_t s) {
    const int nblk = (A_total + TK_CHUNK - 1) / TK_CHUNK, P = next_pow2(k);
    int zrc = zero_fill_async(st, sizeof(TopkState) * (size_t)B, s);
    if (zrc) return zrc;
    const dim3 hgrid(std::min((A_total + 255) / 256, 128), B);
    hipLaunchKernelGGL(topk_hist_kernel<0>, hgrid, dim3(256), 0, s, scores, A_total, st);
    hipLaunchKernelGGL(topk_pick_kernel<0>, dim3(B), dim3(256), 0, s, st, k);
    hipLaunchKernelGGL(topk_hist_kernel<1>, hgrid, dim3(256), 0, s, scores, A_total, st);
    hipLaunchKernelGGL(topk_pick_kernel<1>, dim3(B), dim3(256), 0, s, st, k);
    hipLaunchKernelGGL(topk_hist_kernel<2>, hgrid, dim3(256), 0, s, scores, A_total, st);
    hipLaunchKernelGGL(topk_pick_kernel<2>, dim3(B), dim3(256), 0, s, st, k);
    hipLaunchKernelGGL(topk_count_kernel, dim3(nblk, B), dim3(256), 0, s, scores, A_total, st, cnt, nblk);
    hipLaunchKernelGGL(topk_scan_kernel, dim3(B), dim3(256), 0, s, cnt, nblk);
    hipLaunchKernelGGL(topk_compact_kernel, dim3(nblk, B), dim3(256), 0, s, scores, A_total, st, cnt, nblk, k, cand, (long)P);
    int rc = check_launch("top-k select kernels");
    if (rc) return rc;
    if (P <= TK_LDS_MAX) {
        DC_ENSURE_DYN_LDS(&topk_ranksort_kernel, TK_LDS_MAX * 8);
        hipLaunchKernelGGL(topk_ranksort_kernel, dim3((k + 31) / 32, B), dim3(256), (size_t)((k + 1) & ~1) * 8, s, cand, (long)P, k, vals, keys, (long)A_total);
        return check_launch("topk_ranksort_kernel");
    }
    const dim3 sgrid(std::min((P / 2 + 255) / 256, kNumCU * 4), B);
    hipLaunchKernelGGL(topk_pad_kernel, sgrid, dim3(256), 0, s, cand, (long)P, k, P);
    for (int kk = 2; kk <= P; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) hipLaunchKernelGGL(topk_sort_step_kernel, sgrid, dim3(256), 0, s, cand, (long)P, P, kk, j);
    hipLaunchKernelGGL(topk_unpack_kernel, sgrid, dim3(256), 0, s, cand, (long)P, k, vals, keys, (long)A_total);
    return check_launch("top-k global sort kernels");
}

struct ProposalWs {
    size_t scores, deltas, iota, keys, vals, boxes, mask, tk_state, tk_cnt, tk_cand, total;
};

static ProposalWs proposal_layout(const dc_proposal_desc* d) {
    ProposalWs w{};
    const int k = std::min(d->pre_nms_limit, d->A_total), words = (k + 63) / 64;
    size_t o = 0;
    w.scores = o; o += up256((size_t)d->B * d->A_total * 4);
    w.deltas = o; o += up256((size_t)d->B * d->A_total * 16);
    w.iota = o;   o += up256((size_t)d->A_total * 4);
    w.keys = o;   o += up256((size_t)d->B * d->A_total * 4);
    w.vals = o;   o += up256((size_t)d->B * d->A_total * 4);
    w.boxes = o;  o += up256((size_t)d->B * k * 16);
    w.mask = o;   o += up256((size_t)d->B * k * words * 8);
    w.tk_state = o; o += up256(sizeof(TopkState) * (size_t)d->B);
    w.tk_cnt = o;   o += up256((size_t)d->B * ((d->A_total + TK_CHUNK - 1) / TK_CHUNK) * 2 * sizeof(unsigned));
    w.tk_cand = o;  o += up256((size_t)d->B * next_pow2(k) * sizeof(unsigned long long));
    w.total = o;
    return w;
}

static int proposal_validate(const dc_proposal_desc* d) {
    DC_REQUIRE(d && d->anchors && d->proposals, DC_EINVAL, "dc_proposals: null pointer");
    DC_REQUIRE(d->B > 0 && d->levels >= 1 && d->levels <= 5 && d->anchors_per_loc > 0 && d->A_total > 0 && d->pre_nms_limit > 0 &&
                   d->proposal_count > 0,
               DC_EINVAL, "dc_proposals: bad sizes");
    long a = 0;
    for (int l = 0; l < d->levels; ++l) {
        DC_REQUIRE(d->heads[l] && d->Hs[l] > 0 && d->Ws[l] > 0, DC_EINVAL, "dc_proposals: bad head %d", l);
        a += (long)d->Hs[l] * d->Ws[l] * d->anchors_per_loc;
    }
    DC_REQUIRE(d->head_stride == 0 || d->head_stride >= d->anchors_per_loc * 6, DC_EINVAL, "dc_proposals: head_stride too small");
    DC_REQUIRE(a == d->A_total, DC_EINVAL, "dc_proposals: A_total (%d) != sum of H*W*A over levels (%ld)", d->A_total, a);
    DC_REQUIRE(aligned16(d->anchors) && aligned16(d->proposals), DC_EALIGN, "dc_proposals: anchors/proposals must be 16-byte aligned");
    return DC_OK;
}

}  // namespace dcap

using namespace dcap;

extern "C" size_t dc_proposals_workspace_bytes(const dc_proposal_desc* d) {
    if (proposal_validate(d)) return 0;
    return proposal_layout(d).total;
}

extern "C" int dc_proposals_f32(const dc_proposal_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = proposal_validate(d);
    if (rc) return rc;
    const ProposalWs L = proposal_layout(d);
    DC_REQUIRE(workspace && workspace_bytes >= L.total && aligned16(workspace), DC_EWORKSPACE,
               "dc_proposals: needs %zu workspace bytes, got %zu", L.total, workspace_bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    float* scores = reinterpret_cast<float*>(ws + L.scores);
    float* deltas = reinterpret_cast<float*>(ws + L.deltas);
    int* iota = reinterpret_cast<int*>(ws + L.iota);
    float* keys = reinterpret_cast<float*>(ws + L.keys);
    int* vals = reinterpret_cast<int*>(ws + L.vals);
    float4* boxes = reinterpret_cast<float4*>(ws + L.boxes);
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(ws + L.mask);
    const int k = std::min(d->pre_nms_limit, d->A_total), words = (k + 63) / 64;

    int off = 0;
    for (int l = 0; l < d->levels; ++l) {
        const long n = (long)d->B * d->Hs[l] * d->Ws[l] * d->anchors_per_loc;
        const int blocks = (int)std::min<long>((n + 255) / 256, (long)kNumCU * 8);
        hipLaunchKernelGGL(rpn_score_kernel, dim3(blocks), dim3(256), 0, s, *d, l, off, scores, deltas, iota);
        off += d->Hs[l] * d->Ws[l] * d->anchors_per_loc;
    }
    rc = check_launch("rpn_score_kernel");
    if (rc) return rc;
    rc = topk_select(scores, d->B, d->A_total, k, reinterpret_cast<TopkState*>(ws + L.tk_state), reinterpret_cast<unsigned*>(ws + L.tk_cnt),
                     reinterpret_cast<unsigned long long*>(ws + L.tk_cand), vals, keys, s);
    if (rc) return rc;
    hipLaunchKernelGGL(decode_kernel, dim3((d->B * k + 255) / 256), dim3(256), 0, s, *d, vals, deltas, boxes, k);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words, d->B), dim3(64), 0, s, boxes, mask, k, words, d->nms_threshold);
    if (words <= 128)
        hipLaunchKernelGGL(nms_scan_wave_kernel, dim3(d->B), dim3(NMS_WAVES * 64), 0, s, boxes, mask, k, words, d->proposal_count,
                           reinterpret_cast<float4*>(d->proposals), d->keep_out);
    else
        hipLaunchKernelGGL(nms_scan_kernel, dim3(d->B), dim3(64), (size_t)words * 8, s, boxes, mask, k, words, d->proposal_count,
                           reinterpret_cast<float4*>(d->proposals), d->keep_out);
    rc = check_launch("proposal kernels");
    if (rc) return rc;
    if (d->scores_out) {
        hipError_t e = hipMemcpyAsync(d->scores_out, scores, (size_t)d->B * d->A_total * 4, hipMemcpyDeviceToDevice, s);
        DC_REQUIRE(e == hipSuccess, DC_ELAUNCH, "dc_proposals: copy failed");
    }
    if (d->order_out) {
        hipError_t e = hipMemcpy2DAsync(d->order_out, (size_t)k * 4, vals, (size_t)d->A_total * 4, (size_t)k * 4, d->B, hipMemcpyDeviceToDevice, s);
        DC_REQUIRE(e == hipSuccess, DC_ELAUNCH, "dc_proposals: copy failed");
    }
    return DC_OK;
}

// ------------------------------------------------------------------------------------------------
// DetectionTargetLayer on the device (dense_img_cap/dense_model.py:450-572, detection_targets_graph for ONE image): the IoU
// matrix proposals x GT boxes in float32 (overlaps_graph :421-447), positives = best IoU >= 0.5, negatives = best IoU < 0.5 (a NaN
// row -- 0/0 of two empty boxes -- is neither, as in the graph), tf.random_shuffle of each index list realised as a sort by
// counter-based random keys (Philox(compacted proposal index, offset, seed); without a seed: proposal order), at most
// int(n_rois * ratio) positives, int32(float32(1 / ratio) * float32(n_pos)) - n_pos negatives, every positive takes the caption of
// its best GT box (first maximum, tf.argmax), the rest of the n_rois rows is zero.  ONE workgroup (2000 proposals, <= 100 GT boxes):
// classification, a ballot scan for the compacted index, ranks by counting smaller (class, key, index) triples in LDS.
// Replaces the device -> host -> device hop in the middle of the joint train step: nothing here needs the host.
// ------------------------------------------------------------------------------------------------
constexpr int DT_MAX_PROPOSALS = 4096, DT_MAX_GT = 512, DT_THREADS = 1024;

__global__ __launch_bounds__(DT_THREADS) void detection_targets_kernel(dc_detection_targets_desc d) {
    __shared__ __attribute__((aligned(16))) unsigned long long v[DT_MAX_PROPOSALS + 2];   // (class << 62) | (key << 16 ...) see below
    __shared__ float4 gbox[DT_MAX_GT];
    __shared__ unsigned short best_g[DT_MAX_PROPOSALS];
    __shared__ int wave_nz[DT_THREADS / 64];
    __shared__ int cnt[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = d.n_proposals, G = d.n_gt;
    const float4* props = reinterpret_cast<const float4*>(d.proposals);
    for (int g = tid; g < G; g += DT_THREADS) gbox[g] = reinterpret_cast<const float4*>(d.gt_boxes)[g];
    if (tid < 2) cnt[tid] = 0;
    __syncthreads();
    const unsigned offset = d.offset + (d.offset_dev ? d.offset_dev[0] : 0u);
    int base = 0;                                            // non-zero proposals in the chunks before this one
    for (int i0 = 0; i0 < N; i0 += DT_THREADS) {
        const int i = i0 + tid;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < N) b = props[i];                             // (y1, x1, y2, x2)
        const bool nz = i < N && (fabsf(b.x) + fabsf(b.y) + fabsf(b.z) + fabsf(b.w)) > 0.f;
        // ---- class and best GT box
        int cls = 2;                                         // 0 positive, 1 negative, 2 neither (padding row, NaN row)
        int bg = 0;
        if (nz) {
            float best = 0.f;
            bool any = false, isnan_ = false;
            const float a1 = (b.z - b.x) * (b.w - b.y);
            for (int g = 0; g < G; ++g) {
                const float4 q = gbox[g];
                if (!((fabsf(q.x) + fabsf(q.y) + fabsf(q.z) + fabsf(q.w)) > 0.f)) continue;      // trim_zeros_graph on the GT boxes
                const float y1 = fmaxf(b.x, q.x), x1 = fmaxf(b.y, q.y), y2 = fminf(b.z, q.z), x2 = fminf(b.w, q.w);
                const float inter = fmaxf(x2 - x1, 0.f) * fmaxf(y2 - y1, 0.f);
                const float a2 = (q.z - q.x) * (q.w - q.y);
                const float iou = inter / (a1 + a2 - inter);
                if (iou != iou) isnan_ = true;
                if (!any || iou > best) { best = iou; bg = g; any = true; }
            }
            // no GT box at all: the graph's reduce_max over an empty axis; every proposal is a negative (iou_max = 0 in the oracle)
            if (!any) cls = 1;
            else if (isnan_) cls = 2;
            else cls = best >= 0.5f ? 0 : 1;
        }
        // ---- compacted index (position among the non-zero proposals): ballot scan
        const unsigned long long bal = __ballot(nz);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_nz[wave] = __popcll(bal);
        __syncthreads();
        int wbase = base;
        for (int w = 0; w < wave; ++w) wbase += wave_nz[w];
        int chunk = 0;
        for (int w = 0; w < DT_THREADS / 64; ++w) chunk += wave_nz[w];
        const unsigned cidx = (unsigned)(wbase + before);
        const unsigned key = d.shuffle ? philox2x32(cidx, offset, d.seed) : cidx;
        if (i < N) {
            v[i] = ((unsigned long long)cls << 62) | ((unsigned long long)key << 16) | (unsigned long long)(cidx & 0xFFFFu);
            best_g[i] = (unsigned short)bg;
            if (cls < 2) atomicAdd(&cnt[cls], 1);
        }
        base += chunk;
        __syncthreads();
    }
    // ---- selection sizes
    const int total_pos = cnt[0], total_neg = cnt[1];
    const int npos = min(total_pos, d.max_positive);
    const int want_neg = (int)(d.inv_ratio * (float)npos) - npos;          // tf.cast(r * tf.cast(positive_count, tf.float32), tf.int32) - positive_count
    const int nneg = max(0, min(min(total_neg, want_neg), d.n_rois - npos));
    // ---- zero fill (workgroup 0) and the selected rows (every workgroup its own proposals): disjoint rows, no ordering issue
    const int T = d.T;
    // ---- ranks: number of (class, key, index) triples below mine.  Round 6: the launch is ceil(N / 256) workgroups; every one of them
    // has built the same v[] above (the class / key pass is 8 us of redundant work), and ranks ITS 256 proposals with FOUR lanes per
    // proposal, each lane walking a quarter of the list with independent 16-byte LDS reads in flight (the one-workgroup walk of rounds 4-5
    // was N iterations per thread behind one LDS latency each: 112 us at 2000 proposals, on the step's serial chain).
    const int own_i = blockIdx.x * (DT_THREADS / 4) + (tid >> 2), part = tid & 3;
    const bool have = own_i < N;
    const unsigned long long mine = have ? v[own_i] : 0ull;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int pairs = (N + 1) >> 1;                          // v[N] is written as ~0 below when N is odd: above every real entry
    if ((N & 1) && tid == 0) v[N] = ~0ull;
    __syncthreads();
    const int per = (pairs + 3) >> 2, j0 = part * per, j1 = min(pairs, j0 + per);
    int rank = 0;
    int j = j0;
    for (; j + 4 <= j1; j += 4) {
        const u64x2 a = *reinterpret_cast<const u64x2*>(&v[2 * j]), b = *reinterpret_cast<const u64x2*>(&v[2 * j + 2]);
        const u64x2 c = *reinterpret_cast<const u64x2*>(&v[2 * j + 4]), e = *reinterpret_cast<const u64x2*>(&v[2 * j + 6]);
        rank += (a[0] < mine) + (a[1] < mine) + (b[0] < mine) + (b[1] < mine) + (c[0] < mine) + (c[1] < mine) + (e[0] < mine) + (e[1] < mine);
    }
    for (; j < j1; ++j) {
        const u64x2 a = *reinterpret_cast<const u64x2*>(&v[2 * j]);
        rank += (a[0] < mine) + (a[1] < mine);
    }
    rank += __shfl_xor(rank, 1, 64);
    rank += __shfl_xor(rank, 2, 64);
    if (blockIdx.x == 0) {                                   // zero fill + counts: once (rows the selected proposals do not touch)
        for (int r = tid; r < d.n_rois; r += DT_THREADS)
            if (r >= npos + nneg) reinterpret_cast<float4*>(d.rois)[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = tid; e < d.n_rois * T; e += DT_THREADS)
            if (e / T >= npos) d.captions[e] = 0;
        if (tid == 0) { d.counts[0] = npos; d.counts[1] = nneg; }
    }
    if (!have || part != 0) return;
    const int cls = (int)(mine >> 62);
    if (cls == 2) return;
    if (cls == 0) {
        if (rank < npos) {
            reinterpret_cast<float4*>(d.rois)[rank] = props[own_i];
            const int32_t* src = d.gt_captions + (long)best_g[own_i] * T;
            for (int t = 0; t < T; ++t) d.captions[(long)rank * T + t] = src[t];
        }
    } else {
        rank -= total_pos;
        if (rank < nneg) reinterpret_cast<float4*>(d.rois)[npos + rank] = props[own_i];
    }
}

// The decoder's index tables from device-resident captions [B][T] (dense_img_cap/dense_model.py:1572-1580 + imgcap_caption_loss_graph
// :936-946): time-major token ids and Keras masks (ids != 0), time-major targets = the caption shifted left by one (last = 0),
// row weights = [target > 0] / max(count, 1) (the masked mean of the loss; count summed in a fixed order).  One workgroup.
__global__ __launch_bounds__(1024) void caption_tables_kernel(const int32_t* __restrict__ caps, int B, int T, int32_t* __restrict__ ids_tm,
                                                               unsigned char* __restrict__ mask, int32_t* __restrict__ targets_tm,
                                                               float* __restrict__ row_weights, int32_t* __restrict__ live_count) {
    __shared__ int red[1024];
    const int n = B * T;
    int mine = 0;
    for (int e = threadIdx.x; e < n; e += 1024) {
        const int t = e / B, b = e - t * B;
        const int id = caps[(long)b * T + t];
        const int tg = t + 1 < T ? caps[(long)b * T + t + 1] : 0;
        ids_tm[e] = id;
        mask[e] = id != 0;
        targets_tm[e] = tg;
        mine += tg > 0;
    }
    red[threadIdx.x] = mine;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    const int count = red[0];
    const float inv = 1.f / (float)max(count, 1);
    for (int e = threadIdx.x; e < n; e += 1024) row_weights[e] = targets_tm[e] > 0 ? inv : 0.f;
    if (threadIdx.x == 0 && live_count) live_count[0] = count;
}

extern "C" int dc_detection_targets_f32(const dc_detection_targets_desc* d, void* stream) {
    DC_REQUIRE(d && d->proposals && d->gt_boxes && d->gt_captions && d->rois && d->captions && d->counts, DC_EINVAL,
               "dc_detection_targets: null pointer");
    DC_REQUIRE(d->n_proposals > 0 && d->n_proposals <= DT_MAX_PROPOSALS && d->n_gt >= 0 && d->n_gt <= DT_MAX_GT && d->n_rois > 0 && d->T > 0 &&
                   d->max_positive >= 0 && d->max_positive <= d->n_rois && d->inv_ratio > 0.f,
               DC_EINVAL, "dc_detection_targets: needs 1..%d proposals, 0..%d GT boxes, n_rois > 0, 0 <= max_positive <= n_rois", DT_MAX_PROPOSALS,
               DT_MAX_GT);
    DC_REQUIRE(aligned16(d->proposals) && aligned16(d->gt_boxes) && aligned16(d->rois), DC_EALIGN, "dc_detection_targets: boxes must be 16-byte aligned");
    hipLaunchKernelGGL(detection_targets_kernel, dim3((d->n_proposals + DT_THREADS / 4 - 1) / (DT_THREADS / 4)), dim3(DT_THREADS), 0,
                       static_cast<hipStream_t>(stream), *d);
    return check_launch("detection_targets_kernel");
}

extern "C" int dc_caption_tables_i32(const int32_t* captions, int B, int T, int32_t* ids_tm, uint8_t* mask, int32_t* targets_tm, float* row_weights,
                                     int32_t* live_count, void* stream) {
    DC_REQUIRE(captions && ids_tm && mask && targets_tm && row_weights && B > 0 && T > 0, DC_EINVAL, "dc_caption_tables: bad arguments");
    hipLaunchKernelGGL(caption_tables_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), captions, B, T, ids_tm, mask, targets_tm,
                       row_weights, live_count);
    return check_launch("caption_tables_kernel");
}

extern "C" int dc_rpn_loss_grad_f32(const dc_rpn_loss_desc* d, void* stream) {
    DC_REQUIRE(d && d->losses && d->levels >= 1 && d->levels <= 5 && d->anchors_per_loc > 0 && d->head_stride >= d->anchors_per_loc * 6,
               DC_EINVAL, "dc_rpn_loss_grad: bad descriptor");
    DC_REQUIRE(d->n_sel == 0 || (d->sel_level && d->sel_index && d->sel_match), DC_EINVAL, "dc_rpn_loss_grad: missing selection");
    DC_REQUIRE(d->n_pos == 0 || d->target_deltas, DC_EINVAL, "dc_rpn_loss_grad: missing target deltas");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (d->n_sel == 0) return zero_fill_async(d->losses, 2 * sizeof(float), s);      // (the kernel below writes both losses itself)
    hipLaunchKernelGGL(rpn_loss_grad_kernel, dim3(1), dim3(256), 0, s, *d);
    return check_launch("rpn_loss_grad_kernel");
}

extern "C" int dc_subsample2_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    DC_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, DC_EINVAL, "dc_subsample2: bad arguments");
    DC_REQUIRE(aligned16(x) && aligned16(y), DC_EALIGN, "dc_subsample2: pointers must be 16-byte aligned");
    const long total = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    const int blocks = (int)std::min<long>((total + 255) / 256, (long)kNumCU * 8);
    hipLaunchKernelGGL(subsample2_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), N, H, W, C / 4);
    return check_launch("subsample2_kernel");
}
