/*
 * dcap.h -- C-ABI of libdcap_hip.so: the MI355X (gfx950) kernels of the dense-captioning hot path.
 *
 * The reference (frosinastojanovska/image-captioning) has no FFI of its own: every arithmetic op on
 * the path is a Keras-2.1 / TF-1.x call.  Each entry point below replaces one family of those call
 * sites (cited as file:line relative to the reference root); the Python host side
 * (the modules under image-captioning_amd/) binds them with ctypes and keeps the reference's module/function names.
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - return 0 on success, a negative DC_E* code otherwise; dc_last_error() gives a thread-local text;
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller;
 *   - enqueue-only on `stream` (a hipStream_t passed as void*): no allocation, no synchronisation,
 *     safe under hipGraph stream capture; stateless and re-entrant on distinct streams;
 *   - scratch memory is a caller-provided workspace; dc_*_workspace_bytes() gives the size;
 *   - tensors are row-major, activations NHWC, float32 unless stated.
 */
#ifndef DCAP_H
#define DCAP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DC_OK            0
#define DC_EINVAL       -1   /* bad shape / null pointer / unsupported combination */
#define DC_EALIGN       -2   /* pointer or leading dimension not 16-byte aligned where required */
#define DC_EWORKSPACE   -3   /* workspace too small */
#define DC_ELAUNCH      -4   /* HIP launch error (text in dc_last_error) */

/* ABI version = 100*major + minor.  The major number changes whenever a descriptor struct's layout changes (fields removed or
 * reordered: round 5 removed dc_conv_desc.w_split / w_wino4 and grew dc_amsgrad_desc => major 6); a caller compiled against this
 * header checks dc_version() / 100 == DC_ABI_VERSION / 100 before the first call (image-captioning_amd/_lib.py does at load). */
#define DC_ABI_VERSION 600
int         dc_version(void);
const char* dc_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM  C[M,N] = epilogue(A[M,K] * B[K,N])            (fp32 in, fp32 MFMA accumulate)
 * Replaces KL.Dense / the LSTM kernels' x.W products / the 7x7-valid + 1x1 RoI-head convs:
 *   dense_img_cap_separate_models/text_generation_model.py:143-144,153-154,251-259
 *   dense_img_cap_separate_models/text_generation_model_v2.py:141-150,157,163-164
 * and their gradients (dgrad = dY*W^T: b_trans=1; wgrad = A^T*dY: a_trans=1).
 *   a_trans=0: A is [M][lda] (K contiguous);   a_trans=1: A is [K][lda] (M contiguous)
 *   b_trans=0: B is [K][ldb] (N contiguous, the Keras [in,out] kernel);  b_trans=1: B is [N][ldb]
 *   a_gather (optional, a_trans=0): row m of A is A[a_gather[m]] -- the fused KL.Embedding lookup
 *     (text_generation_model.py:135-140, _v2.py:155-156); with a_trans=1 it indexes the K rows
 *     (A^T of a gathered matrix: the embedding-side wgrad)
 * epilogue: v = acc*scale[n] + shift[n] (each optional), += residual[m][n] (optional),
 *           relu (optional), then C = v  or  C += v (accumulate).
 * split_k > 1 partitions K over blockIdx.z through fp32 slabs in the workspace (deterministic
 * reduction order); 0 lets the library choose.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int M, N, K;
    const float*   A;  int lda;  int a_trans;
    const int32_t* a_gather;
    const float*   B;  int ldb;  int b_trans;
    float*         C;  int ldc;
    const float*   scale;
    const float*   shift;
    const float*   residual;  int ldr;
    int res_rows;             /* >0: residual row = m % res_rows (per-RoI term broadcast over timesteps) */
    int relu;
    int accumulate;
    int split_k;
} dc_gemm_desc;

size_t dc_gemm_workspace_bytes(const dc_gemm_desc* d);
int    dc_gemm_f32(const dc_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GEMM with bf16 operands (bit patterns in uint16_t), fp32 accumulate on the bf16 matrix pipe -- the arithmetic BASELINE
 * configs[4] asks for (joint model in bf16: dense_img_cap/dense_model.py:738-817 RoI head + Model-3 decoder, :936-946
 * vocabulary softmax).  Same operand conventions and epilogue as dc_gemm_f32; the result is written as fp32 (C), as bf16
 * (Cb: the next bf16 GEMM's operand), or both.  K, lda, ldb multiples of 8; a K-major operand (a_trans = 1 / b_trans = 0)
 * needs its row length (M / N) to be a multiple of 8.  a_gather (optional) indexes rows of a table of a_gather_rows rows:
 * with a_trans = 0 tile row m is table row a_gather[m] (embedding lookup), with a_trans = 1 K row k is a_gather[k].
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int M, N, K;
    const uint16_t* A;  int lda;  int a_trans;
    const int32_t*  a_gather;  int a_gather_rows;
    const uint16_t* B;  int ldb;  int b_trans;
    float*          C;  int ldc;
    uint16_t*       Cb; int ldcb;
    const float*    scale;
    const float*    shift;
    const float*    residual;  int ldr;
    int res_rows;
    int relu;
    int accumulate;
    int split_k;
} dc_gemm_bf16_desc;

size_t dc_gemm_bf16_workspace_bytes(const dc_gemm_bf16_desc* d);
int    dc_gemm_bf16(const dc_gemm_bf16_desc* d, void* workspace, size_t workspace_bytes, void* stream);
/* Profiling / test aid: the block tile dc_gemm_bf16 runs `d` on -- 256 (bgemm256_kernel: 256 x 256 x 64, 8 waves, one block per CU,
 * v_mfma_f32_16x16x32_bf16) or 128 (bgemm_kernel: 128 x 128 x 64, 4 waves, two blocks per CU); *split_k (may be NULL) receives the
 * number of split-K slices.  Returns 0 for an invalid descriptor. */
int    dc_gemm_bf16_tile(const dc_gemm_bf16_desc* d, int* split_k);

/* conv2d weight gradient on the bf16 matrix pipe (configs[4]: the joint model's trainable FPN / RPN convolutions,
 * dense_img_cap/dense_model.py:1829-1831): dw[cout][(ky,kx,ci)] (+)= sum over output pixels of dy[p][cout] * x[p*stride + tap - pad][ci],
 * packed like the forward weights, fp32 accumulation and output.  x [N,H,W,Cin] and dy [N,Ho,Wo,Cout] are bf16 (dc_cast_f32_bf16 of the
 * fp32 activations / gradients).  Cin % 128 == 0, Cout % 8 == 0.  Same sums as dc_conv2d_wgrad_f32 on operands rounded to bf16. */
typedef struct {
    int N, H, W, Cin;
    int Cout, kh, kw, stride, pad_t, pad_l;
    int Ho, Wo;
    const uint16_t* x;
    const uint16_t* dy;
    float* dw;
    int accumulate;
    int split_k;
} dc_conv_wgrad_bf16_desc;
size_t dc_conv2d_wgrad_bf16_workspace_bytes(const dc_conv_wgrad_bf16_desc* d);
int    dc_conv2d_wgrad_bf16(const dc_conv_wgrad_bf16_desc* d, void* workspace, size_t workspace_bytes, void* stream);
/* Test / profiling aid: the block tile (256 | 128) and split-K slices dc_conv2d_wgrad_bf16 runs `d` with. */
int    dc_conv2d_wgrad_bf16_tile(const dc_conv_wgrad_bf16_desc* d, int* split_k);

/* conv2d forward with bf16 STORAGE (configs[4]: "bf16"): bf16 NHWC activations and bf16 packed weights in HBM, fp32 accumulation,
 * the same fused epilogue as dc_conv2d_nhwc_f32 (frozen-BN scale / shift, residual (fp32), ReLU), output in fp32 (y), bf16 (y_bf16: the
 * next convolution's input) or both.  Replaces the same Keras layers as dc_conv2d_nhwc_f32 (feature_generation/dense_model.py:85-100,
 * :120-139, :1406-1421) for the joint model's bf16 mode; on weights rotated by dc_conv_weight_dgrad_pack it is the data gradient.
 *   x [N,H,W,Cin] bf16, Cin % 64 == 0;  w PACKED [Cout][kh*kw*Cin] bf16;  y / y_bf16 [N,Ho,Wo,Cout];
 *   res_mode: 0 none, 1 residual of the output's shape, 2 residual on the 2x coarser map (nearest-upsampled), as for the fp32
 *   convolution.  Taps in the padding read zeros. */
typedef struct {
    int N, H, W, Cin;
    int Cout, kh, kw, stride, pad_t, pad_l;
    int Ho, Wo;
    const uint16_t* x;
    const uint16_t* w;
    float*          y;        /* may be NULL when y_bf16 is given */
    uint16_t*       y_bf16;   /* may be NULL when y is given */
    const float* scale;
    const float* shift;
    const float* residual;  int res_mode;
    int relu;
    int split_k;
    int tile;                 /* 0 = the library's cost model picks the block tile; 64 | 128 | 256 = run on that tile where the shape allows it
                               * (tests and benchmarks; dc_conv2d_bf16_tile reports what will run) */
} dc_conv_bf16_desc;
size_t dc_conv2d_bf16_workspace_bytes(const dc_conv_bf16_desc* d);
int    dc_conv2d_bf16(const dc_conv_bf16_desc* d, void* workspace, size_t workspace_bytes, void* stream);
/* Test / profiling aid: the block tile `d` runs on -- 256 (bconv256_kernel, 256 x 256 x 64), 128 (bconv_kernel) or 64 (bconv64_kernel: whole K
 * loop per block, no split-K); *split_k (may be NULL) = slices. */
int    dc_conv2d_bf16_tile(const dc_conv_bf16_desc* d, int* split_k);

/* fp32 -> bf16 (round to nearest even): the bf16 shadow of weights / activations that feed dc_gemm_bf16.
 * _2d: rows x cols with row strides, output columns cols..cols_out-1 zero-filled (pads K to a multiple of 8). */
int dc_cast_f32_bf16(const float* x, uint16_t* out, size_t n, void* stream);
/* bf16 -> fp32 (exact).  Data-parallel joint model with a bf16 gradient exchange (configs[4]'s buckets travel as bf16,
 * half the bytes on xGMI): the all-reduced bf16 bucket goes back into the fp32 gradient bucket the clip norm and AMSGrad read. */
int dc_cast_bf16_f32(const uint16_t* x, float* out, size_t n, void* stream);
int dc_cast_f32_bf16_2d(const float* x, int ld_in, uint16_t* out, int ld_out, int rows, int cols, int cols_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * conv2d NHWC forward as an implicit GEMM (LDS im2col tiles), fused frozen-BN / bias / residual /
 * ReLU epilogue.  Replaces KL.Conv2D + BatchNorm(training=False) + Add + Activation('relu'):
 *   feature_generation/dense_model.py:85-100 (identity_block), :120-139 (conv_block),
 *   :146-149 (stem), :1406-1421 (FPN laterals / 3x3), UpSampling2D+Add :1407-1415 (res_mode 2).
 *   x [N,H,W,Cin]; w PACKED [Cout][kh*kw*Cin] (cin fastest; dc_pack helper on the host side);
 *   y [N,Ho,Wo,Cout];  y = relu?( scale[c]*conv + shift[c] + residual )
 *   res_mode: 0 none, 1 residual [N,Ho,Wo,Cout], 2 residual [N,Ho/2,Wo/2,Cout] nearest-upsampled x2.
 * Cin must be a multiple of 32, except the stem: Cin==4 (RGBX), kh==kw==7, stride 2, with w packed
 * as [Cout][7][8][4] (kx==7 and c==3 entries zero).
 * ------------------------------------------------------------------------------------------------ */
#define DC_MATH_F32 0
#define DC_MATH_BF16X3 1
#define DC_MATH_BF16X2 2
#define DC_MATH_BF16 3
typedef struct {
    int N, H, W, Cin;
    int Cout, kh, kw, stride, pad_t, pad_l;
    int Ho, Wo;
    const float* x;
    const float* w;
    float*       y;
    const float* scale;
    const float* shift;
    const float* residual;  int res_mode;
    int relu;
    int split_k;
    int accumulate;           /* wgrad only: dw += (a weight shared by several inputs, e.g. the RPN over P2..P6) */
    int math;                 /* forward only: DC_MATH_F32 = fp32 MFMA (exact fp32 products), DC_MATH_BF16X3 = every operand
                                 element split into three bf16 pieces, six bf16 MFMA products, fp32 accumulate (fp32-grade
                                 accuracy on the bf16 matrix pipe; csrc/igemm_bf16s.h); DC_MATH_BF16X2 = two pieces, three
                                 products: a 16-bit-mantissa product (2^-16 relative; TF32 is 2^-11) at half the MFMAs;
                                 DC_MATH_BF16 = operands rounded once to bf16, one product, fp32 accumulate -- plain bf16 compute,
                                 the arithmetic BASELINE configs[4] names (2^-9 relative per operand) */
    const float* w_wino;      /* optional with DC_MATH_F32, forward only: the weights of a 3x3 / stride 1 / pad 1 layer transformed by
                                 dc_conv2d_winograd_pack_f32 (16 * Cin * Cout floats).  Non-NULL = run the layer in the Winograd
                                 F(2x2, 3x3) form: fp32 throughout, 16 products per 2x2 output tile instead of 36 (the minimal-filtering
                                 algorithm cuDNN picks for these layers under the reference's TF); needs Cin, Cout multiples of 32, no
                                 residual.  Layers that do not qualify ignore the field */
    const uint16_t* w_wino_b3; /* optional, like w_wino (and preferred when both are given): the same transformed weights split into three bf16
                                 pieces per element by dc_conv2d_winograd_pack_b3 (16 * Cin * Cout * 3 bf16).  The 16 products per tile then
                                 run on the BF16 matrix pipe in split arithmetic -- six bf16 MFMA products per fp32 product, fp32
                                 accumulation: DC_MATH_BF16X3's fp32-grade arithmetic -- transforms in fp32 as before */
} dc_conv_desc;

/* x = p0 + p1 + p2 with bf16 pieces rounded to nearest even: out[0..n) = p0, out[n..2n) = p1, out[2n..3n) = p2. */
int dc_split_bf16x3_f32(const float* x, uint16_t* out, size_t n, void* stream);
size_t dc_conv2d_workspace_bytes(const dc_conv_desc* d);
int    dc_conv2d_nhwc_f32(const dc_conv_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* Profiling aid: the block tile (bm x bn) and split-K factor dc_conv2d_nhwc_f32 picks for `d`, i.e.
 * which igemm_kernel<bm,bn,...> instantiation runs (bench.py maps layers to rocprof kernel names). */
int    dc_conv2d_tile_config(const dc_conv_desc* d, int* bm, int* bn, int* split_k);
/* Profiling aid: rocprof's spelling (without the "void dcap::" prefix and the parameter list) of the kernel template
 * instantiation that dc_conv2d_nhwc_f32 launches for `d`.  buf_bytes >= 96. */
int    dc_conv2d_kernel_name(const dc_conv_desc* d, char* buf, size_t buf_bytes);
/* 1 when `d` runs on the dense (pointwise: 1x1, stride 1, unpadded) A-operand loader instead of the im2col one. */
int    dc_conv2d_is_pointwise(const dc_conv_desc* d);
/* Winograd F(2x2, 3x3) weight transform U = G g G^T of a packed 3x3 kernel w [Cout][9*Cin] into the fragment order the kernel
 * reads (dc_conv_desc.w_wino); done once per frozen weight.  Replaces nothing in the reference: it is how the KL.Conv2D(3x3) layers of
 * feature_generation/dense_model.py:85-100, :1417-1421 are evaluated. */
size_t dc_conv2d_winograd_weight_bytes(int Cin, int Cout);
int    dc_conv2d_winograd_pack_f32(const float* w, float* u, int Cin, int Cout, void* stream);
size_t dc_conv2d_winograd_b3_weight_bytes(int Cin, int Cout);
int    dc_conv2d_winograd_pack_b3(const float* w, uint16_t* u, int Cin, int Cout, void* stream);
/* Two chained pointwise (1x1, stride 1) convolutions in one launch (csrc/conv_chain.hip, round 5): y = act1((x W1^T) * scale1 + shift1
 * [+ residual]) is written out AND contracted on the spot with the second kernel, z = act2((y W2^T) * scale2 + shift2) -- the last
 * convolution of a ResNet bottleneck (branch2c + shortcut + ReLU) and the first of the next block (branch2a + ReLU):
 * feature_generation/dense_model.py:85-100, :120-139, frozen BatchNorm folded into scale / shift like dc_conv2d_nhwc_f32.  fp32 operands,
 * exact fp32 MFMA products.  x [M][K1], y / residual [M][N1], z [M][N2] row-major (NHWC pixels = rows); w1 / w2: the packed kernels
 * [N][K] re-ordered ONCE by dc_pw_chain_pack_f32 (same size).  Shapes: dc_pw_chain_supported (K1 % 32 == 0; N1 -> N2 = 1024 -> 256 or
 * 512 -> 128: a block owns 32 pixels, the intermediate rows stay in LDS; or K1 -> N1 -> N2 = 64 -> 256 -> 64, the stage-2 seam: a
 * streaming kernel whose layer-1 accumulators are layer 2's MFMA operands, fp32 products only).  No workspace. */
typedef struct {
    int M, K1, N1, N2;
    const float* x;
    const float* w1;
    const float* scale1;      /* NULL = 1 */
    const float* shift1;
    const float* residual;    /* NULL = none */
    int relu1;
    float* y;
    const float* w2;
    const float* scale2;
    const float* shift2;
    int relu2;
    float* z;
    const uint16_t* w1_b3;    /* optional, BOTH or neither (then w1 / w2 may be NULL): the kernels split into three bf16 pieces per element by */
    const uint16_t* w2_b3;    /* dc_pw_chain_pack_b3 (3 * N * K bf16): both layers' products on the BF16 matrix pipe in split arithmetic (six
                                 bf16 MFMA products per fp32 product, fp32 accumulation: fp32-grade, DC_MATH_BF16X3's arithmetic) */
} dc_pw_chain_desc;
int dc_pw_chain_supported(int K1, int N1, int N2);
int dc_pw_chain_pack_f32(const float* w, float* out, int N, int K, void* stream);
int dc_pw_chain_pack_b3(const float* w, uint16_t* out, int N, int K, void* stream);
int dc_pw_chain_f32(const dc_pw_chain_desc* d, void* stream);
/* Profiling aid: rocprof's spelling of the kernel instantiation dc_pw_chain_f32 launches for `d` (buf_bytes >= 40). */
int dc_pw_chain_kernel_name(const dc_pw_chain_desc* d, char* buf, size_t buf_bytes);

/* CUs the persistent Winograd grids may occupy (process-wide; a multiple of 8 -- one share per XCD; 0 restores the default:
 * DCAP_WINO_CUS or all 256).  A persistent block holds its CU for the whole launch: in a data-parallel run (parallel_model.py:58-102
 * -> one rank per GPU here) the RCCL all-reduce of another queue needs CUs of its own to overlap the encoder pass, so
 * ParallelModel / bench.py set 248 when world > 1.  Applies to launches of at least eight rounds of work items per block (the
 * long-running ones: fpn_p2 at the benchmark's size); shorter launches keep the full grid, where a smaller one would add a whole
 * round.  Takes effect at the next launch; a captured hipGraph keeps the grid it was captured with. */
int    dc_set_persistent_cus(int n);
int    dc_get_persistent_cus(void);

/* ------------------------------------------------------------------------------------------------
 * conv2d weight gradient (for the layers the joint model trains: fpn_*, rpn_*, dense_img_cap/dense_model.py
 * train(layers="no_backbone") :1829-1831):  dw[cout][(ky,kx,ci)] = sum over output pixels of
 * dy[p][cout] * x[p*stride + (ky,kx) - pad][ci]   -- packed like the forward weights, so the optimizer
 * updates the packed tensor directly.  Uses the fields N,H,W,Cin,Cout,kh,kw,stride,pad_*,Ho,Wo and
 * x (activations), y (= dy, [N,Ho,Wo,Cout]), w (= dw OUT).  Cin % 64 == 0, Cout % 4 == 0, N*Ho*Wo % 32 == 0.
 * The data gradient needs no entry point of its own: for stride-1 convs it is dc_conv2d_nhwc_f32 on dy with
 * the kernel rotated by 180 degrees and cin/cout swapped (packing.pack_conv_kernel_dgrad).
 * ------------------------------------------------------------------------------------------------ */
size_t dc_conv2d_wgrad_workspace_bytes(const dc_conv_desc* d);
int    dc_conv2d_wgrad_f32(const dc_conv_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* KL.MaxPooling2D((3,3), strides 2, 'same') (dense_model.py:150); C % 4 == 0. */
int dc_maxpool3x3s2_same_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);

/* KL.MaxPooling2D((2,2), strides 2) of keras.applications VGG16 (image captioning/vgg16.py:9-18 loads that model; the
 * alternative-backbone benchmark config); C % 4 == 0, H and W even. */
int dc_maxpool2x2s2_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);

/* mold_image (dense_model.py:2050-2055) fused with the RGBX repack the stem kernel wants:
 * out[n,h,w,0..2] = float(img_u8) - mean[c], out[...,3] = 0. */
int dc_mold_image_rgbx_f32(const uint8_t* img, float* out, int N, int H, int W,
                           float mean_r, float mean_g, float mean_b, void* stream);
/* The same with the pixel zero-padded to `channels` floats (channels % 4 == 0): the first convolution of the VGG16 alternative
 * backbone reads 32-channel pixels (the implicit-GEMM loader wants Cin % 32 == 0). */
int dc_mold_image_padded_f32(const uint8_t* img, float* out, int N, int H, int W, int channels,
                             float mean_r, float mean_g, float mean_b, void* stream);

/* ------------------------------------------------------------------------------------------------
 * PyramidROIAlign forward (feature_generation/dense_model.py:317-418): level routing
 * k = clamp(4 + round_half_even(log2(sqrt(h*w) / (224/sqrt(image_area)))), 2, 5) and
 * tf.image.crop_and_resize(bilinear, 1 sample per bin, extrapolation 0) in one gather kernel.
 *   maps[l] = P(2+l) [B,Hl,Wl,C] (C % 4 == 0, C <= 256*4); boxes [B*R,4] normalised (y1,x1,y2,x2), batch-major;
 *   out [B*R,pool,pool,C] in box order (the reference's final reorder, :397-411, is the identity
 *   here because nothing is regrouped by level).  levels_out (optional) [B*R] int32.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int B, R, C, pool;
    const float* maps[4];
    int Hs[4], Ws[4];
    const float* boxes;
    float image_area;
    float* out;
    int32_t* levels_out;
} dc_roialign_desc;

int dc_roi_align_pyramid_f32(const dc_roialign_desc* d, void* stream);

/* KL.MaxPooling2D(pool_size=(1,1), strides=2): P6 = every other pixel of P5 (dense_model.py:1423). */
int dc_subsample2_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RPN scores + ProposalLayer (feature_generation/dense_model.py:684-725, :221-305): per image
 *   fg score = softmax(class logits)[1] per anchor, deltas *= RPN_BBOX_STD_DEV,
 *   tf.nn.top_k(scores, pre_nms_limit) (stable: lower anchor index first among ties),
 *   apply_box_deltas -> clip to the image -> normalise by [h,w,h,w] (float32, TF's operation order),
 *   tf.image.non_max_suppression(threshold) -> first `proposal_count` survivors, zero padded.
 * heads[l] = the fused RPN head output of pyramid level l, NHWC [B,Hl,Wl,A*6]: channels a*2+{bg,fg}
 * (rpn_class_raw) followed by A*2 + a*4 + {dy,dx,dh,dw} (rpn_bbox_pred); anchors [A_total,4] pixels in
 * generate_pyramid_anchors order (level, y, x, ratio).
 * Optional outputs for tests: scores_out [B,A_total], order_out [B,pre_nms_limit] (anchor index of each
 * sorted candidate), keep_out [B,proposal_count] (candidate rank of each kept box, -1 padded).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int B, levels, anchors_per_loc;
    const float* heads[5];
    int Hs[5], Ws[5];
    int head_stride;          /* floats per cell in heads[] (0 = anchors_per_loc*6; > that when the head is padded) */
    const float* anchors;
    int A_total;
    float std_dev[4];
    float image_h, image_w;
    int pre_nms_limit, proposal_count;
    float nms_threshold;
    float* proposals;
    float* scores_out;
    int32_t* order_out;
    int32_t* keep_out;
} dc_proposal_desc;

size_t dc_proposals_workspace_bytes(const dc_proposal_desc* d);
int    dc_proposals_f32(const dc_proposal_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* PyramidROIAlign backward: d(maps) += bilinear scatter of d(out) (gradients to the boxes are stopped in the
 * reference, dense_model.py:378-379).  dmaps[l] must be zero-initialised by the caller; accumulation uses
 * float atomics (order-dependent in the last bits).  Same descriptor as the forward; `out` holds d(out) and
 * `maps` the gradient maps (written). */
int dc_roi_align_pyramid_bwd_f32(const dc_roialign_desc* d, void* stream);

/* FPN top-down backward: out[n,y,x,c] (+)= sum of the 2x2 block of fine[n,2y..2y+1,2x..2x+1,c]
 * (the gradient of UpSampling2D(2) + Add, dense_model.py:1407-1415). */
int dc_downsample2x_sum_f32(const float* fine, float* out, int N, int Ho, int Wo, int C, int accumulate, void* stream);
/* ... and, in the same pass, the bf16 copy of `out` that the bf16 weight-gradient GEMM of the lateral convolution reads (NULL: none). */
int dc_downsample2x_sum_dual_f32(const float* fine, float* out, uint16_t* out_bf16, int N, int Ho, int Wo, int C, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Keras-2.1 LSTM over a whole sequence (gate blocks i,f,c,o; hard-sigmoid gates; tanh; mask carry).
 * Replaces KL.LSTM: text_generation_model.py:141-142,150-151; _v2.py:157,163.
 * Time-major: row (t*B + b).  The caller first computes zx = x*kernel + bias with dc_gemm_f32
 * (optionally with the fused embedding gather); this call runs the recurrence:
 *   z_t = zx_t + h_{t-1}*U ; i,f,o = hs(z) ; g = tanh(z_c) ; c' = f*c + i*g ; h' = o*tanh(c')
 *   masked rows (mask[t*B+b]==0) carry h,c from t-1 (zeros before the first unmasked step).
 * zx is updated IN PLACE to the full pre-activation z (saved for backward).
 *   z [T*B][4U] in/out, U_rec [U][4U], mask [T*B] uint8 or NULL, h_seq/c_seq [T*B][U] out.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int B, T, U;
    float* z;
    const float* U_rec;
    const uint8_t* mask;
    float* h_seq;
    float* c_seq;
    const float* rec_masks;   /* optional [4][B][U]: Keras recurrent_dropout masks (values 0 or 1/(1-rate)), one per gate i,f,c,o,
                                 fixed over the timesteps: z_g = x W_g + (h_{t-1} * m_g) U_g + b_g  (training phase of
                                 LSTM(recurrent_dropout=0.2), text_generation_model.py:141-142).  NULL = no dropout. */
} dc_lstm_fwd_desc;

size_t dc_lstm_seq_workspace_bytes(int B, int T, int U);
int    dc_lstm_seq_fwd_f32(const dc_lstm_fwd_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the recurrence.  dh_seq [T*B][U] (may be NULL) is the gradient w.r.t. every step's
 * output, dh_last [B][U] (may be NULL) an extra gradient on the last step's output.
 * Outputs: dz [T*B][4U] (caller derives dkernel = x^T dz, dbias = colsum dz, dx = dz kernel^T with
 * dc_gemm_f32 / dc_colsum_f32) and dU_rec [U][4U] (written, or accumulated if accumulate_dU).  dU_rec may be NULL: the
 * caller then forms it itself, dU_rec = h_seq[0 .. (T-1)B)^T dz[B .. TB) (with rec_masks: per gate g from h_seq * m_g) -- the
 * bf16 joint model does, on the bf16 matrix pipe from the bf16 copies it already holds. */
typedef struct {
    int B, T, U;
    const float* z;
    const float* U_rec;
    const uint8_t* mask;
    const float* h_seq;
    const float* c_seq;
    const float* dh_seq;
    const float* dh_last;
    float* dz;
    float* dU_rec;
    int accumulate_dU;
    const float* rec_masks;   /* the masks of the forward call (or NULL) */
} dc_lstm_bwd_desc;

int dc_lstm_seq_bwd_f32(const dc_lstm_bwd_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * softmax + K.categorical_crossentropy (clip 1e-7) forward and d/dlogits in one pass per row.
 * Replaces Dense(..., activation='softmax') + keras.losses.categorical_crossentropy /
 * roi_caption_loss: text_generation_model.py:154,286-294; _v2.py:164,267.
 *   logits [M][ld] (V valid columns); targets [M] int32 (the argmax of the reference's one-hot rows);
 *   probs (optional) [M][ld]; loss_rows (optional) [M]; dlogits (optional, may alias logits) [M][ld]:
 *   dlogits = grad_scale * (p - onehot) for rows whose target probability is inside [1e-7, 1-1e-7], else 0.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int M, V, ld;
    const float* logits;
    const int32_t* targets;
    float* probs;
    float* loss_rows;
    float* dlogits;
    float grad_scale;
    const float* row_weights; /* optional [M]: loss_rows and dlogits rows are multiplied by it (0 drops a row) */
    int keras_sparse;         /* 1: K.sparse_categorical_crossentropy on probabilities (dense_img_cap/dense_model.py:943-945):
                                 q = clip(p,1e-7,1-1e-7); loss = -log q_t + log sum(q), gradient through clip and sum */
} dc_softmax_ce_desc;

int dc_softmax_ce_f32(const dc_softmax_ce_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Vocabulary projection FUSED with softmax + cross-entropy: logits = X[M,K] * W[K,V] + bias are reduced tile by tile inside
 * the GEMM and never written.  Replaces Dense(V, activation='softmax') + the losses above in training:
 *   text_generation_model.py:153-154,286-294; text_generation_model_v2.py:164,267; dense_img_cap/dense_model.py:787-817,936-946.
 *   bf16 = 0: X, W float32 (fp32 MFMA), K % 32 == 0, V % 4 == 0;  bf16 = 1: X, W bf16 bit patterns, K % 8 == 0, V % 8 == 0.
 *   loss_rows [M] (optional), dlogits [M][lddl] (optional; float32, or bf16 when dl_bf16 -- then columns V..lddl-1 are written
 *   as zeros so the matrix can feed dc_gemm_bf16 as an operand), dbias [V] (optional, needs dlogits): column sums of dlogits,
 *   combined in a fixed order.  Semantics of loss / gradient / row_weights / keras_sparse exactly as dc_softmax_ce_f32.
 * Cost model: one GEMM pass for the loss plus one for the gradient, against one pass plus four sweeps over a [M,V] float32 matrix
 * unfused; keras_sparse adds a pass that sums the clipped probabilities -- only on the row tiles that hold a probability outside
 * [1e-7, 1 - 1e-7] (the first pass keeps every row's smallest logit; elsewhere nothing is clipped and the sums are 1).
 * materialize_bf16 (round 6, configs[4]'s bf16 arithmetic): one GEMM pass + one in-place elementwise pass over the bf16 [M,V] buffer.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int M, V, K;
    int bf16;
    const void* X;  int ldx;
    const void* W;  int ldw;
    const float* bias;
    const int32_t* targets;
    const float* row_weights;
    float grad_scale;
    int keras_sparse;
    float* loss_rows;
    void* dlogits;  int lddl;  int dl_bf16;
    float* dbias;
    int materialize_bf16;     /* bf16 = dl_bf16 = 1 with dlogits, large problems (the 256-square tile): the first GEMM pass ROUNDS the logits to
                                 bf16 and parks them in the dlogits buffer; statistics, loss and gradient are those of the rounded logits
                                 (what a bf16 framework that materialises its logits computes); the clip sums and the gradient are then
                                 elementwise passes over that buffer, in place -- ONE GEMM pass instead of two or three.  0 (default) and
                                 every other shape: fp32 logits recomputed per pass, never stored. */
} dc_vocab_ce_desc;

size_t dc_vocab_ce_workspace_bytes(const dc_vocab_ce_desc* d);
int    dc_vocab_ce(const dc_vocab_ce_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* tf.argmax over the last axis, lowest index wins ties (text_generation_model.py:222-225). */
int dc_argmax_rows_f32(const float* x, int M, int V, int ld, int32_t* out, void* stream);

/* Row gather (data movement only): out[n][0:width] = idx[n] >= 0 ? src[idx[n]][0:width] : 0.
 * Builds the Concatenate() operands of the decoders (text_generation_model.py:147-152; _v2.py:161)
 * from per-RoI / per-timestep rows, and routes their gradients back (inverse index). width % 4 == 0. */
int dc_gather_rows_f32(const float* src, int ld_src, const int32_t* idx, float* out, int ld_out,
                       int n_rows, int width, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Trainable RoI head (v1 / joint model: TimeDistributed Conv2D + BatchNorm(training=False) + ReLU with
 * trainable kernel, bias, gamma, beta; text_generation_model.py:251-262).  The conv itself is a
 * dc_gemm_f32 without epilogue (acc = x*kernel); these two kernels apply / differentiate
 *   n = (acc + bias - mean) / sqrt(var + eps),  y = relu(gamma*n + beta)        (eps = 1e-3)
 * fwd: y [M][ld] from acc [M][ld].
 * bwd: given dy (gradient w.r.t. y) and the saved acc: dacc [M][ld] (gradient w.r.t. acc, also the
 *      conv-bias gradient's summand) and the column reductions dgamma, dbeta, dbias [N].
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int M, N, ld;
    const float* acc;
    const float* bias; const float* gamma; const float* beta; const float* mean; const float* var;
    float eps;
    float* y;              /* fwd out */
    const float* dy;       /* bwd in  */
    float* dacc;           /* bwd out */
    float* dgamma; float* dbeta; float* dbias;   /* bwd out [N] */
} dc_bn_relu_desc;

int dc_bn_relu_fwd_f32(const dc_bn_relu_desc* d, void* stream);
int dc_bn_relu_bwd_f32(const dc_bn_relu_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Joint-model helpers (dense_img_cap/dense_model.py).
 * dc_conv_weight_dgrad_pack: packed forward weights [Cout][kh*kw*Cin] -> packed data-gradient weights
 *   [Cin][kh*kw*Cout] (taps rotated 180 degrees), re-derived on the device after every optimizer step.
 * dc_rpn_loss_grad: rpn_class_loss_graph + rpn_bbox_loss_graph (:877-933) for ONE image: `sel` lists the n_sel
 *   non-neutral anchors as (level, cell*A + a) pairs with their match (+1/-1) in anchor order; target_deltas
 *   [n_pos][4] in the order of the positive ones.  Writes d(loss)/d(head) into the (pre-zeroed) padded head
 *   gradients dheads[l] [H,W,head_stride] of this image and losses[0..1] = (class, bbox).
 * dc_scatter2_add: fine[n,2y,2x,c] += coarse[n,y,x,c]  (backward of P6 = MaxPooling2D(1, strides 2)(P5)).
 * dc_axpy: y += a*x  (L2 regulariser gradient, compile() :1715-1718).
 * ------------------------------------------------------------------------------------------------ */
int dc_conv_weight_dgrad_pack_f32(const float* w, float* out, int Cout, int kh, int kw, int Cin, void* stream);
typedef struct {
    int levels, anchors_per_loc, head_stride;
    const float* heads[5];
    float*       dheads[5];
    int Hs[5], Ws[5];
    int n_sel, n_pos;
    const int32_t* sel_level;     /* [n_sel] */
    const int32_t* sel_index;     /* [n_sel] cell*A + a within the level */
    const int32_t* sel_match;     /* [n_sel] +1 / -1 */
    const float*   target_deltas; /* [n_pos][4] */
    float* losses;                /* [2] */
    const int32_t* counts_dev;    /* optional: device words {n_sel, n_pos} that override the two fields above (which then give the
                                     CAPACITY of sel_* / target_deltas): the launch is the same whatever the image's counts are, so a
                                     captured hipGraph can replay it */
} dc_rpn_loss_desc;
int dc_rpn_loss_grad_f32(const dc_rpn_loss_desc* d, void* stream);
int dc_scatter2_add_f32(const float* coarse, float* fine, int N, int Hc, int Wc, int C, void* stream);
/* DetectionTargetLayer for one image on the device (dense_img_cap/dense_model.py:450-572 detection_targets_graph; :421-447
 * overlaps_graph in float32).  proposals [n_proposals][4] / gt_boxes [n_gt][4]: (y1,x1,y2,x2) normalised, all-zero rows are padding;
 * gt_captions [n_gt][T] token ids.  Positives: best IoU >= 0.5 (at most max_positive = int(TRAIN_ROIS_PER_IMAGE * ROI_POSITIVE_RATIO)),
 * negatives: best IoU < 0.5, int32(inv_ratio * float32(n_pos)) - n_pos of them (inv_ratio = float32(1 / ROI_POSITIVE_RATIO)).
 * tf.random_shuffle of the two index lists = ascending order of Philox-2x32-10(counter = (position of the proposal among the non-zero
 * ones, offset + *offset_dev), key = seed) with the position breaking ties (shuffle = 0: proposal order); the reference's shuffle is
 * non-deterministic, this one is reproducible from (seed, offset).  offset_dev (optional): a device word added to `offset`, so that a
 * captured hipGraph draws a fresh sample every replay.
 * Writes rois [n_rois][4] (positives, then negatives, zero padded), captions [n_rois][T] (the best GT box's caption for positives,
 * zeros otherwise) and counts = {n_pos, n_neg}.  Limits: n_proposals <= 4096, n_gt <= 512.  ceil(n_proposals / 256) workgroups, each
 * building the same class / key table in its LDS and ranking its own 256 proposals (outputs are disjoint rows); one image per call (a batch:
 * one call per image, dense_img_cap/utils.py batch_slice); no workspace. */
typedef struct {
    int n_proposals, n_gt, n_rois, T;
    const float*   proposals;
    const float*   gt_boxes;
    const int32_t* gt_captions;
    int   max_positive;
    float inv_ratio;
    int   shuffle;
    uint32_t seed, offset;
    const uint32_t* offset_dev;
    float*   rois;
    int32_t* captions;
    int32_t* counts;          /* [2] */
} dc_detection_targets_desc;
int dc_detection_targets_f32(const dc_detection_targets_desc* d, void* stream);
/* Index tables of the Model-3 decoder from device-resident captions [B][T] (dense_img_cap/dense_model.py:1572-1580: target = caption
 * shifted left by one; imgcap_caption_loss_graph :936-946: mean over positions with target > 0): ids_tm / targets_tm [T*B] time-major
 * (row t*B + b), mask = ids != 0, row_weights = [target > 0] / max(count, 1), live_count[0] = count (may be NULL). */
int dc_caption_tables_i32(const int32_t* captions, int B, int T, int32_t* ids_tm, uint8_t* mask, int32_t* targets_tm, float* row_weights,
                          int32_t* live_count, void* stream);
/* keras.regularizers.l2(WEIGHT_DECAY)(w) / size(w) summed over the trainable non-BN weights
 * (dense_img_cap/dense_model.py:1712-1718) over one flat bucket: coef[i] = WEIGHT_DECAY/size of i's tensor (0 where not
 * regularised).  grad[i] = grad[i]*mask[i] + 2*coef[i]*w[i] (grad may be NULL; mask = the 0/1 subset set_trainable() left
 * trainable, :1732-1774, NULL = all of it); loss[0] = sum coef[i]*w[i]^2 (loss may be NULL), summed in a fixed order through
 * `workspace` (dc_l2_reg_workspace_bytes; only needed with a loss): bit-reproducible. */
size_t dc_l2_reg_workspace_bytes(size_t n);
int dc_l2_reg_f32(const float* w, const float* coef, const float* mask, float* grad, size_t n, float* loss, void* workspace,
                  size_t workspace_bytes, void* stream);

int dc_axpy_f32(float a, const float* x, float* y, size_t n, void* stream);

/* out = (y > 0) ? dy : 0 over [M][N] (row strides ld): backward of Activation('relu') given its output. */
int dc_relu_bwd_f32(const float* dy, const float* y, float* out, int M, int N, int ld, void* stream);
/* The same over n contiguous elements (n % 4 == 0) with 16-byte accesses, writing the fp32 result and / or its bf16 copy (either may be
 * NULL): the RPN branch's d(shared) feeds bf16 convolutions, and a separate cast pass would read the fp32 result once more. */
int dc_relu_bwd_dual_f32(const float* dy, const float* y, float* out, uint16_t* out_bf16, size_t n, void* stream);

/* out[b][n] = sum_t x[t*B + b][n]  (time-major fold: the per-RoI gradient of a term that was broadcast
 * over timesteps -- RepeatVector(feature), text_generation_model.py:146). */
int dc_fold_time_f32(const float* x, int T, int B, int N, int ld, float* out, int ld_out, void* stream);

/* out[n] (+)= sum_m x[m][n]  -- bias gradients (K.sum over batch/time inside the Dense/Conv backward of Keras).
 * Tall matrices are summed in row chunks through `workspace` (dc_colsum_workspace_bytes) and combined in a fixed
 * order: bit-reproducible.  Without workspace the kernel falls back to one chunk. */
size_t dc_colsum_workspace_bytes(int M, int N, int ld);
int dc_colsum_f32(const float* x, int M, int N, int ld, float* out, int accumulate, void* workspace, size_t workspace_bytes,
                  void* stream);

/* out[0] (+)= sum(x^2) -- global-norm clipping (Adam(clipnorm=0.5), dense_img_cap/dense_model.py:1699).  Block partials go
 * through `workspace` (dc_sumsq_workspace_bytes) and are combined in a fixed order: every data-parallel rank derives the
 * same clip scale from the same all-reduced gradient, and a run is bit-reproducible. */
size_t dc_sumsq_workspace_bytes(size_t n);
int dc_sumsq_f32(const float* x, size_t n, float* out, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* Inverted-dropout mask of ones: out[i] = 1/(1-rate) with probability 1-rate, else 0 -- K.dropout(K.ones_like(h), rate), the
 * per-gate recurrent_dropout masks Keras' LSTMCell draws in the training phase (recurrent.py _generate_recurrent_dropout_mask;
 * used by text_generation_model.py:141-142 and dense_img_cap/dense_model.py:769-770 with rate 0.2).  Counter-based
 * (Philox-2x32-10): element i of stream (seed, offset) is a pure function of (i, seed, offset).  offset_dev (optional): a device
 * word added to `offset` (a captured hipGraph then draws fresh masks every replay: the host bumps the word between replays). */
int dc_dropout_mask_f32(float* out, size_t n, float rate, uint32_t seed, uint32_t offset, const uint32_t* offset_dev, void* stream);

/* Training ResNet stages (dense_img_cap/dense_model.py:1829-1845: layers "3+" | "4+" | "5+" | "all"; BatchNorm with frozen
 * statistics :51-61, trainable gamma / beta and convolution).
 *   dc_bn_fold_f32: scale[c] = gamma[c] / sqrt(var[c] + eps), shift[c] = beta[c] + (bias[c] - mean[c]) * scale[c] -- the fused
 *     epilogue operands of the forward convolution, refreshed from the trained parameters before every pass.
 *   dc_bn_bwd_f32: given dz [rows][channels] = d(loss)/d(BN output) and the BN output itself as a - b (b may be NULL):
 *     dacc = dz * scale (gradient w.r.t. the convolution output, also the summand of the conv-bias gradient) and
 *     dzn = dz * (a - b - beta) / gamma (0 where dz == 0); column sums of dzn / dz are dgamma / dbeta, dbias = scale * dbeta.
 *   dc_mul_f32: out = a * b elementwise. */
int dc_bn_fold_f32(const float* gamma, const float* beta, const float* bias, const float* mean, const float* var, float eps,
                   float* scale, float* shift, int n, void* stream);
int dc_bn_bwd_f32(const float* dz, const float* a, const float* b, const float* gamma, const float* beta, const float* scale,
                  float* dacc, float* dzn, long rows, int channels, void* stream);
int dc_mul_f32(const float* a, const float* b, float* out, size_t n, void* stream);
/* Backward of dc_maxpool3x3s2_same_f32 (the stem's pool, trained with layers = "all"): dx[pixel] = sum of dy over the windows whose
 * first maximum (row-major, as TF's MaxPoolGrad) is this pixel.  x [N,H,W,C], y / dy [N,ceil(H/2),ceil(W/2),C]. */
int dc_maxpool3x3s2_same_bwd_f32(const float* x, const float* y, const float* dy, float* dx, int N, int H, int W, int C, void* stream);

/* bytes of zeros at p (pointer and size multiples of 4): a kernel, not a memset node -- a captured hipMemsetAsync node faulted on its
 * second replay on this runtime (round 4), so the library and its callers zero buffers with this. */
int dc_zero_fill(void* p, size_t bytes, void* stream);

/* mean of loss rows: out[0] = sum(x)/n. */
int dc_mean_f32(const float* x, size_t n, float* out, void* stream);

/* Piecewise-constant per-element coefficients over a flat parameter bucket: segment s covers elements [start[s], start[s+1]) and
 * carries coef[s] (WEIGHT_DECAY / size(tensor), 0 for BatchNorm gamma / beta and padding) and mask[s] (1 = trainable, 0 = frozen;
 * mask NULL = everything trains).  The arrays live in DEVICE memory; start has nseg + 1 ascending entries, start[0] = 0, start[nseg] = n.
 * Replaces the two n-element float vectors dc_l2_reg_f32 reads (2 x 300 MB per step for the joint model's 77 M parameters). */
typedef struct {
    const int32_t* start;
    const float* coef;
    const float* mask;
    int nseg;
} dc_reg_segments;

/* ------------------------------------------------------------------------------------------------
 * keras.optimizers.Adam(amsgrad=True) fused over one flat parameter bucket
 * (text_generation_model.py:425; _v2.py:266): with g' = g * grad_scale * clip,
 *   m = b1*m + (1-b1)*g' ; v = b2*v + (1-b2)*g'^2 ; vhat = max(vhat, v) ; p -= lr_t*m/(sqrt(vhat)+eps)
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the caller.  If gnorm_sq (device scalar: sum of
 * squares of the UNSCALED gradients) is non-NULL and clipnorm > 0, clip = clipnorm/norm when
 * norm >= clipnorm (norm taken after grad_scale), else 1.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    size_t n;
    float* p;
    const float* g;
    float* m;
    float* v;
    float* vhat;
    float lr_t, beta1, beta2, eps;
    float grad_scale;
    const float* gnorm_sq;
    float clipnorm;
    uint16_t* p_bf16;         /* optional: bf16 shadow of p (the operand copy the bf16 GEMMs read), refreshed in the same pass */
    size_t n_bf16;            /* the shadow covers p[0 .. n_bf16) (a multiple of 4) */
    const float* lr_t_dev;    /* optional: a device word that overrides lr_t (Keras' lr_t depends on the iteration count; a captured
                                 hipGraph reads this step's value from memory the host refreshed before the replay) */
    const dc_reg_segments* reg; /* optional (HOST pointer, read during the call): the joint model's regulariser and trainable mask applied
                                 INSIDE this pass -- the gradient the update sees is g * mask + 2 * coef * p with per-segment constants --
                                 instead of by a dc_l2_reg_f32 pass that rewrites the gradient bucket first; gnorm_sq must then come from
                                 dc_reg_sumsq_f32 (the norm of that same regularised gradient).  The gradient bucket is left as it was */
} dc_amsgrad_desc;

int dc_amsgrad_step_f32(const dc_amsgrad_desc* d, void* stream);

/* One READ-ONLY pass over (w, g) in front of the fused update above (round 5: replaces dc_l2_reg_f32 + dc_sumsq_f32 = three reads and
 * one write of the bucket + one more read, by two reads): loss[0] = sum coef * w^2 (the L2 term of dense_img_cap/dense_model.py:1715-1718),
 * gnorm_sq[0] = sum (g * mask + 2 * coef * w)^2 (what Adam(clipnorm=0.5), :1699, clips by), both as block partials in block order
 * through `workspace`, added by one block in a fixed tree (bit-reproducible).  Either output may be NULL. */
size_t dc_reg_sumsq_workspace_bytes(size_t n);
int dc_reg_sumsq_f32(const float* w, const float* g, const dc_reg_segments* reg, size_t n, float* loss, float* gnorm_sq, void* workspace,
                     size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCAP_H */
