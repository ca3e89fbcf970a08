"""ctypes binding of libdcap_hip.so (the C-ABI in include/dcap.h).

There is NO CPU fallback: if the library is missing, load() raises.  build it with
`python __graft_entry__.py` (hipcc --offload-arch=gfx950).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdcap_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class GemmDesc(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("A", C.c_void_p), ("lda", C.c_int), ("a_trans", C.c_int),
                ("a_gather", C.c_void_p),
                ("B", C.c_void_p), ("ldb", C.c_int), ("b_trans", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int),
                ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("residual", C.c_void_p), ("ldr", C.c_int), ("res_rows", C.c_int),
                ("relu", C.c_int), ("accumulate", C.c_int), ("split_k", C.c_int)]


class GemmBf16Desc(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("A", C.c_void_p), ("lda", C.c_int), ("a_trans", C.c_int),
                ("a_gather", C.c_void_p), ("a_gather_rows", C.c_int),
                ("B", C.c_void_p), ("ldb", C.c_int), ("b_trans", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int),
                ("Cb", C.c_void_p), ("ldcb", C.c_int),
                ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("residual", C.c_void_p), ("ldr", C.c_int), ("res_rows", C.c_int),
                ("relu", C.c_int), ("accumulate", C.c_int), ("split_k", C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int),
                ("Cout", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int),
                ("pad_t", C.c_int), ("pad_l", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
                ("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("residual", C.c_void_p), ("res_mode", C.c_int),
                ("relu", C.c_int), ("split_k", C.c_int), ("accumulate", C.c_int), ("math", C.c_int), ("w_wino", C.c_void_p), ("w_wino_b3", C.c_void_p)]


class ConvWgradBf16Desc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int),
                ("Cout", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int),
                ("pad_t", C.c_int), ("pad_l", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
                ("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p), ("accumulate", C.c_int), ("split_k", C.c_int)]


class ConvBf16Desc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int),
                ("Cout", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int),
                ("pad_t", C.c_int), ("pad_l", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
                ("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p), ("y_bf16", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("residual", C.c_void_p), ("res_mode", C.c_int),
                ("relu", C.c_int), ("split_k", C.c_int), ("tile", C.c_int)]


MATH_F32, MATH_BF16X3, MATH_BF16X2, MATH_BF16 = 0, 1, 2, 3


class RoiAlignDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("R", C.c_int), ("C", C.c_int), ("pool", C.c_int),
                ("maps", C.c_void_p * 4), ("Hs", C.c_int * 4), ("Ws", C.c_int * 4),
                ("boxes", C.c_void_p), ("image_area", C.c_float),
                ("out", C.c_void_p), ("levels_out", C.c_void_p)]


class LstmFwdDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("U", C.c_int),
                ("z", C.c_void_p), ("U_rec", C.c_void_p), ("mask", C.c_void_p),
                ("h_seq", C.c_void_p), ("c_seq", C.c_void_p), ("rec_masks", C.c_void_p)]


class LstmBwdDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("U", C.c_int),
                ("z", C.c_void_p), ("U_rec", C.c_void_p), ("mask", C.c_void_p),
                ("h_seq", C.c_void_p), ("c_seq", C.c_void_p),
                ("dh_seq", C.c_void_p), ("dh_last", C.c_void_p),
                ("dz", C.c_void_p), ("dU_rec", C.c_void_p), ("accumulate_dU", C.c_int), ("rec_masks", C.c_void_p)]


class SoftmaxCeDesc(C.Structure):
    _fields_ = [("M", C.c_int), ("V", C.c_int), ("ld", C.c_int),
                ("logits", C.c_void_p), ("targets", C.c_void_p),
                ("probs", C.c_void_p), ("loss_rows", C.c_void_p), ("dlogits", C.c_void_p),
                ("grad_scale", C.c_float), ("row_weights", C.c_void_p), ("keras_sparse", C.c_int)]


class VocabCeDesc(C.Structure):
    _fields_ = [("M", C.c_int), ("V", C.c_int), ("K", C.c_int), ("bf16", C.c_int),
                ("X", C.c_void_p), ("ldx", C.c_int), ("W", C.c_void_p), ("ldw", C.c_int),
                ("bias", C.c_void_p), ("targets", C.c_void_p), ("row_weights", C.c_void_p),
                ("grad_scale", C.c_float), ("keras_sparse", C.c_int), ("loss_rows", C.c_void_p),
                ("dlogits", C.c_void_p), ("lddl", C.c_int), ("dl_bf16", C.c_int), ("dbias", C.c_void_p), ("materialize_bf16", C.c_int)]


class BnReluDesc(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("ld", C.c_int), ("acc", C.c_void_p),
                ("bias", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p), ("var", C.c_void_p),
                ("eps", C.c_float), ("y", C.c_void_p), ("dy", C.c_void_p), ("dacc", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dbias", C.c_void_p)]


class ProposalDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("levels", C.c_int), ("anchors_per_loc", C.c_int),
                ("heads", C.c_void_p * 5), ("Hs", C.c_int * 5), ("Ws", C.c_int * 5), ("head_stride", C.c_int),
                ("anchors", C.c_void_p), ("A_total", C.c_int), ("std_dev", C.c_float * 4),
                ("image_h", C.c_float), ("image_w", C.c_float),
                ("pre_nms_limit", C.c_int), ("proposal_count", C.c_int), ("nms_threshold", C.c_float),
                ("proposals", C.c_void_p), ("scores_out", C.c_void_p), ("order_out", C.c_void_p), ("keep_out", C.c_void_p)]


class RpnLossDesc(C.Structure):
    _fields_ = [("levels", C.c_int), ("anchors_per_loc", C.c_int), ("head_stride", C.c_int),
                ("heads", C.c_void_p * 5), ("dheads", C.c_void_p * 5), ("Hs", C.c_int * 5), ("Ws", C.c_int * 5),
                ("n_sel", C.c_int), ("n_pos", C.c_int),
                ("sel_level", C.c_void_p), ("sel_index", C.c_void_p), ("sel_match", C.c_void_p),
                ("target_deltas", C.c_void_p), ("losses", C.c_void_p), ("counts_dev", C.c_void_p)]


class DetectionTargetsDesc(C.Structure):
    _fields_ = [("n_proposals", C.c_int), ("n_gt", C.c_int), ("n_rois", C.c_int), ("T", C.c_int),
                ("proposals", C.c_void_p), ("gt_boxes", C.c_void_p), ("gt_captions", C.c_void_p),
                ("max_positive", C.c_int), ("inv_ratio", C.c_float), ("shuffle", C.c_int),
                ("seed", C.c_uint32), ("offset", C.c_uint32), ("offset_dev", C.c_void_p),
                ("rois", C.c_void_p), ("captions", C.c_void_p), ("counts", C.c_void_p)]


class PwChainDesc(C.Structure):
    _fields_ = [("M", C.c_int), ("K1", C.c_int), ("N1", C.c_int), ("N2", C.c_int),
                ("x", C.c_void_p), ("w1", C.c_void_p), ("scale1", C.c_void_p), ("shift1", C.c_void_p), ("residual", C.c_void_p),
                ("relu1", C.c_int), ("y", C.c_void_p), ("w2", C.c_void_p), ("scale2", C.c_void_p), ("shift2", C.c_void_p),
                ("relu2", C.c_int), ("z", C.c_void_p), ("w1_b3", C.c_void_p), ("w2_b3", C.c_void_p)]


class RegSegments(C.Structure):
    _fields_ = [("start", C.c_void_p), ("coef", C.c_void_p), ("mask", C.c_void_p), ("nseg", C.c_int)]


class AmsgradDesc(C.Structure):
    _fields_ = [("n", C.c_size_t), ("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p),
                ("v", C.c_void_p), ("vhat", C.c_void_p),
                ("lr_t", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("grad_scale", C.c_float), ("gnorm_sq", C.c_void_p), ("clipnorm", C.c_float),
                ("p_bf16", C.c_void_p), ("n_bf16", C.c_size_t), ("lr_t_dev", C.c_void_p), ("reg", C.POINTER(RegSegments))]


# name -> (restype, argtypes): every symbol include/dcap.h declares
SYMBOLS = {
    "dc_version": (C.c_int, []),
    "dc_last_error": (C.c_char_p, []),
    "dc_gemm_workspace_bytes": (C.c_size_t, [C.POINTER(GemmDesc)]),
    "dc_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_gemm_bf16_workspace_bytes": (C.c_size_t, [C.POINTER(GemmBf16Desc)]),
    "dc_gemm_bf16_tile": (C.c_int, [C.POINTER(GemmBf16Desc), C.POINTER(C.c_int)]),
    "dc_gemm_bf16": (C.c_int, [C.POINTER(GemmBf16Desc), C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_conv2d_wgrad_bf16_workspace_bytes": (C.c_size_t, [C.POINTER(ConvWgradBf16Desc)]),
    "dc_conv2d_wgrad_bf16_tile": (C.c_int, [C.POINTER(ConvWgradBf16Desc), C.POINTER(C.c_int)]),
    "dc_conv2d_wgrad_bf16": (C.c_int, [C.POINTER(ConvWgradBf16Desc), C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_conv2d_bf16_workspace_bytes": (C.c_size_t, [C.POINTER(ConvBf16Desc)]),
    "dc_conv2d_bf16_tile": (C.c_int, [C.POINTER(ConvBf16Desc), C.POINTER(C.c_int)]),
    "dc_conv2d_bf16": (C.c_int, [C.POINTER(ConvBf16Desc), C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_cast_f32_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_cast_bf16_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_cast_f32_bf16_2d": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dc_split_bf16x3_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_conv2d_workspace_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "dc_conv2d_nhwc_f32": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_conv2d_tile_config": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dc_conv2d_is_pointwise": (C.c_int, [C.POINTER(ConvDesc)]),
    "dc_conv2d_winograd_weight_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "dc_conv2d_winograd_pack_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dc_pw_chain_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "dc_pw_chain_pack_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dc_pw_chain_pack_b3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dc_pw_chain_f32": (C.c_int, [C.POINTER(PwChainDesc), C.c_void_p]),
    "dc_pw_chain_kernel_name": (C.c_int, [C.POINTER(PwChainDesc), C.c_char_p, C.c_size_t]),
    "dc_conv2d_winograd_b3_weight_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "dc_conv2d_winograd_pack_b3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dc_conv2d_kernel_name": (C.c_int, [C.POINTER(ConvDesc), C.c_char_p, C.c_size_t]),
    "dc_conv2d_wgrad_workspace_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "dc_conv2d_wgrad_f32": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_roi_align_pyramid_bwd_f32": (C.c_int, [C.POINTER(RoiAlignDesc), C.c_void_p]),
    "dc_downsample2x_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dc_downsample2x_sum_dual_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dc_maxpool2x2s2_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dc_maxpool3x3s2_same_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dc_mold_image_padded_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "dc_mold_image_rgbx_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "dc_roi_align_pyramid_f32": (C.c_int, [C.POINTER(RoiAlignDesc), C.c_void_p]),
    "dc_subsample2_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dc_proposals_workspace_bytes": (C.c_size_t, [C.POINTER(ProposalDesc)]),
    "dc_proposals_f32": (C.c_int, [C.POINTER(ProposalDesc), C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_lstm_seq_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "dc_lstm_seq_fwd_f32": (C.c_int, [C.POINTER(LstmFwdDesc), C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_lstm_seq_bwd_f32": (C.c_int, [C.POINTER(LstmBwdDesc), C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_softmax_ce_f32": (C.c_int, [C.POINTER(SoftmaxCeDesc), C.c_void_p]),
    "dc_vocab_ce_workspace_bytes": (C.c_size_t, [C.POINTER(VocabCeDesc)]),
    "dc_vocab_ce": (C.c_int, [C.POINTER(VocabCeDesc), C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_argmax_rows_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dc_gather_rows_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dc_bn_relu_fwd_f32": (C.c_int, [C.POINTER(BnReluDesc), C.c_void_p]),
    "dc_bn_relu_bwd_f32": (C.c_int, [C.POINTER(BnReluDesc), C.c_void_p]),
    "dc_conv_weight_dgrad_pack_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dc_rpn_loss_grad_f32": (C.c_int, [C.POINTER(RpnLossDesc), C.c_void_p]),
    "dc_scatter2_add_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dc_detection_targets_f32": (C.c_int, [C.POINTER(DetectionTargetsDesc), C.c_void_p]),
    "dc_caption_tables_i32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dc_set_persistent_cus": (C.c_int, [C.c_int]),
    "dc_get_persistent_cus": (C.c_int, []),
    "dc_l2_reg_workspace_bytes": (C.c_size_t, [C.c_size_t]),
    "dc_l2_reg_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_axpy_f32": (C.c_int, [C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_relu_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dc_relu_bwd_dual_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_fold_time_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "dc_colsum_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "dc_colsum_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_sumsq_workspace_bytes": (C.c_size_t, [C.c_size_t]),
    "dc_sumsq_f32": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_mean_f32": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "dc_bn_fold_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dc_bn_bwd_f32": (C.c_int, [C.c_void_p] * 8 + [C.c_long, C.c_int, C.c_void_p]),
    "dc_mul_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_maxpool3x3s2_same_bwd_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]),
    "dc_dropout_mask_f32": (C.c_int, [C.c_void_p, C.c_size_t, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "dc_amsgrad_step_f32": (C.c_int, [C.POINTER(AmsgradDesc), C.c_void_p]),
    "dc_zero_fill": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "dc_reg_sumsq_workspace_bytes": (C.c_size_t, [C.c_size_t]),
    "dc_reg_sumsq_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(RegSegments), C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
}

ABI_VERSION = 600        # include/dcap.h: DC_ABI_VERSION (the ctypes Structures below mirror that header's layouts)
_lib = None


class DcapError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises if it has not been built -- no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("DCAP_LIB", LIB_PATH)          # DCAP_LIB: an experiment build of the same library (tools/build_variant.sh)
    if not os.path.exists(path):
        raise DcapError("%s is missing: build it with `python __graft_entry__.py` "
                        "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    # torch first: its wheel bundles the HIP / HSA runtime it was built with, and the library must bind to THAT copy (same soname as the
    # system's /opt/rocm one).  Loaded before torch, the library maps the system runtime, torch then brings its own, and the process
    # holds two HSA runtimes -- the library's launches fail with "no ROCm-capable device is detected" (build() followed by smoke() in one
    # process did exactly that).
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.dc_version()
    if got // 100 != ABI_VERSION // 100:                 # descriptor layouts are per major version (include/dcap.h: DC_ABI_VERSION)
        raise DcapError("%s is ABI version %d, these bindings are for %d: rebuild it (python __graft_entry__.py)" % (path, got, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().dc_last_error()
        raise DcapError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else ""))
