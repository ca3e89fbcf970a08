"""Host-side weight repacking for the HIP kernels (done once at load time, in float64 where it folds)."""
import numpy as np

BN_EPS = 1e-3   # Keras BatchNormalization default, used by the reference's BatchNorm (dense_model.py:51-61)


def pack_conv_kernel(k_hwio):
    """Keras HWIO [kh,kw,cin,cout] -> [cout][kh*kw*cin] (cin fastest): the B operand of the implicit GEMM."""
    kh, kw, cin, cout = k_hwio.shape
    return np.ascontiguousarray(np.transpose(k_hwio, (3, 0, 1, 2)).reshape(cout, kh * kw * cin), dtype=np.float32)


def pack_stem_kernel(k_hwio):
    """conv1 [7,7,3,cout] -> [cout][7][8][4]: kx padded to 8 taps, cin to RGBX (zeros in the pads)."""
    kh, kw, cin, cout = k_hwio.shape
    assert (kh, kw, cin) == (7, 7, 3)
    out = np.zeros((cout, 7, 8, 4), np.float32)
    out[:, :, :7, :3] = np.transpose(k_hwio, (3, 0, 1, 2))
    return out.reshape(cout, 224)


def fold_bn(gamma, beta, mean, var, conv_bias=None, eps=BN_EPS):
    """BN(conv + bias) == scale*conv + shift; folded in float64, handed to the epilogue as float32."""
    scale = np.asarray(gamma, np.float64) / np.sqrt(np.asarray(var, np.float64) + eps)
    b = 0.0 if conv_bias is None else np.asarray(conv_bias, np.float64)
    shift = scale * (b - np.asarray(mean, np.float64)) + np.asarray(beta, np.float64)
    return scale.astype(np.float32), shift.astype(np.float32)


def pack_conv_kernel_dgrad(k_hwio):
    """Packed weights that make dc_conv2d_nhwc_f32(dy, ...) the DATA gradient of a stride-1 convolution:
    rotate the taps by 180 degrees and swap cin/cout -> [cin][kh*kw*cout]; call it with padding k-1-pad."""
    kh, kw, cin, cout = k_hwio.shape
    rot = k_hwio[::-1, ::-1]                                   # [kh,kw,cin,cout] rotated
    return np.ascontiguousarray(np.transpose(rot, (2, 0, 1, 3)).reshape(cin, kh * kw * cout), dtype=np.float32)
