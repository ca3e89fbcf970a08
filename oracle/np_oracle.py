"""
oracle/np_oracle.py -- CPU restatement (NumPy, float64) of the arithmetic primitives on the
dense-captioning hot path of frosinastojanovska/image-captioning.

*** TEST INFRASTRUCTURE, NOT PRODUCT CODE. ***
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (image-captioning_amd/) never imports it and fails loudly without its HIP library.

*** PARITY: host-side geometry PINNED by outputs of the reference's own code; layer arithmetic UNPINNED. ***
Pinned (tests/golden/make_reference_vectors.py runs the reference's pure-NumPy functions and its config module in this
container and stores inputs + outputs in tests/golden/reference_host_vectors.npz; tests/test_golden_reference.py): anchor
generation, IoU, NMS, box-delta application / refinement, mold_image -- for this file -- plus, for the product's host
mirrors, Config, the Dataset class, resize/pad, image meta, build_rpn_targets, load_image_gt, the joint and v1
data generators and the vocabulary helpers.
Unpinned: the reference's layer arithmetic lives in Keras 2.1.x / TensorFlow 1.x (un-vendored, un-pinned, not
installable here: no network, Python 3.10), and the reference repo ships no tests, golden vectors or fixtures for it
(SURVEY.md section 4, 8c).  For conv / BN / pooling / crop_and_resize / LSTM / softmax-CE / losses / Adam this file
restates the documented semantics of those libraries at the reference's call sites (cited per function, paths relative to
/root/reference); that part is checked against analytic known-answer tests and an independent torch restatement
(oracle/torch_ref.py), not against outputs of the reference itself.

Everything is float64 unless a function says otherwise (box-coordinate arithmetic follows TF's
float32 kernels so that index decisions -- pyramid level, out-of-range bins -- are bit-faithful).
"""
import numpy as np

F64 = np.float64
F32 = np.float32


# --------------------------------------------------------------------------------------------
# Convolution / normalisation / pooling  (feature_generation/dense_model.py:51-173, :1404-1427)
# --------------------------------------------------------------------------------------------

def same_pad(n, k, s):
    """TF 'SAME' padding: total = max((ceil(n/s)-1)*s + k - n, 0); floor(total/2) goes before."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def _resolve_pad(H, W, kh, kw, stride, padding):
    if padding == 'same':
        (pt, pb), (pl, pr) = same_pad(H, kh, stride), same_pad(W, kw, stride)
    elif padding == 'valid':
        pt = pb = pl = pr = 0
    else:
        pt, pb, pl, pr = padding
    return pt, pb, pl, pr


def conv2d_nhwc(x, w, b=None, stride=1, padding='valid'):
    """Keras Conv2D, channels_last, kernel HWIO [kh,kw,cin,cout]
    (dense_model.py:85-97 identity_block, :120-136 conv_block, :146-147 stem, :1406-1421 FPN).
    padding: 'same' | 'valid' | (top, bottom, left, right)."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    N, H, W, C = x.shape
    kh, kw, C2, O = w.shape
    assert C == C2
    pt, pb, pl, pr = _resolve_pad(H, W, kh, kw, stride, padding)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    Ho = (H + pt + pb - kh) // stride + 1
    Wo = (W + pl + pr - kw) // stride + 1
    out = np.zeros((N * Ho * Wo, O), F64)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky:ky + (Ho - 1) * stride + 1:stride, kx:kx + (Wo - 1) * stride + 1:stride, :]
            out += patch.reshape(-1, C) @ w[ky, kx]
    out = out.reshape(N, Ho, Wo, O)
    if b is not None:
        out = out + np.asarray(b, F64)
    return out


# Winograd F(2x2, 3x3) (Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks", 2015): the form the product evaluates its
# frozen 3x3 / stride 1 / 'same' layers in (csrc/conv_wino.hip).  Not in the reference's source -- it is what cuDNN runs those
# KL.Conv2D layers with under TF -- so this restatement checks the ALGORITHM against conv2d_nhwc above, in the precision asked for.
WINO_G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], F64)
WINO_BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], F64)
WINO_AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], F64)


# F(4x4, 3x3), same paper, interpolation points 0, +-1, +-2, infinity: 36 products per 4x4 output tile and channel pair where the
# direct form needs 144 (4x fewer; F(2x2,3x3): 2.25x).  NOT what the product runs -- restated to size the float32 error of the larger
# transform before a kernel is written for it (tests/test_oracle_kat.py).
WINO4_G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], F64)
WINO4_BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], F64)
WINO4_AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], F64)


def conv2d_winograd_nhwc(x, w, dtype=F64, m=2):
    """3x3 / stride 1 / 'same' convolution as  Y = A^T [ (G g G^T) . (B^T d B) ] A  per m x m output tile (m = 2: the form
    csrc/conv_wino.hip evaluates; m = 4: F(4x4,3x3), see above), every product, sum and transform rounded to `dtype` (float32 = the
    arithmetic of the kernel; sums over channels in one np.matmul)."""
    x = np.asarray(x, dtype)
    w = np.asarray(w, dtype)
    N, H, W, C = x.shape
    assert w.shape[:3] == (3, 3, C) and m in (2, 4)
    O = w.shape[3]
    G, BT, AT = [a.astype(dtype) for a in ((WINO_G, WINO_BT, WINO_AT) if m == 2 else (WINO4_G, WINO4_BT, WINO4_AT))]
    U = np.einsum('ai,ijck,bj->abck', G, w, G).astype(dtype)
    th, tw = (H + m - 1) // m, (W + m - 1) // m
    xp = np.zeros((N, m * th + 2, m * tw + 2, C), dtype)
    xp[:, 1:H + 1, 1:W + 1] = x
    out = np.zeros((N, m * th, m * tw, O), dtype)
    for ty in range(th):
        for tx in range(tw):
            d = xp[:, m * ty:m * ty + m + 2, m * tx:m * tx + m + 2]             # top-left pixel (m ty - 1, m tx - 1) of the image
            V = np.einsum('ai,nijc,bj->nabc', BT, d, BT).astype(dtype)
            M = np.einsum('nabc,abck->nabk', V, U).astype(dtype)
            out[:, m * ty:m * ty + m, m * tx:m * tx + m] = np.einsum('pa,nabk,qb->npqk', AT, M, AT).astype(dtype)
    return out[:, :H, :W]


def conv2d_nhwc_loops(x, w, b=None, stride=1, padding='valid'):
    """Loop-level twin of conv2d_nhwc for tiny shapes (self-check of the vectorised form)."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    N, H, W, C = x.shape
    kh, kw, _, O = w.shape
    pt, pb, pl, pr = _resolve_pad(H, W, kh, kw, stride, padding)
    Ho = (H + pt + pb - kh) // stride + 1
    Wo = (W + pl + pr - kw) // stride + 1
    out = np.zeros((N, Ho, Wo, O), F64)
    for n in range(N):
        for oy in range(Ho):
            for ox in range(Wo):
                for o in range(O):
                    acc = 0.0 if b is None else float(b[o])
                    for ky in range(kh):
                        iy = oy * stride + ky - pt
                        if iy < 0 or iy >= H:
                            continue
                        for kx in range(kw):
                            ix = ox * stride + kx - pl
                            if ix < 0 or ix >= W:
                                continue
                            for c in range(C):
                                acc += x[n, iy, ix, c] * w[ky, kx, c, o]
                    out[n, oy, ox, o] = acc
    return out


def conv2d_nhwc_backward(x, w, dy, stride=1, padding='valid'):
    """Gradients of conv2d_nhwc w.r.t. x, w and bias (the joint model trains fpn_*/rpn_* convs,
    dense_img_cap/dense_model.py:1829-1831).  float64."""
    x, w, dy = np.asarray(x, F64), np.asarray(w, F64), np.asarray(dy, F64)
    N, H, W, C = x.shape
    kh, kw, _, Oc = w.shape
    pt, pb, pl, pr = _resolve_pad(H, W, kh, kw, stride, padding)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dxp = np.zeros_like(xp)
    Ho, Wo = dy.shape[1:3]
    dw = np.zeros_like(w)
    d2 = dy.reshape(-1, Oc)
    for ky in range(kh):
        for kx in range(kw):
            sl = (slice(None), slice(ky, ky + (Ho - 1) * stride + 1, stride), slice(kx, kx + (Wo - 1) * stride + 1, stride), slice(None))
            dw[ky, kx] = xp[sl].reshape(-1, C).T @ d2
            dxp[sl] += (d2 @ w[ky, kx].T).reshape(N, Ho, Wo, C)
    dx = dxp[:, pt:pt + H, pl:pl + W, :]
    return dx, dw, d2.sum(0)


BN_EPS = 1e-3  # Keras BatchNormalization default epsilon


def batchnorm_inference(x, gamma, beta, mean, var, eps=BN_EPS):
    """BatchNorm with training=False hard-coded (dense_model.py:51-61)."""
    x = np.asarray(x, F64)
    return np.asarray(gamma, F64) * (x - np.asarray(mean, F64)) / np.sqrt(np.asarray(var, F64) + eps) \
        + np.asarray(beta, F64)


def bn_scale_shift(gamma, beta, mean, var, conv_bias=None, eps=BN_EPS):
    """y = scale*conv_nobias + shift  ==  BN(conv_nobias + bias).  float64; the product folds the
    same way on the host (in float64) and hands float32 scale/shift to the kernel epilogue."""
    scale = np.asarray(gamma, F64) / np.sqrt(np.asarray(var, F64) + eps)
    b = 0.0 if conv_bias is None else np.asarray(conv_bias, F64)
    shift = scale * (b - np.asarray(mean, F64)) + np.asarray(beta, F64)
    return scale, shift


def relu(x):
    return np.maximum(x, 0.0)


def maxpool3x3s2_same(x):
    """KL.MaxPooling2D((3,3), strides=(2,2), padding='same') (dense_model.py:150).
    TF SAME on even n: pad 0 before, 1 after; padded cells never win (-inf)."""
    x = np.asarray(x, F64)
    N, H, W, C = x.shape
    pt, pb = same_pad(H, 3, 2)
    pl, pr = same_pad(W, 3, 2)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
    Ho, Wo = -(-H // 2), -(-W // 2)
    out = np.full((N, Ho, Wo, C), -np.inf)
    for ky in range(3):
        for kx in range(3):
            out = np.maximum(out, xp[:, ky:ky + (Ho - 1) * 2 + 1:2, kx:kx + (Wo - 1) * 2 + 1:2, :])
    return out


def subsample2(x):
    """KL.MaxPooling2D(pool_size=(1,1), strides=2) == every other pixel (dense_model.py:1423)."""
    return np.asarray(x)[:, ::2, ::2, :]


def upsample2x(x):
    """KL.UpSampling2D(size=(2,2)): nearest neighbour (dense_model.py:1408)."""
    return np.repeat(np.repeat(np.asarray(x), 2, axis=1), 2, axis=2)


def mold_image(images_u8, mean_pixel):
    """mold_image: float32(image) - MEAN_PIXEL (dense_model.py:2050-2055)."""
    return np.asarray(images_u8).astype(F64) - np.asarray(mean_pixel, F64)


# --------------------------------------------------------------------------------------------
# RoIAlign  (feature_generation/dense_model.py:312-418)
# --------------------------------------------------------------------------------------------

def roi_levels(boxes_norm, image_shape):
    """PyramidROIAlign level routing (dense_model.py:351-362), float32 like the TF graph:
    level = clamp(4 + round_half_even(log(sqrt(h*w) / (224/sqrt(area))) / log(2)), 2, 5)."""
    b = np.asarray(boxes_norm, F32)
    h = b[..., 2] - b[..., 0]
    w = b[..., 3] - b[..., 1]
    area = F32(image_shape[0] * image_shape[1])
    denom = F32(224.0) / np.sqrt(area, dtype=F32)
    with np.errstate(divide='ignore', invalid='ignore'):
        ratio = np.sqrt(h * w, dtype=F32) / denom
        lvl = np.log(ratio, dtype=F32) / np.log(F32(2.0), dtype=F32)
    # tf.cast(tf.round(x), int32): round half to even; log(0) = -inf casts to INT_MIN in TF, the
    # clamp then gives 2.  NaN (negative area) is undefined in TF; we map it to 2 as well.
    r = np.rint(lvl)
    r = np.where(np.isfinite(r), r, -100.0)
    return np.minimum(5, np.maximum(2, 4 + r.astype(np.int64))).astype(np.int32)


def crop_and_resize(image, boxes, box_ind, crop_size, extrapolation_value=0.0):
    """tf.image.crop_and_resize(method='bilinear') (dense_model.py:378-380).
    image [B,H,W,C]; boxes [N,4] (y1,x1,y2,x2) normalised; box_ind [N].
    Sample coordinates are float32 in TF's kernel order (mul, div, mul, add -- no fma); the
    interpolation itself is float64 here."""
    image = np.asarray(image, F64)
    boxes = np.asarray(boxes, F32)
    B, H, W, C = image.shape
    ch, cw = crop_size
    N = boxes.shape[0]
    out = np.full((N, ch, cw, C), extrapolation_value, F64)
    for n in range(N):
        y1, x1, y2, x2 = boxes[n]
        bi = int(box_ind[n])
        hs = (y2 - y1) * F32(H - 1) / F32(ch - 1) if ch > 1 else F32(0)
        ws = (x2 - x1) * F32(W - 1) / F32(cw - 1) if cw > 1 else F32(0)
        for y in range(ch):
            in_y = (y1 * F32(H - 1) + F32(y) * hs) if ch > 1 else F32(0.5) * (y1 + y2) * F32(H - 1)
            if not (in_y >= 0 and in_y <= H - 1):
                continue
            top, bot = int(np.floor(in_y)), int(np.ceil(in_y))
            ly = F64(in_y - F32(top))
            for x in range(cw):
                in_x = (x1 * F32(W - 1) + F32(x) * ws) if cw > 1 else F32(0.5) * (x1 + x2) * F32(W - 1)
                if not (in_x >= 0 and in_x <= W - 1):
                    continue
                left, right = int(np.floor(in_x)), int(np.ceil(in_x))
                lx = F64(in_x - F32(left))
                tl, tr = image[bi, top, left], image[bi, top, right]
                bl, br = image[bi, bot, left], image[bi, bot, right]
                t = tl + (tr - tl) * lx
                bt = bl + (br - bl) * lx
                out[n, y, x] = t + (bt - t) * ly
    return out


def pyramid_roi_align(boxes_norm, feature_maps, image_shape, pool=7):
    """PyramidROIAlign.call (dense_model.py:339-415).  boxes_norm [B,R,4]; feature_maps = [P2..P5],
    each [B,Hk,Wk,C].  Returns [B,R,pool,pool,C] (the reference re-packs it as [1,B*R,...] in
    batch-then-box order, :413-415; same data)."""
    boxes_norm = np.asarray(boxes_norm, F32)
    B, R, _ = boxes_norm.shape
    C = feature_maps[0].shape[-1]
    lv = roi_levels(boxes_norm, image_shape)
    out = np.zeros((B, R, pool, pool, C), F64)
    for i, level in enumerate(range(2, 6)):
        bi, ri = np.nonzero(lv == level)
        if bi.size == 0:
            continue
        out[bi, ri] = crop_and_resize(feature_maps[i], boxes_norm[bi, ri], bi, (pool, pool))
    return out


def normalize_boxes(rois_px, h, w):
    """rois / [h, w, h, w] with the *molded* image size (modified_dense_model.py:1522-1527)."""
    return (np.asarray(rois_px, F32) / np.array([h, w, h, w], F32)).astype(F32)


# --------------------------------------------------------------------------------------------
# Dense / embedding / LSTM (Keras 2.1 semantics: SURVEY.md section 9.5, 9.6)
# --------------------------------------------------------------------------------------------

def hard_sigmoid(z):
    """K.hard_sigmoid: clip(0.2*z + 0.5, 0, 1) -- Keras-2.1 LSTM default recurrent_activation."""
    return np.clip(0.2 * z + 0.5, 0.0, 1.0)


def hard_sigmoid_grad(z):
    y = 0.2 * z + 0.5
    return np.where((y >= 0.0) & (y <= 1.0), 0.2, 0.0)


def embedding(ids, table):
    """KL.Embedding(mask_zero=True): casts ids to int32 (truncation), mask = ids != 0
    (text_generation_model.py:135-140; _v2.py:155-156)."""
    ids = np.asarray(ids).astype(np.int32)
    return np.asarray(table, F64)[ids], ids != 0


def recurrent_dropout_masks(rng, B, Uh, rate):
    """Keras LSTMCell._generate_recurrent_dropout_mask (training phase): four masks K.dropout(ones[B,Uh], rate) -- kept units
    scaled by 1/(1-rate) -- one per gate i,f,c,o, drawn once per call and reused at every timestep."""
    return (rng.random((4, B, Uh)) >= rate).astype(F64) / (1.0 - rate)


def lstm_forward(x, mask, W, U, b, h0=None, c0=None, rec_masks=None):
    """Keras LSTM (gate blocks i,f,c,o; hard-sigmoid gates; tanh) with mask carry.
    x [B,T,I], mask [B,T] bool or None.  Returns (H [B,T,Uh] -- output per step with carry, zeros
    before the first unmasked step --, cache).
    rec_masks [4,B,Uh] (optional): recurrent_dropout in the training phase (Keras 2.1 LSTMCell.call, implementation 1):
    h_{t-1} enters gate g as h_{t-1} * rec_masks[g]  (text_generation_model.py:141-142)."""
    x = np.asarray(x, F64)
    B, T, _ = x.shape
    Uh = U.shape[0]
    W, U, b = np.asarray(W, F64), np.asarray(U, F64), np.asarray(b, F64)
    h = np.zeros((B, Uh)) if h0 is None else np.asarray(h0, F64)
    c = np.zeros((B, Uh)) if c0 is None else np.asarray(c0, F64)
    if mask is None:
        mask = np.ones((B, T), bool)
    H = np.zeros((B, T, Uh))
    rm = None if rec_masks is None else np.asarray(rec_masks, F64)
    cache = dict(x=x, mask=mask, W=W, U=U, steps=[], rec_masks=rm)
    for t in range(T):
        if rm is None:
            z = x[:, t] @ W + h @ U + b
        else:
            z = x[:, t] @ W + b + np.concatenate([(h * rm[g]) @ U[:, g * Uh:(g + 1) * Uh] for g in range(4)], axis=1)
        zi, zf, zc, zo = z[:, :Uh], z[:, Uh:2 * Uh], z[:, 2 * Uh:3 * Uh], z[:, 3 * Uh:]
        i, f, g, o = hard_sigmoid(zi), hard_sigmoid(zf), np.tanh(zc), hard_sigmoid(zo)
        cn = f * c + i * g
        tc = np.tanh(cn)
        hn = o * tc
        m = mask[:, t][:, None]
        cache['steps'].append((h, c, z, i, f, g, o, tc))
        h = np.where(m, hn, h)
        c = np.where(m, cn, c)
        H[:, t] = h
    cache['h_last'], cache['c_last'] = h, c
    return H, cache


def lstm_backward(dH, cache, dh_last=None):
    """Backward of lstm_forward.  dH [B,T,Uh] gradient w.r.t. every step's output (or None),
    dh_last [B,Uh] extra gradient on the final output.  Returns dx, dW, dU, db."""
    x, mask, W, U = cache['x'], cache['mask'], cache['W'], cache['U']
    B, T, _ = x.shape
    Uh = U.shape[0]
    dx = np.zeros_like(x)
    dW, dU, db = np.zeros_like(W), np.zeros_like(U), np.zeros(4 * Uh)
    dh = np.zeros((B, Uh)) if dh_last is None else np.asarray(dh_last, F64).copy()
    dc = np.zeros((B, Uh))
    rm = cache.get('rec_masks')
    for t in reversed(range(T)):
        if dH is not None:
            dh = dh + dH[:, t]
        hp, cp, z, i, f, g, o, tc = cache['steps'][t]
        m = mask[:, t][:, None].astype(F64)
        dhn = dh * m
        dcn = dc * m + dhn * o * (1.0 - tc * tc)
        do = dhn * tc
        dz = np.concatenate([
            dcn * g * hard_sigmoid_grad(z[:, :Uh]),
            dcn * cp * hard_sigmoid_grad(z[:, Uh:2 * Uh]),
            dcn * i * (1.0 - g * g),
            do * hard_sigmoid_grad(z[:, 3 * Uh:]),
        ], axis=1)
        dW += x[:, t].T @ dz
        db += dz.sum(0)
        dx[:, t] = dz @ W.T
        if rm is None:
            dU += hp.T @ dz
            dh_rec = dz @ U.T
        else:
            dh_rec = np.zeros_like(dh)
            for g in range(4):
                sl = slice(g * Uh, (g + 1) * Uh)
                dU[:, sl] += (hp * rm[g]).T @ dz[:, sl]
                dh_rec += rm[g] * (dz[:, sl] @ U[:, sl].T)
        dh = dh * (1.0 - m) + dh_rec
        dc = dc * (1.0 - m) + dcn * f
    return dx, dW, dU, db


def softmax(z):
    z = np.asarray(z, F64)
    e = np.exp(z - z.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


KERAS_EPS = 1e-7


def categorical_crossentropy(target_ids, probs):
    """K.categorical_crossentropy(target=one-hot, output=probs) per row (TF backend):
    output /= sum(output); output = clip(output, 1e-7, 1-1e-7); -sum(target*log(output)).
    Targets are one-hot in every reference call site, so they are passed as ids
    (text_generation_model.py:286-294; _v2.py:267)."""
    p = np.asarray(probs, F64)
    p = p / p.sum(-1, keepdims=True)
    p = np.clip(p, KERAS_EPS, 1.0 - KERAS_EPS)
    idx = np.asarray(target_ids).astype(np.int64)
    return -np.log(np.take_along_axis(p, idx[..., None], -1)[..., 0])


def softmax_ce_grad_logits(target_ids, probs, upstream):
    """d(sum_r upstream_r * CE_r)/d logits for softmax followed by categorical_crossentropy.
    Unclipped rows: p - onehot (renormalisation is the identity on a softmax output); rows whose
    target probability was clipped have zero gradient (tf.clip_by_value passes no gradient outside
    its range)."""
    p = np.asarray(probs, F64)
    idx = np.asarray(target_ids).astype(np.int64)
    pt = np.take_along_axis(p, idx[..., None], -1)[..., 0]
    live = ((pt >= KERAS_EPS) & (pt <= 1.0 - KERAS_EPS)).astype(F64)
    g = p.copy()
    np.put_along_axis(g, idx[..., None], np.take_along_axis(g, idx[..., None], -1) - 1.0, -1)
    return g * (live * np.asarray(upstream, F64))[..., None]


def sparse_categorical_crossentropy_keras(target_ids, probs):
    """K.sparse_categorical_crossentropy on probabilities (dense_img_cap/dense_model.py:943-945):
    clip to [1e-7, 1-1e-7], logits = log(output), then TF sparse softmax-CE on those logits."""
    p = np.clip(np.asarray(probs, F64), KERAS_EPS, 1.0 - KERAS_EPS)
    lg = np.log(p)
    lse = np.log(np.exp(lg - lg.max(-1, keepdims=True)).sum(-1)) + lg.max(-1)
    idx = np.asarray(target_ids).astype(np.int64)
    return lse - np.take_along_axis(lg, idx[..., None], -1)[..., 0]


def argmax_rows(x):
    """tf.argmax / np.argmax: lowest index among ties (text_generation_model.py:224)."""
    return np.argmax(np.asarray(x), axis=-1).astype(np.int32)


# --------------------------------------------------------------------------------------------
# Optimizer (Keras 2.1 Adam with amsgrad=True; text_generation_model.py:425, _v2.py:266,
# dense_img_cap/dense_model.py:1699)
# --------------------------------------------------------------------------------------------

def clip_by_global_norm(grads, clipnorm):
    """Keras clipnorm: norm = sqrt(sum_i sum(g_i^2)); g *= clipnorm/norm when norm >= clipnorm."""
    norm = np.sqrt(sum(float((np.asarray(g, F64) ** 2).sum()) for g in grads))
    if norm >= clipnorm:
        return [np.asarray(g, F64) * (clipnorm / norm) for g in grads], norm
    return [np.asarray(g, F64) for g in grads], norm


def amsgrad_step(p, g, m, v, vhat, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
    """One Keras Adam(amsgrad=True) update; t = iterations + 1 (1-based).  Returns new (p,m,v,vhat)."""
    p, g, m, v, vhat = (np.asarray(a, F64) for a in (p, g, m, v, vhat))
    lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    vhat = np.maximum(vhat, v)
    p = p - lr_t * m / (np.sqrt(vhat) + eps)
    return p, m, v, vhat


# --------------------------------------------------------------------------------------------
# RPN anchors and ProposalLayer (feature_generation/utils.py:333-389; dense_model.py:180-305)
# --------------------------------------------------------------------------------------------

def generate_anchors(scales, ratios, shape, feature_stride, anchor_stride):
    """utils.generate_anchors: [H*W*len(ratios), (y1,x1,y2,x2)] float64, index = (y*W + x)*R + ratio."""
    scales, ratios = np.meshgrid(np.array(scales), np.array(ratios))
    scales, ratios = scales.flatten(), ratios.flatten()
    heights = scales / np.sqrt(ratios)
    widths = scales * np.sqrt(ratios)
    shifts_y = np.arange(0, shape[0], anchor_stride) * feature_stride
    shifts_x = np.arange(0, shape[1], anchor_stride) * feature_stride
    shifts_x, shifts_y = np.meshgrid(shifts_x, shifts_y)
    box_widths, box_centers_x = np.meshgrid(widths, shifts_x)
    box_heights, box_centers_y = np.meshgrid(heights, shifts_y)
    box_centers = np.stack([box_centers_y, box_centers_x], axis=2).reshape([-1, 2])
    box_sizes = np.stack([box_heights, box_widths], axis=2).reshape([-1, 2])
    return np.concatenate([box_centers - 0.5 * box_sizes, box_centers + 0.5 * box_sizes], axis=1)


def generate_pyramid_anchors(scales, ratios, feature_shapes, feature_strides, anchor_stride):
    return np.concatenate([generate_anchors(scales[i], ratios, feature_shapes[i], feature_strides[i], anchor_stride)
                           for i in range(len(scales))], axis=0)


def apply_box_deltas_f32(boxes, deltas):
    """apply_box_deltas_graph in float32, operation by operation (no fma)."""
    b, d = np.asarray(boxes, F32), np.asarray(deltas, F32)
    h = b[:, 2] - b[:, 0]
    w = b[:, 3] - b[:, 1]
    cy = b[:, 0] + F32(0.5) * h
    cx = b[:, 1] + F32(0.5) * w
    cy = cy + d[:, 0] * h
    cx = cx + d[:, 1] * w
    h = h * np.exp(d[:, 2], dtype=F32)
    w = w * np.exp(d[:, 3], dtype=F32)
    y1 = cy - F32(0.5) * h
    x1 = cx - F32(0.5) * w
    return np.stack([y1, x1, y1 + h, x1 + w], axis=1).astype(F32)


def clip_boxes_f32(boxes, window):
    wy1, wx1, wy2, wx2 = (F32(v) for v in window)
    b = np.asarray(boxes, F32)
    return np.stack([np.maximum(np.minimum(b[:, 0], wy2), wy1), np.maximum(np.minimum(b[:, 1], wx2), wx1),
                     np.maximum(np.minimum(b[:, 2], wy2), wy1), np.maximum(np.minimum(b[:, 3], wx2), wx1)], axis=1)


def nms_tf(boxes, scores, max_output_size, iou_threshold):
    """tf.image.non_max_suppression (TF 1.x kernel), float32: candidates in descending score order (stable),
    a candidate is kept unless its IoU with an already kept box exceeds the threshold."""
    b = np.asarray(boxes, F32)
    order = np.argsort(-np.asarray(scores, F32), kind='stable')
    ymin, ymax = np.minimum(b[:, 0], b[:, 2]), np.maximum(b[:, 0], b[:, 2])
    xmin, xmax = np.minimum(b[:, 1], b[:, 3]), np.maximum(b[:, 1], b[:, 3])
    area = (ymax - ymin) * (xmax - xmin)
    thr = F32(iou_threshold)
    cap = int(min(max_output_size, len(order)))
    keep = np.zeros(cap, np.int64)
    n = 0
    # (the kept boxes are tested as one float32 vector per candidate: the same elementwise operations as the kernel's loop over them)
    with np.errstate(divide='ignore', invalid='ignore'):
        for i in order:
            if n >= cap:
                break
            if n and area[i] > 0:
                k = keep[:n]
                ih = np.maximum(np.minimum(ymax[i], ymax[k]) - np.maximum(ymin[i], ymin[k]), F32(0))
                iw = np.maximum(np.minimum(xmax[i], xmax[k]) - np.maximum(xmin[i], xmin[k]), F32(0))
                inter = ih * iw
                if np.any((area[k] > 0) & (inter / (area[i] + area[k] - inter) > thr)):
                    continue
            keep[n] = i
            n += 1
    keep = keep[:n]
    return np.array(keep, np.int32)


def proposal_layer(scores, deltas, anchors, image_hw, proposal_count, nms_threshold, bbox_std=(0.1, 0.1, 0.2, 0.2),
                   pre_nms_limit=6000):
    """ProposalLayer.call for one image: scores [A] (fg prob), deltas [A,4], anchors [A,4] pixels.
    Returns normalised proposals [proposal_count,4] (zero padded), float32.  Index decisions (top-k order,
    clipping, NMS) are float32 as in the TF graph; tf.nn.top_k puts the lower index first among ties."""
    scores = np.asarray(scores, F32)
    d = np.asarray(deltas, F32) * np.asarray(bbox_std, F32)
    k = min(pre_nms_limit, anchors.shape[0])
    ix = np.argsort(-scores, kind='stable')[:k]
    boxes = apply_box_deltas_f32(np.asarray(anchors, F32)[ix], d[ix])
    h, w = image_hw
    boxes = clip_boxes_f32(boxes, (0, 0, h, w))
    nb = (boxes / np.array([h, w, h, w], F32)).astype(F32)
    keep = nms_tf(nb, scores[ix], proposal_count, nms_threshold)
    out = np.zeros((proposal_count, 4), F32)
    out[:len(keep)] = nb[keep]
    return out, ix, keep


# --------------------------------------------------------------------------------------------
# Joint model pieces (dense_img_cap/dense_model.py): RoIAlign backward, detection targets, RPN losses
# --------------------------------------------------------------------------------------------

def crop_and_resize_backward(image_shape, boxes, box_ind, dout):
    """d(image) of crop_and_resize (bilinear scatter; no gradient to the boxes, dense_model.py:378-379)."""
    B, H, W, C = image_shape
    boxes = np.asarray(boxes, F32)
    N, ch, cw, _ = dout.shape
    dimg = np.zeros(image_shape, F64)
    for n in range(N):
        y1, x1, y2, x2 = boxes[n]
        bi = int(box_ind[n])
        hs = (y2 - y1) * F32(H - 1) / F32(ch - 1) if ch > 1 else F32(0)
        ws = (x2 - x1) * F32(W - 1) / F32(cw - 1) if cw > 1 else F32(0)
        for y in range(ch):
            in_y = (y1 * F32(H - 1) + F32(y) * hs) if ch > 1 else F32(0.5) * (y1 + y2) * F32(H - 1)
            if not (in_y >= 0 and in_y <= H - 1):
                continue
            top, bot = int(np.floor(in_y)), int(np.ceil(in_y))
            ly = F64(in_y - F32(top))
            for x in range(cw):
                in_x = (x1 * F32(W - 1) + F32(x) * ws) if cw > 1 else F32(0.5) * (x1 + x2) * F32(W - 1)
                if not (in_x >= 0 and in_x <= W - 1):
                    continue
                left, right = int(np.floor(in_x)), int(np.ceil(in_x))
                lx = F64(in_x - F32(left))
                g = dout[n, y, x]
                dimg[bi, top, left] += g * (1 - ly) * (1 - lx)
                dimg[bi, top, right] += g * (1 - ly) * lx
                dimg[bi, bot, left] += g * ly * (1 - lx)
                dimg[bi, bot, right] += g * ly * lx
    return dimg


def pyramid_roi_align_backward(boxes_norm, map_shapes, image_shape, dout):
    boxes_norm = np.asarray(boxes_norm, F32)
    lv = roi_levels(boxes_norm, image_shape)
    grads = [np.zeros(s, F64) for s in map_shapes]
    for i, level in enumerate(range(2, 6)):
        bi, ri = np.nonzero(lv == level)
        if bi.size:
            grads[i] += crop_and_resize_backward(map_shapes[i], boxes_norm[bi, ri], bi, dout[bi, ri])
    return grads


def overlaps_f32(boxes1, boxes2):
    """overlaps_graph (dense_img_cap/dense_model.py:421-447), float32: IoU matrix [len(b1), len(b2)] (0/0 -> nan like TF)."""
    b1, b2 = np.asarray(boxes1, F32)[:, None, :], np.asarray(boxes2, F32)[None, :, :]
    y1, x1 = np.maximum(b1[..., 0], b2[..., 0]), np.maximum(b1[..., 1], b2[..., 1])
    y2, x2 = np.minimum(b1[..., 2], b2[..., 2]), np.minimum(b1[..., 3], b2[..., 3])
    inter = np.maximum(x2 - x1, F32(0)) * np.maximum(y2 - y1, F32(0))
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    with np.errstate(divide='ignore', invalid='ignore'):
        return (inter / (a1 + a2 - inter)).astype(F32)


def detection_targets(proposals, gt_captions, gt_boxes, train_rois, positive_ratio, shuffle=None):
    """detection_targets_graph (dense_img_cap/dense_model.py:450-528) for one image.  proposals [N,4] and
    gt_boxes [G,4] normalised (zero rows = padding), gt_captions [G,T].  `shuffle(indices)` stands for
    tf.random_shuffle (identity when None: the reference's choice is non-deterministic).
    Returns rois [train_rois,4] float32 and captions [train_rois,T] (zero padded)."""
    p = np.asarray(proposals, F32)
    p = p[np.abs(p).sum(1) > 0]
    gb = np.asarray(gt_boxes, F32)
    nz = np.abs(gb).sum(1) > 0
    gb, gc = gb[nz], np.asarray(gt_captions)[nz]
    ov = overlaps_f32(p, gb)
    iou_max = ov.max(axis=1) if gb.shape[0] else np.zeros(len(p), F32)
    pos = np.nonzero(iou_max >= 0.5)[0]
    neg = np.nonzero(iou_max < 0.5)[0]
    sh = shuffle if shuffle is not None else (lambda a: a)
    pcount = int(train_rois * positive_ratio)
    pos = sh(pos)[:pcount]
    ncount = int((1.0 / positive_ratio) * len(pos)) - len(pos)
    neg = sh(neg)[:ncount]
    assign = ov[pos].argmax(axis=1) if len(pos) else np.zeros(0, np.int64)
    rois = np.zeros((train_rois, 4), F32)
    caps = np.zeros((train_rois, gc.shape[1]), gc.dtype)
    rois[:len(pos)] = p[pos]
    rois[len(pos):len(pos) + len(neg)] = p[neg]
    caps[:len(pos)] = gc[assign]
    return rois, caps, len(pos), len(neg)


def rpn_class_loss(rpn_match, logits):
    """rpn_class_loss_graph (:877-900): sparse softmax CE on non-neutral anchors, mean.  Returns (loss, dlogits)."""
    m = np.asarray(rpn_match).reshape(-1)
    lg = np.asarray(logits, F64).reshape(-1, 2)
    idx = np.nonzero(m != 0)[0]
    d = np.zeros_like(lg)
    if idx.size == 0:
        return 0.0, d.reshape(np.shape(logits))
    cls = (m[idx] == 1).astype(np.int64)
    z = lg[idx]
    p = softmax(z)
    loss = -np.log(p[np.arange(len(idx)), cls]).mean()
    g = p.copy()
    g[np.arange(len(idx)), cls] -= 1.0
    d[idx] = g / len(idx)
    return loss, d.reshape(np.shape(logits))


def rpn_bbox_loss(target_deltas, rpn_match, rpn_bbox):
    """rpn_bbox_loss_graph (:903-933): smooth-L1 between the first P target rows and the deltas of the P
    positive anchors (in anchor order), mean over all 4P elements.  Returns (loss, d rpn_bbox)."""
    m = np.asarray(rpn_match).reshape(-1)
    bb = np.asarray(rpn_bbox, F64).reshape(-1, 4)
    idx = np.nonzero(m == 1)[0]
    d = np.zeros_like(bb)
    if idx.size == 0:
        return 0.0, d.reshape(np.shape(rpn_bbox))
    t = np.asarray(target_deltas, F64)[:len(idx)]
    diff = t - bb[idx]
    ad = np.abs(diff)
    lt = ad < 1.0
    loss = np.where(lt, 0.5 * ad ** 2, ad - 0.5).mean()
    g = np.where(lt, -diff, -np.sign(diff)) / diff.size
    d[idx] = g
    return loss, d.reshape(np.shape(rpn_bbox))


def sparse_cce_keras_with_grad(target_ids, probs, weights):
    """K.sparse_categorical_crossentropy(target, probs) (dense_img_cap/dense_model.py:943-945) per row, times
    `weights`, and the gradient of sum(weights*loss) w.r.t. the softmax LOGITS that produced probs:
    q = clip(p,1e-7,1-1e-7); loss = -log q_t + log sum(q)."""
    p = np.asarray(probs, F64)
    idx = np.asarray(target_ids).astype(np.int64)
    w = np.asarray(weights, F64)
    q = np.clip(p, KERAS_EPS, 1.0 - KERAS_EPS)
    inside = (p >= KERAS_EPS) & (p <= 1.0 - KERAS_EPS)                  # u: where the clip has a gradient
    S = q.sum(-1)
    qt = np.take_along_axis(q, idx[..., None], -1)[..., 0]
    del q
    loss = -np.log(qt) + np.log(S)
    ut = np.take_along_axis(inside, idx[..., None], -1)[..., 0].astype(F64)
    # g = u / S - [k == t] u_t / q_t ;  dz = p * (g - sum(g p)).  Written on one work array (the [rows, V] temporaries are what this
    # function costs at V = 50 000): gp_sum = sum(u p) / S - u_t p_t / q_t
    pt = np.take_along_axis(p, idx[..., None], -1)[..., 0]
    gp_sum = np.where(inside, p, 0.0).sum(-1) / S - ut * pt / qt
    dz = inside / S[..., None]                                            # g without the target term
    dz -= gp_sum[..., None]
    dz *= p
    np.put_along_axis(dz, idx[..., None], np.take_along_axis(dz, idx[..., None], -1) - (pt * ut / qt)[..., None], -1)
    return loss * w, dz * w[..., None]


# ------------------------------------------------------------------ bf16 storage (BASELINE configs[4])

def to_bf16(x):
    """Round-to-nearest-even onto the bfloat16 grid (8 significant bits), returned as float64: the value a bf16 tensor
    holds.  configs[4] stores weights / activations of the RoI head, the decoder and the vocabulary layers in bf16 and
    accumulates in fp32; the oracle for those paths rounds the same operands and multiplies in float64."""
    a = np.ascontiguousarray(np.asarray(x, np.float32))
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32).astype(np.float64)
    return np.where(np.isfinite(a), out, a.astype(np.float64)).reshape(np.shape(x))
