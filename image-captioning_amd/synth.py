"""Synthetic weights and Visual-Genome-shaped inputs.

There is no network for datasets or checkpoints, so the benchmark and the parity tests run on
random-init weights of the reference's architecture and on synthetic images / RoIs / captions.
Everything comes from numpy.random.default_rng(seed); weights are float32.
Initialisers follow the Keras defaults the reference relies on: Glorot-uniform kernels, orthogonal
recurrent kernels, zero biases with unit forget-gate bias, and random (frozen) BN statistics.
"""
import numpy as np

from .layers import resnet_fpn_convs, V2_WORD_LSTM

F32 = np.float32


def glorot(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(F32)


def orthogonal(rng, rows, cols):
    a = rng.standard_normal((max(rows, cols), min(rows, cols)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    return (q.T if rows < cols else q).astype(F32)


def conv_weights(rng, name, k, cin, cout, W):
    W[name + "/kernel"] = glorot(rng, (k, k, cin, cout), k * k * cin, k * k * cout)
    W[name + "/bias"] = (0.01 * rng.standard_normal(cout)).astype(F32)


def bn_weights(rng, name, c, W):
    W[name + "/gamma"] = rng.uniform(0.5, 1.5, c).astype(F32)
    W[name + "/beta"] = (0.1 * rng.standard_normal(c)).astype(F32)
    W[name + "/moving_mean"] = (0.1 * rng.standard_normal(c)).astype(F32)
    W[name + "/moving_variance"] = rng.uniform(0.5, 1.5, c).astype(F32)


def lstm_weights(rng, name, n_in, units, W):
    W[name + "/kernel"] = glorot(rng, (n_in, 4 * units), n_in, 4 * units)
    W[name + "/recurrent_kernel"] = orthogonal(rng, units, 4 * units)
    b = np.zeros(4 * units, F32)
    b[units:2 * units] = 1.0                      # unit_forget_bias
    W[name + "/bias"] = b


def dense_weights(rng, name, n_in, n_out, W):
    W[name + "/kernel"] = glorot(rng, (n_in, n_out), n_in, n_out)
    W[name + "/bias"] = np.zeros(n_out, F32)


def encoder_weights(seed=0, stage4_blocks=22):
    rng = np.random.default_rng(seed)
    W = {}
    for s in resnet_fpn_convs(stage4_blocks):
        conv_weights(rng, s.name, s.k, s.cin, s.cout, W)
        if s.bn:
            bn_weights(rng, s.bn, s.cout, W)
            # keep the residual trunk's scale bounded over 33 blocks with random (non-trained) BN
            if s.name.endswith("2c"):
                W[s.bn + "/gamma"] *= F32(0.3)
    return W


def vgg16_weights(seed=0):
    """He-style kernels (std sqrt(2 / fan_in)) keep the activations O(1) through 13 ReLU convolutions with random weights."""
    from .layers import vgg16_convs
    rng = np.random.default_rng(seed)
    W = {}
    for s in vgg16_convs():
        W[s.name + "/kernel"] = (rng.standard_normal((s.k, s.k, s.cin, s.cout)) * np.sqrt(2.0 / (s.k * s.k * s.cin))).astype(F32)
        W[s.name + "/bias"] = (0.01 * rng.standard_normal(s.cout)).astype(F32)
    return W


def rpn_weights(seed=4, anchors_per_loc=3, depth=256):
    """rpn_graph's three convolutions (feature_generation/dense_model.py:701-725)."""
    rng = np.random.default_rng(seed)
    W = {}
    conv_weights(rng, "rpn_conv_shared", 3, depth, 512, W)
    conv_weights(rng, "rpn_class_raw", 1, 512, 2 * anchors_per_loc, W)
    conv_weights(rng, "rpn_bbox_pred", 1, 512, 4 * anchors_per_loc, W)
    return W


def head_weights(seed=1, pool=7, cin=256, width=1024):
    rng = np.random.default_rng(seed)
    W = {}
    conv_weights(rng, "mrcnn_class_conv1", pool, cin, width, W)
    bn_weights(rng, "mrcnn_class_bn1", width, W)
    conv_weights(rng, "mrcnn_class_conv2", 1, width, width, W)
    bn_weights(rng, "mrcnn_class_bn2", width, W)
    return W


def embedding_matrix(seed, vocab, dim=300):
    """preprocess.load_corpus layout (preprocess.py:8-27): row 0 zeros (<unk>/pad), rows 1-2
    (<start>, <end>) uniform(-0.5, 0.5), the rest GloVe-like N(0, 0.4^2)."""
    rng = np.random.default_rng(seed)
    E = (0.4 * rng.standard_normal((vocab, dim))).astype(F32)
    E[0] = 0
    E[1:3] = rng.uniform(-0.5, 0.5, (2, dim)).astype(F32)
    return E


def v2_weights(seed, vocab, emb_dim=300, word_units=1024, units=256, inject=True, feat_dim=1024):
    rng = np.random.default_rng(seed)
    W = {}
    lstm_weights(rng, V2_WORD_LSTM, emb_dim, word_units, W)
    if inject:
        lstm_weights(rng, "imgcap_lstm", feat_dim + word_units, units, W)
        dense_weights(rng, "imgcap_d1", units, vocab, W)
    else:
        dense_weights(rng, "imgcap_d1", feat_dim + word_units, vocab, W)
    return W


def v1_weights(seed, vocab, emb_dim=300, units=512, feat_dim=1024):
    rng = np.random.default_rng(seed)
    W = {}
    lstm_weights(rng, "imgcap_lstm1", emb_dim + feat_dim, units, W)
    lstm_weights(rng, "imgcap_lstm2", units, units, W)
    dense_weights(rng, "imgcap_lstm_d1", units + feat_dim, 1024, W)
    dense_weights(rng, "imgcap_lstm_d2", 1024, vocab, W)
    return W


def images(seed, n, h=1024, w=1024):
    return np.random.default_rng(seed).integers(0, 256, (n, h, w, 3), dtype=np.uint8)


def rois(seed, n_img, n_roi, h=1024, w=1024, lo=32, hi=512):
    """(y1,x1,y2,x2) integer pixels; side lengths log-uniform in [lo,hi] so all four pyramid
    levels are hit; clipped to the image."""
    rng = np.random.default_rng(seed)
    hi_h, hi_w = min(hi, h), min(hi, w)
    lo_h, lo_w = min(lo, hi_h), min(lo, hi_w)
    hh = np.exp(rng.uniform(np.log(lo_h), np.log(hi_h), (n_img, n_roi)))
    ww = np.exp(rng.uniform(np.log(lo_w), np.log(hi_w), (n_img, n_roi)))
    y1 = rng.uniform(0, h - 1, (n_img, n_roi))
    x1 = rng.uniform(0, w - 1, (n_img, n_roi))
    y1, x1 = np.floor(y1), np.floor(x1)
    y2 = np.minimum(np.floor(y1 + hh) + 1, h)
    x2 = np.minimum(np.floor(x1 + ww) + 1, w)
    return np.stack([y1, x1, y2, x2], axis=-1).astype(np.float32)


def captions_v1(seed, n, T, vocab, lmin=3, lmax=13):
    """[1(<start>), w_1..w_L, 2(<end>), 0-pad] float32 [n,T] (text_generation_model.py:108-114)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, T), np.float32)
    for i in range(n):
        L = int(rng.integers(lmin, min(lmax, T - 2) + 1))
        out[i, 0] = 1
        out[i, 1:1 + L] = rng.integers(3, vocab, L)
        out[i, 1 + L] = 2
    return out


def captions_v2(seed, n, T, vocab, full=True, lmin=3):
    """v2 captions are plain word-id lists (no start/end tokens, OOV dropped; _v2.py:101-114).
    full=True: every caption has exactly T words (the benchmark's 15-token captions)."""
    rng = np.random.default_rng(seed)
    caps = []
    for _ in range(n):
        L = T if full else int(rng.integers(lmin, T + 1))
        caps.append(rng.integers(3, vocab, L).astype(np.int32))
    return caps
