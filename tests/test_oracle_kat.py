"""Analytic known-answer tests pinning the oracle's restatement of the Keras-2.1 / TF-1.x semantics
(SURVEY.md 8c, 9).  The reference ships no vectors, so these are the oracle's anchor."""
import numpy as np
import pytest

from oracle import np_oracle as O
from oracle import np_models as M


def test_conv_vectorised_equals_loops():
    rng = np.random.default_rng(0)
    for (k, s, pad) in [(1, 1, 'valid'), (1, 2, 'valid'), (3, 1, 'same'), (3, 2, 'same'), (7, 2, (3, 3, 3, 3)), (7, 1, 'valid')]:
        x = rng.standard_normal((2, 9, 10, 3))
        w = rng.standard_normal((k, k, 3, 4))
        b = rng.standard_normal(4)
        if k == 7 and pad == 'valid':
            x = rng.standard_normal((2, 7, 7, 3))
        np.testing.assert_allclose(O.conv2d_nhwc(x, w, b, s, pad), O.conv2d_nhwc_loops(x, w, b, s, pad), rtol=1e-12, atol=1e-12)


def test_same_padding_rule():
    assert O.same_pad(512, 3, 2) == (0, 1)      # maxpool 3x3/s2 on even size: 0 before, 1 after
    assert O.same_pad(256, 3, 1) == (1, 1)
    assert O.same_pad(7, 3, 2) == (1, 1)


def test_conv7x7_valid_is_flat_gemm():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 7, 7, 5))
    w = rng.standard_normal((7, 7, 5, 6))
    y = O.conv2d_nhwc(x, w, None, 1, 'valid')[:, 0, 0]
    np.testing.assert_allclose(y, x.reshape(3, -1) @ w.reshape(-1, 6), rtol=1e-12)


def test_bn_fold_identity():
    rng = np.random.default_rng(2)
    y = rng.standard_normal((4, 8))
    g, b, m, v, cb = (rng.standard_normal(8) for _ in range(5))
    v = np.abs(v) + 0.5
    sc, sh = O.bn_scale_shift(g, b, m, v, cb)
    np.testing.assert_allclose(sc * y + sh, O.batchnorm_inference(y + cb, g, b, m, v), rtol=1e-12)


def test_maxpool_same_and_upsample():
    x = np.arange(16, dtype=float).reshape(1, 4, 4, 1)
    p = O.maxpool3x3s2_same(x)[0, :, :, 0]
    np.testing.assert_array_equal(p, [[10, 11], [14, 15]])
    u = O.upsample2x(np.array([[1., 2.], [3., 4.]]).reshape(1, 2, 2, 1))[0, :, :, 0]
    np.testing.assert_array_equal(u, [[1, 1, 2, 2], [1, 1, 2, 2], [3, 3, 4, 4], [3, 3, 4, 4]])
    np.testing.assert_array_equal(O.subsample2(x)[0, :, :, 0], [[0, 2], [8, 10]])


def test_crop_and_resize_linear_ramp_is_exact():
    H, W = 16, 32
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    fm = (3.0 * yy + 0.5 * xx)[None, :, :, None].astype(float)
    box = np.array([[0.125, 0.25, 0.75, 0.875]], np.float32)
    out = O.crop_and_resize(fm, box, [0], (7, 7))[0, :, :, 0]
    in_y = 0.125 * (H - 1) + np.arange(7) * (0.75 - 0.125) * (H - 1) / 6
    in_x = 0.25 * (W - 1) + np.arange(7) * (0.875 - 0.25) * (W - 1) / 6
    np.testing.assert_allclose(out, 3.0 * in_y[:, None] + 0.5 * in_x[None, :], rtol=1e-5)


def test_crop_and_resize_out_of_range_is_zero():
    fm = np.ones((1, 8, 8, 2))
    out = O.crop_and_resize(fm, np.array([[-0.5, 0.0, 0.5, 1.0]], np.float32), [0], (7, 7))
    assert np.all(out[0, :3] == 0) and np.all(out[0, 3:] == 1)      # in_y<0 rows extrapolate to 0
    out = O.crop_and_resize(fm, np.array([[0.0, 0.0, 1.0, 1.0]], np.float32), [0], (7, 7))
    assert np.all(out == 1)                                         # the border itself is in range


def test_roi_level_routing():
    def box(side_px):
        s = side_px / 1024.0
        return np.array([[[0.1, 0.1, 0.1 + s, 0.1 + s]]], np.float32)
    shape = (1024, 1024, 3)
    assert O.roi_levels(box(224), shape)[0, 0] == 4
    assert O.roi_levels(box(112), shape)[0, 0] == 3
    assert O.roi_levels(box(448), shape)[0, 0] == 5
    assert O.roi_levels(box(56), shape)[0, 0] == 2
    assert O.roi_levels(box(16), shape)[0, 0] == 2     # clamp low
    assert O.roi_levels(box(900), shape)[0, 0] == 5    # clamp high
    assert O.roi_levels(np.zeros((1, 1, 4), np.float32), shape)[0, 0] == 2   # zero-area padding box
    # round half to even: np.rint(0.5) = 0, rint(1.5) = 2, rint(-0.5) = -0, rint(2.5)=2
    assert list(np.rint([0.5, 1.5, -0.5, 2.5])) == [0, 2, -0, 2]


def test_hard_sigmoid_breakpoints():
    np.testing.assert_allclose(O.hard_sigmoid(np.array([-2.5, 0.0, 2.5, -10, 10])), [0, 0.5, 1, 0, 1])


def test_lstm_all_masked_is_zero_and_zero_weights():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 5, 4))
    W, U, b = rng.standard_normal((4, 12)), rng.standard_normal((3, 12)), rng.standard_normal(12)
    H, _ = O.lstm_forward(x, np.zeros((2, 5), bool), W, U, b)
    assert np.all(H == 0)
    H, _ = O.lstm_forward(x, None, W * 0, U * 0, b * 0)
    assert np.all(H == 0)                                            # c = 0.5*0 + 0.5*tanh(0) = 0


def test_lstm_pre_vs_post_padding_same_final_state():
    rng = np.random.default_rng(4)
    E = rng.standard_normal((10, 4))
    E[0] = 0
    W, U, b = rng.standard_normal((4, 12)), rng.standard_normal((3, 12)), rng.standard_normal(12)
    ids_post = np.array([[3, 5, 7, 0, 0, 0]])
    ids_pre = np.array([[0, 0, 0, 3, 5, 7]])
    Hpost, _ = O.lstm_forward(E[ids_post], ids_post != 0, W, U, b)
    Hpre, _ = O.lstm_forward(E[ids_pre], ids_pre != 0, W, U, b)
    np.testing.assert_allclose(Hpost[:, -1], Hpre[:, -1], rtol=1e-14)
    np.testing.assert_allclose(Hpost[:, 2], Hpost[:, -1])            # carried through masked steps


def test_lstm_backward_matches_finite_differences():
    rng = np.random.default_rng(5)
    B, T, I, Uh = 3, 4, 5, 2
    x = rng.standard_normal((B, T, I))
    mask = np.array([[1, 1, 0, 1], [0, 1, 1, 0], [1, 1, 1, 1]], bool)
    W, U, b = 0.5 * rng.standard_normal((I, 4 * Uh)), 0.5 * rng.standard_normal((Uh, 4 * Uh)), 0.1 * rng.standard_normal(4 * Uh)
    R = rng.standard_normal((B, T, Uh))

    def loss(W_, U_, b_, x_):
        H, _ = O.lstm_forward(x_, mask, W_, U_, b_)
        return (H * R).sum()
    H, cache = O.lstm_forward(x, mask, W, U, b)
    dx, dW, dU, db = O.lstm_backward(R, cache)
    eps = 1e-6
    for arr, grad, idx in [(W, dW, (1, 3)), (U, dU, (0, 5)), (b, db, (2,)), (x, dx, (0, 1, 2)), (x, dx, (1, 0, 0))]:
        a2 = arr.copy(); a2[idx] += eps
        a3 = arr.copy(); a3[idx] -= eps
        args = lambda a: (a if arr is W else W, a if arr is U else U, a if arr is b else b, a if arr is x else x)
        num = (loss(*args(a2)) - loss(*args(a3))) / (2 * eps)
        assert abs(num - grad[idx]) < 1e-6 * max(1, abs(num))


def test_winograd_f2x2_3x3_is_the_same_convolution():
    """The minimal-filtering form of a 3x3 / stride 1 / 'same' convolution (what csrc/conv_wino.hip evaluates): exact in float64
    (odd sizes: half-empty edge tiles), and in float32 arithmetic within a few fp32 roundings of the float64 convolution -- the same
    order as the direct float32 sum, far inside the 2e-5 tolerance the GPU tests hold both kernels to."""
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 7, 9, 24))
    w = rng.standard_normal((3, 3, 24, 8)) / np.sqrt(9 * 24)
    want = O.conv2d_nhwc(x, w, None, 1, 'same')
    np.testing.assert_allclose(O.conv2d_winograd_nhwc(x, w), want, rtol=0, atol=1e-13)
    scale = np.abs(want).max()
    e_wino = np.abs(O.conv2d_winograd_nhwc(x, w, np.float32).astype(np.float64) - want).max() / scale
    x32, w32 = x.astype(np.float32), w.astype(np.float32)
    direct32 = np.zeros(want.shape, np.float32)
    xp = np.pad(x32, ((0, 0), (1, 1), (1, 1), (0, 0)))
    for ky in range(3):
        for kx in range(3):
            direct32 += xp[:, ky:ky + 7, kx:kx + 9] @ w32[ky, kx]
    e_direct = np.abs(direct32.astype(np.float64) - want).max() / scale
    assert e_wino < 2e-6 and e_direct < 2e-6 and e_wino < 8 * e_direct + 1e-7, (e_wino, e_direct)
    # the transform matrices: G and A^T rows sum the taps / outputs the way the 1-D F(2,3) identity needs
    d, g = rng.standard_normal(4), rng.standard_normal(3)
    y = O.WINO_AT @ ((O.WINO_G @ g) * (O.WINO_BT @ d))
    np.testing.assert_allclose(y, [d[0] * g[0] + d[1] * g[1] + d[2] * g[2], d[1] * g[0] + d[2] * g[1] + d[3] * g[2]], atol=1e-14)


def test_winograd_f4x4_3x3_restatement_and_its_float32_error():
    """F(4x4,3x3) (36 products per 4x4 tile instead of 144): exact in float64 incl. ragged sizes; in float32 its error at a
    ResNet-sized reduction (256 post-ReLU input channels) is an order of magnitude above F(2x2,3x3)'s and the direct sum's -- about
    1e-5 of the output scale at the worst element, 6e-7 rms -- which still fits the 2e-5 per-layer / 2e-4 feature tolerances the
    GPU tests hold the encoder to.  This sizes the error budget for a kernel that does not exist yet."""
    rng = np.random.default_rng(14)
    x = rng.standard_normal((2, 7, 9, 24))
    w = rng.standard_normal((3, 3, 24, 8)) / np.sqrt(9 * 24)
    want = O.conv2d_nhwc(x, w, None, 1, 'same')
    np.testing.assert_allclose(O.conv2d_winograd_nhwc(x, w, m=4), want, rtol=0, atol=1e-12)
    d, g = rng.standard_normal(6), rng.standard_normal(3)
    y = O.WINO4_AT @ ((O.WINO4_G @ g) * (O.WINO4_BT @ d))
    np.testing.assert_allclose(y, [sum(d[i + k] * g[k] for k in range(3)) for i in range(4)], atol=1e-13)
    x = np.maximum(rng.standard_normal((1, 16, 16, 256)), 0)
    w = rng.standard_normal((3, 3, 256, 32)) / np.sqrt(9 * 256)
    want = O.conv2d_nhwc(x, w, None, 1, 'same')
    scale = np.abs(want).max()
    e2 = np.abs(O.conv2d_winograd_nhwc(x, w, np.float32, m=2).astype(np.float64) - want)
    e4 = np.abs(O.conv2d_winograd_nhwc(x, w, np.float32, m=4).astype(np.float64) - want)
    assert e2.max() / scale < 2e-6 and e4.max() / scale < 2e-5, (e2.max() / scale, e4.max() / scale)
    assert np.sqrt((e4 ** 2).mean()) / scale < 2e-6
    assert e4.max() > 3 * e2.max()                         # the larger transform does cost accuracy: it is not free


def test_lstm_recurrent_dropout_masks_and_gradients():
    """Keras recurrent_dropout (training phase): masks of ones reproduce the plain LSTM; kept units are scaled by 1/(1-rate); the
    hand-written backward with masks matches finite differences (incl. through masked timesteps)."""
    rng = np.random.default_rng(9)
    B, T, I, Uh = 3, 4, 5, 2
    x = rng.standard_normal((B, T, I))
    mask = np.array([[1, 1, 0, 1], [0, 1, 1, 0], [1, 1, 1, 1]], bool)
    W, U, b = 0.5 * rng.standard_normal((I, 4 * Uh)), 0.5 * rng.standard_normal((Uh, 4 * Uh)), 0.1 * rng.standard_normal(4 * Uh)
    H0, _ = O.lstm_forward(x, mask, W, U, b)
    H1, _ = O.lstm_forward(x, mask, W, U, b, rec_masks=np.ones((4, B, Uh)))
    np.testing.assert_allclose(H0, H1, rtol=0, atol=1e-15)
    rm = O.recurrent_dropout_masks(np.random.default_rng(1), B, Uh, 0.2)
    assert set(np.unique(rm)) <= {0.0, 1.25} and rm.shape == (4, B, Uh)
    big = O.recurrent_dropout_masks(np.random.default_rng(2), 200, 512, 0.2)
    assert abs(big.mean() - 1.0) < 5e-3 and abs((big == 0).mean() - 0.2) < 5e-3           # inverted dropout keeps the expectation
    R = rng.standard_normal((B, T, Uh))

    def loss(W_, U_, b_, x_):
        H, _ = O.lstm_forward(x_, mask, W_, U_, b_, rec_masks=rm)
        return (H * R).sum()
    H, cache = O.lstm_forward(x, mask, W, U, b, rec_masks=rm)
    assert np.abs(H - H0).max() > 1e-3
    dx, dW, dU, db = O.lstm_backward(R, cache)
    eps = 1e-6
    for arr, grad, idx in [(W, dW, (1, 3)), (U, dU, (0, 5)), (U, dU, (1, 2)), (b, db, (2,)), (x, dx, (0, 1, 2)), (x, dx, (2, 0, 0))]:
        a2 = arr.copy(); a2[idx] += eps
        a3 = arr.copy(); a3[idx] -= eps
        args = lambda a: (a if arr is W else W, a if arr is U else U, a if arr is b else b, a if arr is x else x)
        num = (loss(*args(a2)) - loss(*args(a3))) / (2 * eps)
        assert abs(num - grad[idx]) < 1e-6 * max(1, abs(num))


def test_cce_uniform_and_clip():
    V = 1000
    p = np.full((2, V), 1.0 / V)
    np.testing.assert_allclose(O.categorical_crossentropy([3, 7], p), np.log(V), rtol=1e-12)
    p = np.array([[1e-9, 1 - 1e-9]])
    np.testing.assert_allclose(O.categorical_crossentropy([0], p), -np.log(1e-7), rtol=1e-12)   # 16.118
    assert np.all(O.softmax_ce_grad_logits([0], p, [1.0]) == 0)          # clipped row: no gradient
    g = O.softmax_ce_grad_logits([1], O.softmax(np.array([[0., 1., 2.]])), [1.0])
    np.testing.assert_allclose(g, O.softmax(np.array([[0., 1., 2.]])) - np.array([[0, 1, 0]]))


def test_amsgrad_first_step_and_vhat_monotone():
    g = np.array([0.3, -2.0, 1e-3])
    p, m, v, vh = O.amsgrad_step(np.zeros(3), g, 0, 0, 0, 1, lr=1e-3)
    expect = -1e-3 * np.sqrt(1 - 0.999) * g / (np.sqrt(1 - 0.999) * np.abs(g) + 1e-7) * (0.1 / (1 - 0.9))
    np.testing.assert_allclose(p, expect, rtol=1e-12)
    p2, m2, v2, vh2 = O.amsgrad_step(p, g * 1e-3, m, v, vh, 2)
    assert np.all(vh2 >= vh) and np.all(v2 < v)                           # v decays, v-hat does not


def test_clipnorm():
    gs, n = O.clip_by_global_norm([np.array([3.0]), np.array([4.0])], 0.5)
    assert n == 5.0
    np.testing.assert_allclose(np.sqrt(sum((g ** 2).sum() for g in gs)), 0.5)
    gs, n = O.clip_by_global_norm([np.array([0.3]), np.array([0.1])], 0.5)
    np.testing.assert_allclose(gs[0], [0.3])


def test_v2_sample_expansion_and_padding():
    roi, words, tgt = M.v2_expand_samples([[5, 6, 7], [9]], 4)
    assert roi.tolist() == [0, 0, 0, 1]
    assert words.tolist() == [[0, 0, 0, 0], [0, 0, 0, 5], [0, 0, 5, 6], [0, 0, 0, 0]]
    assert tgt.tolist() == [5, 6, 7, 9]
    assert M.pad_sequences_pre([[1, 2, 3, 4, 5, 6]], 4).tolist() == [[3, 4, 5, 6]]     # truncating='pre'


def test_v1_per_prefix_dropout_masks_reduce_to_shared_masks():
    """v1_training_forward with [T,4,B,512] masks (one set per prefix, Keras' TimeDistributed graph): T copies of one set are the
    [4,B,512] form; distinct sets change the output of the prefixes they belong to and only those."""
    from image_captioning_amd import synth
    V, T, B = 16, 3, 2
    Wt = dict(synth.head_weights(1, 7, 256, 1024))
    Wt.update(synth.v1_weights(2, V, 300, 512, 1024))
    Wt['imgcap_embedding_layer/embeddings'] = synth.embedding_matrix(3, V)
    Wt = {k: np.asarray(v, np.float64) for k, v in Wt.items()}
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((B, 7, 7, 256))
    caps = np.array([[1, 5, 2], [1, 7, 9]], np.float64)
    m = tuple(O.recurrent_dropout_masks(np.random.default_rng(10 + l), B, 512, 0.2) for l in range(2))
    shared, _ = M.v1_training_forward(Wt, feat, caps, m)
    tiled, _ = M.v1_training_forward(Wt, feat, caps, tuple(np.stack([x] * T) for x in m))
    np.testing.assert_array_equal(shared, tiled)
    other = tuple(np.stack([x] * (T - 1) + [O.recurrent_dropout_masks(np.random.default_rng(20 + l), B, 512, 0.2)]) for l, x in enumerate(m))
    mixed, _ = M.v1_training_forward(Wt, feat, caps, other)
    np.testing.assert_array_equal(mixed[:, :T - 1], shared[:, :T - 1])
    assert np.abs(mixed[:, T - 1] - shared[:, T - 1]).max() > 1e-6


def test_v1_prefixes_and_targets():
    caps = np.array([[1, 5, 2, 0]], np.float32)
    P = M.v1_prefixes(caps)
    assert P[0].tolist() == [[1, 0, 0, 0], [1, 5, 0, 0], [1, 5, 2, 0], [1, 5, 2, 0]]
    assert M.v1_targets(caps).tolist() == [[5, 2, 0, 0]]


def test_dp_mean_equals_single_rank_on_concatenated_batch():
    from image_captioning_amd import synth
    V = 50
    Wt = dict(synth.head_weights(1), **synth.v2_weights(2, V))
    Wt['imgcap_embedding_layer/embeddings'] = synth.embedding_matrix(3, V)
    rng = np.random.default_rng(6)
    feat = rng.standard_normal((8, 7, 7, 256))
    words = rng.integers(0, V, (8, 5))
    tgt = rng.integers(0, V, 8)
    _, Gall, _ = M.v2_loss_and_grads(Wt, feat, words, tgt)
    parts = [M.v2_loss_and_grads(Wt, feat[i::2], words[i::2], tgt[i::2])[1] for i in range(2)]
    Gm = M.data_parallel_mean(parts)
    for k in Gall:
        np.testing.assert_allclose(Gm[k], Gall[k], rtol=1e-9, atol=1e-12)


def test_anchor_layout_and_counts():
    a = O.generate_pyramid_anchors((32, 64, 128, 256, 512), [0.5, 1, 2], [[256, 256], [128, 128], [64, 64], [32, 32], [16, 16]],
                                   [4, 8, 16, 32, 64], 1)
    assert a.shape == (261888, 4)                         # SURVEY: 261 888 anchors at 1024x1024
    # first cell of P2: centre (0,0); ratio 0.5 -> h = 32/sqrt(.5), w = 32*sqrt(.5)
    h, w = 32 / np.sqrt(0.5), 32 * np.sqrt(0.5)
    np.testing.assert_allclose(a[0], [-h / 2, -w / 2, h / 2, w / 2])
    np.testing.assert_allclose(a[1], [-16, -16, 16, 16])  # ratio 1
    np.testing.assert_allclose(a[3], [-h / 2, 4 - w / 2, h / 2, 4 + w / 2])   # next x (stride 4), ratio 0.5


def test_box_deltas_clip_and_nms():
    boxes = np.array([[0, 0, 10, 10]], np.float32)
    np.testing.assert_allclose(O.apply_box_deltas_f32(boxes, np.zeros((1, 4))), boxes)
    out = O.apply_box_deltas_f32(boxes, np.array([[0.1, -0.2, np.log(2.0), 0.0]]))
    np.testing.assert_allclose(out, [[-4, -2, 16, 8]], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(O.clip_boxes_f32(out, (0, 0, 12, 12)), [[0, 0, 12, 8]], atol=1e-5)
    b = np.array([[0, 0, 10, 10], [0, 0, 10, 9], [20, 20, 30, 30], [0, 0, 10, 6.9], [5, 5, 5, 9]], np.float32)
    s = np.array([0.9, 0.8, 0.7, 0.6, 0.5], np.float32)
    assert O.nms_tf(b, s, 10, 0.7).tolist() == [0, 2, 3, 4]        # IoU(0,1)=0.9 > 0.7 suppressed; 0.69 kept; zero area kept
    assert O.nms_tf(b, s, 2, 0.7).tolist() == [0, 2]
    assert O.nms_tf(b, np.array([0.5, 0.5, 0.5, 0.5, 0.5], np.float32), 1, 0.7).tolist() == [0]   # ties: lowest index first


def test_resnet_trunk_backward_matches_finite_differences():
    """The oracle's backward through trainable ResNet stages (train(layers="3+" | "4+" | "5+" | "all"), dense_img_cap/dense_model.py:
    1829-1845): bottleneck blocks with identity and projection shortcuts, strided stage entries, frozen-statistics BatchNorm with
    trainable gamma / beta, the 3x3/2 max pool and the stem -- against central differences of the float64 forward."""
    from image_captioning_amd import synth
    from oracle import np_models as M
    Wt = {k: np.asarray(v, np.float64) for k, v in synth.encoder_weights(0, 1).items()}
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 64, 64, 3))
    Cs, caches = M.resnet_graph_cached(x, Wt, 1)
    plain = M.resnet_graph(x, Wt, 1)
    assert all(np.allclose(Cs[s], plain[s - 1], rtol=0, atol=1e-12) for s in (2, 3, 4, 5))
    R = {s: rng.standard_normal(Cs[s].shape) for s in (2, 3, 4, 5)}
    loss = lambda W: sum((M.resnet_graph_cached(x, W, 1)[0][s] * R[s]).sum() for s in (2, 3, 4, 5))
    G = M.resnet_backward(R, caches, Wt, 1)
    assert set(G) == set(M.backbone_trainable(Wt, 1, 1))
    assert set(M.backbone_trainable(Wt, 4, 1)) == {k for k in G if k.split('/')[0][3 if k.startswith('res') else 2] in '45'}
    checked = 0
    for key in ('res4a_branch2a/bias', 'bn3b_branch2c/beta', 'res2a_branch1/kernel', 'conv1/kernel', 'bn_conv1/gamma', 'res5a_branch2b/kernel',
                'bn4a_branch1/gamma', 'res3a_branch1/kernel', 'bn2c_branch2a/beta'):
        w = Wt[key]
        for _ in range(3):
            idx = tuple(int(rng.integers(0, n)) for n in w.shape)
            if abs(G[key][idx]) < 1e-9:
                continue                                   # a dead unit (zero on both sides; finite differences agree trivially)
            vals = []
            for sgn in (1, -1):
                Wp = dict(Wt)
                a = w.copy()
                a[idx] += sgn * 1e-6
                Wp[key] = a
                vals.append(loss(Wp))
            fd = (vals[0] - vals[1]) / 2e-6
            assert abs(fd - G[key][idx]) < 5e-3 * max(1.0, abs(fd)), (key, idx, fd, G[key][idx])     # (ReLU / max-pool kinks inside the step)
            checked += 1
    assert checked >= 12


def test_maxpool_backward_routes_to_the_first_maximum():
    from oracle import np_models as M
    x = np.zeros((1, 4, 4, 1))
    x[0, 1, 1, 0], x[0, 1, 2, 0] = 5.0, 5.0                # a tie inside window (0,0) of the SAME-padded 3x3/2 pool
    y = O.maxpool3x3s2_same(x)
    dy = np.arange(1.0, 5.0).reshape(1, 2, 2, 1)
    dx = M.maxpool3x3s2_same_backward(x, y, dy)
    assert dx.sum() == dy.sum()                             # every window's gradient lands on exactly one pixel
    assert dx[0, 1, 1, 0] >= dy[0, 0, 0, 0] and (dx[0, 1, 2, 0] == 0 or dx[0, 1, 2, 0] == dy[0, 0, 1, 0])
