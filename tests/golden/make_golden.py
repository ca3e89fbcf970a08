"""Regenerates tests/golden/oracle_regression.npz: outputs of the NumPy oracle on small seeded inputs.

The reference ships no vectors and cannot run here (parity unpinned), so this file
does NOT pin the oracle to the reference; it pins the oracle to ITSELF across rounds, so that an
accidental change of the restated semantics shows up as a diff of committed numbers.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np_models as M       # noqa: E402
from oracle import np_oracle as O       # noqa: E402
from image_captioning_amd import synth  # noqa: E402


def build():
    rng = np.random.default_rng(2024)
    out = {}
    x = rng.standard_normal((1, 6, 7, 4))
    w = rng.standard_normal((3, 3, 4, 5))
    out["conv_same_s1"] = O.conv2d_nhwc(x, w, rng.standard_normal(5), 1, "same")
    out["conv_same_s2"] = O.conv2d_nhwc(x, w, None, 2, "same")
    out["maxpool"] = O.maxpool3x3s2_same(x)
    fm = rng.standard_normal((1, 9, 11, 3))
    boxes = np.array([[0.1, 0.2, 0.7, 0.9], [0.0, 0.0, 1.0, 1.0], [-0.2, 0.1, 0.4, 1.3]], np.float32)
    out["crop_and_resize"] = O.crop_and_resize(fm, boxes, [0, 0, 0], (7, 7))
    out["roi_levels"] = O.roi_levels(np.array([[[0.1, 0.1, 0.15, 0.2], [0.0, 0.0, 0.5, 0.5], [0.2, 0.2, 0.9, 0.95]]], np.float32),
                                     (1024, 1024, 3))
    xs = rng.standard_normal((2, 5, 3))
    mask = np.array([[1, 1, 0, 1, 0], [0, 1, 1, 1, 1]], bool)
    W, U, b = rng.standard_normal((3, 8)), rng.standard_normal((2, 8)), rng.standard_normal(8)
    H, cache = O.lstm_forward(xs, mask, W, U, b)
    out["lstm_H"] = H
    dx, dW, dU, db = O.lstm_backward(rng.standard_normal(H.shape), cache)
    out["lstm_dW"], out["lstm_dU"] = dW, dU
    p = O.softmax(rng.standard_normal((4, 9)))
    out["cce"] = O.categorical_crossentropy([1, 0, 8, 3], p)
    pp, m, v, vh = O.amsgrad_step(np.ones(4), np.array([0.5, -1.0, 2.0, 0.0]), 0, 0, 0, 1)
    pp, m, v, vh = O.amsgrad_step(pp, np.array([0.1, 0.1, -3.0, 1.0]), m, v, vh, 2)
    out["amsgrad_p2"], out["amsgrad_vhat2"] = pp, vh
    V = 24
    Wt = dict(synth.head_weights(1), **synth.v2_weights(2, V))
    Wt["imgcap_embedding_layer/embeddings"] = synth.embedding_matrix(3, V)
    feat = rng.standard_normal((3, 7, 7, 256))
    words = np.array([[0, 0, 5], [0, 4, 6], [7, 8, 9]])
    loss, G, probs = M.v2_loss_and_grads(Wt, feat, words, np.array([4, 7, 3]))
    out["v2_loss"], out["v2_probs"], out["v2_gbias"] = np.array([loss]), probs, G["imgcap_d1/bias"]
    a = O.generate_anchors(32, [0.5, 1, 2], [2, 2], 4, 1)
    out["anchors"] = a
    bx = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 31], [0, 0, 10, 7.2]], np.float32)
    out["nms_keep"] = O.nms_tf(bx, np.array([0.9, 0.8, 0.7, 0.6], np.float32), 10, 0.7)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.npz"), **build())
    print("written")
