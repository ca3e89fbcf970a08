"""Full-size joint train step (configs[4] shapes: 1024x1024 image, ResNet-101+FPN+RPN, 2000 proposals -> 200 RoIs,
T=15, V=50000) timed on one GPU.  Synthetic weights/inputs.  Usage: python tools/joint_bench.py [--steps K] [--vocab V]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from image_captioning_amd import synth
from image_captioning_amd.config import Config
from image_captioning_amd.dense_model import DenseImageCapRCNN, build_rpn_targets
from image_captioning_amd import utils


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--vocab", type=int, default=50000)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--conv-math", default=None, help="f32 | bf16x3 | bf16x2 (forward convs and data gradients)")
    a = ap.parse_args()
    S, V, T = a.size, a.vocab, 15

    class Cfg(Config):
        NAME = "joint"
        IMAGES_PER_GPU = 1
        IMAGE_MIN_DIM = S
        IMAGE_MAX_DIM = S
        PADDING_SIZE = T
        VOCABULARY_SIZE = V
        EMBEDDING_SIZE = 300
        RECURRENT_DROPOUT = 0.0
    cfg = Cfg()
    cfg.EMBEDDING_WEIGHTS = synth.embedding_matrix(3, V)
    model = DenseImageCapRCNN("training", cfg, "logs", conv_math=a.conv_math)
    # random FPN maps are O(10): keep the RPN / head activations in a trained network's range
    w = model.get_weights_dict()
    model.set_weights({"rpn_conv_shared/kernel": w["rpn_conv_shared/kernel"] * np.float32(0.02),
                       "rpn_bbox_pred/kernel": w["rpn_bbox_pred/kernel"] * np.float32(0.3),
                       "mrcnn_class_conv1/kernel": w["mrcnn_class_conv1/kernel"] * np.float32(0.05)})
    model.compile(1e-5)
    rng = np.random.RandomState(0)
    img = synth.images(7, 1, S, S)
    n_gt = 40
    y, x = rng.randint(0, S - 64, n_gt), rng.randint(0, S - 64, n_gt)
    hh, ww = rng.randint(32, 400, n_gt), rng.randint(32, 400, n_gt)
    boxes = np.stack([y, x, np.minimum(y + hh, S), np.minimum(x + ww, S)], axis=1).astype(np.int32)
    caps = synth.captions_v1(9, n_gt, T, V, lmin=3, lmax=12).astype(np.int32)
    anchors = utils.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, cfg.BACKBONE_SHAPES, cfg.BACKBONE_STRIDES, 1)
    match, deltas = build_rpn_targets(img[0].shape, anchors, caps, boxes, cfg, rng)
    gt_caps = np.zeros((1, cfg.MAX_GT_INSTANCES, T), np.int32)
    gt_boxes = np.zeros((1, cfg.MAX_GT_INSTANCES, 4), np.int32)
    gt_caps[0, :n_gt], gt_boxes[0, :n_gt] = caps, boxes
    inputs = [img, np.zeros((1, 12)), match[None, :, None], deltas[None], gt_caps, gt_boxes]
    for _ in range(a.warmup):
        out = model.train_on_batch(inputs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = model.train_on_batch(inputs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    t = model.last_targets
    print("joint step %.2f ms  (%.1f images/s, %d RoIs: %d pos / %d neg)  losses %s  params %.1f M" %
          (dt * 1e3, 1.0 / dt, t['npos'] + t['nneg'], t['npos'], t['nneg'], ["%.4f" % v for v in out], model.store.n_train / 1e6))
    print("peak memory %.2f GB" % (torch.cuda.max_memory_allocated() / 2 ** 30))


if __name__ == "__main__":
    main()
