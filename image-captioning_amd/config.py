"""Config: the reference's class-attribute flag system, attribute for attribute
(dense_img_cap_separate_models/config.py:18-172; the five copies in the reference are identical on
the hot path).  Sub-class it and override attributes; __init__ derives BATCH_SIZE, IMAGE_SHAPE and
BACKBONE_SHAPES exactly as the reference does, so scripts written against it run unchanged
(defaults and derived values are pinned to the reference's module by tests/test_golden_reference.py).

The defaults are kept as one table (name, value, meaning) and installed as class attributes below, which is
also what display() and the docs walk."""
import math

import numpy as np

DEFAULTS = (
    # ---- devices / batching
    ("NAME", None, "experiment name, set by sub-classes"),
    ("GPU_COUNT", 1, "processes (one per GPU) of a data-parallel run"),
    ("IMAGES_PER_GPU", 2, "BATCH_SIZE = IMAGES_PER_GPU * GPU_COUNT"),
    ("STEPS_PER_EPOCH", 1000, "training steps between checkpoints"),
    ("VALIDATION_STEPS", 50, "validation batches per epoch"),
    # ---- backbone pyramid and RPN anchors
    ("BACKBONE_STRIDES", [4, 8, 16, 32, 64], "stride of P2..P6"),
    ("RPN_ANCHOR_SCALES", (32, 64, 128, 256, 512), "anchor side per pyramid level, pixels"),
    ("RPN_ANCHOR_RATIOS", [0.5, 1, 2], "width/height ratios at every cell"),
    ("RPN_ANCHOR_STRIDE", 1, "anchors at every cell (1) or every other cell (2)"),
    ("RPN_NMS_THRESHOLD", 0.7, "IoU above which a lower-scored proposal is dropped"),
    ("RPN_TRAIN_ANCHORS_PER_IMAGE", 256, "anchors that enter the RPN losses"),
    ("POST_NMS_ROIS_TRAINING", 2000, "proposals kept after NMS, training graph"),
    ("POST_NMS_ROIS_INFERENCE", 1000, "proposals kept after NMS, inference graph"),
    # ---- image molding
    ("IMAGE_MIN_DIM", 800, "short side after scaling up"),
    ("IMAGE_MAX_DIM", 1024, "long side limit and padded square size"),
    ("IMAGE_PADDING", True, "zero-pad to IMAGE_MAX_DIM x IMAGE_MAX_DIM"),
    ("MEAN_PIXEL", np.array([123.7, 116.8, 103.9]), "RGB mean subtracted by mold_image"),
    # ---- RoI heads
    ("TRAIN_ROIS_PER_IMAGE", 200, "RoIs the detection-target layer hands to the heads"),
    ("ROI_POSITIVE_RATIO", 0.33, "share of positive RoIs among them"),
    ("POOL_SIZE", 7, "PyramidROIAlign output side"),
    ("MASK_POOL_SIZE", 14, "unused on the captioning path"),
    ("MASK_SHAPE", [28, 28], "unused on the captioning path"),
    ("MAX_GT_INSTANCES", 100, "ground-truth regions per image (zero padded)"),
    ("RPN_BBOX_STD_DEV", np.array([0.1, 0.1, 0.2, 0.2]), "RPN delta normalisation"),
    ("BBOX_STD_DEV", np.array([0.1, 0.1, 0.2, 0.2]), "head delta normalisation"),
    ("DETECTION_MAX_INSTANCES", 100, "generations kept per image at inference"),
    ("DETECTION_MIN_CONFIDENCE", 0.7, "unused on the captioning path"),
    ("DETECTION_NMS_THRESHOLD", 0.3, "NMS over generated regions"),
    # ---- optimisation
    ("LEARNING_RATE", 0.001, "Adam step size"),
    ("LEARNING_MOMENTUM", 0.9, "kept for SGD-era scripts"),
    ("WEIGHT_DECAY", 0.0001, "L2(w)/size(w) regulariser of the joint model"),
    ("USE_RPN_ROIS", True, "train the heads on RPN proposals"),
    # ---- captions
    ("PADDING_SIZE", 15, "tokens per caption incl. <start>/<end>"),
    ("EMBEDDING_SIZE", 100, "word embedding width"),
    ("EMBEDDING_WEIGHTS", None, "embedding matrix [vocab, width]"),
    ("VOCABULARY_SIZE", 0, "softmax width"),
    # not a field of the reference's Config: the reference hard-codes it in the model (dense_img_cap/dense_model.py:769-770,
    # text_generation_model.py:141-142: KL.LSTM(..., recurrent_dropout=0.2)); a field here so that parity runs can switch it off
    ("RECURRENT_DROPOUT", 0.2, "recurrent_dropout of imgcap_lstm1/2 in the training phase (the reference's hard-coded 0.2)"),
    ("DROPOUT_ROWS", "roi", "'roi': one recurrent-dropout mask set per RoI (single masked pass); 'prefix': per (RoI, prefix) row as Keras draws them"),
)


class Config(object):
    def __init__(self):
        side = self.IMAGE_MAX_DIM
        self.BATCH_SIZE = self.IMAGES_PER_GPU * self.GPU_COUNT
        self.IMAGE_SHAPE = np.array([side, side, 3])
        self.BACKBONE_SHAPES = np.array([[int(math.ceil(side / s)), int(math.ceil(side / s))] for s in self.BACKBONE_STRIDES])

    def display(self):
        print("\nConfigurations:")
        for a in dir(self):
            if not a.startswith("__") and not callable(getattr(self, a)):
                print("{:30} {}".format(a, getattr(self, a)))
        print("\n")


for _name, _value, _doc in DEFAULTS:
    setattr(Config, _name, _value)
del _name, _value, _doc
