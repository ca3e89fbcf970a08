"""Config: the reference's class-attribute flag system, attribute for attribute
(dense_img_cap_separate_models/config.py:18-172; the five copies in the reference are identical on
the hot path).  Sub-class it and override attributes; __init__ derives BATCH_SIZE, IMAGE_SHAPE and
BACKBONE_SHAPES exactly as the reference does, so scripts written against it run unchanged."""
import math

import numpy as np


class Config(object):
    NAME = None                     # experiment name, set by sub-classes

    # -- devices / batching: BATCH_SIZE = IMAGES_PER_GPU * GPU_COUNT
    GPU_COUNT = 1
    IMAGES_PER_GPU = 2
    STEPS_PER_EPOCH = 1000
    VALIDATION_STEPS = 50

    # -- backbone pyramid and RPN anchors
    BACKBONE_STRIDES = [4, 8, 16, 32, 64]
    RPN_ANCHOR_SCALES = (32, 64, 128, 256, 512)
    RPN_ANCHOR_RATIOS = [0.5, 1, 2]
    RPN_ANCHOR_STRIDE = 1
    RPN_NMS_THRESHOLD = 0.7
    RPN_TRAIN_ANCHORS_PER_IMAGE = 256
    POST_NMS_ROIS_TRAINING = 2000
    POST_NMS_ROIS_INFERENCE = 1000

    # -- image molding
    IMAGE_MIN_DIM = 800
    IMAGE_MAX_DIM = 1024
    IMAGE_PADDING = True
    MEAN_PIXEL = np.array([123.7, 116.8, 103.9])

    # -- RoI heads
    TRAIN_ROIS_PER_IMAGE = 200
    ROI_POSITIVE_RATIO = 0.33
    POOL_SIZE = 7
    MASK_POOL_SIZE = 14
    MASK_SHAPE = [28, 28]
    MAX_GT_INSTANCES = 100
    RPN_BBOX_STD_DEV = np.array([0.1, 0.1, 0.2, 0.2])
    BBOX_STD_DEV = np.array([0.1, 0.1, 0.2, 0.2])
    DETECTION_MAX_INSTANCES = 100
    DETECTION_MIN_CONFIDENCE = 0.7
    DETECTION_NMS_THRESHOLD = 0.3

    # -- optimisation
    LEARNING_RATE = 0.001
    LEARNING_MOMENTUM = 0.9
    WEIGHT_DECAY = 0.0001
    USE_RPN_ROIS = True

    # -- captions
    PADDING_SIZE = 15
    EMBEDDING_SIZE = 100
    EMBEDDING_WEIGHTS = None
    VOCABULARY_SIZE = 0

    def __init__(self):
        self.BATCH_SIZE = self.IMAGES_PER_GPU * self.GPU_COUNT
        self.IMAGE_SHAPE = np.array([self.IMAGE_MAX_DIM, self.IMAGE_MAX_DIM, 3])
        self.BACKBONE_SHAPES = np.array(
            [[int(math.ceil(self.IMAGE_SHAPE[0] / s)), int(math.ceil(self.IMAGE_SHAPE[1] / s))]
             for s in self.BACKBONE_STRIDES])

    def display(self):
        print("\nConfigurations:")
        for a in dir(self):
            if not a.startswith("__") and not callable(getattr(self, a)):
                print("{:30} {}".format(a, getattr(self, a)))
        print("\n")
