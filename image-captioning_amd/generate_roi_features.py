"""Image-level features of the reference's feature_generation/ package
(feature_generation/generate_roi_features.py:60-75): ResNet-101 + FPN + RPN + ProposalLayer, PyramidROIAlign on
the POST_NMS_ROIS_INFERENCE proposals, mean over RoIs, flattened to 12 544 floats."""
import os

import numpy as np

from .config import Config
from .modified_dense_model import DenseImageCapRCNN

ROOT_DIR = os.getcwd()
MODEL_DIR = os.path.join(ROOT_DIR, "logs")
MODEL_PATH = os.path.join(ROOT_DIR, "img_cap_dense.npz")      # the reference's img_cap_dense.h5, converted


DenseCapConfig = type("DenseCapConfig", (Config,), dict(NAME="dense image captioning", GPU_COUNT=1, IMAGES_PER_GPU=3, STEPS_PER_EPOCH=500,
                                                     VALIDATION_STEPS=50, EMBEDDING_SIZE=100, PADDING_SIZE=5, REDUCE_EMBEDDINGS=True))
InferenceConfig = type("InferenceConfig", (DenseCapConfig,), dict(GPU_COUNT=1, IMAGES_PER_GPU=1))

config = InferenceConfig()


def load_model(weights=None, model_path=None, **kw):
    model = DenseImageCapRCNN(mode="inference", model_dir=MODEL_DIR, config=kw.pop("config", config),
                              use_generated_rois=True, **kw)
    if weights is not None:
        model.set_weights(weights)
    else:
        model.load_weights(model_path or MODEL_PATH, by_name=True)
    return model


def generate_features(image, model):
    """image: [H,W,3] uint8 array (the reference reads a JPEG path with skimage, which is absent here)."""
    results = model.generate_captions([np.asarray(image)], verbose=0)
    return np.mean(results[0]['features'], axis=0).flatten()
