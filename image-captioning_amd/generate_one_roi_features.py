"""load_model() / generate_features() of the reference
(dense_img_cap_separate_models/generate_one_roi_features.py:59-75): per-RoI 7x7x256 features of the
ground-truth regions of one image; and the image-level variant
(feature_generation/generate_roi_features.py:60-75): mean over RoIs, flattened to 12 544."""
import os

import numpy as np

from .config import Config
from .modified_dense_model import DenseImageCapRCNN

ROOT_DIR = os.getcwd()
MODEL_DIR = os.path.join(ROOT_DIR, "logs")
MODEL_PATH = os.path.join(ROOT_DIR, "rcnn_coco.npz")      # the reference's rcnn_coco.h5, converted


# DenseCapConfig / InferenceConfig of the reference script (:22-54): batch of one image at inference
DenseCapConfig = type("DenseCapConfig", (Config,), dict(NAME="dense image captioning", GPU_COUNT=1, IMAGES_PER_GPU=3, STEPS_PER_EPOCH=500,
                                                     VALIDATION_STEPS=50, EMBEDDING_SIZE=100, PADDING_SIZE=5, REDUCE_EMBEDDINGS=True))
InferenceConfig = type("InferenceConfig", (DenseCapConfig,), dict(GPU_COUNT=1, IMAGES_PER_GPU=1))

config = InferenceConfig()


def load_model(weights=None, model_path=None, **kw):
    """Inference-mode feature model.  `weights` (a dict) or `model_path` (.npz / .h5) seed it; the
    reference hard-codes MODEL_PATH."""
    model = DenseImageCapRCNN(mode="inference", model_dir=MODEL_DIR, config=config, **kw)
    if weights is not None:
        model.set_weights(weights)
    else:
        model.load_weights(model_path or MODEL_PATH, by_name=True)
    return model


def generate_features(dataset, image_id, model, device_features=False):
    """[N,7,7,256] features of the image's ground-truth regions (one batch of one image); device_features=True: a torch
    tensor that stays on the GPU (text_generation_model_v2.data_generator(device_resident=True))."""
    boxes = dataset.load_captions_and_rois(image_id)[0]
    kw = {"device_features": True} if device_features else {}          # (feature models with the reference's plain signature keep working)
    out = model.generate_captions([dataset.load_image(image_id)], boxes[np.newaxis], verbose=0, **kw)
    return out[0]['features']


def generate_image_level_features(dataset, image_id, model):
    """feature_generation/generate_roi_features.py: np.mean(features, axis=0).flatten()."""
    return np.mean(generate_features(dataset, image_id, model), axis=0).flatten()
