"""DenseImageCapRCNN: the RoI feature extractor behind the reference's own class interface
(dense_img_cap_separate_models/modified_dense_model.py:1342 DenseImageCapRCNN, :1583 load_weights,
:1806-1840 mold_inputs, :1886-1923 generate_captions; image-level variant
feature_generation/dense_model.py:1868-1905).

Inference variants: use_generated_rois=False (ground-truth RoIs: what the caption decoders' data
generators call; the RPN is not evaluated) and use_generated_rois=True (RPN + ProposalLayer ->
POST_NMS_ROIS_INFERENCE proposals, the feature_generation/ image-level path).  The joint training graph
is a SURVEY.md section 8(f) "next" row and raises NotImplementedError.
"""
import numpy as np
import torch

from . import utils
from .encoder import EncoderPlan
from .layers import resnet_fpn_convs


class BatchNorm(object):
    """Marker for the reference's BatchNorm(training=False) layer (modified_dense_model.py BatchNorm):
    frozen statistics are folded into the conv epilogue (packing.fold_bn); nothing to run."""


def compose_image_meta(image_id, image_shape, window):
    return utils.compose_image_meta(image_id, image_shape, window)


def mold_image(images, config):
    return utils.mold_image(images, config)


def load_weight_file(filepath):
    """Checkpoint reader: '<layer>/<weight>' -> ndarray.  .npz files hold those keys directly; Keras HDF5 weight files
    (mask_rcnn_coco.h5, rcnn_coco.h5, model-47-1.74.h5, ... -- dense_img_cap/dense_model.py:1656-1692) are parsed by the
    package's own HDF5 reader (hdf5_lite: no h5py needed)."""
    if filepath.endswith(".npz"):
        with np.load(filepath) as z:
            return {k: z[k] for k in z.files}
    if filepath.endswith(".h5") or filepath.endswith(".hdf5"):
        from .hdf5_lite import load_keras_weights
        return load_keras_weights(filepath)
    raise ValueError("unknown weight file type: %s" % filepath)


def save_weight_file(filepath, weights, layer_groups=None, group_member_order=None):
    """Checkpoint writer: .npz, or a Keras-layout HDF5 weight file (Keras' ModelCheckpoint(save_weights_only=True) format)
    when the name ends in .h5 / .hdf5 (layer_groups: hdf5_lite.save_keras_weights -- layers nested in a wrapper layer)."""
    if filepath.endswith(".h5") or filepath.endswith(".hdf5"):
        from .hdf5_lite import save_keras_weights
        save_keras_weights(filepath, weights, layer_groups=layer_groups, group_member_order=group_member_order)
    else:
        np.savez(filepath, **weights)


class DenseImageCapRCNN(object):
    def __init__(self, mode, config, model_dir, use_generated_rois=False, device=None, stage4_blocks=22, conv_math=None):
        assert mode in ['training', 'inference']
        if mode == 'training':
            raise NotImplementedError("the joint training graph (dense_img_cap/dense_model.py) is a SURVEY 8(f) 'next' row")
        self.mode = mode
        self.config = config
        self.model_dir = model_dir
        self.use_generated_rois = use_generated_rois
        self.stage4_blocks = stage4_blocks
        self.conv_math = conv_math          # None: encoder.DEFAULT_CONV_MATH / DCAP_CONV_MATH; 'f32' | 'bf16x3'
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self._weights = None
        self._plans = {}

    # ---- weights -------------------------------------------------------------------------
    def weight_names(self):
        names = []
        for s in resnet_fpn_convs(self.stage4_blocks):
            names += [s.name + "/kernel", s.name + "/bias"]
            if s.bn:
                names += [s.bn + "/" + w for w in ("gamma", "beta", "moving_mean", "moving_variance")]
        if self.use_generated_rois:
            for n in ("rpn_conv_shared", "rpn_class_raw", "rpn_bbox_pred"):
                names += [n + "/kernel", n + "/bias"]
        return names

    def _rpn_config(self):
        if not self.use_generated_rois:
            return None
        c = self.config
        return dict(scales=c.RPN_ANCHOR_SCALES, ratios=c.RPN_ANCHOR_RATIOS, strides=c.BACKBONE_STRIDES,
                    anchor_stride=c.RPN_ANCHOR_STRIDE, bbox_std=[float(v) for v in c.RPN_BBOX_STD_DEV],
                    nms_threshold=c.RPN_NMS_THRESHOLD, proposal_count=c.POST_NMS_ROIS_INFERENCE)

    def set_weights(self, weights):
        missing = [n for n in self.weight_names() if n not in weights]
        if missing:
            raise KeyError("encoder weights missing: %s ..." % missing[:5])
        self._weights = {n: np.asarray(weights[n], np.float32) for n in self.weight_names()}
        self._plans = {}

    def load_weights(self, filepath, by_name=False, exclude=None):
        """by_name loading as keras topology.load_weights_from_hdf5_group_by_name: layers absent from
        the file keep their current values; `exclude` skips layers."""
        loaded = load_weight_file(filepath)
        cur = dict(self._weights or {})
        for k, v in loaded.items():
            if exclude and k.split("/")[0] in exclude:
                continue
            cur[k] = v
        self.set_weights(cur)

    # ---- plan ----------------------------------------------------------------------------
    def plan(self, batch, h, w):
        key = (batch, h, w)
        if key not in self._plans:
            if self._weights is None:
                raise RuntimeError("load_weights()/set_weights() must be called before inference")
            self._plans[key] = EncoderPlan(self._weights, batch, h, w, self.device, self.stage4_blocks,
                                           self.config.MEAN_PIXEL, rpn=self._rpn_config(), math=self.conv_math)
        return self._plans[key]

    def extract_features(self, images_u8, rois_px=None):
        """Device-resident fast path: images [B,H,W,3] uint8 (numpy or torch, already the model's
        size), rois [B,R,4] pixels (or None with use_generated_rois: the RPN's proposals)
        -> torch [B,R,7,7,256] on the GPU (no host round trip)."""
        imgs = torch.as_tensor(images_u8)
        B, H, W, _ = imgs.shape
        p = self.plan(B, H, W)
        p.forward(imgs)
        if self.use_generated_rois:
            self.last_proposals = p.proposals()
            return p.roi_features(boxes_norm=self.last_proposals)
        return p.roi_features(rois_px)

    # ---- reference API -------------------------------------------------------------------
    def mold_inputs(self, images):
        molded, metas, windows = [], [], []
        for image in images:
            m, window, scale, padding = utils.resize_image(image, min_dim=self.config.IMAGE_MIN_DIM,
                                                           max_dim=self.config.IMAGE_MAX_DIM,
                                                           padding=self.config.IMAGE_PADDING)
            molded.append(m)                                # mean subtraction happens on the GPU
            metas.append(compose_image_meta(0, image.shape, window))
            windows.append(window)
        return np.stack(molded), np.stack(metas), np.stack(windows)

    def generate_captions(self, images, rois=None, verbose=0, device_features=False):
        """images: list of [H,W,3] uint8; rois: [len(images), N, 4] (y1,x1,y2,x2) pixels.
        Returns [{'features': float32 [N,7,7,256]}] like the reference (whose slice
        features[i][1000*i:1000*(i+1)] only works for BATCH_SIZE == 1; image i > 0 gets its own RoIs here).
        device_features=True keeps the features on the GPU (torch tensors, a copy the caller owns): a training generator that
        feeds them back to the decoder then never moves [N,7,7,256] floats through host memory."""
        assert self.mode == "inference", "Create model in inference mode."
        assert len(images) == self.config.BATCH_SIZE, "len(images) must be equal to BATCH_SIZE"
        molded, metas, windows = self.mold_inputs(images)
        rois = None if self.use_generated_rois else np.asarray(rois, np.float32)
        feats = self.extract_features(molded, rois)
        feats = feats.clone() if device_features else feats.cpu().numpy()
        n = self.config.POST_NMS_ROIS_INFERENCE
        results = [{"features": feats[i][:n]} for i in range(len(images))]
        if self.use_generated_rois:
            # evaluate_models/modified_dense_model.py:1922-1929: the evaluation's copy also hands back the proposals the features were
            # pooled from (normalised y1, x1, y2, x2) -- what test_score_dense_captions.generate_features reads as results[0]['rois']
            props = self.last_proposals.cpu().numpy()
            for i, r in enumerate(results):
                r["rois"] = props[i][:n]
        return results

    def train(self, *a, **k):
        raise NotImplementedError("joint training is a SURVEY 8(f) 'next' row")
