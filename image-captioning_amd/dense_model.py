"""The joint dense-captioning model (configs[4]) behind the reference's own module interface:
dense_img_cap/dense_model.py  (DenseImageCapRCNN :1408, build('training') :1429-1600, compile :1694-1730,
train :1810-1888, data_generator :1260-1403, build_rpn_targets :1095-1183, detection_targets_graph :450-528,
rpn_class_loss_graph :877-900, rpn_bbox_loss_graph :903-933, imgcap_caption_loss_graph :936-946).

One train step on the GPU (IMAGES_PER_GPU images; the reference's training script uses 1, train_dense_captions.py:27):

  frozen ResNet-101  ->  FPN (trainable)  ->  RPN on P2..P6 (trainable)  ->  ProposalLayer (2000 boxes)
  -> detection targets (<= TRAIN_ROIS_PER_IMAGE RoIs, positives carry their GT caption)
  -> PyramidROIAlign -> trainable RoI head -> Model-3 decoder
  losses: masked sparse CE over caption positions with target > 0, RPN class CE, RPN smooth-L1, L2(w)/size(w)
  backward: decoder + head (text_generation_model.CaptionModelV1) -> RoIAlign scatter -> RPN and FPN convolutions
  (data gradients = forward conv kernel on rotated weights, weight gradients = dc_conv2d_wgrad_f32)
  Adam(amsgrad, clipnorm 0.5) over ONE flat bucket holding every trainable weight.

Trainable set = train(layers="no_backbone"): imgcap_*, rpn_*, fpn_*, mrcnn_* (dense_model.py:1829-1831).
The detection-target sampling uses tf.random_shuffle in the reference (order is not reproducible there); here a seeded
numpy permutation on the host, the only host round trip of the step (2000x4 floats down, 200x4 up).
"""
import datetime
import math
import os
import re

import numpy as np
import torch

from . import ops, step_graph, synth, utils
from .encoder import EncoderPlan, fuse_rpn_head
from .layers import resnet_fpn_convs
from .modified_dense_model import load_weight_file, save_weight_file
from .packing import pack_conv_kernel, pack_stem_kernel
from .params import Adam
from .text_generation_model import CaptionModelV1, caption_targets

FPN_CONVS = (("fpn_c5p5", 1, 2048), ("fpn_c4p4", 1, 1024), ("fpn_c3p3", 1, 512), ("fpn_c2p2", 1, 256),
             ("fpn_p2", 3, 256), ("fpn_p3", 3, 256), ("fpn_p4", 3, 256), ("fpn_p5", 3, 256))
HEAD_PAD = 20            # 2A + 4A = 18 RPN head channels padded to a multiple of 4 (A = 3 anchors per location)


# ------------------------------------------------------------------------------------------------
# host-side target building (the reference does these in numpy / TF ops on tiny tensors)
# ------------------------------------------------------------------------------------------------

def compute_overlaps(boxes1, boxes2):
    """IoU matrix [len(boxes1), len(boxes2)] of (y1,x1,y2,x2) boxes (utils.compute_overlaps of the reference)."""
    b1 = np.asarray(boxes1, np.float64)[:, None, :]
    b2 = np.asarray(boxes2, np.float64)[None, :, :]
    ih = np.clip(np.minimum(b1[..., 2], b2[..., 2]) - np.maximum(b1[..., 0], b2[..., 0]), 0, None)
    iw = np.clip(np.minimum(b1[..., 3], b2[..., 3]) - np.maximum(b1[..., 1], b2[..., 1]), 0, None)
    inter = ih * iw
    area = lambda b: (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    return inter / (area(b1) + area(b2) - inter)


def build_rpn_targets(image_shape, anchors, gt_captions, gt_boxes, config, rng=np.random):
    """rpn_match [A] int32 (1 positive / -1 negative / 0 neutral) and rpn_bbox [RPN_TRAIN_ANCHORS_PER_IMAGE, 4]
    (dense_model.py:1095-1183): negatives IoU < 0.3, every GT box claims its best anchor(s), IoU >= 0.7 positive;
    positives are capped at half the budget, negatives fill the rest; deltas of the positives (anchor order)
    against their best GT box, divided by RPN_BBOX_STD_DEV."""
    budget = config.RPN_TRAIN_ANCHORS_PER_IMAGE
    match = np.zeros(anchors.shape[0], np.int32)
    deltas = np.zeros((budget, 4))
    iou = compute_overlaps(anchors, gt_boxes)
    best_gt = iou.argmax(axis=1)
    best_iou = iou[np.arange(iou.shape[0]), best_gt]
    match[best_iou < 0.3] = -1
    match[np.argwhere(iou == iou.max(axis=0))[:, 0]] = 1
    match[best_iou >= 0.7] = 1
    pos = np.where(match == 1)[0]
    surplus = len(pos) - budget // 2
    if surplus > 0:
        match[rng.choice(pos, surplus, replace=False)] = 0
    neg = np.where(match == -1)[0]
    surplus = len(neg) - (budget - int(np.sum(match == 1)))
    if surplus > 0:
        match[rng.choice(neg, surplus, replace=False)] = 0
    pos = np.where(match == 1)[0]
    a, g = anchors[pos].astype(np.float64), np.asarray(gt_boxes, np.float64)[best_gt[pos]]
    ah, aw, gh, gw = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1], g[:, 2] - g[:, 0], g[:, 3] - g[:, 1]
    d = np.stack([((g[:, 0] + 0.5 * gh) - (a[:, 0] + 0.5 * ah)) / ah, ((g[:, 1] + 0.5 * gw) - (a[:, 1] + 0.5 * aw)) / aw,
                  np.log(gh / ah), np.log(gw / aw)], axis=1) / np.asarray(config.RPN_BBOX_STD_DEV, np.float64)
    deltas[:len(pos)] = d
    return match, deltas


def box_iou_f32(boxes1, boxes2):
    """overlaps_graph (:421-447) in float32 like the TF graph."""
    f = np.float32
    b1, b2 = np.asarray(boxes1, f)[:, None, :], np.asarray(boxes2, f)[None, :, :]
    ih = np.maximum(np.minimum(b1[..., 2], b2[..., 2]) - np.maximum(b1[..., 0], b2[..., 0]), f(0))
    iw = np.maximum(np.minimum(b1[..., 3], b2[..., 3]) - np.maximum(b1[..., 1], b2[..., 1]), f(0))
    inter = iw * ih
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / (a1 + a2 - inter)).astype(f)


def detection_targets(proposals, gt_captions, gt_boxes, config, shuffle=None):
    """DetectionTargetLayer for one image (:450-528): proposals [N,4] / gt_boxes [G,4] normalised, zero rows are
    padding; gt_captions [G,T].  Positives: IoU >= 0.5 with some GT box (at most TRAIN_ROIS_PER_IMAGE *
    ROI_POSITIVE_RATIO of them), negatives fill up to the 1/ratio proportion; a positive RoI takes the caption of its
    best GT box, negatives and padding get zeros.  shuffle(idx) -> permuted idx (None keeps proposal order)."""
    f = np.float32
    p = np.asarray(proposals, f)
    p = p[np.abs(p).sum(axis=1) > 0]
    g = np.asarray(gt_boxes, f)
    keep = np.abs(g).sum(axis=1) > 0
    g, caps = g[keep], np.asarray(gt_captions)[keep]
    iou = box_iou_f32(p, g)
    best = iou.max(axis=1) if g.shape[0] else np.zeros(len(p), f)
    mix = shuffle if shuffle is not None else (lambda a: a)
    n_rois, ratio = config.TRAIN_ROIS_PER_IMAGE, config.ROI_POSITIVE_RATIO
    pos = mix(np.nonzero(best >= 0.5)[0])[:int(n_rois * ratio)]
    neg = mix(np.nonzero(best < 0.5)[0])[:int((1.0 / ratio) * len(pos)) - len(pos)]
    rois = np.zeros((n_rois, 4), f)
    out_caps = np.zeros((n_rois, caps.shape[1]), caps.dtype)
    rois[:len(pos)] = p[pos]
    rois[len(pos):len(pos) + len(neg)] = p[neg]
    if len(pos):
        out_caps[:len(pos)] = caps[iou[pos].argmax(axis=1)]
    return rois, out_caps, len(pos), len(neg)


def mold_image(images, config):
    return utils.mold_image(images, config)


def non_max_suppression(boxes, scores, threshold):
    """Greedy NMS on (y1,x1,y2,x2) boxes, best score first (utils.non_max_suppression of the reference); returns the
    kept indices in score order."""
    boxes = np.asarray(boxes, np.float64)
    if boxes.shape[0] == 0:
        return np.zeros(0, np.int32)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    order = np.argsort(scores)[::-1]
    keep = []
    while order.size:
        i = order[0]
        keep.append(i)
        rest = order[1:]
        ih = np.maximum(np.minimum(boxes[i, 2], boxes[rest, 2]) - np.maximum(boxes[i, 0], boxes[rest, 0]), 0)
        iw = np.maximum(np.minimum(boxes[i, 3], boxes[rest, 3]) - np.maximum(boxes[i, 1], boxes[rest, 1]), 0)
        inter = ih * iw
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (area[i] + area[rest] - inter)
        order = rest[~(iou > threshold)]
    return np.asarray(keep, np.int32)


def clip_to_window(window, boxes):
    boxes = np.array(boxes, np.float64)
    boxes[:, [0, 2]] = np.clip(boxes[:, [0, 2]], window[0], window[2])
    boxes[:, [1, 3]] = np.clip(boxes[:, [1, 3]], window[1], window[3])
    return boxes


def refine_generations(rois, word_scores, window, config):
    """GenerationMatchLayer for one image (:593-630): caption score = sum over positions of log(max word probability)
    (word_scores [N,T] holds those maxima); boxes to pixels of the molded image, clipped to the window, rounded;
    NMS(DETECTION_NMS_THRESHOLD) on the clipped boxes by caption score; the best DETECTION_MAX_INSTANCES survive.
    (The reference then indexes `keep` by its own leading values, which raises for most inputs; the evident intent --
    the leading entries of `keep` -- is what runs here.)  Returns (int32 boxes [K,4], kept indices [K])."""
    with np.errstate(divide="ignore"):
        scores = np.log(np.asarray(word_scores, np.float64)).sum(axis=1)
    h, w = config.IMAGE_SHAPE[:2]
    boxes = clip_to_window(window, np.asarray(rois, np.float64) * np.array([h, w, h, w], np.float64))
    keep = non_max_suppression(boxes, scores, config.DETECTION_NMS_THRESHOLD)[:config.DETECTION_MAX_INSTANCES]
    return np.rint(boxes[keep]).astype(np.int32), keep


def unmold_generations(boxes, image_shape, window):
    """Boxes of the molded image -> the original image's pixels (:1925-1962); returns (boxes, mask of non-empty ones)."""
    scale = min(image_shape[0] / (window[2] - window[0]), image_shape[1] / (window[3] - window[1]))
    shift = np.array([window[0], window[1], window[0], window[1]])
    out = ((np.asarray(boxes) - shift) * scale).astype(np.int32)
    return out, (out[:, 2] - out[:, 0]) * (out[:, 3] - out[:, 1]) > 0


def load_image_gt(dataset, config, image_id, augment=False, rng=np.random):
    """image (resized + padded), image_meta, gt_captions [G,T], gt_boxes [G,4] (dense_model.py:953-984).
    As in the reference the boxes are handed on exactly as the dataset stores them: they are NOT rescaled or padded
    with the image, and the horizontal flip mirrors only the image (quirk kept so that the same dataset yields the
    same training inputs)."""
    image = dataset.load_image(image_id)
    boxes, captions = dataset.load_captions_and_rois(image_id)
    shape = image.shape
    image, window, scale, padding = utils.resize_image(image, min_dim=config.IMAGE_MIN_DIM, max_dim=config.IMAGE_MAX_DIM,
                                                       padding=config.IMAGE_PADDING)
    if augment and rng.randint(0, 2):
        image = image[:, ::-1]
    return image, utils.compose_image_meta(image_id, shape, window), captions, boxes


def data_generator(dataset, config, shuffle=True, augment=True, batch_size=1, rng=np.random):
    """Infinite generator of ([images f32 molded, image_meta, rpn_match [B,A,1], rpn_bbox [B,256,4], gt_captions
    [B,MAX_GT,T], gt_boxes [B,MAX_GT,4]], []) -- the six training inputs of the reference's generator (:1260-1403)."""
    image_ids = np.copy(dataset.image_ids)
    anchors = utils.generate_pyramid_anchors(config.RPN_ANCHOR_SCALES, config.RPN_ANCHOR_RATIOS, config.BACKBONE_SHAPES,
                                             config.BACKBONE_STRIDES, config.RPN_ANCHOR_STRIDE)
    b, index, errors = 0, -1, 0
    while True:
        index = (index + 1) % len(image_ids)
        if shuffle and index == 0:
            rng.shuffle(image_ids)
        image_id = image_ids[index]
        try:
            image, meta, caps, boxes = load_image_gt(dataset, config, image_id, augment, rng)
            match, deltas = build_rpn_targets(image.shape, anchors, caps, boxes, config, rng)
        except (GeneratorExit, KeyboardInterrupt):
            raise
        except Exception:
            errors += 1
            if errors > 5:
                raise
            continue
        if boxes.shape[0] > config.MAX_GT_INSTANCES:
            pick = rng.choice(np.arange(boxes.shape[0]), config.MAX_GT_INSTANCES, replace=False)
            caps, boxes = caps[pick], boxes[pick]
        if b == 0:
            images = np.zeros((batch_size,) + image.shape, np.float32)
            metas = np.zeros((batch_size,) + meta.shape, meta.dtype)
            matches = np.zeros((batch_size, anchors.shape[0], 1), match.dtype)
            bboxes = np.zeros((batch_size, config.RPN_TRAIN_ANCHORS_PER_IMAGE, 4), deltas.dtype)
            gt_caps = np.zeros((batch_size, config.MAX_GT_INSTANCES, config.PADDING_SIZE), caps.dtype)
            gt_boxes = np.zeros((batch_size, config.MAX_GT_INSTANCES, 4), boxes.dtype)
        images[b] = mold_image(image.astype(np.float32), config)
        metas[b], matches[b], bboxes[b] = meta, match[:, None], deltas
        gt_caps[b, :caps.shape[0]], gt_boxes[b, :boxes.shape[0]] = caps, boxes
        b += 1
        if b >= batch_size:
            yield [images, metas, matches, bboxes, gt_caps, gt_boxes], []
            b = 0


# ------------------------------------------------------------------------------------------------
# the model
# ------------------------------------------------------------------------------------------------

class StepInputs(step_graph.PackedInputs):
    """Everything the host contributes to one joint train step, packed into ONE buffer of 4-byte words and moved with ONE asynchronous
    copy at the start of the step (step_graph.PackedInputs): the RPN selection (counts, level / index / match of the non-neutral anchors,
    target deltas -- fixed capacity, the image's own counts travel as words), the normalised GT boxes, the GT captions and the step
    scalars (Keras' lr_t, the dropout-mask and detection-target stream positions)."""

    def __init__(self, device, cap, n_gt, T, B=1):
        """cap: selected anchors / target rows of the WHOLE batch (the images' selections travel concatenated); n_gt, T: per image."""
        self.cap, self.n_gt, self.T, self.B = cap, n_gt, T, B
        step_graph.PackedInputs.__init__(self, device, [("counts", 2), ("lvl", cap), ("idx", cap), ("mt", cap), ("deltas", 4 * cap), ("gt", 4 * n_gt * B),
                                                         ("gtc", n_gt * T * B), ("scalars", 4)])


class DenseImageCapRCNN(object):
    LOSS_NAMES = ("rpn_class_loss", "rpn_bbox_loss", "imgcap_loss")
    LAYER_REGEX = {                       # dense_img_cap/dense_model.py:1829-1845
        "no_backbone": r"(imgcap\_.*)|(rpn\_.*)|(fpn\_.*)|(mrcnn\_.*)",
        "3+": r"(res3.*)|(bn3.*)|(res4.*)|(bn4.*)|(res5.*)|(bn5.*)|(imgcap\_.*)|(rpn\_.*)|(fpn\_.*)|(mrcnn\_.*)",
        "4+": r"(res4.*)|(bn4.*)|(res5.*)|(bn5.*)|(imgcap\_.*)|(rpn\_.*)|(fpn\_.*)|(mrcnn\_.*)",
        "5+": r"(res5.*)|(bn5.*)|(imgcap\_.*)|(rpn\_.*)|(fpn\_.*)|(mrcnn\_.*)",
        "all": ".*",
        "caption_only": r"imgcap\_.*",
    }

    def __init__(self, mode, config, model_dir, device=None, stage4_blocks=22, seed=0, lstm_units=512, conv_math=None,
                 compute_dtype="f32", backbone_from=None):
        """compute_dtype: 'f32', or 'bf16' = BASELINE configs[4]'s arithmetic for the RoI head, the caption decoder and the
        vocabulary layers (bf16 copies of weights / activations on the bf16 matrix pipe, fp32 master weights and
        accumulation; text_generation_model.CaptionModelV1).  conv_math selects the convolutions' arithmetic separately."""
        assert mode in ['training', 'inference']
        # IMAGES_PER_GPU images per step and GPU (the reference's graph is batched: config.py:35, DetectionTargetLayer's batch_slice
        # dense_model.py:531-572; its training script uses 1, train_dense_captions.py:27); ParallelModel scales out over GPUs
        self.images_per_gpu = int(config.IMAGES_PER_GPU)
        if self.images_per_gpu < 1:
            raise ValueError("IMAGES_PER_GPU must be >= 1")
        h, w = config.IMAGE_SHAPE[:2]
        if h % 64 or w % 64:
            raise Exception("Image size must be dividable by 2 at least 6 times to avoid fractions when downscaling and up-scaling.")
        self.mode, self.config, self.model_dir = mode, config, model_dir
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.stage4_blocks, self.units = stage4_blocks, lstm_units
        self.conv_math = conv_math          # None -> encoder default / DCAP_CONV_MATH; forward convs and data gradients
        self.compute_dtype = compute_dtype
        self.epoch = 0
        self.A = len(config.RPN_ANCHOR_RATIOS)
        if 6 * self.A > HEAD_PAD:
            raise ValueError("at most 3 anchors per location")
        self._seed = int(seed)
        # DCAP_STEP_GRAPH: 0 = every step eagerly, 1 = the captured step graph, unset / auto (round 6) = capture, then KEEP whichever of the
        # two measured faster on this machine over the first steps (HIP events around whole steps: _choose_step_path) -- on the pool's
        # boxes the eager step has been 1 - 2.5 % faster in two rounds of measurements (the replay of a two-branch graph costs more than
        # issuing its launches from a host that stays ahead); a slow or busy host turns that around.  Assigning use_step_graph pins it.
        self._use_step_graph = step_graph.enabled()
        self._step_path_auto = os.environ.get("DCAP_STEP_GRAPH", "auto") not in ("0", "1")
        self._path_events = {"eager": [], "graph": []}
        self.step_path_choice = None                         # dict(eager_ms, graph_ms, kept) once the automatic choice has been made
        self.use_side_stream = True                          # RPN backward beside the proposals / decoder-forward chain (False: serial order)
        self._side_stream = None
        self.step_graph_fallback = None                      # set to the error text when a step-graph capture failed and the model went eager
        self._dt_step = self._dt_val_step = 0                # detection-target sampling streams (training / forward-only validation passes)
        self._dt_rank = 0                                    # ParallelModel sets the tower's rank: towers shuffle their proposals independently
        self._last_targets = None
        self._step_in = None
        self.optimizer = None
        self.grad_sync = None
        self.is_chief = True               # ParallelModel clears it on ranks > 0: one rank prints and writes checkpoints
        self._outer = self                 # ParallelModel points it at the wrapper: train() then feeds GLOBAL batches through
                                           # the wrapper's train_on_batch (tf.split over towers, losses averaged over towers)
        self._trainable_regex = self.LAYER_REGEX["no_backbone"]
        # backbone_from: the lowest ResNet stage (5, 4, 3, 2; 1 = the stem) whose convolutions and BatchNorm gamma / beta are
        # trainable parameters of this model (train(layers="5+" | "4+" | "3+" | "all") sets it, rebuilding the parameter
        # bucket when needed); None = the backbone is frozen and folded (layers="no_backbone", the reference's training script)
        self.backbone_from = backbone_from
        self._seed = seed
        self.set_log_dir()
        self._build(seed)

    @property
    def use_step_graph(self):
        return self._use_step_graph

    @use_step_graph.setter
    def use_step_graph(self, value):
        self._use_step_graph = bool(value)
        self._step_path_auto = False                          # an explicit choice is kept

    def _choose_step_path(self):
        """Automatic mode: two eager steps and two replays have been timed (events around the whole step incl. the encoder pass): keep the
        faster path.  Called at the start of a later step, when those events completed long ago."""
        ev = self._path_events
        if not self._step_path_auto or len(ev["eager"]) < 2 or len(ev["graph"]) < 2:
            return
        t = {k: min(a.elapsed_time(b) for a, b in v) for k, v in ev.items()}
        self._step_path_auto = False
        keep_graph = t["graph"] <= t["eager"]
        self._use_step_graph = keep_graph
        self.step_path_choice = dict(eager_ms=round(t["eager"], 4), graph_ms=round(t["graph"], 4), kept="graph" if keep_graph else "eager")

    # ---- construction -----------------------------------------------------------------------
    @staticmethod
    def _stage_of(layer):
        """ResNet stage of a trunk layer name (conv1 / bn_conv1 -> 1, res4b_branch2a / bn4b_branch2a -> 4), else None."""
        if layer in ("conv1", "bn_conv1"):
            return 1
        m = re.match(r"(?:res|bn)(\d)[a-z]+_branch", layer)
        return int(m.group(1)) if m else None

    def _trunk_specs(self):
        """ConvSpecs of the trainable ResNet stages (empty while the backbone is frozen)."""
        if self.backbone_from is None:
            return []
        return [s for s in resnet_fpn_convs(self.stage4_blocks) if (self._stage_of(s.name) or 0) >= self.backbone_from]

    def _build(self, seed, weights=None):
        cfg, dev = self.config, self.device
        W = dict(synth.encoder_weights(seed, self.stage4_blocks))
        W.update(synth.rpn_weights(seed + 4, self.A))
        if weights is not None:                          # a rebuild (train(layers=...) moved stages into the bucket): keep every weight
            W.update({k: v for k, v in weights.items() if k in W})
        trunk = self._trunk_specs()
        moved = set()
        for sp in trunk:
            moved.update({sp.name + "/kernel", sp.name + "/bias", sp.bn + "/gamma", sp.bn + "/beta", sp.bn + "/moving_mean", sp.bn + "/moving_variance"})
        self._backbone = {k: v for k, v in W.items() if not k.startswith(("fpn_", "rpn_")) and k not in moved}
        extra = []
        for sp in trunk:                                 # trainable ResNet layers: packed kernel, bias, gamma, beta in the bucket; statistics frozen
            kern = W[sp.name + "/kernel"]
            extra.append((sp.name + "/kernel", pack_stem_kernel(kern) if sp.name == "conv1" else pack_conv_kernel(kern), True))
            extra.append((sp.name + "/bias", W[sp.name + "/bias"], True))
            extra.append((sp.bn + "/gamma", W[sp.bn + "/gamma"], True))
            extra.append((sp.bn + "/beta", W[sp.bn + "/beta"], True))
            extra.append((sp.bn + "/moving_mean", W[sp.bn + "/moving_mean"], False))
            extra.append((sp.bn + "/moving_variance", W[sp.bn + "/moving_variance"], False))
        for name, k, cin in FPN_CONVS:
            extra.append((name + "/kernel", pack_conv_kernel(W[name + "/kernel"]), True))
            extra.append((name + "/bias", W[name + "/bias"], True))
        extra.append(("rpn_conv_shared/kernel", pack_conv_kernel(W["rpn_conv_shared/kernel"]), True))
        extra.append(("rpn_conv_shared/bias", W["rpn_conv_shared/bias"], True))
        hk, hb = fuse_rpn_head(W, HEAD_PAD)
        extra.append(("rpn_head/kernel", pack_conv_kernel(hk), True))
        extra.append(("rpn_head/bias", hb, True))
        self.caption_model = CaptionModelV1([cfg.POOL_SIZE, cfg.POOL_SIZE, 256], cfg, self.units, 'training', dev, seed,
                                            extra_params=extra, compute_dtype=self.compute_dtype)
        # data parallel: a decoder layer's gradient range gets its L2 term and trainable mask right when its backward is done, then
        # its all-reduce starts (forward_backward installs the hook when a gradient exchange is attached); the single-GPU step keeps
        # the one fused regulariser pass over the whole bucket
        self.caption_model.overlap_sync = False
        self._reg_done = []
        self.caption_model.recurrent_dropout = float(getattr(cfg, "RECURRENT_DROPOUT", 0.2))    # dense_img_cap/dense_model.py:769-770: recurrent_dropout=0.2
        self.caption_model.dropout_rows = str(getattr(cfg, "DROPOUT_ROWS", "roi"))
        self.store = self.caption_model.store
        self._plan = None                                    # the encoder plan the step runs on
        self._plans = []                                     # ... and its siblings (pipeline.JointTrainPipeline alternates between two)
        self._reg_coef = None
        self._bufs = {}
        self._bf16_cache = {}
        self._invalidate_graphs()
        if weights is not None:
            self.set_weights({k: v for k, v in weights.items() if k not in self._backbone})

    def _invalidate_graphs(self):
        """Captured step graphs bake buffer addresses, the plan's outputs, the trainable subset and the optimizer state: anything
        that replaces one of those drops them (the next steps run eagerly and re-capture)."""
        self._graphs, self._graph_warm, self._graph_out = {}, {}, {}

    def _buf(self, key, shape, dtype=torch.float32, zero=False):
        b = self._bufs.get(key)
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != dtype:
            b = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._bufs[key] = b
        return b

    @property
    def conv_math_name(self):
        from .encoder import conv_math_name
        return conv_math_name(self.plan().math)

    def plan(self):
        if self._plan is None:
            self._plan = self._new_plan()
            self._plans = [self._plan]
        return self._plan

    def plan_pair(self):
        """Two encoder plans on the same weights (frozen backbone packed per plan, FPN / RPN weights read in place from the parameter
        bucket by both), each with its own activation buffers: batch i + 1's backbone pass may write one while batch i's step still
        reads the other (pipeline.JointTrainPipeline).  use_plan(j) makes one of them the plan the step runs on."""
        self.plan()
        while len(self._plans) < 2:
            self._plans.append(self._new_plan())
        return self._plans

    def use_plan(self, j):
        self._plan = self._plans[j]
        return self._plan

    def _new_plan(self):
        cfg, w = self.config, self.store.w
        ext = {name: (w[name + "/kernel"], None, w[name + "/bias"]) for name, _, _ in FPN_CONVS}
        ext["rpn_conv_shared"] = (w["rpn_conv_shared/kernel"], None, w["rpn_conv_shared/bias"])
        ext["rpn_head"] = (w["rpn_head/kernel"], None, w["rpn_head/bias"])
        count = cfg.POST_NMS_ROIS_TRAINING if self.mode == "training" else cfg.POST_NMS_ROIS_INFERENCE
        rpn = dict(scales=cfg.RPN_ANCHOR_SCALES, ratios=cfg.RPN_ANCHOR_RATIOS, strides=cfg.BACKBONE_STRIDES,
                   anchor_stride=cfg.RPN_ANCHOR_STRIDE, bbox_std=[float(v) for v in cfg.RPN_BBOX_STD_DEV],
                   nms_threshold=cfg.RPN_NMS_THRESHOLD, proposal_count=count, head_channels=HEAD_PAD)
        h, wd = [int(v) for v in cfg.IMAGE_SHAPE[:2]]
        ext_bn = {}
        for sp in self._trunk_specs():             # trainable ResNet layers: kernel in place, BN folded on the device every forward
            ext[sp.name] = (w[sp.name + "/kernel"], None, None)
            ext_bn[sp.name] = dict(gamma=w[sp.bn + "/gamma"], beta=w[sp.bn + "/beta"], bias=w[sp.name + "/bias"],
                                   mean=w[sp.bn + "/moving_mean"], var=w[sp.bn + "/moving_variance"])
        stages = () if self.backbone_from is None else tuple(range(max(self.backbone_from, 2), 6))
        # the FPN/RPN weights change every step: the plan reads them in place from the parameter bucket
        return EncoderPlan(self._backbone, self.images_per_gpu, h, wd, self.device, self.stage4_blocks, cfg.MEAN_PIXEL, rpn=rpn, external=ext,
                           math=self.conv_math, external_bn=ext_bn, train_stages=stages)

    # ---- weights ----------------------------------------------------------------------------
    def _unpacked(self, name, packed):
        cout = packed.shape[0]
        if name == "conv1":                              # the stem's [cout][7][8][4] layout (packing.pack_stem_kernel)
            return np.ascontiguousarray(packed.reshape(cout, 7, 8, 4)[:, :, :7, :3].transpose(1, 2, 3, 0))
        trunk = {sp.name: sp.k for sp in self._trunk_specs()}
        k = trunk.get(name) or dict((n, kk) for n, kk, _ in FPN_CONVS).get(name, 3 if name == "rpn_conv_shared" else 1)
        return np.ascontiguousarray(packed.reshape(cout, k, k, -1).transpose(1, 2, 3, 0))

    def get_weights_dict(self):
        """'<layer>/<weight>' -> ndarray with the reference's Keras layer names and HWIO kernels."""
        out = dict(self._backbone)
        for k, v in self.store.to_numpy().items():
            layer, wname = k.split("/")
            if layer == "rpn_head":
                a2, a6 = 2 * self.A, 6 * self.A
                if wname == "kernel":
                    hwio = self._unpacked(layer, v)
                    out["rpn_class_raw/kernel"], out["rpn_bbox_pred/kernel"] = hwio[..., :a2], hwio[..., a2:a6]
                else:
                    out["rpn_class_raw/bias"], out["rpn_bbox_pred/bias"] = v[:a2], v[a2:a6]
            elif wname == "kernel" and (layer.startswith(("fpn_", "rpn_")) or self._stage_of(layer) is not None):
                out[k] = self._unpacked(layer, v)
            else:
                out[k] = v
        return out

    def set_weights(self, weights):
        """Assign any subset of the model's weights (reference names, HWIO kernels)."""
        W = dict(weights)
        st = self.store
        if any(k.startswith(("rpn_class_raw/", "rpn_bbox_pred/")) for k in W):
            cur = self.get_weights_dict()
            full = {k: W.get(k, cur[k]) for k in ("rpn_class_raw/kernel", "rpn_class_raw/bias", "rpn_bbox_pred/kernel", "rpn_bbox_pred/bias")}
            hk, hb = fuse_rpn_head(full, HEAD_PAD)
            st.assign("rpn_head/kernel", pack_conv_kernel(hk), refresh=False)
            st.assign("rpn_head/bias", hb, refresh=False)
        backbone_changed = False
        for k, v in W.items():
            layer = k.split("/")[0]
            if layer in ("rpn_class_raw", "rpn_bbox_pred"):
                continue
            if k in st.w:
                if k == "conv1/kernel":
                    v = pack_stem_kernel(np.asarray(v, np.float32))
                elif k.endswith("/kernel") and (layer.startswith(("fpn_", "rpn_")) or self._stage_of(layer) is not None):
                    v = pack_conv_kernel(np.asarray(v, np.float32))
                st.assign(k, v, refresh=False)
            elif k in self._backbone:
                self._backbone[k] = np.asarray(v, np.float32)
                backbone_changed = True
        if backbone_changed:
            self._plan, self._plans = None, []
            self._invalidate_graphs()
        st.refresh_shadow()                  # the bf16 operand copies: once per call, not once per weight

    def load_weights(self, filepath, by_name=False, exclude=None):
        loaded = load_weight_file(filepath)
        known = set(self.get_weights_dict())
        pick = {}
        for k, v in loaded.items():
            if exclude and k.split("/")[0] in exclude:
                continue
            if k not in known:
                if by_name:
                    continue
                raise KeyError("weight %s is not part of this model" % k)
            pick[k] = v
        self.set_weights(pick)
        self.set_log_dir(filepath)          # like the reference (:1692): a checkpoint of train() carries its epoch

    def save_weights(self, path):
        """Atomic: written beside the target and renamed, so a reader (or a second rank) never sees a torn file."""
        tmp = path + (".tmp.h5" if path.endswith((".h5", ".hdf5")) else ".tmp.npz")
        # the reference keeps the caption decoder inside TimeDistributed(caption_model, name='imgcap_caption_td'): a Keras-layout
        # file stores those layers under that group, where its load_weights(by_name=True) finds them
        nested = {l: "imgcap_caption_td" for l in ("imgcap_embedding_layer", "imgcap_lstm1", "imgcap_lstm2", "imgcap_lstm_d1", "imgcap_lstm_d2")}
        # ... in the order of TimeDistributed(Model).weights, which Keras' by-name loader assigns BY POSITION: the wrapped model's
        # trainable weights in layer order (word_generation_model, dense_model.py:758-784: lstm1, lstm2, d1, d2), then the frozen embedding
        order = {"imgcap_caption_td": ["imgcap_lstm1", "imgcap_lstm2", "imgcap_lstm_d1", "imgcap_lstm_d2", "imgcap_embedding_layer"]}
        save_weight_file(tmp, self.get_weights_dict(), layer_groups=nested, group_member_order=order)
        os.replace(tmp, path)

    def set_log_dir(self, model_path=None):
        """Log directory and epoch counter (dense_model.py:1776-1798): <model_dir>/<name><YYYYMMDDTHHMM>/ with checkpoints
        img_cap_<name>_<epoch:04d>.npz.  A model_path of that form resumes its directory and sets self.epoch to the
        number of epochs that checkpoint has behind it (Keras numbers checkpoints from 1), i.e. train() continues with the next epoch.  (The reference's pattern
        uses \\w+ for the name, which cannot match its own NAME "dense image captioning" -- its resume silently
        restarts at epoch 0; names with spaces are accepted here.)"""
        self.epoch = 0
        now = datetime.datetime.now()
        if model_path:
            m = re.match(r".*/[\w ]+(\d{4})(\d{2})(\d{2})T(\d{2})(\d{2})/img\_cap\_[\w ]+?\_(\d{4})\.(?:npz|h5)$", str(model_path).replace(os.sep, "/"))
            if m:
                now = datetime.datetime(int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)))
                self.epoch = int(m.group(6))
        name = self.config.NAME.lower()
        self.log_dir = os.path.join(self.model_dir, "{}{:%Y%m%dT%H%M}".format(name, now))
        self.checkpoint_path = os.path.join(self.log_dir, "img_cap_{}_{{epoch:04d}}.npz".format(name))

    def find_last(self):
        """(log_dir of the last trained model, its newest checkpoint) -- dense_model.py:1631-1654: the last directory under
        model_dir whose name starts with the config name, the last 'img_cap*' file in it."""
        if not os.path.isdir(self.model_dir):
            return None, None
        key = self.config.NAME.lower()
        dirs = sorted(d for d in next(os.walk(self.model_dir))[1] if d.startswith(key))
        if not dirs:
            return None, None
        dir_name = os.path.join(self.model_dir, dirs[-1])
        cps = sorted(f for f in next(os.walk(dir_name))[2] if f.startswith("img_cap") and f.endswith((".npz", ".h5")))
        return dir_name, (os.path.join(dir_name, cps[-1]) if cps else None)

    def summary(self):
        rows = ["%-40s %-24s %s" % (k, tuple(v.shape), "trainable" if k in self.store.grad else "frozen")
                for k, v in sorted(self.store.w.items())]
        rows.append("backbone (frozen, BN folded): %d tensors" % len(self._backbone))
        rows.append("trainable parameters: %d" % self.store.n_train)
        return "\n".join(rows)

    def _weights_changed(self):
        """Master weights rewritten from outside the optimizer (ParallelModel.broadcast_weights): re-cast the bf16 mirror."""
        self.store.refresh_shadow()

    @property
    def trainable_weights(self):
        rx = re.compile(self._trainable_regex)
        return [k for k in self.store.trainable_names if rx.fullmatch(k.split("/")[0])]

    def set_trainable(self, layer_regex, keras_model=None, indent=0, verbose=1):
        """Only the regexes that leave the backbone frozen are available (the reference trains "no_backbone",
        train_dense_captions.py:199-203); a frozen subset is realised by masking its gradient."""
        rx = re.compile(layer_regex)
        stages = [self._stage_of(n) for sp in resnet_fpn_convs(self.stage4_blocks) if sp.bn for n in (sp.name, sp.bn) if rx.fullmatch(n)]
        need = min(stages) if stages else None
        if need is not None and (self.backbone_from is None or need < self.backbone_from):
            # ResNet stages join the trainable set: their weights move from the folded, frozen backbone into the parameter
            # bucket (kernels, biases, BN gamma / beta; moving statistics stay frozen), the encoder plan is rebuilt around them
            # (conv_math='bf16', configs[4]'s arithmetic: the trainable stages' forward runs in bf16 storage like the rest of the trunk --
            # their packed kernels are re-cast inside every forward, fp32 copies of their activations are kept for the backward --, the
            # 3x3 data gradients and the stride-1 weight gradients take the bf16 matrix pipe, the 1x1 data gradients and the strided entry
            # convolutions' weight gradients stay exact fp32 GEMMs on the master weights; BatchNorm's backward is fp32 throughout)
            self.backbone_from = need
            self._build(self._seed, weights=self.get_weights_dict())
            self.optimizer = None
        self._trainable_regex = layer_regex
        self._reg_coef = None
        self._invalidate_graphs()

    # ---- compile ----------------------------------------------------------------------------
    def compile(self, learning_rate):
        """Adam(lr, clipnorm=0.5, amsgrad=True); losses = the three graph losses + L2(WEIGHT_DECAY)(w)/size(w) over the
        trainable non-BN weights (:1694-1730)."""
        self.optimizer = Adam(lr=learning_rate, clipnorm=0.5, amsgrad=True)
        self.caption_model.compile(self.optimizer)          # (drops the caption model's own captured steps: they hold the old m / v / vhat)
        self._invalidate_graphs()

    def _masks(self):
        """Per-element L2 coefficient over the flat bucket, and a 0/1 gradient mask when set_trainable() froze a
        subset (None when everything in the bucket trains)."""
        if self._reg_coef is None:
            st = self.store
            rx = re.compile(self._trainable_regex)
            coef = np.zeros(st.flat.numel(), np.float32)
            mask = np.zeros(st.flat.numel(), np.float32)
            base = st.flat.data_ptr()
            a2, a6 = 2 * self.A, 6 * self.A
            wd = float(self.config.WEIGHT_DECAY)
            frozen = False
            for name in st.trainable_names:
                t = st.w[name]
                off, n = (t.data_ptr() - base) // 4, t.numel()
                layer, wname = name.split("/")
                if layer == "rpn_head":                               # two Keras layers fused into one padded tensor
                    per = n // HEAD_PAD                                # elements per output channel (512 or 1)
                    for sub, c0, c1 in (("rpn_class_raw", 0, a2), ("rpn_bbox_pred", a2, a6)):
                        if rx.fullmatch(sub):
                            mask[off + c0 * per:off + c1 * per] = 1.0
                            coef[off + c0 * per:off + c1 * per] = wd / ((c1 - c0) * per)
                        else:
                            frozen = True
                elif rx.fullmatch(layer):
                    mask[off:off + n] = 1.0
                    if wname not in ("gamma", "beta"):
                        coef[off:off + n] = wd / (7 * 7 * 3 * 64 if name == "conv1/kernel" else n)     # (the stem's packed layout carries zero padding)
                else:
                    frozen = True
            self._reg_coef = torch.tensor(coef, device=self.device)
            self._train_mask = torch.tensor(mask, device=self.device) if frozen else None
            # the same two vectors in run-length form (a few dozen segments): what the fused optimizer pass reads instead of 2 x n floats
            self._reg_segs = ops.RegSegmentTable(coef, mask if frozen else None, self.device)
        return self._reg_coef, self._train_mask

    def _reg_segments(self):
        """The run-length table the fused optimizer passes read, or None when the trainable set cuts the bucket into more runs than the
        kernels' LDS table holds (ops.RegSegmentTable.MAX_SEGMENTS): the step then takes the unfused dc_l2_reg_f32 + sumsq passes."""
        self._masks()
        return self._reg_segs if self._reg_segs.nseg <= ops.RegSegmentTable.MAX_SEGMENTS else None

    # ---- one training step ------------------------------------------------------------------
    def _cast_cached(self, t, key):
        """bf16 copy of `t`; with a key, one cast per step (callers pass the same key only while `t` is unchanged)."""
        if key is not None and key in self._bf16_cache:
            return self._bf16_cache[key]
        b = ops.to_bf16(t, out=self._buf(("castb", key or tuple(t.shape)), tuple(t.shape), torch.bfloat16))
        if key is not None:
            self._bf16_cache[key] = b
        return b

    def _bf16_slot(self, t, key):
        """The buffer _cast_cached(t, key) would fill, registered as filled: for producers that write the bf16 copy of `t` themselves
        (relu_bwd, downsample2x_sum, conv2d_bf16's second output) instead of leaving it to a cast pass.  None when nothing reads bf16."""
        if not (self.compute_dtype == "bf16" or self.plan().fast_bf16):
            return None
        b = self._buf(("castb", key), tuple(t.shape), torch.bfloat16)
        self._bf16_cache[key] = b
        return b

    def _wgrad(self, x, dy, k, pad, out, accumulate=False, key=None, dy_key=None):
        """Packed weight gradient of a k x k / stride 1 convolution.  bf16 model: operands cast to bf16, products on the bf16
        matrix pipe (fp32 accumulation) when the shapes allow; exact fp32 products otherwise.  key: cache slot of x's bf16 copy
        (the shared RPN convolution reads every P level twice); dy_key: cache slot of dy's (shared with the data gradient)."""
        if self.compute_dtype == "bf16" and ops.wgrad_bf16_supported(x.shape, dy.shape):
            xb = self._bf16_cache.get(key) if key is not None else None
            if xb is None:
                xb = self.plan().bf16_of(x)            # the forward already wrote a bf16 copy of this plan buffer
            if xb is None:
                xb = ops.to_bf16(x, out=self._buf(("xb", key or id(x), tuple(x.shape)), tuple(x.shape), torch.bfloat16))
                if key is not None:
                    self._bf16_cache[key] = xb
            dyb = self._cast_cached(dy, dy_key)
            return ops.conv2d_wgrad_bf16(xb, dyb, k, k, 1, pad, pad, out=out, accumulate=accumulate)
        return ops.conv2d_wgrad(x, dy, k, k, 1, pad, pad, out=out, accumulate=accumulate)

    def _dgrad(self, dy, wd, k, out, residual=None, key=None, dy_key=None, out_key=None):
        """Data gradient of a k x k / stride-1 'same' convolution = the forward convolution of dy with the rotated, transposed
        kernel `wd` (packed [Cin, k*k*Cout] fp32).  bf16 model with bf16 storage: dy and wd are cast and the product runs on
        dc_conv2d_bf16; otherwise dc_conv2d_nhwc_f32 in the plan's conv arithmetic.  residual: added (the accumulation into a
        gradient map that already holds another path's contribution)."""
        p = self.plan()
        _, h_, w_, cout = dy.shape
        pad = (k - 1) // 2
        res_mode = 0 if residual is None else 1
        if p.fast_bf16 and ops.conv_bf16_supported(cout):
            dyb = self._cast_cached(dy, dy_key)
            wdb = self._cast_cached(wd, None if key is None else "wd_" + key)      # the RPN's rotated kernel serves five levels
            ob = None if out_key is None else self._bf16_slot(out, out_key)         # out_key: the result's bf16 copy from the same epilogue
            return ops.conv2d_bf16(dyb, wdb, k, k, 1, pad, pad, h_, w_, residual=residual, res_mode=res_mode, out=out, out_bf16=ob)[0]
        return ops.conv2d(dy, wd, k, k, 1, pad, pad, h_, w_, residual=residual, res_mode=res_mode, out=out, math=p.math)

    def _images_u8(self, images):
        """uint8 [B,H,W,3] as a torch tensor: device-resident uint8 tensors pass through, host arrays are wrapped."""
        if isinstance(images, torch.Tensor) and images.dtype == torch.uint8:
            return images
        a = np.asarray(images)
        if a.dtype == np.uint8:
            return torch.as_tensor(a)
        # the generator yields molded float images (image - MEAN_PIXEL); the GPU molds from the uint8 original
        return torch.as_tensor(np.clip(np.rint(a.astype(np.float64) + np.asarray(self.config.MEAN_PIXEL, np.float64)), 0, 255).astype(np.uint8))

    def _rpn_selection(self, rpn_match, image=0):
        """Non-neutral anchors of one image as (level, index inside the level's [B, h, w, A] head tensor, match).  The heads of a batch are
        contiguous per level, so image b's anchor i of a level is anchor b * h * w * A + i of that level's tensor: the selections of all
        images, concatenated in image order, are ONE selection over the batched heads -- and the means of the reference's RPN losses run
        over exactly that union (rpn_class_loss_graph / rpn_bbox_loss_graph gather over the whole batch, dense_model.py:877-933)."""
        m = np.asarray(rpn_match).reshape(-1)
        p = self.plan()
        sizes = [h.shape[1] * h.shape[2] * self.A for h in p.rpn_heads]
        if m.size != sum(sizes):
            raise ValueError("rpn_match has %d anchors, the pyramid has %d" % (m.size, sum(sizes)))
        idx = np.nonzero(m != 0)[0]
        bounds = np.cumsum([0] + sizes)
        level = np.searchsorted(bounds, idx, side="right") - 1
        return level.astype(np.int32), (idx - bounds[level] + image * np.asarray(sizes)[level]).astype(np.int32), m[idx].astype(np.int32)

    def _pinned(self, key, shape, dtype=torch.float32):
        """A page-locked host buffer owned by the model (asynchronous device -> host copies land here)."""
        b = self._pins.get(key) if hasattr(self, "_pins") else None
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != dtype:
            if not hasattr(self, "_pins"):
                self._pins = {}
            b = torch.empty(shape, dtype=dtype, pin_memory=True)
            self._pins[key] = b
        return b

    def _step_uploads(self, p, rpn_match, rpn_bbox, gt_norm, gt_caps, training):
        """This step's host inputs -> the device, one asynchronous copy (StepInputs).  Returns the device views the step's kernels read.
        rpn_match / rpn_bbox / gt_norm / gt_caps: the generator's arrays with the image axis first (IMAGES_PER_GPU entries)."""
        cfg = self.config
        B = self.images_per_gpu
        sel, rows = [], []
        for b in range(B):
            l_, i_, m_ = self._rpn_selection(rpn_match[b], b)
            t_ = np.asarray(rpn_bbox[b], np.float32).reshape(-1, 4)
            npos_b = int((m_ == 1).sum())
            if npos_b > t_.shape[0]:
                raise ValueError("%d positive anchors but only %d target rows" % (npos_b, t_.shape[0]))
            sel.append((l_, i_, m_))
            rows.append(t_ if B == 1 else t_[:npos_b])       # batch_pack_graph: the images' first-count rows, concatenated
        lvl, idx, mt = (np.concatenate([s_[k] for s_ in sel]) for k in range(3))
        tdl = np.concatenate(rows)
        n_pos = int((mt == 1).sum())
        gt_norm = np.asarray(gt_norm, np.float32).reshape(B, -1, 4)
        gtc = np.asarray(gt_caps).astype(np.int32).reshape(B, gt_norm.shape[1], -1)
        cap = max(B * int(cfg.RPN_TRAIN_ANCHORS_PER_IMAGE), len(lvl), tdl.shape[0], 1)
        si = self._step_in
        if si is None or si.cap < cap or si.n_gt != gt_norm.shape[1] or si.T != gtc.shape[2] or si.B != B:
            si = self._step_in = StepInputs(self.device, cap, gt_norm.shape[1], gtc.shape[2], B)
            self._invalidate_graphs()                       # captured graphs hold the old views
        if training:
            self._dt_step += 1
        else:
            self._dt_val_step += 1
        opt = self.optimizer
        lr_next = 0.0
        if training and opt is not None:
            t = opt.iterations + 1                          # the update at the end of THIS step
            lr_next = opt.lr * math.sqrt(1.0 - opt.beta_2 ** t) / (1.0 - opt.beta_1 ** t)
        cm = self.caption_model
        scal = np.zeros(4, np.int32)
        scal[0:1] = np.array([lr_next], np.float32).view(np.int32)
        # Philox stream positions as 32-bit words, masked like the eager launches' host arguments (ops.dropout_mask / detection_targets)
        scal[1:3] = np.array([(2 * (cm._drop_step + 1)) & 0xFFFFFFFF,            # this step's recurrent-dropout masks (lstm l: + l)
                              (self._dt_step if training else self._dt_val_step) & 0xFFFFFFFF], np.uint32).view(np.int32)
        si.upload({"counts": np.array([len(lvl), n_pos], np.int32), "lvl": lvl, "idx": idx, "mt": mt, "deltas": tdl[:si.cap],
                   "gt": gt_norm, "gtc": gtc, "scalars": scal})
        sc = si.view("scalars")
        return dict(counts=si.view("counts"), lvl=si.view("lvl"), idx=si.view("idx"), mt=si.view("mt"),
                    deltas=si.view("deltas", torch.float32).view(si.cap, 4), gt=si.view("gt", torch.float32).view(si.B, si.n_gt, 4),
                    gtc=si.view("gtc").view(si.B, si.n_gt, si.T), cap=si.cap, lr_t=sc[0:1].view(torch.float32), drop_offset=sc[1:2], dt_offset=sc[2:3])

    def _rpn_backward(self, p, rpn_up, losses):
        """RPN losses (dense_model.py:1008-1075) and the backward of the RPN branch: gradients of the fused head and of the shared
        3x3 convolution (accumulated over the five pyramid levels) and the data gradients into dP2..dP6, which this call creates
        (zeroed) and returns together with the pyramid maps.  Independent of the detection targets."""
        st = self.store
        w, g = st.w, st.grad
        maps = list(p.P) + [p.P6]
        # dP2..dP6 and the five head gradients start from zero: ten tensors cut from ONE buffer, zeroed by ONE launch of the library's
        # zero-fill kernel (they were ten torch fill launches)
        shapes = [tuple(m.shape) for m in maps] + [tuple(h.shape) for h in p.rpn_heads]
        sizes = [(int(np.prod(sh)) + 3) // 4 * 4 for sh in shapes]
        pool = self._buf("rpn_bwd_zero", (sum(sizes),))
        ops.zero_fill(pool)
        cuts, off = [], 0
        for sh, n in zip(shapes, sizes):
            cuts.append(pool[off:off + int(np.prod(sh))].view(sh))
            off += n
        dP, dheads = cuts[:len(maps)], cuts[len(maps):]
        # ---- RPN losses and their gradients w.r.t. the fused head outputs (selection and counts: this step's StepInputs views)
        ops.rpn_loss_grad(p.rpn_heads, dheads, rpn_up["lvl"], rpn_up["idx"], rpn_up["mt"], rpn_up["deltas"], rpn_up["cap"], losses[0:2],
                          anchors_per_loc=self.A, counts_dev=rpn_up["counts"], batched=True)

        # ---- RPN backward (shared weights over the five levels: gradients accumulate)
        wd_shared = ops.conv_weight_dgrad_pack(w["rpn_conv_shared/kernel"], 3, 3, 256, out=self._buf("wd_shared", (256, 9 * 512)))
        for i, (pm, sh, dh) in enumerate(zip(maps, p.rpn_shared, dheads)):
            _, h_, w_, _ = pm.shape
            acc = i > 0
            ops.conv2d_wgrad(sh, dh, 1, 1, 1, 0, 0, out=g["rpn_head/kernel"], accumulate=acc)          # 20 output channels: fp32
            ops.colsum(dh.view(-1, HEAD_PAD), out=g["rpn_head/bias"], accumulate=acc)
            dsh = self._buf("dsh%d" % i, tuple(sh.shape))          # 1x1 head: its data gradient is a K = 20 GEMM on the packed weights
            ops.gemm(dh.view(-1, HEAD_PAD), w["rpn_head/kernel"], out=dsh.view(-1, 512))
            dshb = self._bf16_slot(dsh, "dsh%d" % i)            # bf16 model: the copy the two bf16 products below read, written by the same pass
            ops.relu_bwd(dsh.view(-1, 512), sh.view(-1, 512), dsh.view(-1, 512), out_bf16=None if dshb is None else dshb.view(-1, 512))
            self._wgrad(pm, dsh, 3, 1, g["rpn_conv_shared/kernel"], accumulate=acc, key="P%d" % i, dy_key="dsh%d" % i)
            ops.colsum(dsh.view(-1, 512), out=g["rpn_conv_shared/bias"], accumulate=acc)
            self._dgrad(dsh, wd_shared, 3, dP[i], residual=dP[i], key="rpn_shared", dy_key="dsh%d" % i)     # dP += dgrad
        return maps, dP

    # ---- backward through trainable ResNet stages (train(layers = "5+" | "4+" | "3+" | "all")) ------------------------
    def _bn_conv_backward(self, conv, bn, dz, bn_out, bn_sub, x, k, stride, key):
        """Backward of y_bn = BN_frozen_stats(conv(x)) given dz = d(loss)/d(y_bn) and y_bn = bn_out - bn_sub (bn_sub may be None):
        writes the gradients of kernel, bias, gamma, beta into the bucket and returns dacc = d(loss)/d(conv output)."""
        p, w, g = self.plan(), self.store.w, self.store.grad
        scale = p._w[conv][1]                                # gamma / sqrt(var + eps), folded by this step's forward
        cout = dz.shape[-1]
        dacc = self._buf(("tb_dacc", key, tuple(dz.shape)), tuple(dz.shape))
        dzn = self._buf(("tb_dzn", tuple(dz.shape)), tuple(dz.shape))
        ops.bn_bwd(dz, bn_out, bn_sub, w[bn + "/gamma"], w[bn + "/beta"], scale, dacc, dzn)
        ops.colsum(dz.view(-1, cout), out=g[bn + "/beta"])
        ops.colsum(dzn.view(-1, cout), out=g[bn + "/gamma"])
        ops.mul(scale, g[bn + "/beta"], g[conv + "/bias"])   # sum(dz * scale) over the pixels = scale * dbeta
        pad = (k - 1) // 2
        if stride == 1:
            self._wgrad(x, dacc, k, pad, g[conv + "/kernel"])
        else:                                                # the strided 1x1 convolutions at a stage's entry
            ops.conv2d_wgrad(x, dacc, k, k, stride, pad, pad, out=g[conv + "/kernel"])
        return dacc

    def _stage_backward(self, p, s, d_out):
        """d_out: gradient w.r.t. the stage's output; returns the gradient w.r.t. its input.  Blocks in reverse order; per block
        out = relu(bn2c(conv2c(m2)) + shortcut), m2 = relu(bn2b(conv2b(m1))), m1 = relu(bn2a(conv2a(x))), shortcut = x or bn1(conv1(x))."""
        w = self.store.w
        for bi in range(len(p.saved[s]) - 1, -1, -1):
            b = p.saved[s][bi]
            cn, bn = b["name"], b["name"].replace("res", "bn", 1)
            x, m1, m2, sc, out, st = b["x"], b["m1"], b["m2"], b["sc"], b["out"], b["stride"]
            mid, cout, cin = m1.shape[-1], out.shape[-1], x.shape[-1]
            ds = ops.relu_bwd(d_out.view(-1, cout), out.view(-1, cout), self._buf(("tb_ds", tuple(out.shape)), tuple(out.shape)).view(-1, cout)).view(out.shape)
            # branch 2c (no activation of its own: its BN output is out - shortcut wherever ds != 0)
            dacc = self._bn_conv_backward(cn + "2c", bn + "2c", ds, out, sc if sc is not None else x, m2, 1, 1, "2c")
            dm2 = ops.gemm(dacc.view(-1, cout), w[cn + "2c/kernel"], out=self._buf(("tb_dm", tuple(m2.shape)), tuple(m2.shape)).view(-1, mid))
            dz = ops.relu_bwd(dm2, m2.view(-1, mid), dm2).view(m2.shape)
            dacc = self._bn_conv_backward(cn + "2b", bn + "2b", dz, m2, None, m1, 3, 1, "2b")
            wd = ops.conv_weight_dgrad_pack(w[cn + "2b/kernel"], 3, 3, mid, out=self._buf(("tb_wd", mid), (mid, 9 * mid)))
            dm1 = self._dgrad(dacc, wd, 3, self._buf(("tb_dm1", tuple(m1.shape)), tuple(m1.shape)))
            dz = ops.relu_bwd(dm1.view(-1, mid), m1.view(-1, mid), dm1.view(-1, mid)).view(m1.shape)
            dacc = self._bn_conv_backward(cn + "2a", bn + "2a", dz, m1, None, x, 1, st, "2a")
            # gradient w.r.t. the block input at the block's OUTPUT resolution: branch 2a + the shortcut
            dxc = self._buf(("tb_dxc", tuple(m1.shape[:3]) + (cin,)), tuple(m1.shape[:3]) + (cin,))
            if sc is None:                                   # identity shortcut (stride 1): dx = W2a^T dacc + ds
                ops.gemm(dacc.view(-1, mid), w[cn + "2a/kernel"], out=dxc.view(-1, cin), residual=ds.view(-1, cout))
            else:
                ops.gemm(dacc.view(-1, mid), w[cn + "2a/kernel"], out=dxc.view(-1, cin))
                dacc1 = self._bn_conv_backward(cn + "1", bn + "1", ds, sc, None, x, 1, st, "1")
                ops.gemm(dacc1.view(-1, cout), w[cn + "1/kernel"], out=dxc.view(-1, cin), accumulate=True)
            if st == 1:
                d_out = dxc
            else:                                            # the 1x1 / stride-2 entry convolutions read every other pixel of x
                d_out = self._buf(("tb_dx", tuple(x.shape)), tuple(x.shape))
                ops.zero_fill(d_out)
                ops.scatter2_add(dxc, d_out)
            # (the next, earlier block turns d_out into its ds before it writes its own dxc, which may be this very buffer)
        return d_out

    def _trunk_backward(self, p, dpre):
        """Gradients of the trainable ResNet stages: C_s receives the data gradient of its FPN lateral plus what the stage above
        passes down; stages are walked top-down (5 -> backbone_from), then the stem when layers = "all"."""
        w, g = self.store.w, self.store.grad
        low = max(self.backbone_from, 2)
        d_c = None
        for s in (5, 4, 3, 2):
            if s < low:
                break
            cmap, name = p.C[s - 2], "fpn_c%dp%d" % (s, s)
            cin = cmap.shape[-1]
            dlat = self._buf(("tb_dC", s), tuple(cmap.shape))          # dC_s = lateral^T dpre_s (+ the gradient from stage s + 1)
            ops.gemm(dpre[s - 2].view(-1, 256), w[name + "/kernel"], out=dlat.view(-1, cin), residual=None if d_c is None else d_c.view(-1, cin))
            d_c = self._stage_backward(p, s, dlat)
        if self.backbone_from == 1:                              # the stem: maxpool 3x3/2 <- relu <- bn_conv1 <- conv1 7x7/2 (ZeroPadding 3)
            sv = p.saved[1]
            c1, pooled = sv["c1"], sv["pooled"]
            dc1 = ops.maxpool3x3s2_same_bwd(c1, pooled, d_c, out=self._buf("tb_dc1", tuple(c1.shape)))
            dz = ops.relu_bwd(dc1.view(-1, 64), c1.view(-1, 64), dc1.view(-1, 64)).view(c1.shape)
            scale = p._w["conv1"][1]
            dacc = self._buf("tb_dacc_stem", tuple(c1.shape))
            dzn = self._buf("tb_dzn_stem", tuple(c1.shape))
            ops.bn_bwd(dz, c1, None, w["bn_conv1/gamma"], w["bn_conv1/beta"], scale, dacc, dzn)
            ops.colsum(dz.view(-1, 64), out=g["bn_conv1/beta"])
            ops.colsum(dzn.view(-1, 64), out=g["bn_conv1/gamma"])
            ops.mul(scale, g["bn_conv1/beta"], g["conv1/bias"])
            x64 = ops.mold_image_padded(p.images, p.mean_pixel, self._buf("tb_x64", tuple(p.images.shape[:3]) + (64,)))   # Cin % 64 == 0 for the wgrad kernel
            gw = ops.conv2d_wgrad(x64, dacc, 7, 7, 2, 3, 3, out=self._buf("tb_gw_stem", (64, 49 * 64)))
            gk = g["conv1/kernel"].view(64, 7, 8, 4)                       # the stem's packed layout (pads stay zero)
            ops.zero_fill(g["conv1/kernel"])
            gk[:, :, :7, :3].copy_(gw.view(64, 7, 7, 64)[..., :3])

    @property
    def last_targets(self):
        """The last step's detection targets as host values: dict(rois [R,4], caps [R,T], npos, nneg) -- with IMAGES_PER_GPU > 1 the
        arrays carry the image axis first ([B,R,4], [B,R,T]) and npos / nneg are int arrays [B].  The step itself leaves them on the
        device; reading this property is what copies them (and waits for the step)."""
        t = self._last_targets
        if t is None or isinstance(t, dict):
            return t
        rois, caps, counts = t
        c, r, k = counts.cpu().numpy(), rois.cpu().numpy(), caps.cpu().numpy()
        if c.shape[0] == 1:
            return dict(rois=r[0], caps=k[0], npos=int(c[0, 0]), nneg=int(c[0, 1]))
        return dict(rois=r, caps=k, npos=c[:, 0].astype(int), nneg=c[:, 1].astype(int))

    def forward_backward(self, inputs, shuffle="rng", backward=True, targets=None, trunk_done=False):
        """Losses and gradients of one step's IMAGES_PER_GPU images into the flat gradient bucket (no optimizer step).
        Returns the device tensor [rpn_class_loss, rpn_bbox_loss, imgcap_loss, reg_loss].
        backward=False: the forward graph only (validation, Keras' test_on_batch): no gradient is computed, the gradient bucket is
        left alone, no recurrent dropout, and the detection-target sample is drawn from a generator of its own so that an
        evaluation between two train steps does not change the training run.
        targets = (rois [R,4] normalised, caps [R,T]): the DetectionTargetLayer's sample handed in by the caller instead of drawn (one
        image per step; parity tests check a second model against the oracle result of a sample another model drew)."""
        images, _meta, rpn_match, rpn_bbox, gt_caps, gt_boxes = inputs[:6]
        self._targets_given = None if targets is None else (np.asarray(targets[0], np.float32), np.asarray(targets[1]))
        p = self.plan()
        gt_norm = self._check_batch(p, images, gt_boxes)
        # ---- this step's host inputs go to the device FIRST (GT boxes, GT captions, the RPN selection, the step scalars: one asynchronous
        # copy, StepInputs): nothing the host contributes may sit in the middle of the step
        rpn_up = self._step_uploads(p, rpn_match, rpn_bbox, gt_norm, gt_caps, backward)

        # ---- forward: backbone + FPN + RPN (the plan's hipGraph), then everything behind the encoder
        if trunk_done:                                       # the backbone pass of these images has run on this plan (JointTrainPipeline)
            p.forward_top()
        else:
            p.forward(self._images_u8(images))
        return self._after_encoder(p, rpn_up, shuffle, backward, gt_caps, gt_norm)

    def _check_batch(self, p, images, gt_boxes):
        """The step takes exactly IMAGES_PER_GPU images (static graph, like the reference's KL.Input batch); returns the GT boxes
        normalised by the molded image's size, float32 [B, G, 4]."""
        if len(images) != self.images_per_gpu:
            raise ValueError("%d image(s) handed to a model built for IMAGES_PER_GPU = %d" % (len(images), self.images_per_gpu))
        return (np.asarray(gt_boxes, np.float32).reshape(self.images_per_gpu, -1, 4) / np.array([p.H, p.W, p.H, p.W], np.float32)).astype(np.float32)

    def _after_encoder(self, p, rpn_up, shuffle, backward, gt_caps, gt_norm, fuse_reg=False):
        """The step behind the encoder pass: proposals, detection targets, RoIAlign, head + decoder, the four losses and (backward)
        every gradient into the flat bucket.  With device-side targets (shuffle None / "rng") nothing in here depends on a host value
        that changes from step to step -- counts, stream positions and lr_t are device words of StepInputs -- so train_on_batch_device
        captures it (plus the optimizer) as ONE hipGraph."""
        cfg, st, cm = self.config, self.store, self.caption_model
        w, g = st.w, st.grad
        dev = self.device
        H, W = p.H, p.W
        up = lambda a, dt=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
        given = getattr(self, "_targets_given", None)
        self._targets_given = None
        device_targets = (shuffle is None or shuffle == "rng") and given is None
        gt_dev, gtc_dev = rpn_up["gt"], rpn_up["gtc"]
        B = self.images_per_gpu
        self._bf16_cache = {}
        losses = self._buf("losses", (4,))
        # The RPN branch's backward (RPN losses, head / shared-convolution weight gradients, data gradients into dP2..dP6: ~1.2 ms of
        # large kernels) needs only the encoder's outputs and the step's RPN targets, while the chain proposals -> top-k -> NMS scan ->
        # detection targets -> RoIAlign -> head + LSTM forward is a string of small, latency-bound launches (the NMS scan alone is one
        # wave for 0.28 ms).  They run side by side: the RPN backward on a second stream, forked here and joined before the RoIAlign
        # backward adds into dP.  (Captured: two branches of the step's hipGraph.)
        overlap_dp = self.grad_sync is not None and hasattr(self.grad_sync, "ready") and getattr(self.grad_sync, "world", 1) > 1
        # Data parallel (round 6): the same fork.  The RPN ranges' regulariser pass and their all-reduce are issued from INSIDE the side
        # stream's context right behind the RPN backward -- torch.distributed orders a collective behind the stream that is current when
        # it is issued -- so the exchange of the RPN gradients starts while the main stream still runs proposals -> targets -> decoder.
        fork = backward and device_targets and self.use_side_stream
        self._reg_done = []
        early = None
        if backward and overlap_dp:
            coef_, mask_ = self._masks()

            def early(lo, hi):                              # regulariser gradient + mask of one layer range, then it may travel
                ops.l2_reg(st.flat[lo:hi], coef_[lo:hi], st.flat_grad[lo:hi], mask=None if mask_ is None else mask_[lo:hi])
                self._reg_done.append((lo, hi))

        def rpn_ranges_travel():
            for layer in ("rpn_conv_shared", "rpn_head"):
                lo, hi = st.layer_range(layer)
                early(lo, hi)
                self.grad_sync.ready(st.flat_grad, lo, hi)
        maps = dP = None
        if fork:
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=dev)
            cur = torch.cuda.current_stream(dev)
            self._side_stream.wait_stream(cur)
            with torch.cuda.stream(self._side_stream):
                maps, dP = self._rpn_backward(p, rpn_up, losses)
                if overlap_dp:
                    rpn_ranges_travel()
        proposals = p.proposals()
        R = cfg.TRAIN_ROIS_PER_IMAGE
        if device_targets:
            # DetectionTargetLayer on the device (dc_detection_targets_f32): IoU, the >= 0.5 / < 0.5 split, the shuffle (Philox keys drawn
            # from (model seed, step): reproducible, where tf.random_shuffle is not), the 1:2 sample and the caption gather.  Nothing
            # comes back to the host: counts travel as device words, every shape downstream is static (TRAIN_ROIS_PER_IMAGE rows).
            seed = None if shuffle is None else ((self._seed + (0 if backward else 1)) * 2654435761 + 12345 + self._dt_rank * 0x9E3779B9) & 0xFFFFFFFF
            T = int(np.asarray(gt_caps).shape[-1])
            # one launch per image (the reference's utils.batch_slice over DetectionTargetLayer, dense_model.py:531-572), each image with
            # its own Philox key; the sampled RoIs, captions and counts of the batch are rows of ONE set of buffers
            rois_d, caps_d, counts_d = self._buf("dt_rois", (B, R, 4)), self._buf("dt_caps", (B, R, T), torch.int32), self._buf("dt_counts", (B, 2), torch.int32)
            for b in range(B):
                ops.detection_targets(proposals[b], gt_dev[b], gtc_dev[b], R, cfg.ROI_POSITIVE_RATIO,
                                      seed=None if seed is None else (seed + b * 0x85EBCA6B) & 0xFFFFFFFF, offset=0, offset_dev=rpn_up["dt_offset"],
                                      out=(rois_d[b], caps_d[b], counts_d[b]))
            self._last_targets = (rois_d, caps_d, counts_d)
            boxes = rois_d
            feats = p.roi_features(boxes_norm=boxes, out=self._buf("feats", (B, R, cfg.POOL_SIZE, cfg.POOL_SIZE, 256)))
            feats = feats.view(B * R, cfg.POOL_SIZE, cfg.POOL_SIZE, 256)
            # the caption loss is the mean over every live position of the BATCH (imgcap_caption_loss_graph gathers over all images,
            # dense_model.py:936-946): the tables' row weights are 1 / (live positions of all B * R captions)
            R_all = B * R
            tables = ops.caption_tables(caps_d.view(R_all, T), out=(self._buf("ct_ids", (T * R_all,), torch.int32), self._buf("ct_mask", (T * R_all,), torch.uint8),
                                                                    self._buf("ct_tg", (T * R_all,), torch.int32), self._buf("ct_rw", (T * R_all,))))
            if cm._prefix_rows(backward):
                # DROPOUT_ROWS = 'prefix' (one mask per (RoI, prefix) row, as the reference's TimeDistributed graph draws them): the T-fold
                # prefix tables are built on the host, so this non-default mode reads the sampled captions back (one synchronisation)
                cm._drop_offset_dev = None
                caps_h = caps_d.view(R_all, T).cpu().numpy()
                tg_h = caption_targets(caps_h)
                live = (tg_h > 0).astype(np.float32)
                loss_rows, _ = cm._forward_train(feats, caps_h, tg_h, want_grad=backward, row_weights=live / max(float(live.sum()), 1.0),
                                                 keras_sparse=True)
            else:
                cm._drop_offset_dev = rpn_up["drop_offset"]
                try:
                    loss_rows, _ = cm._forward_train(feats, None, want_grad=backward, keras_sparse=True, device_tables=tables + (R_all, T))
                finally:
                    cm._drop_offset_dev = None                  # a later standalone step of the shared caption model draws from ITS counter
            if backward and not fork:
                maps, dP = self._rpn_backward(p, rpn_up, losses)
        else:
            # a caller-supplied permutation (shuffle = callable): the sample is drawn on the host, as until round 3.  The proposals start
            # their way to the host first; the RPN branch's backward is enqueued behind that copy, so the GPU works while the host samples.
            mix = shuffle
            if B != 1:
                raise ValueError("a caller-supplied shuffle samples on the host: one image per step only (use shuffle=None or 'rng')")
            if given is not None:
                props_np = None
                if backward:
                    maps, dP = self._rpn_backward(p, rpn_up, losses)
            elif backward:
                host_props = self._pinned("props", tuple(proposals[0].shape))
                host_props.copy_(proposals[0], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                maps, dP = self._rpn_backward(p, rpn_up, losses)
                ev.synchronize()
                props_np = host_props.numpy()
            else:
                props_np = proposals[0].cpu().numpy()
            cm._drop_offset_dev = None
            if given is not None:
                rois, caps = given
                npos = int((np.asarray(caps)[:, 1:] > 0).any(axis=1).sum())
                nneg = int((np.abs(rois).sum(axis=1) > 0).sum()) - npos
            else:
                rois, caps, npos, nneg = detection_targets(props_np, gt_caps[0], gt_norm[0], cfg, mix)
            self._last_targets = dict(rois=rois, caps=caps, npos=npos, nneg=nneg)
            boxes = up(rois[None])
            feats = p.roi_features(boxes_norm=boxes, out=self._buf("feats", (1, R, cfg.POOL_SIZE, cfg.POOL_SIZE, 256)))
            tg = caption_targets(caps)
            live = (tg > 0).astype(np.float32)
            count = float(live.sum())
            loss_rows, _ = cm._forward_train(feats[0], caps, tg, want_grad=backward, row_weights=live / max(count, 1.0), keras_sparse=True)
        ops.mean(loss_rows, out=losses[2:3])                 # x rows below: the weights already carry 1/count
        self._loss_scale = float(loss_rows.numel())
        if not backward:
            # RPN losses need the heads only (their gradient goes to scratch), the regulariser the weights only
            scratch = [self._buf("dhead%d" % i, tuple(h.shape)) for i, h in enumerate(p.rpn_heads)]
            for t in scratch:
                ops.zero_fill(t)
            ops.rpn_loss_grad(p.rpn_heads, scratch, rpn_up["lvl"], rpn_up["idx"], rpn_up["mt"], rpn_up["deltas"], rpn_up["cap"], losses[0:2],
                              anchors_per_loc=self.A, counts_dev=rpn_up["counts"], batched=True)
            coef, _ = self._masks()
            ops.l2_reg(st.flat, coef, None, loss=losses[3:4])
            return losses

        # ---- backward: decoder + head -> RoI features -> pyramid
        overlap = overlap_dp
        if overlap:
            cm.before_sync, cm.grad_sync, cm.overlap_sync = early, self.grad_sync, True
        else:
            cm.before_sync, cm.overlap_sync = None, False
        if overlap and not fork:                             # serial order: the RPN's gradients are final here, they travel first
            rpn_ranges_travel()
        # (tried and measured without gain, round 4: the decoder's weight-gradient GEMMs on the side stream beside its LSTM backward
        # chain -- 8.77 ms against 8.68 captured, 8.57 against 8.58 eager)
        dX = cm._backward(want_dx=True)
        if fork:
            torch.cuda.current_stream(dev).wait_stream(self._side_stream)      # join: dP, the RPN gradients and losses[0:2] are complete
            # (a SECOND fork was tried and removed, round 4: the decoder's / head's share of the regulariser pass -- HBM-bound, 290 of the
            # bucket's 308 MB -- on the side stream beside the RoIAlign / FPN backward: 8.59 ms against 8.58 eager, 9.07 against 8.68
            # captured -- every extra branch costs the graph replay more than the overlap returns)
        # dP already holds the RPN branch's data gradients; the RoI features' gradient is added on top (a fixed-order gather per pyramid pixel: reproducible)
        ops.roi_align_pyramid_bwd(dP[:4], boxes, float(H * W), dX.view(B, R, cfg.POOL_SIZE, cfg.POOL_SIZE, 256), cfg.POOL_SIZE)
        ops.scatter2_add(dP[4], dP[3])                       # P6 = MaxPooling2D(1, strides=2)(P5)

        # ---- FPN backward (data parallel: every layer's gradient range starts its all-reduce as soon as it is complete)
        def announce(layer):
            if overlap:
                lo, hi = st.layer_range(layer)
                early(lo, hi)
                self.grad_sync.ready(st.flat_grad, lo, hi)
        dpre = []
        for i in range(4):
            name = "fpn_p%d" % (i + 2)
            wd = ops.conv_weight_dgrad_pack(w[name + "/kernel"], 3, 3, 256, out=self._buf("wd_" + name, (256, 9 * 256)))
            _, h_, w_, _ = dP[i].shape
            self._wgrad(p.pre[i], dP[i], 3, 1, g[name + "/kernel"], dy_key="dP%d" % i)
            ops.colsum(dP[i].view(-1, 256), out=g[name + "/bias"])
            announce(name)
            # (dpre[0] is final here: its bf16 copy comes out of the same epilogue; the coarser ones still receive the top-down sums below)
            dpre.append(self._dgrad(dP[i], wd, 3, self._buf("dpre%d" % i, tuple(dP[i].shape)), key=name, dy_key="dP%d" % i,
                                    out_key="dpre0" if i == 0 else None))
        for i in range(3):                                   # pre[k] = upsample(pre[k+1]) + lateral(C_k)
            ops.downsample2x_sum(dpre[i], out=dpre[i + 1], accumulate=True, out_bf16=self._bf16_slot(dpre[i + 1], "dpre%d" % (i + 1)))
        for i, cmap in enumerate(p.C):
            name = "fpn_c%dp%d" % (i + 2, i + 2)
            self._wgrad(cmap, dpre[i], 1, 0, g[name + "/kernel"], dy_key="dpre%d" % i)
            ops.colsum(dpre[i].view(-1, 256), out=g[name + "/bias"])
            announce(name)
        if self.backbone_from is not None:
            self._trunk_backward(p, dpre)
        if fuse_reg and not self._reg_done:
            # single-GPU train step: the regulariser's gradient, the trainable mask, the clip norm and losses[3] are the optimizer's
            # passes (Adam.apply(reg=...)): flat_grad keeps the plain loss gradient and is not rewritten here
            self._loss_scale = float(loss_rows.numel())
            return losses
        coef, mask = self._masks()
        if self._reg_done:                                  # the ranges that did not go early (FPN / RPN, anything the decoder skipped)
            n, pos = st.flat.numel(), 0
            for lo, hi in sorted(self._reg_done) + [(n, n)]:
                if lo > pos:
                    ops.l2_reg(st.flat[pos:lo], coef[pos:lo], st.flat_grad[pos:lo], mask=None if mask is None else mask[pos:lo])
                pos = max(pos, hi)
            ops.l2_reg(st.flat, coef, None, loss=losses[3:4])      # the loss term alone (weights only)
        else:
            ops.l2_reg(st.flat, coef, st.flat_grad, loss=losses[3:4], mask=mask)          # one pass: trainable subset, regulariser gradient, loss term
        self._loss_scale = float(loss_rows.numel())
        return losses

    def _loss_list(self, losses):
        """losses: the step's raw loss terms [rpn_class, rpn_bbox, imgcap (unscaled), reg], a device tensor or its host copy."""
        v = (losses.cpu().numpy() if isinstance(losses, torch.Tensor) else np.asarray(losses)).astype(np.float64)
        out = dict(rpn_class_loss=v[0], rpn_bbox_loss=v[1], imgcap_loss=v[2] * self._loss_scale, reg_loss=v[3])
        out["loss"] = out["rpn_class_loss"] + out["rpn_bbox_loss"] + out["imgcap_loss"] + out["reg_loss"]
        return out

    def _losses_to_api(self, v):
        self.last_losses = d = self._loss_list(v)
        return [d["loss"], d["rpn_class_loss"], d["rpn_bbox_loss"], d["imgcap_loss"]]

    def train_on_batch_device(self, inputs, targets=None, trunk_done=False):
        """One optimizer step; the raw loss terms as a float32 device tensor [4] (the loss all-reduce of ParallelModel and the
        epoch sums of train() work on it; _losses_to_api makes the Keras return value from its host copy).
        trunk_done: plan().forward_trunk() has already run on these images (pipeline.JointTrainPipeline): the step starts at the FPN."""
        assert self.mode == "training", "Create model in training mode."
        if self.optimizer is None:
            raise RuntimeError("compile(learning_rate) first")
        # a gradient hook without `.world` (a custom callable) counts as an exchange: it takes the eager path and is called
        world = 1 if self.grad_sync is None else getattr(self.grad_sync, "world", None)
        cm = self.caption_model
        if world != 1 or cm._prefix_rows(True):
            # eager, serial: the data-parallel step (its collectives are issued from Python as layer groups finish), DROPOUT_ROWS='prefix'
            losses = self.forward_backward(inputs, trunk_done=trunk_done)
            scale = self.grad_sync(self.store.flat_grad) if self.grad_sync is not None else 1.0
            self.optimizer.apply(self.store, grad_scale=scale)
            return losses
        # ---- single GPU: [one async upload] -> [encoder hipGraph] -> [step hipGraph: proposals .. losses .. gradients .. AMSGrad]
        images, _meta, rpn_match, rpn_bbox, gt_caps, gt_boxes = inputs[:6]
        p = self.plan()
        dev = self.device
        gt_norm = self._check_batch(p, images, gt_boxes)
        rpn_up = self._step_uploads(p, rpn_match, rpn_bbox, gt_norm, gt_caps, True)
        self._choose_step_path()
        timing = self._step_path_auto and self._use_step_graph
        if timing:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if trunk_done:
            p.forward_top()
        else:
            p.forward(self._images_u8(images))
        opt = self.optimizer

        def timed(kind, out):
            if timing and len(self._path_events[kind]) < 2:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self._path_events[kind].append((e0, e1))
            return out

        def body():
            segs = self._reg_segments()
            losses = self._after_encoder(p, rpn_up, "rng", True, gt_caps, gt_norm, fuse_reg=segs is not None)
            opt.apply(self.store, grad_scale=1.0, lr_t_dev=rpn_up["lr_t"], reg=segs, reg_loss=None if segs is None else losses[3:4])
            return losses

        def step():
            if not self.use_step_graph:
                return body()
            key = ("train", float(cm.recurrent_dropout or 0.0), opt.baked_key())      # (with dropout the mask kernels are launches of the step)
            graph = self._graphs.get(key)
            if graph is not None:
                graph.replay()
                opt.iterations += 1                              # what the captured Python did once: the host-side counters
                if float(cm.recurrent_dropout or 0.0) > 0.0:
                    cm._drop_step += 1
                return timed("graph", self._graph_out[key])
            warm_steps = 3 if self._step_path_auto else 2        # (automatic mode: the first eager step allocates; the next two are timed)
            if self._graph_warm.get(key, 0) < warm_steps:
                self._graph_warm[key] = n_warm = self._graph_warm.get(key, 0) + 1  # eager: sizes every buffer and workspace, builds the masks
                out = body()
                return timed("eager", out) if n_warm > 1 else out
            saved = (opt.iterations, cm._drop_step)
            try:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with ops.no_gc_during_capture(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    out = body()
            except RuntimeError as e:                            # a capture error (torch raises RuntimeError): stay eager from here on
                import warnings
                warnings.warn("joint step: hipGraph capture failed (%s); running eagerly" % (repr(e)[:200],))
                opt.iterations, cm._drop_step = saved
                self.use_step_graph = False
                self.step_graph_fallback = repr(e)[:200]         # queryable (bench.py reports it): the eager step is a different schedule
                # the failed capture may have pulled the side stream in (between fork and join): it is in an invalidated-capture state,
                # so the eager retry forks onto a fresh one
                torch.cuda.synchronize()
                if self._side_stream is not None:
                    ops.WORKSPACE.release(self._side_stream)
                self._side_stream = None
                return body()
            self._graphs[key], self._graph_out[key] = graph, out
            graph.replay()                                       # (capture records, it does not run: this is the step itself)
            return out
        return step()

    def train_on_batch(self, inputs, targets=None):
        """One optimizer step; returns [loss, rpn_class_loss, rpn_bbox_loss, imgcap_loss] like the compiled Keras model
        (metrics_names order, :1722-1730)."""
        return self._losses_to_api(self.train_on_batch_device(inputs, targets))

    def test_on_batch_device(self, inputs, targets=None):
        return self.forward_backward(inputs, backward=False)

    def test_on_batch(self, inputs, targets=None):
        """Forward only (Keras test_on_batch): losses of the batch, no gradient, no effect on the training state."""
        return self._losses_to_api(self.test_on_batch_device(inputs, targets))

    # ---- inference -------------------------------------------------------------------------
    def mold_inputs(self, images):
        molded, metas, windows = [], [], []
        for image in images:
            m, window, scale, padding = utils.resize_image(image, min_dim=self.config.IMAGE_MIN_DIM, max_dim=self.config.IMAGE_MAX_DIM,
                                                           padding=self.config.IMAGE_PADDING)
            molded.append(m)                                # mean subtraction happens on the GPU
            metas.append(utils.compose_image_meta(0, image.shape, window))
            windows.append(window)
        return np.stack(molded), np.stack(metas), np.stack(windows)

    def generate_captions(self, images, verbose=0, return_probabilities=True):
        """The inference graph (:1602-1622) + generate_captions (:1964-2003): RPN proposals (POST_NMS_ROIS_INFERENCE) ->
        RoI features -> greedy ROICaptionInferenceLayer -> GenerationMatchLayer -> boxes in the original image.
        Returns [{'rois': int32 [K,4], 'captions': f32 [K,T,V] word probabilities, 'ids': int32 [K,T]}]; with
        return_probabilities=False the [K,T,V] tensor (3 GB at 1000 RoIs x 15 x 50 000) stays on the GPU and is dropped."""
        assert self.mode == "inference", "Create model in inference mode."
        assert len(images) == self.config.BATCH_SIZE, "len(images) must be equal to BATCH_SIZE"
        molded, metas, windows = self.mold_inputs(images)
        p = self.plan()
        p.forward(torch.as_tensor(molded))
        proposals = p.proposals()
        self.last_proposals = proposals
        feats = p.roi_features(boxes_norm=proposals)
        results = []
        for b in range(len(images)):
            probs, ids, word_scores = self.caption_model.generate(feats[b], return_probabilities=return_probabilities)
            boxes, keep = refine_generations(proposals[b].cpu().numpy(), word_scores, windows[b], self.config)
            final, ok = unmold_generations(boxes, images[b].shape, windows[b])
            keep = keep[ok]
            out = {"rois": final[ok], "ids": ids[keep]}
            if return_probabilities:
                out["captions"] = probs[keep]
            results.append(out)
        return results

    # ---- training loop ----------------------------------------------------------------------
    def train(self, train_dataset, val_dataset, learning_rate, epochs, layers):
        """fit_generator over data_generator with a checkpoint per epoch (:1810-1888)."""
        assert self.mode == "training", "Create model in training mode."
        layers = self.LAYER_REGEX.get(layers, layers)
        cfg = self.config
        train_generator = data_generator(train_dataset, cfg, shuffle=True, batch_size=cfg.BATCH_SIZE)
        val_generator = data_generator(val_dataset, cfg, shuffle=True, batch_size=cfg.BATCH_SIZE, augment=False)
        self.set_trainable(layers)
        self.compile(learning_rate)
        val_batch = next(val_generator)[0]
        names = ("loss",) + self.LOSS_NAMES
        history = []
        # Frozen ResNet (the script's layers: 'heads'-like sets): the backbone pass of batch i + 1 runs beside the rest of batch i's step
        # (pipeline.JointTrainPipeline: same updates bit for bit, tests/test_gpu_models.py; DCAP_JOINT_PIPELINE=0 keeps the serial loop).
        pipe = None
        if self.backbone_from is None and os.environ.get("DCAP_JOINT_PIPELINE", "1") != "0":
            from .pipeline import JointTrainPipeline
            pipe = JointTrainPipeline(self._outer)
        for epoch in range(self.epoch, epochs):
            acc = None                                           # raw loss terms summed on the device: one host copy per epoch
            for i in range(cfg.STEPS_PER_EPOCH + (1 if pipe is not None else 0)):
                if pipe is None:
                    step = self._outer.train_on_batch_device(next(train_generator)[0])
                elif i < cfg.STEPS_PER_EPOCH:
                    step = pipe.step(next(train_generator)[0])   # the losses of the batch before (None on the epoch's first call)
                else:
                    step = pipe.flush()                          # the epoch's last batch: every update is in before validation and checkpoint
                if step is not None:
                    acc = step.clone() if acc is None else acc.add_(step)
            logs = dict(zip(names, self._losses_to_api((acc / cfg.STEPS_PER_EPOCH).cpu().numpy())))
            # the reference validates on ONE fixed batch too: validation_data=next(val_generator) (:1878)
            logs.update({"val_" + n: v for n, v in zip(names, self._outer.test_on_batch(val_batch))})
            history.append(logs)
            if self.is_chief:
                print("Epoch %d/%d - " % (epoch + 1, epochs) + " - ".join("%s: %.4f" % kv for kv in sorted(logs.items())))
                os.makedirs(self.log_dir, exist_ok=True)
                self.save_weights(self.checkpoint_path.format(epoch=epoch + 1))      # Keras ModelCheckpoint numbers epochs from 1
            if self.grad_sync is not None:
                self.grad_sync.barrier()                     # every rank sees the finished checkpoint before going on
        self.epoch = max(self.epoch, epochs)
        return history
