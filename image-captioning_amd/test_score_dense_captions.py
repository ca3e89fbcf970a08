"""The caption-generation half of the reference's evaluation script (evaluate_models/test_score_dense_captions.py): the greedy decode of
every ground-truth region of an image through the v2 decoder on the device, then the script's own NumPy post-processing of the boxes.
SURVEY 8(f2) cites :207-283; the language metrics behind it (SPICE / METEOR / ... : Java jars under eval/) are outside the hot path.

This file keeps the reference module's name and the evaluator's method names.  The NumPy pieces are restated so that their outputs
equal the reference functions' own outputs on the same inputs -- tests/test_golden_reference.py checks them against vectors made by
running the reference's functions (tests/golden/make_reference_vectors.py) -- INCLUDING what differs from the training-side copies:

  * evaluate_models/utils.py computes `2 * intersection / (area_a + area_b)` (the Dice coefficient) where the two other copies of
    utils.py compute intersection / union (:30-48); NMS and the ground-truth merge of the evaluation run on that overlap;
  * refine_generations (:245-283) leaves the boxes as handed in (normalised or not, no clipping, no rounding) and orders the kept
    boxes by `argsort(scores[keep])[::-1]`, i.e. a REVERSED STABLE ascending sort: among equal scores the later one comes first;
  * unmold_generations (:158-183) returns boxes only and drops nothing.
"""
import numpy as np

from . import utils as _utils
from .preprocess import decode_word
from .text_generation_model_v2 import pad_sequences


def compute_iou(box, boxes, box_area, boxes_area):
    """evaluate_models/utils.py:30-48 -- the evaluation's overlap: 2 * intersection / (sum of the two areas)."""
    y1 = np.maximum(box[0], boxes[:, 0])
    y2 = np.minimum(box[2], boxes[:, 2])
    x1 = np.maximum(box[1], boxes[:, 1])
    x2 = np.minimum(box[3], boxes[:, 3])
    inter = np.maximum(x2 - x1, 0) * np.maximum(y2 - y1, 0)
    return 2 * inter / (box_area + boxes_area)


def compute_overlaps(boxes1, boxes2):
    """[len(boxes1), len(boxes2)] float64 matrix of the overlap above (evaluate_models/utils.py:51-67)."""
    a1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    a2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    out = np.zeros((boxes1.shape[0], boxes2.shape[0]))
    for i in range(boxes2.shape[0]):
        out[:, i] = compute_iou(boxes2[i], boxes1, a2[i], a1)
    return out


def non_max_suppression(boxes, scores, threshold):
    """Greedy NMS in the boxes' own float dtype (integer boxes -> float32), best score first (`argsort()[::-1]`: among equal scores
    the LATER index leads), suppressing overlap > threshold with the evaluation's overlap (evaluate_models/utils.py:70-104)."""
    assert boxes.shape[0] > 0
    if boxes.dtype.kind != "f":
        boxes = boxes.astype(np.float32)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    ixs = scores.argsort()[::-1]
    pick = []
    while len(ixs) > 0:
        i, rest = ixs[0], ixs[1:]
        pick.append(i)
        ixs = rest[~(compute_iou(boxes[i], boxes[rest], area[i], area[rest]) > threshold)]
    return np.array(pick, dtype=np.int32)


def generate_features(image, dataset, image_id, model):
    """:40-47 -- boxes and [N,7,7,256] RoI features of the image's ground-truth regions from the feature model."""
    rois, _ = dataset.load_captions_and_rois(image_id)
    results = model.generate_captions([image], np.expand_dims(rois, axis=0), verbose=0)
    return results[0]['rois'], results[0]['features']


class DenseCaptioningEvaluator(object):
    """DenseCaptioningEvaluator(model, feature_model, text_metrics, dataset, id_to_word, word_to_id, config, model_name) (:50-83)."""

    def __init__(self, model, feature_model, text_metrics, dataset, id_to_word, word_to_id, config, model_name):
        self.model, self.feature_model, self.text_metrics = model, feature_model, text_metrics
        self.dataset, self.id_to_word, self.word_to_id = dataset, id_to_word, word_to_id
        self.config, self.model_name = config, model_name
        self.predictions = self.ground_truths = None

    @staticmethod
    def merge_boxes(boxes, captions, thresh):
        """:85-129 -- ground-truth boxes overlapping by more than `thresh` become one box (integer mean of the group) carrying all of
        the group's captions; groups are taken greedily, the box with the most partners first (`argmax`: the first such box)."""
        assert thresh > 0
        ov = compute_overlaps(boxes, boxes)
        groups = []
        while True:
            good = (ov > thresh).astype(int)
            count = good.sum(axis=0)
            if count.max() == 0:
                break
            members = np.nonzero(good[np.argmax(count)])
            groups.append(members)
            ov[members] = 0
            ov[:, members] = 0
        new_boxes = np.zeros((len(groups), 4))
        old = np.array(captions)
        new_captions = []
        for i, m in enumerate(groups):
            b = boxes[m]
            new_boxes[i] = np.mean(b, axis=0).astype(int) if b.shape[0] > 1 else b[0]
            new_captions.append(old[m].tolist())
        return new_boxes, new_captions

    def unmold_generations(self, boxes, image_shape, window):
        """:158-183 -- boxes of the molded image to the original image's pixels (truncated to int32); nothing is dropped."""
        scale = min(image_shape[0] / (window[2] - window[0]), image_shape[1] / (window[3] - window[1]))
        shifts = np.array([window[0], window[1], window[0], window[1]])
        return np.multiply(boxes - shifts, np.array([scale] * 4)).astype(np.int32)

    def refine_generations(self, rois, captions, window, config):
        """:245-283 -- caption score = sum over positions of log(max word probability); NMS(DETECTION_NMS_THRESHOLD) on the boxes AS
        HANDED IN; the kept boxes in descending score order as `argsort(...)[::-1]` gives it; the best DETECTION_MAX_INSTANCES.
        Returns (boxes[keep], captions[keep])."""
        scores = np.sum(np.log(np.max(captions, axis=2)), axis=1)
        keep = non_max_suppression(rois, scores, config.DETECTION_NMS_THRESHOLD)
        keep = keep[np.argsort(scores[keep])[::-1][:config.DETECTION_MAX_INSTANCES]]
        return rois[keep], captions[keep]

    def clip_to_window(self, window, boxes):
        """:285-294 -- in place, like the reference."""
        boxes[:, 0] = np.maximum(np.minimum(boxes[:, 0], window[2]), window[0])
        boxes[:, 1] = np.maximum(np.minimum(boxes[:, 1], window[3]), window[1])
        boxes[:, 2] = np.maximum(np.minimum(boxes[:, 2], window[2]), window[0])
        boxes[:, 3] = np.maximum(np.minimum(boxes[:, 3], window[3]), window[1])
        return boxes

    def decode_rois(self, features):
        """The reference's inner loop (:213-224) for all RoIs of an image: start from the all-zero word, PADDING_SIZE - 1 times predict
        the next word's distribution from the argmax ids so far.  [N, PADDING_SIZE - 1, V] probabilities.  The reference calls
        model.predict once per RoI and step; here every step is ONE device pass over all N RoIs (rows are independent)."""
        T = self.config.PADDING_SIZE
        feats = np.asarray(features)
        ids = np.zeros((feats.shape[0], 1), np.int64)                   # argmax of the zero start word
        rows = []
        for _ in range(T - 1):
            probs = self.model.predict([feats, pad_sequences(ids.tolist(), T)])
            rows.append(probs)
            ids = np.concatenate([ids, np.argmax(probs, axis=1)[:, None]], axis=1)
        return np.stack(rows, axis=1)

    def get_generated_captions(self, num_images, images):
        """:185-243 without the pickle cache: per image (boxes int32 [K,4], caption strings [K], log probabilities [K])."""
        boxes, captions, log_probs = [], [], []
        for i in range(num_images):
            im_id = self.dataset._image_ids[i]
            _, window, _, _ = _utils.resize_image(images[i], min_dim=self.config.IMAGE_MIN_DIM, max_dim=self.config.IMAGE_MAX_DIM,
                                                  padding=self.config.IMAGE_PADDING)
            img_boxes, img_features = generate_features(images[i], self.dataset, im_id, self.feature_model)
            rois, img_caps = self.refine_generations(img_boxes, self.decode_rois(img_features), window, self.config)
            rois = self.unmold_generations(rois, images[i].shape, window)
            texts = []
            for cap in img_caps:
                text = ' '.join(decode_word(c, self.id_to_word) for c in cap)
                texts.append(text.split(' .', maxsplit=1)[0])
            boxes.append(rois)
            log_probs.append(np.sum(np.log(np.max(img_caps, axis=2)), axis=1))
            captions.append(texts)
        return boxes, captions, log_probs

    @staticmethod
    def assign_detections_to_ground_truth(num_images, gt_boxes, gt_captions, boxes, captions, log_probs):
        """:296-345 -- detections in descending log-probability order, each assigned to the ground-truth box it overlaps most; the first
        detection assigned to a box is 'ok', the later ones are not."""
        results = []
        for i in range(num_images):
            order = np.argsort(log_probs[i])[::-1]
            ov = compute_overlaps(boxes[i], gt_boxes[i])
            assign, best = np.argmax(ov, axis=1), np.amax(ov, axis=1)
            used, records = set(), []
            for ind in order[:log_probs[i].shape[0]]:
                ok = 0
                if assign[ind] not in used:
                    used.add(assign[ind])
                    ok = 1
                records.append({'ok': ok, 'ov': best[ind], 'candidate': captions[i][ind],
                                'references': gt_captions[i][assign[ind]] if best[ind] > 0 else []})
            results.append(records)
        return results
