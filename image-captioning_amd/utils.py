"""Dataset base class and image-molding helpers the hot path needs
(dense_img_cap_separate_models/utils.py:187-287 Dataset, :290-340 resize_image;
modified_dense_model.py:2028-2042 compose_image_meta, :2068-2073 mold_image).
Box/anchor/NMS utilities of the reference's utils.py belong to the RPN path and are not here."""
import numpy as np


class Dataset(object):
    """Image registry: add_image() records, prepare() assigns dense ids; sub-classes add
    load_image / load_captions_and_rois."""

    def __init__(self):
        self._image_ids = []
        self.image_info = []

    def add_image(self, source, image_id, path, **kwargs):
        info = {"id": image_id, "source": source, "path": path}
        info.update(kwargs)
        self.image_info.append(info)

    def image_reference(self, image_id):
        return ""

    def prepare(self):
        self.num_images = len(self.image_info)
        self._image_ids = np.arange(self.num_images)

    @property
    def image_ids(self):
        return self._image_ids

    def source_image_link(self, image_id):
        return self.image_info[image_id]["path"]

    def load_image(self, image_id):
        """[H,W,3] uint8 (utils.py:255-262: skimage.io.imread + gray2rgb for single-channel files).  Decoded with PIL, the
        library behind skimage's default imread plugin; datasets that already hold decoded pixels put them under
        image_info[...]['pixels']."""
        info = self.image_info[image_id]
        if "pixels" in info:
            img = np.asarray(info["pixels"])
        else:
            img = imread(info["path"])
        if img.ndim != 3:
            img = np.stack([img] * 3, axis=-1)      # skimage.color.gray2rgb
        return img

    def load_captions_and_rois(self, image_id):
        return np.empty([0, 0, 0, 0]), np.empty([0], np.float32)


def imread(path):
    """skimage.io.imread(path) for the formats the path meets (JPEG / PNG): the file's own channels as a uint8 array
    ([H,W] for grayscale, [H,W,3] RGB, [H,W,4] RGBA); palette images are expanded like skimage's PIL plugin does."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode == "P":
            im = im.convert("RGBA" if "transparency" in im.info else "RGB")
        elif im.mode not in ("L", "RGB", "RGBA"):
            im = im.convert("RGB")
        return np.asarray(im).copy()


def imresize(arr, size):
    """scipy.misc.imresize(arr, size) with its defaults (interp='bilinear', mode=None), size = (rows, cols): scipy's
    function (removed in SciPy 1.3) was a thin wrapper over PIL -- toimage(arr) (uint8 data unchanged; other dtypes byte-scaled
    to their min..max), Image.resize((cols, rows), resample=BILINEAR), back to an array -- and PIL is what runs here."""
    from PIL import Image
    a = np.asarray(arr)
    if a.dtype != np.uint8:
        lo, hi = float(a.min()), float(a.max())
        span = (hi - lo) or 1.0
        a = ((a - lo) * (255.0 / span)).clip(0, 255) + 0.5
        a = a.astype(np.uint8)
    im = Image.fromarray(a)
    bilinear = getattr(Image, "Resampling", Image).BILINEAR
    return np.asarray(im.resize((int(size[1]), int(size[0])), resample=bilinear))


def resize_image(image, min_dim=None, max_dim=None, padding=False):
    """Scale so the short side reaches min_dim (never down-scale for it) while the long side stays
    within max_dim, then zero-pad to max_dim x max_dim.  Returns (image, window, scale, padding)
    (utils.py:290-340; the resampling is scipy.misc.imresize's: PIL bilinear, see imresize)."""
    dtype = image.dtype
    h, w = image.shape[:2]
    window = (0, 0, h, w)
    scale = 1
    if min_dim:
        scale = max(1, min_dim / min(h, w))
    if max_dim:
        if round(max(h, w) * scale) > max_dim:
            scale = max_dim / max(h, w)
    if scale != 1:
        image = imresize(image, (round(h * scale), round(w * scale)))
    if padding:
        h, w = image.shape[:2]
        top = (max_dim - h) // 2
        left = (max_dim - w) // 2
        padding = [(top, max_dim - h - top), (left, max_dim - w - left), (0, 0)]
        if any(p != (0, 0) for p in padding):            # (an image that already has the molded size is handed on as it is: no copies)
            image = np.pad(image, padding, mode='constant', constant_values=0)
        window = (top, left, h + top, w + left)
    return image.astype(dtype, copy=False), window, scale, padding


def compose_image_meta(image_id, image_shape, window):
    """[id, h, w, c, y1, x1, y2, x2]"""
    return np.array([image_id] + list(image_shape) + list(window))


def mold_image(images, config):
    return images.astype(np.float32) - config.MEAN_PIXEL


def generate_anchors(scales, ratios, shape, feature_stride, anchor_stride):
    """Anchors of one pyramid level, [H*W*len(ratios), (y1,x1,y2,x2)] in image pixels, ordered
    (y, x, ratio) -- the order rpn_graph's reshape gives the RPN outputs (utils.py:333-369)."""
    scales, ratios = np.meshgrid(np.array(scales), np.array(ratios))
    scales, ratios = scales.flatten(), ratios.flatten()
    heights, widths = scales / np.sqrt(ratios), scales * np.sqrt(ratios)
    ys = np.arange(0, shape[0], anchor_stride) * feature_stride
    xs = np.arange(0, shape[1], anchor_stride) * feature_stride
    xs, ys = np.meshgrid(xs, ys)
    bw, cx = np.meshgrid(widths, xs)
    bh, cy = np.meshgrid(heights, ys)
    centers = np.stack([cy, cx], axis=2).reshape([-1, 2])
    sizes = np.stack([bh, bw], axis=2).reshape([-1, 2])
    return np.concatenate([centers - 0.5 * sizes, centers + 0.5 * sizes], axis=1)


def generate_pyramid_anchors(scales, ratios, feature_shapes, feature_strides, anchor_stride):
    """All levels concatenated, scale i on level i (utils.py:372-389)."""
    return np.concatenate([generate_anchors(scales[i], ratios, feature_shapes[i], feature_strides[i], anchor_stride)
                           for i in range(len(scales))], axis=0)


# ------------------------------------------------------------------------------------------------
# Box helpers of the reference's utils namespace (dense_img_cap_separate_models/utils.py:30-184, :410-417; the dense_img_cap copy is
# identical; evaluate_models/utils.py differs in compute_iou: see test_score_dense_captions.py).  Callers of the drop-in surface reach
# them as `utils.<name>`; outputs equal the reference functions' own on the same inputs (tests/test_golden_reference.py).
# ------------------------------------------------------------------------------------------------

def compute_iou(box, boxes, box_area, boxes_area):
    """IoU of one (y1,x1,y2,x2) box with an array of boxes; the areas are passed in (utils.py:30-48)."""
    ih = np.maximum(np.minimum(box[2], boxes[:, 2]) - np.maximum(box[0], boxes[:, 0]), 0)
    iw = np.maximum(np.minimum(box[3], boxes[:, 3]) - np.maximum(box[1], boxes[:, 1]), 0)
    inter = iw * ih
    return inter / (box_area + boxes_area - inter)


def compute_overlaps(boxes1, boxes2):
    """float64 IoU matrix [len(boxes1), len(boxes2)] (utils.py:51-67)."""
    a1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    a2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    out = np.zeros((boxes1.shape[0], boxes2.shape[0]))
    for i in range(boxes2.shape[0]):
        out[:, i] = compute_iou(boxes2[i], boxes1, a2[i], a1)
    return out


def non_max_suppression(boxes, scores, threshold):
    """Kept indices (int32) in descending score order; integer boxes are taken as float32 (utils.py:70-104)."""
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


def _center_form(b):
    h, w = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    return b[:, 0] + 0.5 * h, b[:, 1] + 0.5 * w, h, w


def apply_box_deltas(boxes, deltas):
    """boxes shifted by (dy, dx) of their size and scaled by exp(dh), exp(dw); float32 boxes (utils.py:107-128)."""
    cy, cx, h, w = _center_form(boxes.astype(np.float32))
    cy, cx = cy + deltas[:, 0] * h, cx + deltas[:, 1] * w
    h, w = h * np.exp(deltas[:, 2]), w * np.exp(deltas[:, 3])
    y1, x1 = cy - 0.5 * h, cx - 0.5 * w
    return np.stack([y1, x1, y1 + h, x1 + w], axis=1)


def box_refinement(box, gt_box):
    """The (dy, dx, log dh, log dw) that takes box to gt_box, computed in float32 (utils.py:157-180)."""
    cy, cx, h, w = _center_form(box.astype(np.float32))
    gy, gx, gh, gw = _center_form(gt_box.astype(np.float32))
    return np.stack([(gy - cy) / h, (gx - cx) / w, np.log(gh / h), np.log(gw / w)], axis=1)


def trim_zeros(x):
    """Rows that are not all zero (utils.py:410-417)."""
    assert len(x.shape) == 2
    return x[~np.all(x == 0, axis=1)]


def compute_recall(pred_boxes, gt_boxes, iou):
    """(share of GT boxes that are some prediction's best match at IoU >= iou, indices of those predictions) (utils.py:420-436)."""
    overlaps = compute_overlaps(pred_boxes, gt_boxes)
    best, arg = np.max(overlaps, axis=1), np.argmax(overlaps, axis=1)
    positive_ids = np.where(best >= iou)[0]
    return len(set(arg[positive_ids])) / gt_boxes.shape[0], positive_ids
