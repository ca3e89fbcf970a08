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
