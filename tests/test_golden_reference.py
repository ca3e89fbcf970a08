"""Host-side code against vectors produced by RUNNING the reference's own NumPy functions
(tests/golden/make_reference_vectors.py -> tests/golden/reference_host_vectors.npz): the product's mirror modules and the
oracle's geometry must reproduce them.  This is the pinned part of the oracle; Keras/TF layer arithmetic stays unpinned."""
import os

import numpy as np
import pytest

from oracle import np_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_host_vectors.npz"))


def test_config_defaults_and_derived_attributes():
    from image_captioning_amd.config import Config
    for key in G.files:
        if key.startswith("config_default/"):
            name = key.split("/")[1]
            np.testing.assert_array_equal(np.asarray(getattr(Config, name), np.float64), G[key], err_msg=name)

    class Cfg(Config):
        NAME = "golden"
        IMAGES_PER_GPU = 3
        GPU_COUNT = 2
        IMAGE_MAX_DIM = 512
        IMAGE_MIN_DIM = 384
    c = Cfg()
    assert c.BATCH_SIZE == int(G["config_derived/BATCH_SIZE"])
    np.testing.assert_array_equal(c.IMAGE_SHAPE, G["config_derived/IMAGE_SHAPE"])
    np.testing.assert_array_equal(c.BACKBONE_SHAPES, G["config_derived/BACKBONE_SHAPES"])


@pytest.mark.parametrize("impl", ["product", "oracle"])
def test_anchor_generation(impl):
    from image_captioning_amd import utils
    from image_captioning_amd.config import Config
    gen = utils.generate_pyramid_anchors if impl == "product" else O.generate_pyramid_anchors
    shapes = lambda S: np.array([[-(-S // s), -(-S // s)] for s in Config.BACKBONE_STRIDES])
    a = gen(Config.RPN_ANCHOR_SCALES, Config.RPN_ANCHOR_RATIOS, shapes(256), Config.BACKBONE_STRIDES, 1)
    np.testing.assert_allclose(a, G["anchors256"], rtol=0, atol=1e-9)
    b = gen(Config.RPN_ANCHOR_SCALES, Config.RPN_ANCHOR_RATIOS, shapes(1024), Config.BACKBONE_STRIDES, 1)
    assert b.shape[0] == int(G["anchors1024/count"]) == 261888
    np.testing.assert_allclose(b[:64], G["anchors1024/head"], atol=1e-9)
    np.testing.assert_allclose(b[-64:], G["anchors1024/tail"], atol=1e-9)
    np.testing.assert_allclose(b[::4099], G["anchors1024/every4099"], atol=1e-9)
    np.testing.assert_allclose(b.sum(axis=0), G["anchors1024/colsum"], rtol=1e-12)
    if impl == "product":
        np.testing.assert_allclose(utils.generate_anchors([32, 64], [0.5, 1, 2], [3, 5], 16, 2), G["anchors_single"], atol=1e-9)


def test_overlaps_product_and_oracle():
    from image_captioning_amd.dense_model import box_iou_f32, compute_overlaps
    want = G["iou/out"]
    np.testing.assert_allclose(compute_overlaps(G["iou/b1"], G["iou/b2"]), want, rtol=1e-12, atol=1e-15)
    assert want[10, 3] == 1.0
    np.testing.assert_allclose(box_iou_f32(G["iou/b1"], G["iou/b2"]), want, atol=2e-6)         # the TF graph's float32 IoU
    np.testing.assert_allclose(O.overlaps_f32(G["iou/b1"], G["iou/b2"]), want, atol=2e-6)


def test_non_max_suppression():
    from image_captioning_amd.dense_model import non_max_suppression
    boxes, scores = G["nms/boxes"], G["nms/scores"]
    for t in (0.3, 0.5, 0.7):
        np.testing.assert_array_equal(non_max_suppression(boxes, scores, t), G["nms/keep_%02d" % int(t * 10)])
    np.testing.assert_array_equal(non_max_suppression(G["nms/int_boxes"], scores[:30], 0.5), G["nms/int_keep"])
    # the oracle's TF-style NMS agrees with the NumPy one wherever no score tie decides (scores 5 and 6 tie here: the
    # NumPy version visits the higher index first, tf.image.non_max_suppression the lower)
    s = scores.copy()
    s[6] = np.nextafter(s[6], 2.0)                            # 6 before 5, as the NumPy argsort()[::-1] orders the tie
    for t in (0.3, 0.7):
        keep = O.nms_tf(boxes.astype(np.float32), s.astype(np.float32), len(s), t)
        np.testing.assert_array_equal(np.asarray(keep), G["nms/keep_%02d" % int(t * 10)])


def test_box_deltas_and_refinement():
    boxes, deltas = G["iou/b1"], G["deltas/in"]
    got = O.apply_box_deltas_f32(boxes.astype(np.float32), deltas.astype(np.float32))
    np.testing.assert_allclose(got, G["deltas/applied"], rtol=2e-6, atol=2e-5)
    # box_refinement is what build_rpn_targets divides by RPN_BBOX_STD_DEV: invert it through apply_box_deltas
    back = O.apply_box_deltas_f32(boxes.astype(np.float32), G["refine/out"].astype(np.float32))
    np.testing.assert_allclose(back, G["refine/gt"], rtol=1e-5, atol=2e-4)


def test_image_meta_and_mold():
    from image_captioning_amd import utils
    from image_captioning_amd.config import Config
    np.testing.assert_array_equal(utils.compose_image_meta(17, (480, 640, 3), (0, 16, 384, 496)), G["meta/one"])
    np.testing.assert_array_equal(G["meta/window"][0], [0, 16, 384, 496])
    img = G["mold/img"]
    np.testing.assert_allclose(utils.mold_image(img, Config), G["mold/out"], rtol=0, atol=0)
    np.testing.assert_allclose(O.mold_image(img[None], Config.MEAN_PIXEL)[0], G["mold/out"], atol=1e-12)
    np.testing.assert_array_equal(G["mold/back"], img)


def test_build_rpn_targets_reproduces_reference_sampling():
    from image_captioning_amd.config import Config
    from image_captioning_amd.dense_model import build_rpn_targets

    class TCfg(Config):
        NAME = "t"
        RPN_TRAIN_ANCHORS_PER_IMAGE = 64
    cfg = TCfg()
    for seed in (0, 1):
        np.random.seed(100 + seed)
        match, bbox = build_rpn_targets((256, 256, 3), G["anchors256"], None, G["rpn_targets/gt"], cfg, rng=np.random)
        np.testing.assert_array_equal(match, G["rpn_targets/match_%d" % seed])
        np.testing.assert_allclose(bbox, G["rpn_targets/bbox_%d" % seed], rtol=1e-12, atol=1e-12)
    assert not np.array_equal(G["rpn_targets/match_0"], G["rpn_targets/match_1"])         # the sub-sampling is exercised


def test_window_clipping_and_unmolding():
    from image_captioning_amd.dense_model import clip_to_window, unmold_generations
    np.testing.assert_allclose(clip_to_window(G["clip/window"], G["clip/in"]), G["clip/out"], atol=1e-12)
    gen = G["unmold/in"]
    boxes, ok = unmold_generations(gen[:, :4], (256, 192, 3), G["clip/window"])
    np.testing.assert_array_equal(boxes[ok], G["unmold/boxes"])
    np.testing.assert_array_equal(gen[ok, 4:], G["unmold/captions"])
    assert not ok[4]


# ---------------------------------------------------------------------------------------------
# data pipelines: the same toy datasets as tests/golden/make_reference_vectors.py, fed to the mirror modules
# ---------------------------------------------------------------------------------------------

def _toy_image(i):
    return np.random.RandomState(1000 + i).randint(0, 256, (96 if i % 2 == 0 else 128, 128, 3)).astype(np.uint8)


def _toy_regions(i):
    r = np.random.RandomState(2000 + i)
    n = 6 if i == 0 else 2
    y, x = r.randint(0, 60, n), r.randint(0, 60, n)
    bx = np.stack([y, x, y + r.randint(8, 60, n), x + r.randint(8, 60, n)], axis=1)
    return bx, r.randint(1, 9, (n, 6)).astype(np.float32)


def test_dataset_resize_and_joint_data_generator_match_reference():
    import random
    from image_captioning_amd import utils
    from image_captioning_amd.config import Config
    from image_captioning_amd.dense_model import data_generator

    class GCfg(Config):
        NAME = "gen"
        IMAGES_PER_GPU = 1
        IMAGE_MIN_DIM = 96
        IMAGE_MAX_DIM = 128
        RPN_TRAIN_ANCHORS_PER_IMAGE = 32
        MAX_GT_INSTANCES = 4
        PADDING_SIZE = 6
    cfg = GCfg()

    class Toy(utils.Dataset):
        def load_image(self, image_id):
            return _toy_image(image_id)

        def load_captions_and_rois(self, image_id):
            return _toy_regions(image_id)
    ds = Toy()
    for i in range(3):
        ds.add_image("toy", image_id=i, path="img%d" % i, width=128, height=96)
    ds.prepare()
    np.testing.assert_array_equal(ds.image_ids, G["dataset/image_ids"])
    assert ds.num_images == int(G["dataset/num_images"])
    img, window, scale, padding = utils.resize_image(_toy_image(0), min_dim=96, max_dim=128, padding=True)
    np.testing.assert_array_equal(img, G["resize/image"])
    np.testing.assert_array_equal(np.asarray(window), G["resize/window"])
    assert scale == float(G["resize/scale"])
    np.testing.assert_array_equal(np.asarray(padding), G["resize/padding"])
    np.random.seed(7)
    random.seed(7)
    gen = data_generator(ds, cfg, shuffle=False, augment=False, batch_size=1, rng=np.random)
    for b in range(3):
        inputs, outputs = next(gen)
        assert outputs == []
        for j, name in enumerate(("images", "image_meta", "rpn_match", "rpn_bbox", "gt_captions", "gt_boxes")):
            want = G["joint_gen/%d/%s" % (b, name)]
            assert inputs[j].shape == want.shape and inputs[j].dtype == want.dtype, (b, name, inputs[j].dtype, want.dtype)
            np.testing.assert_allclose(inputs[j], want, rtol=0, atol=0, err_msg="%d %s" % (b, name))


def test_v1_roi_info_and_data_generator_match_reference():
    import types
    import image_captioning_amd.text_generation_model as T
    table = {i: np.random.RandomState(3000 + i).standard_normal((4, 2, 2, 3)).astype(np.float32) for i in range(3)}

    class ToyV1:
        _image_ids = np.arange(3)

        def load_captions_and_rois(self, image_id):
            r = np.random.RandomState(4000 + image_id)
            caps = np.zeros((2 + image_id, 5), np.float32)
            for k in range(caps.shape[0]):
                n = r.randint(1, 4)
                caps[k, 0], caps[k, 1:1 + n], caps[k, 1 + n] = 1, r.randint(3, 11, n), 2
            return None, caps
    ds = ToyV1()
    ds.rois = T.create_roi_info(ds)
    assert len(ds.rois) == int(G["v1_gen/roi_count"])
    np.testing.assert_array_equal([r[0] for r in ds.rois], G["v1_gen/roi_image_ids"])
    import image_captioning_amd.generate_one_roi_features as GF
    orig = GF.generate_features
    GF.generate_features = lambda dataset, image_id, model: table[image_id]      # the generator's input, as in the fixture
    try:
        gen = T.data_generator(ds, None, types.SimpleNamespace(VOCABULARY_SIZE=12), 4)
        for b in range(3):
            (feat, words), onehot = next(gen)
            for got, name in ((feat, "features"), (words, "words"), (onehot, "onehot")):
                want = G["v1_gen/%d/%s" % (b, name)]
                assert got.shape == want.shape and got.dtype == want.dtype, (b, name, got.dtype, want.dtype)
                np.testing.assert_array_equal(got, want)
    finally:
        GF.generate_features = orig


def test_vocabulary_helpers_match_reference():
    from image_captioning_amd import preprocess as P
    emb = {w: np.random.RandomState(50 + i).standard_normal(8) for i, w in enumerate(["a", "red", "car", "dog"])}
    np.random.seed(11)
    w2i, i2w, mat = P.load_corpus(["a", "red", "car", "dog"], emb, 8)
    np.testing.assert_array_equal(mat, G["vocab/matrix"])
    np.testing.assert_array_equal([w2i[w] for w in ["a", "red", "car", "dog"]], G["vocab/ids"])
    names = sorted(k for k in w2i if k.startswith("<"))
    assert names == list(G["vocab/special_names"])
    np.testing.assert_array_equal([w2i[k] for k in names], G["vocab/specials"])
    np.testing.assert_array_equal([P.encode_word("car", w2i), P.encode_word("zebra", w2i)], G["vocab/encode_known_unknown"])
    onehots = np.eye(len(i2w))[[w2i["red"], w2i["dog"]]]
    assert P.decode_caption(onehots, i2w) == str(G["vocab/decode_caption"])


def test_mold_inputs_matches_reference():
    """The mirror keeps the images uint8 (the mean is subtracted on the GPU): reference molded == mirror - MEAN_PIXEL."""
    from image_captioning_amd.config import Config
    from image_captioning_amd.modified_dense_model import DenseImageCapRCNN

    class GCfg(Config):
        NAME = "gen"
        IMAGES_PER_GPU = 1
        IMAGE_MIN_DIM = 96
        IMAGE_MAX_DIM = 128
    holder = type("H", (), {"config": GCfg()})()
    molded, metas, windows = DenseImageCapRCNN.mold_inputs(holder, [_toy_image(0), _toy_image(1)])
    assert molded.dtype == np.uint8
    np.testing.assert_allclose(molded.astype(np.float32) - GCfg.MEAN_PIXEL, G["mold_inputs/molded"], atol=0)
    np.testing.assert_array_equal(metas, G["mold_inputs/metas"])
    np.testing.assert_array_equal(windows, G["mold_inputs/windows"])
