"""Host-side mirror of the reference interface (no GPU): Config values, datasets, sequence expansion,
batch layouts of both data generators, sample tables, image molding, weight packing."""
import os

import numpy as np
import pytest
import torch

from oracle import np_models as M


class FakeFeatures:
    """Stands in for DenseImageCapRCNN in the generators: features = f(image id, roi index)."""
    calls = 0

    def generate_captions(self, images, rois, verbose=0):
        FakeFeatures.calls += 1
        n = rois.shape[1]
        base = float(images[0][0, 0, 0])
        return [{"features": (base + np.arange(n, dtype=np.float32))[:, None, None, None] * np.ones((n, 7, 7, 256), np.float32)}]


def test_config_matches_reference_values():
    from image_captioning_amd.config import Config
    c = Config()
    assert c.BATCH_SIZE == 2 and c.IMAGE_SHAPE.tolist() == [1024, 1024, 3]
    assert c.BACKBONE_SHAPES.tolist() == [[256, 256], [128, 128], [64, 64], [32, 32], [16, 16]]
    assert (c.POOL_SIZE, c.PADDING_SIZE, c.LEARNING_RATE, c.WEIGHT_DECAY) == (7, 15, 0.001, 0.0001)
    assert c.MEAN_PIXEL.tolist() == [123.7, 116.8, 103.9] and c.RPN_ANCHOR_SCALES == (32, 64, 128, 256, 512)
    assert (c.TRAIN_ROIS_PER_IMAGE, c.ROI_POSITIVE_RATIO, c.POST_NMS_ROIS_INFERENCE) == (200, 0.33, 1000)
    from image_captioning_amd.text_generation_model_v2 import DenseCapConfig as C2
    from image_captioning_amd.text_generation_model import DenseCapConfig as C1
    E = np.zeros((12, 300), np.float32)
    # v2 quirk kept: the class attribute BATCH_SIZE = 64 is shadowed by Config.__init__ (IMAGES_PER_GPU*GPU_COUNT = 1)
    assert (C2(12, E).BATCH_SIZE, C2.BATCH_SIZE, C2(12, E).PADDING_SIZE, C2(12, E).EMBEDDING_SIZE) == (1, 64, 10, 300)
    assert (C1(12, E, 256).BATCH_SIZE, C1(12, E, 256).PADDING_SIZE) == (256, 10)


def _v2_dataset():
    from image_captioning_amd.text_generation_model_v2 import VisualGenomeDataset
    w2i = {"<unk>": 0, "<start>": 1, "<end>": 2, "a": 3, "red": 4, "car": 5, "dog": 6}
    ds = VisualGenomeDataset(w2i, 10)
    for i, caps in enumerate([["a red car", "dog"], ["a dog zebra"]]):
        ds.add_image("VisualGenome", image_id=100 + i, path="none", width=8, height=8,
                     rois=[[0, 0, 4, 4]] * len(caps), captions=[[c] for c in caps],
                     pixels=np.full((8, 8, 3), 10 * (i + 1), np.uint8))
    ds.prepare()
    return ds


def test_v2_sequences_and_generator_batch_layout():
    from image_captioning_amd.text_generation_model_v2 import DenseCapConfig, data_generator, load_sequences
    ds = _v2_dataset()
    seqs = load_sequences(ds)
    # "a red car" -> 3 samples, "dog" -> 1, "a dog zebra" (zebra OOV dropped) -> 2
    assert [(s[0], s[1], s[2], s[3]) for s in seqs] == [(0, 0, [0], 3), (0, 0, [3], 4), (0, 0, [3, 4], 5), (0, 1, [0], 6),
                                                        (1, 0, [0], 3), (1, 0, [3], 6)]
    ds.add_sequences(seqs)
    cfg = DenseCapConfig(7, np.zeros((7, 300), np.float32))
    cfg.PADDING_SIZE = 4
    FakeFeatures.calls = 0
    gen = data_generator(ds, FakeFeatures(), cfg, 4)
    (feat, words), y = next(gen)
    assert feat.shape == (4, 7, 7, 256) and feat.dtype == np.float32
    assert words.tolist() == [[0, 0, 0, 0], [0, 0, 0, 3], [0, 0, 3, 4], [0, 0, 0, 0]] and words.dtype == np.int32
    assert y.shape == (4, 7) and y.dtype == np.float64 and y.argmax(1).tolist() == [3, 4, 5, 6]
    assert feat[:, 0, 0, 0].tolist() == [10, 10, 10, 11]          # roi 0,0,0 then roi 1 of image 0
    assert FakeFeatures.calls == 1                                  # features recomputed only on image change
    (feat, words), y = next(gen)                                    # wraps around the sequence list
    assert feat[:, 0, 0, 0].tolist() == [20, 20, 10, 10] and FakeFeatures.calls == 3
    roi, w2, t2 = M.v2_expand_samples([[3, 4, 5], [6], [3, 6]], 4)  # the oracle's expansion agrees
    assert w2[:4].tolist() == [[0, 0, 0, 0], [0, 0, 0, 3], [0, 0, 3, 4], [0, 0, 0, 0]] and t2.tolist() == [3, 4, 5, 6, 3, 6]


def test_v1_dataset_and_generator_batch_layout():
    from image_captioning_amd.text_generation_model import (DenseCapConfig, VisualGenomeDataset, caption_targets,
                                                            create_roi_info, data_generator)
    w2i = {"<unk>": 0, "<start>": 1, "<end>": 2, "a": 3, "red": 4, "car": 5, "dog": 6}
    ds = VisualGenomeDataset(w2i, 5)
    ds.add_image("VisualGenome", image_id=7, path="none", width=8, height=8, rois=[[0, 0, 4, 4], [1, 1, 3, 3]],
                 captions=[["a red car dog a"], ["dog"]], pixels=np.full((8, 8, 3), 5, np.uint8))
    ds.prepare()
    rois, caps = ds.load_captions_and_rois(0)
    assert caps.dtype == np.float32 and caps.tolist() == [[1, 3, 4, 5, 2], [1, 6, 2, 0, 0]]    # truncated to T-2 words
    ds.add_rois(create_roi_info(ds))
    cfg = DenseCapConfig(7, np.zeros((7, 300), np.float32), 2)
    cfg.PADDING_SIZE = 5
    (feat, words), y = next(data_generator(ds, FakeFeatures(), cfg, 2))
    assert feat.shape == (2, 7, 7, 256) and words.tolist() == caps.tolist()
    assert y.shape == (2, 5, 7) and y.dtype == np.float64
    assert y.argmax(-1).tolist() == [[3, 4, 5, 2, 0], [6, 2, 0, 0, 0]]     # shifted left, pads -> class 0
    np.testing.assert_array_equal(caption_targets(caps), M.v1_targets(caps))


def test_sample_tables_single_pass_indexing():
    from image_captioning_amd.text_generation_model_v2 import SampleTables
    tb = SampleTables.from_captions([[5, 6, 7], [9], [3, 4]], "cpu")
    assert (tb.Bw, tb.T, tb.N) == (3, 2, 6)
    assert tb.ids_tm.tolist() == [5, 0, 3, 6, 0, 0]                 # time-major, last word never an input
    assert tb.mask.tolist() == [1, 0, 1, 1, 0, 0]
    assert tb.roi_idx.tolist() == [0, 0, 0, 1, 2, 2] and tb.targets.tolist() == [5, 6, 7, 9, 3, 4]
    assert tb.hrow_idx.tolist() == [-1, 0, 3, -1, -1, 2]            # h after j tokens = row (j-1)*R + r
    assert tb.inv_hrow.tolist() == [1, -1, 5, 2, -1, -1]
    ts = SampleTables.from_samples(np.array([[0, 0, 3], [0, 3, 4]]), [4, 5], "cpu")
    assert ts.ids_tm.tolist() == [0, 0, 0, 3, 3, 4] and ts.hrow_idx.tolist() == [4, 5]
    with pytest.raises(ValueError):
        SampleTables(np.zeros(2, np.int32), 1, 2, [0, 0], [1, 1], [3, 4], "cpu")   # one state row feeding two samples


def test_resize_and_meta():
    from image_captioning_amd import utils
    img = np.ones((1024, 1024, 3), np.uint8)
    out, window, scale, padding = utils.resize_image(img, 800, 1024, True)
    assert out.shape == (1024, 1024, 3) and window == (0, 0, 1024, 1024) and scale == 1
    out, window, scale, padding = utils.resize_image(np.ones((400, 200, 3), np.uint8), 800, 1024, True)
    assert out.shape == (1024, 1024, 3) and scale == 1024 / 400 and window == (0, 256, 1024, 768)
    assert utils.compose_image_meta(0, (400, 200, 3), window).tolist() == [0, 400, 200, 3, 0, 256, 1024, 768]


def test_weight_packing_roundtrip():
    from image_captioning_amd.packing import fold_bn, pack_conv_kernel, pack_stem_kernel
    from oracle import np_oracle as O
    rng = np.random.default_rng(0)
    k = rng.standard_normal((3, 3, 8, 5)).astype(np.float32)
    p = pack_conv_kernel(k)
    assert p.shape == (5, 72) and p[2, (1 * 3 + 2) * 8 + 4] == k[1, 2, 4, 2]
    s = pack_stem_kernel(rng.standard_normal((7, 7, 3, 4)).astype(np.float32)).reshape(4, 7, 8, 4)
    assert np.all(s[:, :, 7, :] == 0) and np.all(s[:, :, :, 3] == 0)
    g, b, m, v, cb = (rng.standard_normal(6) for _ in range(5))
    sc, sh = fold_bn(g, b, m, np.abs(v) + 0.1, cb)
    so, sho = O.bn_scale_shift(g, b, m, np.abs(v) + 0.1, cb)
    np.testing.assert_allclose(sc, so, rtol=1e-6)
    np.testing.assert_allclose(sh, sho, rtol=1e-6, atol=1e-7)


def test_param_store_layout_on_cpu():
    from image_captioning_amd.params import ParamStore
    st = ParamStore("cpu")
    st.add("a/kernel", np.arange(6, dtype=np.float32).reshape(2, 3), True)
    st.add("b/bias", np.ones(5, np.float32), True)
    st.add("emb", np.zeros((2, 2), np.float32), False)
    st.finalize()
    assert st.trainable_names == ["a/kernel", "b/bias"] and st.frozen_names == ["emb"]
    assert st.flat.numel() == 8 + 8 and st.w["b/bias"].data_ptr() % 16 == st.flat.data_ptr() % 16     # 16-byte aligned views
    st.grad["a/kernel"].fill_(2.0)
    assert float(st.flat_grad[:6].sum()) == 12.0 and float(st.flat_grad[6:8].sum()) == 0.0


# ---------------------------------------------------------------------------------------------
# joint model host logic (dense_img_cap/dense_model.py): detection targets, RPN targets, generator
# ---------------------------------------------------------------------------------------------

def _joint_cfg(S=128, T=5):
    from image_captioning_amd.config import Config

    class Cfg(Config):
        IMAGES_PER_GPU = 1
        IMAGE_MIN_DIM = S
        IMAGE_MAX_DIM = S
        TRAIN_ROIS_PER_IMAGE = 12
        PADDING_SIZE = T
        MAX_GT_INSTANCES = 5
        RPN_TRAIN_ANCHORS_PER_IMAGE = 64
    return Cfg()


def test_detection_targets_match_oracle():
    from oracle import np_oracle as O
    from image_captioning_amd.dense_model import detection_targets
    cfg = _joint_cfg()
    rng = np.random.default_rng(0)
    gt = np.zeros((5, 4), np.float32)
    gt[:3] = [[0.1, 0.1, 0.5, 0.6], [0.3, 0.2, 0.9, 0.9], [0.0, 0.0, 0.4, 0.3]]
    caps = np.zeros((5, 5), np.int32)
    caps[:3] = rng.integers(1, 20, (3, 5))
    props = np.zeros((60, 4), np.float32)
    for i in range(50):
        b = gt[i % 3] + rng.normal(0, 0.08, 4)
        props[i] = np.clip([min(b[0], b[2]), min(b[1], b[3]), max(b[0], b[2]) + 0.01, max(b[1], b[3]) + 0.01], 0, 1)
    for mix in (None, np.random.RandomState(3).permutation):
        mix2 = None if mix is None else np.random.RandomState(3).permutation
        want = O.detection_targets(props, caps, gt, cfg.TRAIN_ROIS_PER_IMAGE, cfg.ROI_POSITIVE_RATIO, mix2)
        got = detection_targets(props, caps, gt, cfg, mix)
        assert got[2:] == want[2:] and got[2] > 0 and got[3] > 0
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # no positives: everything zero-captioned, no negatives either (the reference's ratio rule)
    far = np.tile(np.array([[0.9, 0.9, 1.0, 1.0]], np.float32), (4, 1))
    rois, c, npos, nneg = detection_targets(far, caps, gt, cfg)
    assert npos == 0 and nneg == 0 and not rois.any() and not c.any()


def test_build_rpn_targets_rules():
    from image_captioning_amd import utils
    from image_captioning_amd.dense_model import build_rpn_targets, compute_overlaps
    cfg = _joint_cfg()
    anchors = utils.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, cfg.BACKBONE_SHAPES,
                                             cfg.BACKBONE_STRIDES, cfg.RPN_ANCHOR_STRIDE)
    gt = np.array([[10, 12, 70, 90], [40, 30, 120, 128], [0, 0, 50, 40]], np.int32)
    match, deltas = build_rpn_targets((128, 128, 3), anchors, None, gt, cfg, np.random.RandomState(0))
    assert match.shape == (anchors.shape[0],) and deltas.shape == (64, 4)
    n_pos, n_neg = int((match == 1).sum()), int((match == -1).sum())
    assert 0 < n_pos <= 32 and n_pos + n_neg == 64
    iou = compute_overlaps(anchors, gt)
    assert np.all(iou[match == -1].max(axis=1) < 0.3)
    pos = np.where(match == 1)[0]
    # applying the (de-normalised) deltas to the positive anchors reproduces their GT boxes
    a = anchors[pos]
    d = deltas[:n_pos] * cfg.RPN_BBOX_STD_DEV
    h, w = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
    cy, cx = a[:, 0] + 0.5 * h + d[:, 0] * h, a[:, 1] + 0.5 * w + d[:, 1] * w
    hh, ww = h * np.exp(d[:, 2]), w * np.exp(d[:, 3])
    rebuilt = np.stack([cy - 0.5 * hh, cx - 0.5 * ww, cy + 0.5 * hh, cx + 0.5 * ww], axis=1)
    np.testing.assert_allclose(rebuilt, gt[iou[pos].argmax(axis=1)], atol=1e-9)
    assert not deltas[n_pos:].any()


def test_joint_data_generator_layout():
    from image_captioning_amd.dense_model import data_generator
    from image_captioning_amd.utils import Dataset
    cfg = _joint_cfg()

    class Toy(Dataset):
        def load_image(self, image_id):
            return np.random.RandomState(image_id).randint(0, 255, (96, 128, 3)).astype(np.uint8)

        def load_captions_and_rois(self, image_id):
            n = 7 if image_id == 0 else 2
            r = np.random.RandomState(image_id)
            y, x = r.randint(0, 60, n), r.randint(0, 60, n)
            boxes = np.stack([y, x, y + r.randint(8, 60, n), x + r.randint(8, 60, n)], axis=1)
            return boxes, r.randint(1, 9, (n, cfg.PADDING_SIZE)).astype(np.float32)
    ds = Toy()
    for i in range(2):
        ds.add_image("toy", image_id=i, path=None)
    ds.prepare()
    gen = data_generator(ds, cfg, shuffle=False, augment=False, batch_size=1, rng=np.random.RandomState(0))
    inputs, outputs = next(gen)
    images, metas, match, bbox, caps, boxes = inputs
    assert outputs == [] and images.shape == (1, 128, 128, 3) and images.dtype == np.float32
    assert match.shape[0] == 1 and match.shape[2] == 1 and bbox.shape == (1, 64, 4)
    assert caps.shape == (1, 5, cfg.PADDING_SIZE) and boxes.shape == (1, 5, 4)          # 7 instances sub-sampled to MAX_GT_INSTANCES
    assert np.all(np.abs(boxes[0]).sum(axis=1) > 0)
    inputs, _ = next(gen)
    assert int((np.abs(inputs[5][0]).sum(axis=1) > 0).sum()) == 2                      # zero-padded
    # molded = image - MEAN_PIXEL: the model recovers the uint8 image exactly
    from image_captioning_amd.dense_model import DenseImageCapRCNN
    u8 = DenseImageCapRCNN._images_u8(type("M", (), {"config": cfg})(), inputs[0]).numpy()      # a host uint8 torch tensor
    assert u8.dtype == np.uint8 and np.allclose(u8.astype(np.float32) - cfg.MEAN_PIXEL, inputs[0], atol=1e-4)


def test_train_dense_captions_dataset_and_vocabulary(tmp_path):
    import json
    from image_captioning_amd import preprocess
    from image_captioning_amd.train_dense_captions import DenseCapConfig, VisualGenomeDataset, load_vocabulary
    glove = tmp_path / "glove.txt"
    words = ["a", "red", "car", "dog", "on", "grass", "rare"]
    glove.write_text("\n".join("%s %s" % (w, " ".join("%.3f" % (0.01 * (i + j)) for j in range(300))) for i, w in enumerate(words)))
    regions = [{"id": 1, "regions": [{"phrase": "A red car.", "x": 5, "y": 6, "width": 30, "height": 20},
                                     {"phrase": "!!!", "x": 0, "y": 0, "width": 4, "height": 4},
                                     {"phrase": "a dog on grass " * 8, "x": 1, "y": 2, "width": 3, "height": 4}] * 15},
               {"id": 2, "regions": [{"phrase": "rare", "x": 0, "y": 0, "width": 9, "height": 9}]}]
    data = tmp_path / "regions.json"
    data.write_text(json.dumps(regions))
    meta = tmp_path / "meta.json"
    meta.write_text(json.dumps([{"image_id": 1, "width": 64, "height": 48}, {"image_id": 2, "width": 10, "height": 10}]))
    emb = preprocess.load_embeddings(str(glove))
    assert set(emb) == set(words) and emb["car"].shape == (300,)
    vocab = preprocess.tokenize_corpus(str(data), [1], emb)
    assert vocab == {"a", "red", "car", "dog", "on", "grass"}            # 'rare' is not in the training split, '.' is punctuation
    id_to_word, word_to_id, matrix = load_vocabulary(str(tmp_path / "cache"), str(glove), str(data), [1])
    assert matrix.shape[0] % 4 == 0 and matrix.shape[0] == len(id_to_word) and word_to_id["<start>"] == 1
    again = load_vocabulary(str(tmp_path / "cache"), str(glove), str(data), [1])                   # cached pickles
    assert again[1] == word_to_id and np.array_equal(again[2], matrix)
    cfg = DenseCapConfig(len(id_to_word), matrix)
    assert cfg.BATCH_SIZE == 1 and cfg.EMBEDDING_SIZE == 300 and cfg.PADDING_SIZE == 15
    ds = VisualGenomeDataset(word_to_id, cfg.PADDING_SIZE)
    ds.load_visual_genome(str(tmp_path), [1], str(meta), str(data))
    ds.prepare()
    rois, caps = ds.load_captions_and_rois(0)
    assert rois.shape == (30, 4) and caps.shape == (30, 15) and caps.dtype == np.float32           # the '!!!' regions encode to nothing
    assert list(rois[0]) == [6, 5, 26, 35]
    assert caps[0, 0] == 1 and caps[0, 4] == 2 and not caps[0, 5:].any()                          # <start> a red car <end>
    assert caps[1, 0] == 1 and caps[1, 14] == 2 and np.all(caps[1, 1:14] > 2)                      # truncated to T-2 words


def test_refine_and_unmold_generations():
    from image_captioning_amd.dense_model import non_max_suppression, refine_generations, unmold_generations
    cfg = _joint_cfg()
    cfg.DETECTION_MAX_INSTANCES = 2
    rois = np.array([[0.1, 0.1, 0.5, 0.5], [0.12, 0.1, 0.5, 0.52], [0.6, 0.6, 0.9, 1.2], [0.0, 0.0, 0.05, 0.05], [0.3, 0.5, 0.8, 0.9]])
    word = np.array([[0.9, 0.9], [0.95, 0.95], [0.5, 0.5], [0.2, 0.9], [0.7, 0.7]])
    keep = non_max_suppression(rois, np.log(word).sum(1), 0.3)
    assert list(keep) == [1, 4, 2, 3]                                   # box 0 is suppressed by the better-scored box 1
    window = (0, 16, 128, 112)
    boxes, kept = refine_generations(rois, word, window, cfg)
    assert list(kept) == [1, 4] and boxes.dtype == np.int32
    assert list(boxes[0]) == [15, 16, 64, 67] and list(boxes[1]) == [38, 64, 102, 112]   # pixels, clipped to the window, rounded
    final, ok = unmold_generations(boxes, (256, 192, 3), window)
    assert list(final[0]) == [30, 0, 128, 102] and ok.all()
    degenerate, ok = unmold_generations(np.array([[5, 16, 5, 40]]), (256, 192, 3), window)
    assert not ok.any()


def test_joint_model_log_dir_and_epoch_from_checkpoint_name(tmp_path):
    """set_log_dir / find_last follow dense_img_cap/dense_model.py:1631-1654, :1776-1798 (no GPU needed: unbound methods
    on a stand-in object carrying config and model_dir)."""
    import types
    from image_captioning_amd.dense_model import DenseImageCapRCNN as J
    cfg = types.SimpleNamespace(NAME="dense image captioning")
    m = types.SimpleNamespace(config=cfg, model_dir=str(tmp_path))
    J.set_log_dir(m)
    assert m.epoch == 0 and os.path.dirname(m.log_dir) == str(tmp_path)
    assert os.path.basename(m.log_dir).startswith("dense image captioning20")
    run = tmp_path / "dense image captioning20260102T0304"
    run.mkdir()
    for e in (1, 2, 11):
        (run / ("img_cap_dense image captioning_%04d.npz" % e)).write_bytes(b"")
    (tmp_path / "other20260102T0304").mkdir()
    assert J.find_last(m) == (str(run), str(run / "img_cap_dense image captioning_0011.npz"))
    J.set_log_dir(m, str(run / "img_cap_dense image captioning_0011.npz"))
    assert m.epoch == 11 and m.log_dir == str(run)
    assert m.checkpoint_path.format(epoch=12) == str(run / "img_cap_dense image captioning_0012.npz")
    J.set_log_dir(m, str(tmp_path / "mask_rcnn_coco.npz"))          # a foreign file: fresh run, epoch 0
    assert m.epoch == 0 and m.log_dir != str(run)


# ---------------------------------------------------------------------------------------------
# tokenisation, image decode and resize: the host-side steps in front of the GPU path
# ---------------------------------------------------------------------------------------------

def test_treebank_tokenizer_known_answers():
    """The examples of NLTK's own documentation (TreebankWordTokenizer / word_tokenize docstrings) -- the published known
    answers of the algorithm the reference's preprocess.py:52-72 calls -- plus the Visual-Genome-style cases that make the
    difference to the vocabulary: possessives, n't, 'm / 're / 've / 'll / 'd, cannot, quotes, brackets, dashes."""
    from image_captioning_amd.treebank import treebank_tokenize, word_tokenize
    s = "Good muffins cost $3.88\nin New York.  Please buy me\ntwo of them.\nThanks."
    assert treebank_tokenize(s, False) == ['Good', 'muffins', 'cost', '$', '3.88', 'in', 'New', 'York.', 'Please', 'buy', 'me', 'two', 'of',
                                            'them.', 'Thanks', '.']
    assert word_tokenize(s) == ['Good', 'muffins', 'cost', '$', '3.88', 'in', 'New', 'York', '.', 'Please', 'buy', 'me', 'two', 'of', 'them', '.',
                                'Thanks', '.']
    assert treebank_tokenize("They'll save and invest more.", False) == ['They', "'ll", 'save', 'and', 'invest', 'more', '.']
    assert treebank_tokenize("hi, my name can't hello,", False) == ['hi', ',', 'my', 'name', 'ca', "n't", 'hello', ',']
    assert word_tokenize("a man's hat. it's red, isn't it?") == ['a', 'man', "'s", 'hat', '.', 'it', "'s", 'red', ',', 'is', "n't", 'it', '?']
    assert word_tokenize("the dog (brown) cannot jump -- i'm sure we've seen they're gonna") == \
        ['the', 'dog', '(', 'brown', ')', 'can', 'not', 'jump', '--', 'i', "'m", 'sure', 'we', "'ve", 'seen', 'they', "'re", 'gon', 'na']
    assert word_tokenize('the sign says "stop"') == ['the', 'sign', 'says', '``', 'stop', "''"]
    # interior periods (the Punkt approximation: listed abbreviations and initials do not end a sentence, anything else does)
    assert word_tokenize("sign says no. 5 on it") == ['sign', 'says', 'no.', '5', 'on', 'it']
    assert word_tokenize("open until 5 p.m. sign on st. marks ave.") == ['open', 'until', '5', 'p.m.', 'sign', 'on', 'st.', 'marks', 'ave', '.']
    assert word_tokenize("approx. ten birds. one flies away") == ['approx.', 'ten', 'birds', '.', 'one', 'flies', 'away']
    assert word_tokenize("j. smith's cafe") == ['j.', 'smith', "'s", 'cafe']
    assert word_tokenize("clock reads 10:30, price is 1,000 dollars") == ['clock', 'reads', '10:30', ',', 'price', 'is', '1,000', 'dollars']
    assert word_tokenize("") == [] and word_tokenize("   ") == []
    assert word_tokenize("dr. who's tardis") == ['dr.', 'who', "'s", 'tardis']          # abbreviation: no sentence break


def test_caption_encoding_uses_treebank_tokens():
    from image_captioning_amd.preprocess import encode_caption, encode_caption_v2
    w2i = {'<unk>': 0, '<start>': 1, '<end>': 2, 'man': 3, "'s": 4, 'hat': 5, 'is': 6, "n't": 7, 'red': 8}
    np.testing.assert_array_equal(encode_caption("Man's hat isn't RED.", w2i), [3, 4, 5, 6, 7, 8])      # '.' is out of vocabulary: dropped
    rows = encode_caption_v2("man's blue hat", w2i)                                                     # 'blue' -> <unk> row -> dropped
    assert rows.shape == (3, len(w2i)) and rows.argmax(1).tolist() == [3, 4, 5]


def test_imresize_is_pil_bilinear_and_resize_image_geometry():
    """resize_image's resampling is scipy.misc.imresize's, i.e. PIL Image.resize(BILINEAR) (utils.py:327): pinned to PIL's
    documented call and to the properties bilinear resampling has; the geometry (scale, window, padding) per utils.py:290-340."""
    from PIL import Image
    from image_captioning_amd import utils
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    got = utils.imresize(img, (90, 120))
    want = np.asarray(Image.fromarray(img).resize((120, 90), resample=getattr(Image, "Resampling", Image).BILINEAR))
    np.testing.assert_array_equal(got, want)
    assert utils.imresize(np.full((10, 12, 3), 77, np.uint8), (25, 31)).tolist() == np.full((25, 31, 3), 77).tolist()      # constants survive
    ramp = np.tile(np.arange(0, 200, 2, dtype=np.uint8)[None, :, None], (8, 1, 3))
    up = utils.imresize(ramp, (16, 200)).astype(int)
    assert (np.diff(up[0, :, 0]) >= 0).all() and abs(int(up[0, 100, 0]) - 100) <= 2                                        # a ramp stays a ramp
    f = utils.imresize(np.linspace(0.0, 1.0, 48).reshape(6, 8), (6, 8))                   # non-uint8: byte-scaled to min..max like scipy's toimage
    assert f.dtype == np.uint8 and f.min() == 0 and f.max() == 255
    out, window, scale, padding = utils.resize_image(rng.integers(0, 256, (600, 800, 3), dtype=np.uint8), min_dim=800, max_dim=1024, padding=True)
    assert out.shape == (1024, 1024, 3) and window == (128, 0, 896, 1024) and scale == 1.28 and padding == [(128, 128), (0, 0), (0, 0)]
    assert not out[:128].any() and not out[896:].any() and out[128:896].any()
    small, window, scale, _ = utils.resize_image(np.zeros((200, 100, 3), np.uint8), min_dim=800, max_dim=1024, padding=True)
    assert scale == 1024 / 200 and window == (0, 256, 1024, 768)                          # the long side caps the up-scaling


def test_jpeg_decode_and_grayscale_expansion(tmp_path):
    from PIL import Image
    from image_captioning_amd import utils
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / "a.png")
    Image.fromarray(rgb[..., 0]).save(tmp_path / "g.jpg", quality=95)
    ds = utils.Dataset()
    ds.add_image("t", image_id=0, path=str(tmp_path / "a.png"))
    ds.add_image("t", image_id=1, path=str(tmp_path / "g.jpg"))
    ds.add_image("t", image_id=2, path=None, pixels=rgb)
    ds.prepare()
    np.testing.assert_array_equal(ds.load_image(0), rgb)                                   # lossless round trip
    g = ds.load_image(1)
    assert g.shape == (40, 56, 3) and g.dtype == np.uint8 and (g[..., 0] == g[..., 1]).all() and (g[..., 1] == g[..., 2]).all()
    np.testing.assert_array_equal(ds.load_image(2), rgb)


def test_reference_sample_images_decode_and_resize_to_the_committed_fixture():
    """The six Visual Genome JPEGs the reference ships (dataset/visual genome/): decoded size, pixel checksum and the resized
    1024 x 1024 input of the model are pinned in tests/golden/vg_sample_images.json (made by tests/golden/make_image_fixtures.py).
    Needs the reference checkout; skipped where it is absent (the GPU box)."""
    import json
    ref = "/root/reference/dataset/visual genome"
    if not os.path.isdir(ref):
        pytest.skip("reference checkout not present")
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_image_fixtures", os.path.join(sys_path, "make_image_fixtures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(sys_path, "vg_sample_images.json")))
    assert len(want) == 6
    for name, row in want.items():
        got = mod.describe(os.path.join(ref, name))
        assert got["shape"] == row["shape"] and got["sha1"] == row["sha1"], name
        assert got["resized_shape"] == [1024, 1024, 3] and got["window"] == row["window"] and got["scale"] == row["scale"], name
        assert got["resized_sha1"] == row["resized_sha1"], name
        assert abs(got["resized_mean"] - got["mean"]) < 0.5                                # bilinear resampling preserves the mean


# ---------------------------------------------------------------------------------------------
# Keras HDF5 weight files without h5py (hdf5_lite)
# ---------------------------------------------------------------------------------------------

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fixture_value(L, shape):
    n = int(np.prod(shape))
    return (((7 * np.arange(n) + 3 * L) % 101).astype(np.float32) / np.float32(8.0) - np.float32(5.0)).reshape(shape)


def test_hdf5_reader_on_a_file_written_by_libhdf5_in_keras_layout():
    """tests/golden/keras_weights_sample.h5 was written by the real HDF5 library (tests/golden/make_h5_fixture.c) in the layout
    Keras' save_weights produces: layer_names / weight_names attributes, nested weight paths, 13 layers (a multi-leaf group
    B-tree), a weightless layer, a nested sub-model layer, one chunked dataset, variable-length string attributes.  The reader
    must return every array by its own name scope with the values the generator's formula gives."""
    from image_captioning_amd import hdf5_lite as H
    path = os.path.join(GOLDEN, "keras_weights_sample.h5")
    f = H.H5File(path)
    layers = [v.decode() for v in f.attrs["layer_names"]]
    assert layers[:3] == ["input_1", "conv1", "bn_conv1"] and len(layers) == 13 and f.keys() == sorted(layers)
    assert f.attrs["backend"] is None                                      # variable-length string: present, not decoded
    assert f["input_1"].attrs["weight_names"].shape == (0,)
    assert f["conv1/conv1/kernel:0"].shape == (3, 3, 2, 6) and f["conv1/conv1/kernel:0"].dtype == np.dtype("<f4")
    w = H.load_keras_weights(path)
    assert len(w) == 34
    order = []                                                              # the generator's dataset order
    for name in layers[1:]:
        if name.startswith("bn") or "_bn" in name:
            order += [(name + "/" + k, (6,)) for k in ("gamma", "beta", "moving_mean", "moving_variance")]
        elif name == "imgcap_caption_td":
            order += [("imgcap_lstm1/kernel", (10, 16)), ("imgcap_lstm1/recurrent_kernel", (4, 16)), ("imgcap_lstm1/bias", (16,)),
                      ("imgcap_lstm_d2/kernel", (8, 12)), ("imgcap_lstm_d2/bias", (12,))]
        elif name == "imgcap_embedding_layer":
            order += [("imgcap_embedding_layer/embeddings", (9, 5))]
        else:
            order += [(name + "/kernel", (3, 3, 2, 6)), (name + "/bias", (6,))]
    assert sorted(k for k, _ in order) == sorted(w)
    for L, (key, shape) in enumerate(order):
        np.testing.assert_array_equal(w[key], _fixture_value(L, shape), err_msg=key)


def test_hdf5_writer_round_trip_and_libhdf5_tools_read_it(tmp_path):
    """save_keras_weights -> load_keras_weights is the identity (incl. > 8 layers: multi-level group B-tree, and a
    layer_names attribute split like Keras splits it); where libhdf5's command line tools exist (the build container's
    /opt/conda) h5ls / h5dump must list every dataset and print a weight's values."""
    import shutil
    import subprocess
    from image_captioning_amd import hdf5_lite as H
    from image_captioning_amd.modified_dense_model import load_weight_file, save_weight_file
    rng = np.random.default_rng(0)
    W = {}
    for i in range(300):
        W["res%03d_branch2a_with_a_long_layer_name/kernel" % i] = rng.standard_normal((1, 1, 4, 3)).astype(np.float32)
        W["res%03d_branch2a_with_a_long_layer_name/bias" % i] = rng.standard_normal(3).astype(np.float32)
    W["imgcap_lstm1/recurrent_kernel"] = rng.standard_normal((8, 32)).astype(np.float32)
    path = str(tmp_path / "w.h5")
    save_weight_file(path, W)
    back = load_weight_file(path)
    assert set(back) == set(W) and all(np.array_equal(back[k], W[k]) and back[k].dtype == np.float32 for k in W)
    big = {"l%05d/w" % i: np.float32([i]) for i in range(12000)}            # layer_names > 64 KB: split into layer_names0, 1, ...
    H.save_keras_weights(str(tmp_path / "big.h5"), big)
    f = H.H5File(str(tmp_path / "big.h5"))
    assert "layer_names" not in f.attrs and "layer_names0" in f.attrs and "layer_names1" in f.attrs
    assert H.load_keras_weights(str(tmp_path / "big.h5"))["l11999/w"] == np.float32(11999)
    h5ls = shutil.which("h5ls") or ("/opt/conda/bin/h5ls" if os.path.exists("/opt/conda/bin/h5ls") else None)
    if h5ls is None:
        return
    listing = subprocess.run([h5ls, "-r", path], capture_output=True, text=True)
    assert listing.returncode == 0 and sum("Dataset" in l for l in listing.stdout.splitlines()) == len(W), listing.stderr[:500]
    dump = subprocess.run([os.path.join(os.path.dirname(h5ls), "h5dump"), "-d", "/imgcap_lstm1/imgcap_lstm1/recurrent_kernel:0", path],
                          capture_output=True, text=True)
    assert dump.returncode == 0 and "H5T_IEEE_F32LE" in dump.stdout and "( 8, 32 )" in dump.stdout
    first = float(dump.stdout.split("(0,0):")[1].split(",")[0])
    assert abs(first - float(W["imgcap_lstm1/recurrent_kernel"][0, 0])) < 1e-4 * max(1.0, abs(first))


def test_hdf5_writer_nests_wrapped_layers_like_keras(tmp_path):
    """Layers inside a wrapper (the joint model's decoder under TimeDistributed 'imgcap_caption_td') go under the wrapper's group
    with their inner names -- /imgcap_caption_td/imgcap_lstm1/kernel:0, weight_names 'imgcap_lstm1/kernel:0' -- and read back
    by name; plain layers keep /<layer>/<layer>/<weight>:0."""
    from image_captioning_amd import hdf5_lite as H
    rng = np.random.default_rng(1)
    W = {"fpn_p2/kernel": rng.standard_normal((3, 3, 4, 4)).astype(np.float32), "fpn_p2/bias": rng.standard_normal(4).astype(np.float32),
         "imgcap_lstm1/kernel": rng.standard_normal((6, 8)).astype(np.float32), "imgcap_lstm1/recurrent_kernel": rng.standard_normal((2, 8)).astype(np.float32),
         "imgcap_lstm_d2/kernel": rng.standard_normal((4, 5)).astype(np.float32), "rpn_conv_shared/bias": rng.standard_normal(3).astype(np.float32)}
    path = str(tmp_path / "nested.h5")
    H.save_keras_weights(path, W, layer_groups={"imgcap_lstm1": "imgcap_caption_td", "imgcap_lstm_d2": "imgcap_caption_td"})
    f = H.H5File(path)
    assert H._attr_list(f, "layer_names") == ["fpn_p2", "imgcap_caption_td", "rpn_conv_shared"]
    td = f["imgcap_caption_td"]
    assert H._attr_list(td, "weight_names") == ["imgcap_lstm1/kernel:0", "imgcap_lstm1/recurrent_kernel:0", "imgcap_lstm_d2/kernel:0"]
    assert np.array_equal(td["imgcap_lstm1/recurrent_kernel:0"].read(), W["imgcap_lstm1/recurrent_kernel"])
    assert np.array_equal(f["fpn_p2"]["fpn_p2/kernel:0"].read(), W["fpn_p2/kernel"])
    back = H.load_keras_weights(path)
    assert set(back) == set(W) and all(np.array_equal(back[k], W[k]) for k in W)


def test_hdf5_writer_lists_weights_in_keras_layer_weights_order(tmp_path):
    """Keras' load_weights_from_hdf5_group_by_name matches a GROUP by name and then assigns weight_values[i] to layer.weights[i] by
    POSITION: Conv2D / Dense = kernel, bias; LSTM = kernel, recurrent_kernel, bias; BatchNorm = gamma, beta, moving_mean,
    moving_variance; TimeDistributed(Model) = the model's trainable weights in layer order, then its non-trainable ones
    (word_generation_model, dense_img_cap/dense_model.py:758-784: lstm1, lstm2, d1, d2, then the frozen embedding).  The writer
    must list them so whatever order the parameter store hands them over in (sorted by name here, like ParamStore)."""
    from image_captioning_amd import hdf5_lite as H
    rng = np.random.default_rng(2)
    r = lambda *s: rng.standard_normal(s).astype(np.float32)
    W = {"imgcap_embedding_layer/embeddings": r(9, 3),
         "imgcap_lstm1/bias": r(8), "imgcap_lstm1/kernel": r(5, 8), "imgcap_lstm1/recurrent_kernel": r(2, 8),
         "imgcap_lstm2/bias": r(8), "imgcap_lstm2/kernel": r(2, 8), "imgcap_lstm2/recurrent_kernel": r(2, 8),
         "imgcap_lstm_d1/bias": r(4), "imgcap_lstm_d1/kernel": r(4, 4), "imgcap_lstm_d2/bias": r(9), "imgcap_lstm_d2/kernel": r(4, 9),
         "mrcnn_class_bn1/beta": r(4), "mrcnn_class_bn1/gamma": r(4), "mrcnn_class_bn1/moving_mean": r(4), "mrcnn_class_bn1/moving_variance": r(4),
         "mrcnn_class_conv1/bias": r(4), "mrcnn_class_conv1/kernel": r(7, 7, 2, 4)}
    W = {k: W[k] for k in sorted(W)}
    inner = ["imgcap_lstm1", "imgcap_lstm2", "imgcap_lstm_d1", "imgcap_lstm_d2", "imgcap_embedding_layer"]
    path = str(tmp_path / "order.h5")
    H.save_keras_weights(path, W, layer_groups={l: "imgcap_caption_td" for l in inner}, group_member_order={"imgcap_caption_td": inner})
    f = H.H5File(path)
    assert H._attr_list(f["imgcap_caption_td"], "weight_names") == [
        "imgcap_lstm1/kernel:0", "imgcap_lstm1/recurrent_kernel:0", "imgcap_lstm1/bias:0",
        "imgcap_lstm2/kernel:0", "imgcap_lstm2/recurrent_kernel:0", "imgcap_lstm2/bias:0",
        "imgcap_lstm_d1/kernel:0", "imgcap_lstm_d1/bias:0", "imgcap_lstm_d2/kernel:0", "imgcap_lstm_d2/bias:0",
        "imgcap_embedding_layer/embeddings:0"]
    assert H._attr_list(f["mrcnn_class_bn1"], "weight_names") == ["mrcnn_class_bn1/gamma:0", "mrcnn_class_bn1/beta:0",
                                                                  "mrcnn_class_bn1/moving_mean:0", "mrcnn_class_bn1/moving_variance:0"]
    assert H._attr_list(f["mrcnn_class_conv1"], "weight_names") == ["mrcnn_class_conv1/kernel:0", "mrcnn_class_conv1/bias:0"]
    back = H.load_keras_weights(path)
    assert set(back) == set(W) and all(np.array_equal(back[k], W[k]) for k in W)


def test_hdf5_reader_rejects_what_it_does_not_implement(tmp_path):
    from image_captioning_amd import hdf5_lite as H
    with pytest.raises(H.Hdf5Error):
        H.H5File(b"not an hdf5 file" * 64)
    good = open(os.path.join(GOLDEN, "keras_weights_sample.h5"), "rb").read()
    bad = bytearray(good)
    bad[8] = 3                                                              # superblock version 3 (libver='latest')
    with pytest.raises(NotImplementedError):
        H.H5File(bytes(bad))


def test_vgg16_layer_table_flops():
    """The alternative backbone's 13-conv table: Keras-applications names, 641.43 GFLOP per 1024x1024 image."""
    from image_captioning_amd.layers import vgg16_convs
    L = vgg16_convs()
    assert [s.name for s in L][:3] == ["block1_conv1", "block1_conv2", "block2_conv1"] and len(L) == 13
    assert all(s.k == 3 and s.stride == 1 and s.padding == "same" and s.bn is None for s in L)
    fl = sum(2.0 * (1024 // 2 ** (int(s.name[5]) - 1)) ** 2 * 9 * s.cin * s.cout for s in L)
    assert abs(fl / 1e9 - 641.43) < 0.01


def test_fit_generator_runs_the_generator_on_a_background_thread_like_keras():
    """keras fit_generator(workers=1, max_queue_size=q) as the reference calls it (text_generation_model_v2.py:300-313): batches
    are produced on ONE background thread ahead of the training loop, never more than q (+ the one in hand) ahead; order is the
    generator's; workers=0 runs it inline; a generator error reaches the caller; losses are averaged per epoch."""
    import threading
    import time
    import torch
    from image_captioning_amd.keras_like import KerasLikeModel, GeneratorEnqueuer

    main = threading.get_ident()
    produced, consumed, threads = [], [], set()

    def gen(n=10 ** 6, fail_at=None):
        for i in range(n):
            if fail_at is not None and i == fail_at:
                raise RuntimeError("bad batch %d" % i)
            threads.add(threading.get_ident())
            produced.append(i)
            yield [np.full((2, 3), i, np.float32)], np.array([i])

    class Toy(KerasLikeModel):
        def train_on_batch_device(self, inputs, targets):
            consumed.append(int(targets[0]))
            assert len(produced) - len(consumed) <= self.q + 2          # bounded run-ahead: queue + one in each hand
            time.sleep(0.002)
            return torch.tensor([float(inputs[0][0, 0])])

        def test_on_batch(self, inputs, targets):
            return 7.0

    m = Toy()
    m.q = 3
    hist = m.fit_generator(gen(), epochs=2, steps_per_epoch=5, max_queue_size=3, workers=1, verbose=0, validation_data=([0], [0]))
    assert consumed == list(range(10)) and threads and main not in threads
    assert [h["loss"] for h in hist] == [2.0, 7.0] and hist[0]["val_loss"] == 7.0
    time.sleep(0.1)
    assert len(produced) <= 10 + 3 + 2                                   # the worker stopped with the loop (bounded queue)
    produced.clear(); consumed.clear(); threads.clear()
    m.q = 0
    m.fit_generator(gen(), epochs=1, steps_per_epoch=4, workers=0, verbose=0)
    assert threads == {main} and produced == consumed == [0, 1, 2, 3]   # inline: exactly one batch per step
    with pytest.raises(RuntimeError, match="bad batch 2"):
        m.q = 10
        m.fit_generator(gen(fail_at=2), epochs=1, steps_per_epoch=5, workers=1, verbose=0)
    with pytest.raises(NotImplementedError):
        m.fit_generator(gen(), epochs=1, steps_per_epoch=1, use_multiprocessing=True)
    e = GeneratorEnqueuer(iter([1, 2, 3]), workers=1, max_queue_size=2)
    assert list(e.get()) == [1, 2, 3]                                    # a finite generator ends the stream cleanly
    e.stop()


def test_encoder_plan_per_layer_arithmetic_rule():
    """EncoderPlan._layer_math (host logic, no GPU): in the fp32-grade plan the stem, the projection shortcuts and the short-K pointwise
    layers that measured faster in split-bf16 arithmetic are the ones handed to DC_MATH_BF16X3; everything else, external layers and
    every layer of the other arithmetic modes keep the plan's arithmetic."""
    from types import SimpleNamespace
    from image_captioning_amd import _lib
    from image_captioning_amd.encoder import EncoderPlan
    from image_captioning_amd.layers import resnet_fpn_convs
    specs = {s.name: s for s in resnet_fpn_convs(22)}
    plan = SimpleNamespace(math=_lib.MATH_F32, layer_math=True, _external={})
    pick = {n for n, s in specs.items() if EncoderPlan._layer_math(plan, s, n) == _lib.MATH_BF16X3}
    assert {"conv1", "res3a_branch1", "res4a_branch1", "res5a_branch1", "fpn_c2p2", "fpn_c3p3", "res3d_branch2c", "res4w_branch2c",
            "res5a_branch2c", "res5b_branch2c", "res5c_branch2c"} <= pick
    # faster on the fp32 pipe (tools/conv_bench.py): deep K, the strided 2a layers, Cin = 64, every 3x3 layer
    assert not pick & {"res4b_branch2a", "res5b_branch2a", "fpn_c4p4", "fpn_c5p5", "res3a_branch2a", "res4a_branch2a", "res5a_branch2a",
                       "res2a_branch1", "res2a_branch2a", "res2a_branch2c", "res4b_branch2b", "fpn_p2"}
    plan.layer_math = False
    assert all(EncoderPlan._layer_math(plan, s, n) == _lib.MATH_F32 for n, s in specs.items())
    plan.layer_math, plan._external = True, {"conv1": None}
    assert EncoderPlan._layer_math(plan, specs["conv1"], "conv1") == _lib.MATH_F32               # trainable / external weights: the plan's arithmetic
    plan.math, plan._external = _lib.MATH_BF16X2, {}
    assert all(EncoderPlan._layer_math(plan, s, n) == _lib.MATH_BF16X2 for n, s in specs.items())


def test_evaluator_decode_rois_equals_the_reference_per_roi_loop():
    """DenseCaptioningEvaluator.decode_rois batches the reference's loop (evaluate_models/test_score_dense_captions.py:213-224: per RoI,
    PADDING_SIZE - 1 predict calls on the pre-padded argmax ids so far, starting from the zero word): same probabilities row by row."""
    import types
    from image_captioning_amd.test_score_dense_captions import DenseCaptioningEvaluator
    from image_captioning_amd.text_generation_model_v2 import pad_sequences
    V, T, N = 7, 5, 4
    rng = np.random.default_rng(0)
    table = rng.standard_normal((V, V, 3))

    class Fake(object):
        calls = 0

        def predict(self, inputs):
            feat, words = inputs
            Fake.calls += 1
            z = np.stack([table[w[-1], :, 0] * f[0] + table[w[-2], :, 1] + (w > 0).sum() * table[0, :, 2] for f, w in zip(feat, np.asarray(words))])
            e = np.exp(z - z.max(axis=1, keepdims=True))
            return e / e.sum(axis=1, keepdims=True)
    feats = rng.standard_normal((N, 2))
    ev = DenseCaptioningEvaluator(Fake(), None, "METEOR", None, None, None, types.SimpleNamespace(PADDING_SIZE=T), "t")
    got = ev.decode_rois(feats)
    assert got.shape == (N, T - 1, V) and Fake.calls == T - 1
    for j in range(N):                                              # the reference's loop, one RoI at a time
        prev = [np.zeros(V)]
        for _ in range(T - 1):
            res = Fake().predict([np.array([feats[j]]), np.array([pad_sequences([[np.argmax(c) for c in prev]], T)[0]])])
            prev.append(res[0])
        np.testing.assert_allclose(got[j], np.vstack(prev[1:]), rtol=1e-12)
