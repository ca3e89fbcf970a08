"""Golden vectors produced by RUNNING THE REFERENCE'S OWN CODE in this container (not a restatement).

The reference's Keras/TensorFlow graph code cannot run here (tensorflow / keras / skimage / nltk are not installed and there
is no network), but its host-side geometry is plain NumPy.  This script
  * imports  dense_img_cap/config.py  as a module (it only needs numpy), and
  * takes the pure-NumPy FUNCTIONS it needs out of  dense_img_cap/utils.py  and  dense_img_cap/dense_model.py  by parsing
    those files with `ast` at run time and compiling just these function definitions into a namespace that holds `np`,
    `math` and `random` -- the modules' top-level `import tensorflow / keras / skimage` lines are never executed, no
    stand-in for a missing library is written, and no reference source text is stored in this repository;
then calls them on seeded inputs and stores inputs + outputs in  tests/golden/reference_host_vectors.npz .

tests/test_golden_reference.py checks the oracle (oracle/np_oracle.py) and the product's host code against these
vectors.  What this pins: Config's derived attributes, anchor generation, IoU, NMS, box-delta application / refinement,
image meta, mold/unmold, RPN target building (incl. its use of np.random), clip_to_window, unmold_generations, the Dataset
base class, resize_image's padding/window logic (scale == 1: scipy.misc.imresize no longer exists), load_image_gt and the
joint model's data_generator (all six input arrays), the v1 create_roi_info / data_generator batch layout (features come
from a seeded table standing in for the Keras feature model, which is the generator's INPUT, not code under test) and the
vocabulary helpers load_corpus / encode_word / encode_word_v2 / decode_word / decode_caption; since round 6 also the evaluation script's
post-processing (evaluate_models/test_score_dense_captions.py: refine_generations, unmold_generations, clip_to_window, merge_boxes,
assign_detections_to_ground_truth, with evaluate_models/utils.py's own overlap and NMS), v2 load_sequences, and box_refinement /
compute_overlaps of the separate-models utils.py.  NOT pinnable this way: the v2 data_generator (it calls keras' pad_sequences).
What stays unpinned: every Keras/TF layer's arithmetic (conv, BN, LSTM, crop_and_resize, losses, optimizer).

Run:  python tests/golden/make_reference_vectors.py        (needs /root/reference; the GPU box never runs this)
"""
import ast
import importlib.util
import math
import os
import random
import types

import numpy as np

REF = "/root/reference/dense_img_cap"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_host_vectors.npz")


def functions_from(path, names, extra_globals=None):
    """Compile the named top-level (or class-level) function definitions of a reference file, and nothing else."""
    with open(path, "r", encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)
    ns = {"np": np, "math": math, "random": random}
    ns.update(extra_globals or {})
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names and node.name not in found:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), ns)
            found[node.name] = ns[node.name]
    missing = set(names) - set(found)
    if missing:
        raise RuntimeError("not found in %s: %s" % (path, sorted(missing)))
    return types.SimpleNamespace(**found)


def main():
    spec = importlib.util.spec_from_file_location("ref_config", os.path.join(REF, "config.py"))
    ref_config = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_config)
    U = functions_from(os.path.join(REF, "utils.py"),
                       ["compute_iou", "compute_overlaps", "non_max_suppression", "apply_box_deltas", "box_refinement",
                        "generate_anchors", "generate_pyramid_anchors", "trim_zeros"])
    D = functions_from(os.path.join(REF, "dense_model.py"),
                       ["compose_image_meta", "parse_image_meta", "mold_image", "unmold_image", "build_rpn_targets",
                        "clip_to_window", "unmold_generations"], extra_globals={"utils": U})
    out = {}

    # ---- Config: class defaults and the attributes __init__ derives
    class Cfg(ref_config.Config):
        NAME = "golden"
        IMAGES_PER_GPU = 3
        GPU_COUNT = 2
        IMAGE_MAX_DIM = 512
        IMAGE_MIN_DIM = 384
    c = Cfg()
    base = ref_config.Config
    for k in ("BACKBONE_STRIDES", "RPN_ANCHOR_SCALES", "RPN_ANCHOR_RATIOS", "RPN_ANCHOR_STRIDE", "RPN_NMS_THRESHOLD",
              "RPN_TRAIN_ANCHORS_PER_IMAGE", "POST_NMS_ROIS_TRAINING", "POST_NMS_ROIS_INFERENCE", "IMAGE_MIN_DIM", "IMAGE_MAX_DIM",
              "MEAN_PIXEL", "TRAIN_ROIS_PER_IMAGE", "ROI_POSITIVE_RATIO", "POOL_SIZE", "MAX_GT_INSTANCES", "RPN_BBOX_STD_DEV",
              "BBOX_STD_DEV", "DETECTION_MAX_INSTANCES", "DETECTION_NMS_THRESHOLD", "LEARNING_RATE", "LEARNING_MOMENTUM",
              "WEIGHT_DECAY", "STEPS_PER_EPOCH", "VALIDATION_STEPS", "GPU_COUNT", "IMAGES_PER_GPU"):
        out["config_default/" + k] = np.asarray(getattr(base, k), dtype=np.float64)
    out["config_derived/BATCH_SIZE"] = np.asarray(c.BATCH_SIZE)
    out["config_derived/IMAGE_SHAPE"] = np.asarray(c.IMAGE_SHAPE)
    out["config_derived/BACKBONE_SHAPES"] = np.asarray(c.BACKBONE_SHAPES)

    # ---- anchors
    for S in (256, 1024):
        shapes = np.array([[int(math.ceil(S / s)), int(math.ceil(S / s))] for s in base.BACKBONE_STRIDES])
        a = U.generate_pyramid_anchors(base.RPN_ANCHOR_SCALES, base.RPN_ANCHOR_RATIOS, shapes, base.BACKBONE_STRIDES, 1)
        if S == 256:
            out["anchors256"] = a
        else:
            out["anchors1024/count"] = np.asarray(a.shape[0])
            out["anchors1024/head"], out["anchors1024/tail"] = a[:64], a[-64:]
            out["anchors1024/every4099"] = a[::4099]
            out["anchors1024/colsum"] = a.sum(axis=0)
    out["anchors_single"] = U.generate_anchors([32, 64], [0.5, 1, 2], [3, 5], 16, 2)

    # ---- IoU / NMS / box deltas
    rng = np.random.RandomState(0)

    def boxes(n, lo=0, hi=200):
        y, x = rng.uniform(lo, hi, n), rng.uniform(lo, hi, n)
        return np.stack([y, x, y + rng.uniform(4, 90, n), x + rng.uniform(4, 90, n)], axis=1)
    b1, b2 = boxes(60), boxes(7)
    b1[10] = b2[3]                                           # an exact match
    out["iou/b1"], out["iou/b2"], out["iou/out"] = b1, b2, U.compute_overlaps(b1, b2)
    nb = boxes(120)
    nb[40:80] = nb[:40] + rng.normal(0, 3, (40, 4))          # heavy overlaps
    ns = rng.uniform(0, 1, 120)
    ns[5] = ns[6]                                            # a tie
    out["nms/boxes"], out["nms/scores"] = nb, ns
    for t in (0.3, 0.5, 0.7):
        out["nms/keep_%02d" % int(t * 10)] = U.non_max_suppression(nb.copy(), ns.copy(), t)
    ib = np.round(boxes(30)).astype(np.int32)
    out["nms/int_boxes"], out["nms/int_keep"] = ib, U.non_max_suppression(ib, ns[:30].copy(), 0.5)
    d = rng.normal(0, 0.4, (60, 4))
    out["deltas/in"], out["deltas/applied"] = d, U.apply_box_deltas(b1.copy(), d)
    g = b1 + rng.normal(0, 6, b1.shape)
    g[:, 2:] = np.maximum(g[:, 2:], g[:, :2] + 2)
    out["refine/gt"], out["refine/out"] = g, U.box_refinement(b1.copy(), g)
    tz = np.round(boxes(9)).astype(np.int32)
    tz[[2, 5, 8]] = 0
    out["trim/in"], out["trim/out"] = tz, U.trim_zeros(tz)

    # ---- image meta / mold
    win = (0, 16, 384, 496)
    meta = D.compose_image_meta(17, (480, 640, 3), win)
    out["meta/one"] = meta
    mid, mshape, mwin = D.parse_image_meta(np.stack([meta, D.compose_image_meta(3, (100, 50, 3), (1, 2, 3, 4))]))
    out["meta/id"], out["meta/shape"], out["meta/window"] = mid, mshape, mwin
    img = rng.randint(0, 256, (5, 7, 3)).astype(np.uint8)
    molded = D.mold_image(img, base)
    out["mold/img"], out["mold/out"], out["mold/back"] = img, molded, D.unmold_image(molded, base)

    # ---- RPN targets (consumes np.random exactly as the reference does)
    class TCfg(ref_config.Config):
        NAME = "t"
        RPN_TRAIN_ANCHORS_PER_IMAGE = 64
    tcfg = TCfg()
    gt = np.array([[10, 12, 70, 90], [40, 30, 120, 128], [0, 0, 50, 40], [100, 100, 250, 240], [5, 150, 60, 250]], np.int32)
    for seed in (0, 1):
        np.random.seed(100 + seed)
        match, bbox = D.build_rpn_targets((256, 256, 3), out["anchors256"], None, gt, tcfg)
        out["rpn_targets/match_%d" % seed], out["rpn_targets/bbox_%d" % seed] = match, bbox
    out["rpn_targets/gt"] = gt

    # ---- inference-side box handling
    w2 = np.array([0, 16, 128, 112])
    rb = boxes(12, -20, 140)
    out["clip/window"], out["clip/in"], out["clip/out"] = w2, rb, D.clip_to_window(w2, rb.copy())
    gen = np.concatenate([np.rint(out["clip/out"]), rng.uniform(0, 1, (12, 6))], axis=1)
    gen[4, :4] = [5, 20, 5, 60]                              # zero area: dropped
    ub, uc = D.unmold_generations(None, gen.copy(), (256, 192, 3), w2)
    out["unmold/in"], out["unmold/boxes"], out["unmold/captions"] = gen, ub, uc

    # ---- Dataset base class, resize_image (scale == 1), load_image_gt, the joint data_generator
    import logging
    U2 = functions_from(os.path.join(REF, "utils.py"),
                        ["Dataset", "resize_image", "compute_iou", "compute_overlaps", "generate_anchors", "generate_pyramid_anchors"])
    D2 = functions_from(os.path.join(REF, "dense_model.py"),
                        ["compose_image_meta", "mold_image", "build_rpn_targets", "load_image_gt", "data_generator"],
                        extra_globals={"utils": U2, "logging": logging})

    class GCfg(ref_config.Config):
        NAME = "gen"
        IMAGES_PER_GPU = 1
        IMAGE_MIN_DIM = 96
        IMAGE_MAX_DIM = 128
        RPN_TRAIN_ANCHORS_PER_IMAGE = 32
        MAX_GT_INSTANCES = 4
        PADDING_SIZE = 6
    gcfg = GCfg()

    def toy_image(i):
        return np.random.RandomState(1000 + i).randint(0, 256, (96 if i % 2 == 0 else 128, 128, 3)).astype(np.uint8)

    def toy_regions(i):
        r = np.random.RandomState(2000 + i)
        n = 6 if i == 0 else 2
        y, x = r.randint(0, 60, n), r.randint(0, 60, n)
        bx = np.stack([y, x, y + r.randint(8, 60, n), x + r.randint(8, 60, n)], axis=1)
        return bx, r.randint(1, 9, (n, 6)).astype(np.float32)

    class Toy(U2.Dataset):
        def load_image(self, image_id):
            return toy_image(image_id)

        def load_captions_and_rois(self, image_id):
            return toy_regions(image_id)
    ds = Toy()
    for i in range(3):
        ds.add_image("toy", image_id=i, path="img%d" % i, width=128, height=96)
    ds.prepare()
    out["dataset/image_ids"] = np.asarray(ds.image_ids)
    out["dataset/num_images"] = np.asarray(ds.num_images)
    img0, win0, sc0, pad0 = U2.resize_image(toy_image(0), min_dim=96, max_dim=128, padding=True)
    out["resize/image"], out["resize/window"], out["resize/scale"], out["resize/padding"] = img0, np.asarray(win0), np.asarray(sc0), np.asarray(pad0)
    np.random.seed(7)
    random.seed(7)
    gen = D2.data_generator(ds, gcfg, shuffle=False, augment=False, batch_size=1)
    for b in range(3):
        inputs, outputs = next(gen)
        assert outputs == []
        for j, name in enumerate(("images", "image_meta", "rpn_match", "rpn_bbox", "gt_captions", "gt_boxes")):
            out["joint_gen/%d/%s" % (b, name)] = inputs[j]

    # ---- v1 generator (dense_img_cap_separate_models/text_generation_model.py): batch layout and one-hot targets
    SEP = "/root/reference/dense_img_cap_separate_models"
    table = {i: np.random.RandomState(3000 + i).standard_normal((4, 2, 2, 3)).astype(np.float32) for i in range(3)}
    V1 = functions_from(os.path.join(SEP, "text_generation_model.py"), ["create_roi_info", "data_generator"],
                        extra_globals={"generate_features": lambda dataset, image_id, model: table[image_id]})

    class ToyV1:
        _image_ids = np.arange(3)

        def load_captions_and_rois(self, image_id):
            r = np.random.RandomState(4000 + image_id)
            caps = np.zeros((2 + image_id, 5), np.float32)
            for k in range(caps.shape[0]):
                n = r.randint(1, 4)
                caps[k, 0], caps[k, 1:1 + n], caps[k, 1 + n] = 1, r.randint(3, 11, n), 2
            return None, caps
    dv1 = ToyV1()
    dv1.rois = V1.create_roi_info(dv1)
    out["v1_gen/roi_count"] = np.asarray(len(dv1.rois))
    out["v1_gen/roi_image_ids"] = np.asarray([r[0] for r in dv1.rois])
    vcfg = types.SimpleNamespace(VOCABULARY_SIZE=12)
    g1 = V1.data_generator(dv1, None, vcfg, 4)
    for b in range(3):
        (feat, words), onehot = next(g1)
        out["v1_gen/%d/features" % b], out["v1_gen/%d/words" % b], out["v1_gen/%d/onehot" % b] = feat, words, onehot

    # ---- mold_inputs of the RoI feature extractor (dense_img_cap_separate_models/modified_dense_model.py)
    US = functions_from(os.path.join(SEP, "utils.py"), ["resize_image"])
    MD = functions_from(os.path.join(SEP, "modified_dense_model.py"), ["mold_inputs", "mold_image", "compose_image_meta"],
                        extra_globals={"utils": US})
    holder = types.SimpleNamespace(config=gcfg)
    molded, metas, windows = MD.mold_inputs(holder, [toy_image(0), toy_image(1)])
    out["mold_inputs/molded"], out["mold_inputs/metas"], out["mold_inputs/windows"] = molded, metas, windows

    # ---- vocabulary helpers
    P = functions_from(os.path.join(SEP, "preprocess.py"), ["load_corpus", "encode_word", "encode_word_v2", "decode_word", "decode_caption"])
    emb = {w: np.random.RandomState(50 + i).standard_normal(8) for i, w in enumerate(["a", "red", "car", "dog"])}
    np.random.seed(11)
    w2i, i2w, mat = P.load_corpus(["a", "red", "car", "dog"], emb, 8)
    out["vocab/matrix"] = mat
    out["vocab/ids"] = np.asarray([w2i[w] for w in ["a", "red", "car", "dog"]])
    out["vocab/specials"] = np.asarray([w2i[k] for k in sorted(k for k in w2i if k.startswith("<"))])
    out["vocab/special_names"] = np.asarray(sorted(k for k in w2i if k.startswith("<")))
    out["vocab/encode_known_unknown"] = np.asarray([P.encode_word("car", w2i), P.encode_word("zebra", w2i)])
    onehots = np.eye(len(i2w))[[w2i["red"], w2i["dog"]]]
    out["vocab/decode_caption"] = np.asarray(P.decode_caption(onehots, i2w))


    # ---- round 6: the evaluation script's NumPy post-processing (evaluate_models/test_score_dense_captions.py + its own utils.py, whose
    # compute_iou is 2 * intersection / (area + area)), v2 load_sequences, and the separate-models copy of box_refinement / compute_iou
    EV = "/root/reference/evaluate_models"
    UE = functions_from(os.path.join(EV, "utils.py"), ["compute_iou", "compute_overlaps", "non_max_suppression"])
    E = functions_from(os.path.join(EV, "test_score_dense_captions.py"), ["DenseCaptioningEvaluator"],
                       extra_globals={"compute_overlaps": UE.compute_overlaps, "non_max_suppression": UE.non_max_suppression})
    ev = E.DenseCaptioningEvaluator(None, None, "METEOR", None, None, None, None, "golden")
    er = np.random.RandomState(61)
    out["eval/iou_b1"], out["eval/iou_b2"], out["eval/overlaps"] = b1, b2, UE.compute_overlaps(b1, b2)
    for t in (0.3, 0.5):
        out["eval/nms_keep_%02d" % int(t * 10)] = UE.non_max_suppression(nb.copy(), ns.copy(), t)
    out["eval/nms_int_keep"] = UE.non_max_suppression(ib, ns[:30].copy(), 0.5)
    N, Tm1, V = 40, 5, 9
    y, x = er.uniform(0, 0.7, N), er.uniform(0, 0.7, N)
    rois_n = np.stack([y, x, y + er.uniform(0.05, 0.3, N), x + er.uniform(0.05, 0.3, N)], axis=1).astype(np.float32)
    rois_n[20:30] = rois_n[:10] + er.normal(0, 0.01, (10, 4)).astype(np.float32)      # heavy overlaps
    probs = er.dirichlet(np.ones(V) * 0.3, (N, Tm1))
    probs[7] = probs[3]                                      # two captions with the same score
    probs[31] = probs[12]
    ecfg = types.SimpleNamespace(DETECTION_NMS_THRESHOLD=0.5, DETECTION_MAX_INSTANCES=12)
    rb_, rc_ = ev.refine_generations(rois_n, probs, np.array([0, 0, 96, 128]), ecfg)
    out["eval/refine_rois_in"], out["eval/refine_probs_in"] = rois_n, probs
    out["eval/refine_boxes"], out["eval/refine_captions"] = rb_, rc_
    ecfg2 = types.SimpleNamespace(DETECTION_NMS_THRESHOLD=0.3, DETECTION_MAX_INSTANCES=100)
    rb2, rc2 = ev.refine_generations(rois_n * 128.0, probs, np.array([0, 0, 96, 128]), ecfg2)
    out["eval/refine2_boxes"], out["eval/refine2_captions"] = rb2, rc2
    w3 = np.array([16, 0, 112, 128])
    out["eval/unmold_in"], out["eval/unmold_window"] = rb2, w3
    out["eval/unmold_out"] = ev.unmold_generations(rb2.copy(), (300, 400, 3), w3)
    cb = boxes(15, -30, 150)
    out["eval/clip_in"], out["eval/clip_out"] = cb, ev.clip_to_window(w3, cb.copy())
    gb = np.round(boxes(14, 0, 100)).astype(np.int64)
    gb[5:9] = gb[:4] + er.randint(-3, 4, (4, 4))             # near-duplicates that merge
    gb[9] = gb[0] + 1
    caps_txt = [["caption %d" % i] for i in range(14)]
    mb, mc = E.DenseCaptioningEvaluator.merge_boxes(gb.copy(), caps_txt, 0.7)
    out["eval/merge_in"], out["eval/merge_boxes"] = gb, mb
    out["eval/merge_caption_ids"] = np.asarray([-1 if j is None else j for grp in mc for j in
                                                [int(c[0].split()[1]) for c in grp] + [None]])
    det = [np.round(boxes(10, 0, 100)), np.round(boxes(6, 0, 100))]
    gtb = [det[0][[1, 4, 4, 7]] + er.randint(-4, 5, (4, 4)), np.round(boxes(3, 200, 300))]        # image 1: no overlap at all
    lp = [er.uniform(-9, -1, 10), er.uniform(-9, -1, 6)]
    lp[0][2] = lp[0][6]
    dcap = [["det %d" % i for i in range(10)], ["det %d" % i for i in range(6)]]
    gcap = [[["ref %d" % i] for i in range(4)], [["ref %d" % i] for i in range(3)]]
    rec = E.DenseCaptioningEvaluator.assign_detections_to_ground_truth(2, gtb, gcap, det, dcap, lp)
    for i in range(2):
        out["eval/assign%d_det" % i], out["eval/assign%d_gt" % i], out["eval/assign%d_lp" % i] = det[i], gtb[i], lp[i]
        out["eval/assign%d_ok" % i] = np.asarray([r["ok"] for r in rec[i]])
        out["eval/assign%d_ov" % i] = np.asarray([r["ov"] for r in rec[i]])
        out["eval/assign%d_candidate" % i] = np.asarray([int(r["candidate"].split()[1]) for r in rec[i]])
        out["eval/assign%d_reference" % i] = np.asarray([int(r["references"][0].split()[1]) if r["references"] else -1 for r in rec[i]])

    from tqdm import tqdm
    V2 = functions_from(os.path.join(SEP, "text_generation_model_v2.py"), ["load_sequences"], extra_globals={"tqdm": tqdm})

    class ToyV2:
        _image_ids = np.array([4, 0, 2])

        def load_captions_and_rois(self, image_id):
            r = np.random.RandomState(5000 + image_id)
            caps = []
            for _ in range(1 + image_id % 3):
                ids = r.randint(1, 11, r.randint(1, 5))
                caps.append(np.eye(11)[ids])
            return None, np.array(caps, dtype=object) if len({len(c) for c in caps}) > 1 else np.array(caps)
    seqs = V2.load_sequences(ToyV2())
    out["v2_seq/count"] = np.asarray(len(seqs))
    out["v2_seq/image_roi_next"] = np.asarray([[s_[0], s_[1], s_[3]] for s_ in seqs])
    out["v2_seq/prefix_flat"] = np.asarray([w for s_ in seqs for w in list(s_[2]) + [-1]])

    USEP = functions_from(os.path.join(SEP, "utils.py"), ["compute_iou", "compute_overlaps", "box_refinement", "compute_recall"])
    for thr in (0.3, 0.5):
        rec, pos = USEP.compute_recall(b1, b2, thr)
        out["sep/recall_%02d" % int(thr * 10)], out["sep/recall_pos_%02d" % int(thr * 10)] = np.asarray(rec), pos
    out["sep/overlaps"] = USEP.compute_overlaps(b1, b2)
    out["sep/refine"] = USEP.box_refinement(b1.copy(), g)
    ib2 = np.round(b1).astype(np.int32)
    out["sep/refine_int_in"], out["sep/refine_int_gt"] = ib2, np.round(g).astype(np.int32)
    out["sep/refine_int"] = USEP.box_refinement(ib2, np.round(g).astype(np.int32))

    # encode_word_v2 on a word outside the vocabulary: the reference looks up '<UNK>' in a table whose key is '<unk>'
    try:
        P.encode_word_v2("zebra", w2i)
        out["vocab/encode_v2_oov_raises"] = np.asarray(0)
    except KeyError:
        out["vocab/encode_v2_oov_raises"] = np.asarray(1)
    out["vocab/encode_v2_known"] = P.encode_word_v2("car", w2i)

    np.savez_compressed(OUT, **out)
    print("wrote %s: %d arrays, %.1f KB" % (OUT, len(out), os.path.getsize(OUT) / 1024))


if __name__ == "__main__":
    main()
