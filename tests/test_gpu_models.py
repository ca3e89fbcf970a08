"""Model-level parity on the GPU (-m gpu): encoder, v2 decoders (as-written batches and the
single-pass caption form), optimizer trajectory and greedy decode against the NumPy oracle.
Tolerance per north_star: logits/probabilities within 1e-3 in fp32; token ids bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import np_models as M
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu
MEAN = [123.7, 116.8, 103.9]


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from image_captioning_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.fixture(params=["f32", "bf16x3"])
def conv_math(request, monkeypatch):
    """Every model-level test that runs convolutions runs twice: with exact fp32 MFMA products and with the split-bf16
    arithmetic (three bf16 pieces per operand, six matrix-pipe products) -- held to the SAME tolerances, which is the
    "fp32-grade" claim of DESIGN.md section 5.  The encoder plan reads DCAP_CONV_MATH when it is built."""
    monkeypatch.setenv("DCAP_CONV_MATH", request.param)
    return request.param


def rel_err(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    return float(np.abs(got - want).max()) / max(1e-30, float(np.abs(want).max()))


def make_v2(V, inject, Tw, seed=0):
    from image_captioning_amd import synth
    from image_captioning_amd.text_generation_model_v2 import DenseCapConfig, build_model, Adam
    E = synth.embedding_matrix(seed + 3, V)
    cfg = DenseCapConfig(V, E)
    cfg.PADDING_SIZE = Tw
    model = build_model((7, 7, 256), (Tw,), cfg, 256, inject, seed=seed)
    model.compile(optimizer=Adam(amsgrad=True), loss="categorical_crossentropy")
    Wt = {k: v.astype(np.float64) for k, v in model.get_weights_dict().items()}
    return model, Wt


def test_encoder_matches_oracle(gpu, conv_math):
    from image_captioning_amd import synth
    from image_captioning_amd.config import Config
    from image_captioning_amd.modified_dense_model import DenseImageCapRCNN

    class Cfg(Config):
        IMAGES_PER_GPU = 2
        IMAGE_MIN_DIM = 256
        IMAGE_MAX_DIM = 256
    Wt = synth.encoder_weights(0, stage4_blocks=2)
    img = synth.images(0, 2, 256, 256)
    rois = synth.rois(1, 2, 16, 256, 256, lo=16, hi=256)
    want, maps = M.encoder_features(img, rois, Wt, MEAN, stage4_blocks=2, return_maps=True)
    model = DenseImageCapRCNN("inference", Cfg(), "logs", stage4_blocks=2)
    model.set_weights(Wt)
    for rep in range(3):                      # eager, graph capture, graph replay
        got = model.extract_features(img, rois).cpu().numpy()
        plan = model.plan(2, 256, 256)
        for a, b in zip(plan.P, maps[4:]):
            assert rel_err(a.cpu().numpy(), b) < 2e-4, "pyramid map, rep %d" % rep
        assert rel_err(got, want) < 2e-4, "rep %d" % rep
    out = model.generate_captions([img[0], img[1]], rois)
    assert out[0]['features'].shape == (16, 7, 7, 256)
    assert rel_err(out[1]['features'], want[1]) < 2e-4


def test_encoder_winograd_and_direct_plans_agree(gpu):
    """EncoderPlan(winograd=True) (the default for conv_math='f32': frozen 3x3 / stride 1 layers in the Winograd F(2x2,3x3) form) against
    winograd=False (the direct implicit GEMM everywhere): the same layers report the Winograd kernels, every pyramid map and the RoI
    features agree to fp32 rounding, and both stay within the oracle tolerance of test_encoder_matches_oracle."""
    from image_captioning_amd import synth
    from image_captioning_amd.encoder import EncoderPlan
    Wt = synth.encoder_weights(0, stage4_blocks=2)
    img = torch.tensor(synth.images(0, 2, 256, 256), device="cuda")
    rois = synth.rois(1, 2, 16, 256, 256, lo=16, hi=256)
    want = M.encoder_features(img.cpu().numpy(), rois, Wt, MEAN, stage4_blocks=2)
    feats, maps = {}, {}
    for wino in (True, False):
        plan = EncoderPlan(Wt, 2, 256, 256, "cuda", stage4_blocks=2, mean_pixel=MEAN, winograd=wino)
        names = {name: key for name, fl, bm, bn, sk, key in plan.conv_table()}
        wl = sorted(n for n, k in names.items() if k.startswith("wino"))
        if wino:
            assert "res2a_branch2b" in wl and "res4b_branch2b" in wl and "fpn_p2" in wl and "fpn_p5" in wl and len(wl) == 3 + 4 + 3 + 3 + 4
            assert not any(n.endswith("branch2a") or n.endswith("branch2c") or n.startswith("fpn_c") or n == "conv1" for n in wl)
        else:
            assert not wl
        for _ in range(2):                                 # eager, then the captured graph
            plan.forward(img)
            feats[wino] = plan.roi_features(rois).cpu().numpy()
        maps[wino] = [t.cpu().numpy() for t in plan.P]
        assert rel_err(feats[wino], want) < 2e-4
    assert rel_err(feats[True], feats[False]) < 2e-5
    for a, b in zip(maps[True], maps[False]):
        assert rel_err(a, b) < 2e-5


@pytest.mark.parametrize("inject", [True, False])
def test_v2_as_written_batch(gpu, inject):
    """The reference's own batch layout: predict, loss, every trainable gradient, three optimizer steps."""
    from image_captioning_amd import synth
    V, Tw, R = 1000, 6, 5
    model, Wt = make_v2(V, inject, Tw)
    rng = np.random.default_rng(1)
    feat_r = rng.standard_normal((R, 7, 7, 256)).astype(np.float32)
    caps = synth.captions_v2(2, R, Tw + 2, V, full=False, lmin=1)          # some prefixes longer than the window
    roi, words, tgt = M.v2_expand_samples(caps, Tw)
    feat = feat_r[roi]
    onehot = np.eye(V)[tgt]
    probs = model.predict([feat, words])
    want_p, _ = M.v2_forward(Wt, feat, words, inject)
    assert np.abs(probs - want_p).max() < 1e-5
    assert np.abs(np.log(probs + 1e-30) - np.log(want_p + 1e-30)).max() < 1e-3        # logits within 1e-3
    opt = M.AMSGrad()
    for step in range(3):
        loss, G, _ = M.v2_loss_and_grads(Wt, feat, words, tgt, inject)
        got_loss = model.train_on_batch([feat, words], onehot)
        assert abs(got_loss - loss) < 1e-4 * max(1.0, abs(loss))
        for k in G:
            g = model.store.grad[k].cpu().numpy()
            assert rel_err(g, G[k]) < 2e-4 or np.abs(G[k]).max() < 1e-12, (k, step)
        opt.step(Wt, G)
        for k in G:
            assert np.abs(model.store.w[k].cpu().numpy() - Wt[k]).max() < 2e-5, (k, step)
    assert model.optimizer.iterations == 3


@pytest.mark.parametrize("inject", [True, False])
def test_v2_single_pass_equals_expanded_batch(gpu, inject):
    """train_on_captions (one teacher-forced pass) == the reference's per-prefix batch."""
    from image_captioning_amd import synth
    V, Tw, R = 1000, 15, 12
    model, Wt = make_v2(V, inject, Tw, seed=3)
    rng = np.random.default_rng(4)
    feat_r = rng.standard_normal((R, 7, 7, 256)).astype(np.float32)
    caps = synth.captions_v2(5, R, Tw, V, full=False, lmin=1)
    roi, words, tgt = M.v2_expand_samples(caps, Tw)
    loss, G, _ = M.v2_loss_and_grads(Wt, feat_r[roi], words, tgt, inject)
    got = float(model.train_on_captions(feat_r, caps).item())
    assert abs(got - loss) < 1e-4 * max(1.0, abs(loss))
    for k in G:
        assert rel_err(model.store.grad[k].cpu().numpy(), G[k]) < 2e-4 or np.abs(G[k]).max() < 1e-12, k


def test_v2_greedy_decode_ids_bit_exact(gpu):
    V, Tw = 1000, 8
    model, Wt = make_v2(V, True, Tw, seed=7)
    rng = np.random.default_rng(8)
    for r in range(3):
        feat = rng.standard_normal((7, 7, 256)).astype(np.float32)
        ids, rows = model.greedy_decode(feat)
        want_ids, want_rows = M.v2_greedy_decode(Wt, feat, Tw, Tw - 1)
        np.testing.assert_array_equal(ids, want_ids)
        assert np.abs(rows - want_rows).max() < 1e-5


def test_full_size_properties(gpu, conv_math):
    """BASELINE-size decoder step (64 captions x 15 tokens, V=10000): properties that need no oracle run:
    probabilities sum to 1, loss at step 0 ~ ln V for near-uniform init, the loss falls over steps,
    and two models fed the same data stay bit-identical (determinism)."""
    from image_captioning_amd import synth
    V, T, R = 10000, 15, 64
    losses = []
    flats = []
    for rep in range(2):
        model, _ = make_v2(V, True, T, seed=11)
        rng = np.random.default_rng(12)
        feat = torch.tensor(rng.standard_normal((R, 7, 7, 256)).astype(np.float32), device=gpu)
        caps = synth.captions_v2(13, R, T, V, full=True)
        ls = [float(model.train_on_captions(feat, caps).item()) for _ in range(8)]
        losses.append(ls)
        flats.append(model.store.flat.cpu().numpy().copy())
    assert abs(losses[0][0] - np.log(V)) < 0.5
    assert losses[0][-1] < losses[0][0]
    assert losses[0] == losses[1]
    np.testing.assert_array_equal(flats[0], flats[1])


def make_v1(V, T, B, seed=0):
    from image_captioning_amd import synth
    from image_captioning_amd.text_generation_model import DenseCapConfig, build_lstm_model, Adam, roi_caption_loss
    cfg = DenseCapConfig(V, synth.embedding_matrix(seed + 3, V), B)
    cfg.PADDING_SIZE = T
    model = build_lstm_model([7, 7, 256], cfg, 512, 'training', seed=seed)
    assert model.recurrent_dropout == 0.2         # the reference's training default (text_generation_model.py:141-142)
    model.recurrent_dropout = 0.0                 # parity against the deterministic oracle graph
    model.compile(optimizer=Adam(amsgrad=True), loss=roi_caption_loss)
    Wt = {k: v.astype(np.float64) for k, v in model.get_weights_dict().items()}
    return model, Wt, cfg


def test_v1_training_graph_matches_as_written_oracle(gpu):
    """Model 3: the single masked pass == the reference's T-prefix TimeDistributed graph
    (probabilities, roi_caption_loss, every gradient incl. the trainable RoI head, two AMSGrad steps)."""
    from image_captioning_amd import synth
    from image_captioning_amd.text_generation_model import caption_targets
    V, T, B = 1000, 6, 4
    model, Wt, cfg = make_v1(V, T, B)
    rng = np.random.default_rng(21)
    feat = rng.standard_normal((B, 7, 7, 256)).astype(np.float32)
    caps = synth.captions_v1(22, B, T, V, lmin=1, lmax=3)
    probs = model.predict([feat, caps])
    want_p, _ = M.v1_training_forward(Wt, feat, caps)
    assert probs.shape == (B, T, V)
    assert np.abs(probs - want_p).max() < 1e-5
    assert np.abs(np.log(probs + 1e-30) - np.log(want_p + 1e-30)).max() < 1e-3
    np.testing.assert_array_equal(caption_targets(caps), M.v1_targets(caps))
    onehot = caption_targets(caps, V)
    opt = M.AMSGrad()
    for step in range(2):
        loss, G, _ = M.v1_loss_and_grads(Wt, feat, caps)
        got = model.train_on_batch([feat, caps], onehot)
        assert abs(got - loss) < 1e-4 * max(1.0, abs(loss))
        assert set(G) == set(model.trainable_weights)
        for k in G:
            assert rel_err(model.store.grad[k].cpu().numpy(), G[k]) < 3e-4, (k, step)
        opt.step(Wt, G)
        for k in G:
            assert np.abs(model.store.w[k].cpu().numpy() - Wt[k]).max() < 2e-5, (k, step)


def test_v1_recurrent_dropout_default_is_seeded_and_matches_the_oracle_given_its_masks(gpu):
    """recurrent_dropout=0.2 (text_generation_model.py:141-142) is the training default; a train step draws seeded
    per-gate masks for both LSTMs, and loss + every gradient equal the oracle's T-prefix graph run with those masks;
    test_on_batch / predict (learning phase 0) never apply them."""
    from image_captioning_amd import synth
    from image_captioning_amd.text_generation_model import DenseCapConfig, build_lstm_model, Adam, caption_targets
    V, T, B = 40, 6, 4
    cfg = DenseCapConfig(V, synth.embedding_matrix(3, V), B)
    cfg.PADDING_SIZE = T
    model = build_lstm_model([7, 7, 256], cfg, 512, 'training', seed=0)
    model.compile(optimizer=Adam(amsgrad=True), loss="roi_caption_loss")
    assert model.recurrent_dropout == 0.2
    model.recurrent_dropout = 0.0
    Wt = {k: v.astype(np.float64) for k, v in model.get_weights_dict().items()}
    rng = np.random.default_rng(1)
    feat = rng.standard_normal((B, 7, 7, 256)).astype(np.float32)
    caps = synth.captions_v1(2, B, T, V, lmin=1, lmax=3)
    tg = caption_targets(caps)
    plain = model.test_on_batch([feat, caps], tg)
    model.recurrent_dropout = 0.2
    assert model.test_on_batch([feat, caps], tg) == plain and model.last_rec_masks is None      # evaluation: no dropout
    loss_rows, _ = model._forward_train(model._dev_feat(feat), caps, tg, want_grad=True)
    model._backward()
    masks = [m.astype(np.float64) for m in model.last_rec_masks]
    assert all(set(np.unique(m)) <= {0.0, 1.25} and m.shape == (4, B, 512) for m in masks) and abs(masks[0].mean() - 1.0) < 0.05
    want_loss, G, _ = M.v1_loss_and_grads(Wt, feat.astype(np.float64), caps, rec_masks=tuple(masks))
    got_loss = float(loss_rows.mean().item())
    assert abs(got_loss - want_loss) < 1e-5 * max(1.0, abs(want_loss)) and abs(got_loss - plain) > 2e-6            # dropout did change the forward pass
    for k, g in G.items():
        assert rel_err(model.store.grad[k].cpu().numpy(), g) < 3e-4, k
    again = build_lstm_model([7, 7, 256], cfg, 512, 'training', seed=0)                          # same seed: same masks (the default rate)
    again._forward_train(again._dev_feat(feat), caps, tg, want_grad=True)
    assert all(np.array_equal(a, b) for a, b in zip(again.last_rec_masks, model.last_rec_masks))


def _run_v2_steps(monkeypatch, graph, inject=True):
    """Seven train steps of the v2 decoder over two alternating batch shapes, a predict() in between; DCAP_STEP_GRAPH on or off."""
    from image_captioning_amd import synth
    from image_captioning_amd.text_generation_model_v2 import SampleTables
    monkeypatch.setenv("DCAP_STEP_GRAPH", "1" if graph else "0")
    V, Tw = 400, 6
    model, _ = make_v2(V, inject, Tw)
    model.use_step_graph = True                             # (opt-in for this model: the as-written batch is GPU-bound either way)
    rng = np.random.default_rng(4)
    losses = []
    for step in range(7):
        R = 6 if step % 3 != 2 else 4                      # shape A: steps 0 1 3 4 6 (captured on its third call); shape B: steps 2 5
        feat = torch.tensor(rng.standard_normal((R, 7, 7, 256)).astype(np.float32), device="cuda")
        caps = synth.captions_v2(10 + step, R, 5, V, full=True)
        losses.append(model.train_on_captions(feat, SampleTables.from_captions(caps, "cuda")))
        losses[-1] = float(losses[-1].item())
        if step == 3:
            model.predict([feat.cpu().numpy(), np.zeros((R, Tw), np.int32)])   # other buffer shapes in between: the graph keeps its own
    captured = sorted(k[0][0] for k, cs in model._steps.items() if cs.graph is not None)
    return losses, {k: v.cpu().numpy() for k, v in model.store.w.items()}, model.optimizer.iterations, captured


@pytest.mark.parametrize("inject", [True, False])
def test_v2_captured_train_step_is_bit_equal_to_the_eager_step(gpu, monkeypatch, inject):
    """CaptionModelV2.train_step replayed from a hipGraph (step_graph.py: third call with a batch shape captures, later calls
    replay; the batch arrives through two device copies and one word) against DCAP_STEP_GRAPH=0: every loss and every weight after
    seven steps identical bit for bit, Keras' iteration counter advanced by the replays."""
    le, we, ite, cap_e = _run_v2_steps(monkeypatch, False, inject)
    lg, wg, itg, cap_g = _run_v2_steps(monkeypatch, True, inject)
    assert cap_e == [] and cap_g == [6]                     # shape A was captured, shape B (two calls) was still warming up
    assert ite == itg == 7
    assert le == lg
    for k in we:
        assert np.array_equal(we[k], wg[k]), k


def _run_v1_steps(monkeypatch, graph, rate):
    from image_captioning_amd import synth
    from image_captioning_amd.text_generation_model import caption_targets
    monkeypatch.setenv("DCAP_STEP_GRAPH", "1" if graph else "0")
    V, T, B = 200, 6, 4
    model, _, _ = make_v1(V, T, B)
    model.recurrent_dropout = rate
    rng = np.random.default_rng(8)
    losses, masks = [], []
    for step in range(6):
        feat = rng.standard_normal((B, 7, 7, 256)).astype(np.float32)
        caps = synth.captions_v1(30 + step, B, T, V, lmin=1, lmax=4)
        losses.append(model.train_on_batch([feat, caps], caption_targets(caps, V)))
        masks.append(None if model.last_rec_masks is None else model.last_rec_masks[1].copy())
    n_graphs = sum(cs.graph is not None for cs in model._steps.values())
    return losses, masks, {k: v.cpu().numpy() for k, v in model.store.w.items()}, (model.optimizer.iterations, model._drop_step), n_graphs


@pytest.mark.parametrize("rate", [0.0, 0.2])
def test_v1_captured_train_step_is_bit_equal_to_the_eager_step(gpu, monkeypatch, rate):
    """CaptionModelV1.train_step (BASELINE configs[0]'s step) replayed from a hipGraph against the eager step: losses, weights and
    -- with the reference's recurrent_dropout -- the masks of every step identical (a replay draws fresh masks: the stream position is
    a device word of the step's one upload), host counters advanced."""
    le, me, we, ce, ne = _run_v1_steps(monkeypatch, False, rate)
    lg, mg, wg, cg, ng = _run_v1_steps(monkeypatch, True, rate)
    assert ne == 0 and ng == 1 and ce == cg == (6, 6 if rate else 0)
    assert le == lg
    for a, b in zip(me, mg):
        assert (a is None and b is None) or np.array_equal(a, b)
    if rate:
        assert not np.array_equal(mg[4], mg[5])            # two replays, two different mask draws
    for k in we:
        assert np.array_equal(we[k], wg[k]), k


def test_v1_captured_step_is_keyed_on_the_dropout_rate(gpu, monkeypatch):
    """Changing recurrent_dropout between train steps must not replay a graph captured at the other rate: the rate is part of the
    captured step's key (with dropout the mask kernels are launches of the step)."""
    from image_captioning_amd import synth
    from image_captioning_amd.text_generation_model import caption_targets
    V, T, B = 200, 6, 4
    rng = np.random.default_rng(11)
    feat = rng.standard_normal((B, 7, 7, 256)).astype(np.float32)
    caps = synth.captions_v1(41, B, T, V, lmin=1, lmax=4)
    tg = caption_targets(caps, V)
    out = {}
    for graph in (True, False):
        monkeypatch.setenv("DCAP_STEP_GRAPH", "1" if graph else "0")
        model, _, _ = make_v1(V, T, B)
        model.recurrent_dropout = 0.2
        losses = [model.train_on_batch([feat, caps], tg) for _ in range(4)]          # the fourth step replays the rate-0.2 graph
        assert model.last_rec_masks is not None
        model.recurrent_dropout = 0.0
        losses += [model.train_on_batch([feat, caps], tg) for _ in range(2)]
        assert model.last_rec_masks is None                                           # no mask kernels ran: not the 0.2 graph
        model.recurrent_dropout = 0.2
        losses.append(model.train_on_batch([feat, caps], tg))                        # back on the captured 0.2 graph, fresh masks
        out[graph] = (losses, model.store.flat.cpu().numpy(), model._drop_step)
    assert out[True][0] == out[False][0] and np.array_equal(out[True][1], out[False][1]) and out[True][2] == out[False][2] == 5


def test_v1_recurrent_dropout_per_prefix_rows_match_the_as_written_graph(gpu):
    """dropout_rows='prefix': every (RoI, prefix) row of the TimeDistributed batch gets its own masks, as Keras draws them
    (text_generation_model.py:179-187, :141-142); the LSTMs run over the B*T zero-padded prefixes.  Loss and every gradient equal
    the oracle's T-prefix graph given those [T,4,B,512] masks; the weights after AMSGrad steps follow the oracle's; evaluation
    and the dropout-free graph are untouched by the switch."""
    from image_captioning_amd import synth
    from image_captioning_amd.text_generation_model import DenseCapConfig, build_lstm_model, Adam, caption_targets
    V, T, B = 40, 6, 4
    cfg = DenseCapConfig(V, synth.embedding_matrix(3, V), B)
    cfg.PADDING_SIZE = T
    model = build_lstm_model([7, 7, 256], cfg, 512, 'training', seed=0)
    model.compile(optimizer=Adam(amsgrad=True), loss="roi_caption_loss")
    model.dropout_rows = "prefix"
    Wt = {k: v.astype(np.float64) for k, v in model.get_weights_dict().items()}
    rng = np.random.default_rng(1)
    feat = rng.standard_normal((B, 7, 7, 256)).astype(np.float32)
    caps = synth.captions_v1(2, B, T, V, lmin=1, lmax=3)
    tg = caption_targets(caps)
    model.recurrent_dropout = 0.0
    plain = model.test_on_batch([feat, caps], tg)
    model.recurrent_dropout = 0.2
    assert model.test_on_batch([feat, caps], tg) == plain                      # learning phase 0: single pass, no masks
    opt = M.AMSGrad()
    for step in range(2):
        loss = model.train_on_batch([feat, caps], np.eye(V)[tg])
        masks = [m.astype(np.float64).reshape(4, T, B, 512).transpose(1, 0, 2, 3) for m in model.last_rec_masks]   # row j*B+b -> [j][:, b]
        assert all(set(np.unique(m)) <= {0.0, 1.25} for m in masks)
        assert not np.array_equal(masks[0][0], masks[0][1])                    # prefixes of a RoI do not share masks
        want_loss, G, _ = M.v1_loss_and_grads(Wt, feat.astype(np.float64), caps, rec_masks=tuple(masks))
        assert abs(loss - want_loss) < 1e-5 * max(1.0, abs(want_loss)), (step, loss, want_loss)
        for k in G:
            assert rel_err(model.store.grad[k].cpu().numpy(), G[k]) < 3e-4, (k, step)
        opt.step(Wt, G)
        for k in G:
            assert np.abs(model.store.w[k].cpu().numpy() - Wt[k]).max() < 2e-5, (k, step)
    with pytest.raises(ValueError):
        model.dropout_rows = "rows"
        model.train_on_batch([feat, caps], np.eye(V)[tg])


def test_v1_greedy_inference_ids_bit_exact(gpu):
    from image_captioning_amd import synth
    from image_captioning_amd.text_generation_model import DenseCapConfig, build_lstm_model
    V, T, B = 1000, 6, 3
    cfg = DenseCapConfig(V, synth.embedding_matrix(33, V), B)
    cfg.PADDING_SIZE = T
    model = build_lstm_model([7, 7, 256], cfg, 512, 'inference', seed=30)
    Wt = {k: v.astype(np.float64) for k, v in model.get_weights_dict().items()}
    feat = np.random.default_rng(31).standard_normal((B, 7, 7, 256)).astype(np.float32)
    probs = model.predict(feat)
    want_p, want_ids = M.v1_greedy_decode(Wt, feat, T)
    _, ids = model.generate(feat)
    np.testing.assert_array_equal(ids, want_ids)
    assert np.abs(probs - want_p).max() < 1e-5


def test_image_level_features_with_rpn_proposals(gpu, conv_math):
    """feature_generation/ variant: RPN + ProposalLayer + RoIAlign + mean over RoIs.  Scores reach the sort with
    fp32 rounding differences, so near-tied candidates may swap: the proposal SETS must agree almost everywhere and
    the pooled 12 544-vector within 1e-3."""
    from image_captioning_amd import synth
    from image_captioning_amd.generate_roi_features import InferenceConfig, generate_features, load_model
    S = 256

    class Cfg(InferenceConfig):
        IMAGE_MIN_DIM = S
        IMAGE_MAX_DIM = S
        POST_NMS_ROIS_INFERENCE = 200
    cfg = Cfg()
    Wt = dict(synth.encoder_weights(0, 2), **synth.rpn_weights(4))
    Wt['rpn_conv_shared/kernel'] = Wt['rpn_conv_shared/kernel'] * np.float32(0.02)   # random FPN maps are O(10): keep the RPN
    Wt['rpn_bbox_pred/kernel'] = Wt['rpn_bbox_pred/kernel'] * np.float32(0.3)        # logits / deltas in a trained net's range
    img = synth.images(3, 1, S, S)
    model = load_model(weights=Wt, config=cfg, stage4_blocks=2)
    vec = generate_features(img[0], model)
    assert vec.shape == (12544,)
    cl = dict(scales=cfg.RPN_ANCHOR_SCALES, ratios=cfg.RPN_ANCHOR_RATIOS, strides=cfg.BACKBONE_STRIDES, count=200, nms=0.7)
    feats, props, _ = M.image_level_encoder_features(img, Wt, MEAN, cl, stage4_blocks=2)
    got_props = model.last_proposals.cpu().numpy()[0]
    a = {tuple(np.round(r.astype(np.float64), 4)) for r in got_props}
    b = {tuple(np.round(r, 4)) for r in props[0].astype(np.float64)}
    assert len(b) > 100, "degenerate test: the oracle produced only %d distinct proposals" % len(b)
    assert len(a & b) >= 0.9 * len(b), "only %d of %d proposals agree" % (len(a & b), len(b))
    want = M.image_level_features(feats[0])
    assert np.abs(vec - want).max() < 2e-2 * np.abs(want).max()


# ---------------------------------------------------------------------------------------------
# joint model (configs[4]): dense_img_cap/dense_model.py
# ---------------------------------------------------------------------------------------------

def make_joint(S=128, V=24, T=5, blocks=1, rois=12, compute_dtype="f32", conv_math=None):
    from image_captioning_amd import synth
    from image_captioning_amd.config import Config
    from image_captioning_amd.dense_model import DenseImageCapRCNN

    class Cfg(Config):
        NAME = "joint"
        IMAGES_PER_GPU = 1
        IMAGE_MIN_DIM = S
        IMAGE_MAX_DIM = S
        POST_NMS_ROIS_TRAINING = 60
        TRAIN_ROIS_PER_IMAGE = rois
        PADDING_SIZE = T
        VOCABULARY_SIZE = V
        EMBEDDING_SIZE = 300
        RECURRENT_DROPOUT = 0.0          # parity against the deterministic oracle graph (the training default is the reference's 0.2)
    cfg = Cfg()
    Wt = dict(synth.encoder_weights(0, blocks), **synth.rpn_weights(4))
    Wt['rpn_conv_shared/kernel'] = Wt['rpn_conv_shared/kernel'] * np.float32(0.05)
    Wt['rpn_bbox_pred/kernel'] = Wt['rpn_bbox_pred/kernel'] * np.float32(0.3)
    Wt.update(synth.head_weights(1))
    Wt['mrcnn_class_conv1/kernel'] = Wt['mrcnn_class_conv1/kernel'] * np.float32(0.05)   # random FPN maps are O(10): keep the
    Wt.update(synth.v1_weights(2, V))                                                    # vocabulary softmax out of saturation
    Wt['imgcap_embedding_layer/embeddings'] = synth.embedding_matrix(3, V)
    cfg.EMBEDDING_WEIGHTS = Wt['imgcap_embedding_layer/embeddings']
    model = DenseImageCapRCNN("training", cfg, "logs", stage4_blocks=blocks, compute_dtype=compute_dtype, conv_math=conv_math)
    model.set_weights(Wt)
    return model, cfg, Wt


def joint_inputs(S, V, T, seed=8):
    from image_captioning_amd import synth
    rng = np.random.default_rng(seed)
    img = synth.images(7, 1, S, S)
    gt_boxes = np.zeros((1, 6, 4), np.float32)
    gt_boxes[0, :3] = np.array([[10, 12, 70, 90], [40, 30, 120, 128], [0, 0, 50, 40]], np.float32) * (S / 128.0)
    gt_caps = np.zeros((1, 6, T), np.int32)
    gt_caps[0, :3] = synth.captions_v1(9, 3, T, V, lmin=1, lmax=3)
    n_anchor = sum((S // s) ** 2 for s in (4, 8, 16, 32, 64)) * 3
    match = np.zeros((1, n_anchor, 1), np.int32)
    match[0, rng.choice(n_anchor, 40, replace=False), 0] = np.where(rng.random(40) < 0.4, 1, -1)
    tdelta = rng.standard_normal((1, 64, 4))
    return [img, np.zeros((1, 12)), match, tdelta, gt_caps, gt_boxes]


def joint_oracle(Wt, cfg, inputs, targets, blocks, backbone_from=None):
    img, _, match, tdelta, gt_caps, gt_boxes = inputs
    oc = dict(mean_pixel=MEAN, scales=cfg.RPN_ANCHOR_SCALES, ratios=cfg.RPN_ANCHOR_RATIOS, strides=cfg.BACKBONE_STRIDES,
              proposal_count=cfg.POST_NMS_ROIS_TRAINING, nms=cfg.RPN_NMS_THRESHOLD, train_rois=cfg.TRAIN_ROIS_PER_IMAGE,
              positive_ratio=cfg.ROI_POSITIVE_RATIO, weight_decay=cfg.WEIGHT_DECAY, T=cfg.PADDING_SIZE)
    return M.joint_loss_and_grads(Wt, img[0], match[0, :, 0], tdelta[0], gt_caps[0], gt_boxes[0], oc, stage4_blocks=blocks,
                                  targets_override=targets, backbone_from=backbone_from)


def joint_grads_as_reference(model):
    """The flat gradient bucket re-expressed with the reference's layer names and HWIO kernels."""
    st = model.store
    saved = st.flat.clone()
    st.flat.copy_(st.flat_grad)
    try:
        out = model.get_weights_dict()
    finally:
        st.flat.copy_(saved)
    return out


def test_joint_model_step_matches_oracle(gpu, conv_math):
    """One image through FPN + RPN + proposals + detection targets + RoIAlign + head + Model-3 decoder: the four loss
    terms and the gradient of every trainable weight against the oracle's hand-written backward (which
    tests/test_oracle_vs_torch.py ties to torch autograd), given the RoI sample the device drew."""
    S, V, T, blocks = 128, 24, 5, 1
    model, cfg, Wt = make_joint(S, V, T, blocks)
    inputs = joint_inputs(S, V, T)
    for rep in range(2):                                   # eager plan, then captured graph
        losses = model._loss_list(model.forward_backward(inputs, shuffle=None))
    tg = model.last_targets
    assert tg['npos'] > 0, "no positive RoI: the caption loss is not exercised"
    want, G, aux = joint_oracle(Wt, cfg, inputs, (tg['rois'], tg['caps']), blocks)
    # the device's own proposals reproduce the oracle's sample unless near-tied scores swapped
    agree = sum(1 for r in tg['rois'] if np.abs(aux['proposals'] - r).sum(1).min() < 1e-5)
    assert agree >= 0.8 * (tg['npos'] + tg['nneg'])
    for k in ('imgcap_loss', 'rpn_class_loss', 'rpn_bbox_loss', 'reg_loss', 'loss'):
        assert abs(losses[k] - want[k]) < 1e-4 * max(1.0, abs(want[k])), (k, losses[k], want[k])
    got = joint_grads_as_reference(model)
    for k in M.joint_trainable(Wt):
        assert rel_err(got[k], G[k]) < 2e-4, (k, rel_err(got[k], G[k]))


@pytest.mark.parametrize("layers,stage", [("5+", 5), ("4+", 4), ("3+", 3), ("all", 1)])
def test_joint_model_trains_resnet_stages(gpu, layers, stage):
    """train(layers="5+" | "4+" | "3+" | "all") (dense_img_cap/dense_model.py:1829-1845): the ResNet stages' convolutions and BatchNorm
    gammas / betas (frozen statistics) join the trainable set.  set_trainable moves them into the parameter bucket without changing
    any weight; losses and the gradient of EVERY trainable weight -- through bottleneck blocks, strided stage entries, projection
    shortcuts, the max pool and the stem for "all" -- equal the oracle's backward; an optimizer step then moves them, the moving
    statistics stay put, and the weights round-trip through get_weights_dict / set_weights."""
    S, V, T, blocks = 128, 24, 5, 1
    model, cfg, Wt = make_joint(S, V, T, blocks)
    inputs = joint_inputs(S, V, T)
    base = model._loss_list(model.forward_backward(inputs, shuffle=None))
    model.set_trainable(model.LAYER_REGEX[layers])
    assert model.backbone_from == stage
    back = model.get_weights_dict()
    assert set(back) == set(Wt) and all(np.array_equal(back[k], np.asarray(Wt[k], np.float32)) for k in Wt)
    for rep in range(2):                                   # eager plan, then captured graph
        losses = model._loss_list(model.forward_backward(inputs, shuffle=None))
    tg = model.last_targets
    want, G, aux = joint_oracle(Wt, cfg, inputs, (tg['rois'], tg['caps']), blocks, backbone_from=stage)
    trunk = M.backbone_trainable(Wt, stage, blocks)
    assert trunk and set(trunk) <= set(model.trainable_weights) and len(trunk) == 4 * sum(1 for k in Wt if k.endswith('/gamma') and k in trunk)
    for k in ('imgcap_loss', 'rpn_class_loss', 'rpn_bbox_loss', 'loss'):
        assert abs(losses[k] - want[k]) < 1e-4 * max(1.0, abs(want[k])), (k, losses[k], want[k])
        if k != 'loss':
            assert abs(losses[k] - base[k]) < 1e-5 * max(1.0, abs(base[k]))         # moving weights into the bucket changed no arithmetic
    assert losses['reg_loss'] > base['reg_loss'] and abs(losses['reg_loss'] - want['reg_loss']) < 1e-5 * want['reg_loss']
    got = joint_grads_as_reference(model)
    worst = {k: rel_err(got[k], G[k]) for k in M.joint_trainable(Wt) + trunk if np.abs(G[k]).max() > 1e-12}
    assert max(worst.values()) < 5e-4, sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    for k in trunk:                                        # dead units: exactly zero on both sides
        if np.abs(G[k]).max() <= 1e-12:
            assert np.abs(got[k]).max() < 1e-9, k
    model.compile(1e-4)
    model.train_on_batch(inputs)
    after = model.get_weights_dict()
    moved = [k for k in trunk if not np.array_equal(after[k], back[k])]
    assert len(moved) > 0.5 * len(trunk)
    assert all(np.array_equal(after[k], back[k]) for k in Wt if 'moving_' in k)
    frozen = [k for k in Wt if k.split('/')[0].startswith(('res', 'bn', 'conv1')) and k not in trunk and 'moving_' not in k]
    assert all(np.array_equal(after[k], back[k]) for k in frozen)               # stages below the first trainable one stay frozen


@pytest.mark.parametrize("layers,stage", [("4+", 4), ("all", 1)])
def test_joint_model_trains_resnet_stages_in_bf16(gpu, layers, stage):
    """configs[4]'s arithmetic with trainable ResNet stages (train(layers="4+" | "all"), dense_img_cap/dense_model.py:1829-1845):
    compute_dtype='bf16' and the convolutions in bf16 storage (conv_math 'bf16').  Against the float64 oracle at the bf16 tolerances
    of test_joint_model_bf16_step_tracks_the_fp32_oracle: losses 1e-2; every gradient tensor -- decoder, head, FPN / RPN AND the
    trunk's kernels, biases, gammas, betas -- within 1.5e-1 in relative L2 norm.  The error grows with the distance from the loss, as
    rounding through a chain of bf16 products and ReLU kinks does: measured 3.5e-2 for the decoder, 6e-2 at the stage's exit
    (bn4a_branch1), 1.1e-1 for the layers deepest below it (res4a_branch2a / 2b, ~10 bf16 convolutions from the RoI features); the
    same run with conv_math='f32' holds 5e-4 (test_joint_model_trains_resnet_stages).  An optimizer step then moves the trunk; the
    moving statistics stay put."""
    S, V, T, blocks = 128, 24, 8, 1
    model, cfg, Wt = make_joint(S, V, T, blocks, rois=16, compute_dtype="bf16", conv_math="bf16")
    assert model.conv_math_name == "bf16" and model.plan().fast_bf16
    inputs = joint_inputs(S, V, T)
    model.set_trainable(model.LAYER_REGEX[layers])
    assert model.backbone_from == stage and model.conv_math_name == "bf16" and model.plan().fast_bf16
    back = model.get_weights_dict()
    for rep in range(2):
        losses = model._loss_list(model.forward_backward(inputs, shuffle=None))
    tg = model.last_targets
    assert tg['npos'] > 0
    want, G, aux = joint_oracle(Wt, cfg, inputs, (tg['rois'], tg['caps']), blocks, backbone_from=stage)
    for k in ('imgcap_loss', 'rpn_class_loss', 'rpn_bbox_loss', 'reg_loss', 'loss'):
        assert abs(losses[k] - want[k]) < 1e-2 * max(1.0, abs(want[k])), (k, losses[k], want[k])
    got = joint_grads_as_reference(model)
    trunk = M.backbone_trainable(Wt, stage, blocks)
    l2 = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(1e-30, np.linalg.norm(b)))
    worst = {k: l2(got[k], G[k]) for k in M.joint_trainable(Wt) + trunk if np.abs(G[k]).max() > 1e-12}
    assert all(np.isfinite(v) for v in worst.values())
    assert max(worst.values()) < 1.5e-1, sorted(worst.items(), key=lambda kv: -kv[1])[:8]
    model.compile(1e-4)
    out = model.train_on_batch(inputs)
    assert np.isfinite(out).all()
    after = model.get_weights_dict()
    assert sum(1 for k in trunk if not np.array_equal(after[k], back[k])) > 0.5 * len(trunk)
    assert all(np.array_equal(after[k], back[k]) for k in Wt if 'moving_' in k)


def test_joint_model_bf16_step_tracks_the_fp32_oracle(gpu):
    """BASELINE configs[4] arithmetic: RoI head, decoder and vocabulary GEMMs from bf16 copies of weights / activations (fp32
    master, fp32 accumulate).  Same step as above against the float64 oracle; the tolerance is bf16's: every GEMM operand
    carries 2^-9 relative rounding through a chain of ~20 products, so losses agree to 1e-2 and each gradient tensor to 5e-2 in
    relative L2 norm (measured 0.5-3.3e-2).  The norm, not the largest entry: a ReLU / hard-sigmoid unit whose pre-activation
    sits within rounding of its kink switches its whole gradient contribution on or off (single entries off by 10-16 % of the
    tensor's maximum with 3 positive RoIs), which says nothing about the arithmetic."""
    S, V, T, blocks = 128, 24, 8, 1
    model, cfg, Wt = make_joint(S, V, T, blocks, rois=16, compute_dtype="bf16")
    assert model.caption_model.store.flat_bf16 is not None
    inputs = joint_inputs(S, V, T)
    losses = model._loss_list(model.forward_backward(inputs, shuffle=None))
    tg = model.last_targets
    assert tg['npos'] > 0
    want, G, aux = joint_oracle(Wt, cfg, inputs, (tg['rois'], tg['caps']), blocks)
    for k in ('imgcap_loss', 'rpn_class_loss', 'rpn_bbox_loss', 'reg_loss', 'loss'):
        assert abs(losses[k] - want[k]) < 1e-2 * max(1.0, abs(want[k])), (k, losses[k], want[k])
    got = joint_grads_as_reference(model)
    l2 = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(1e-30, np.linalg.norm(b)))
    worst = {k: l2(got[k], G[k]) for k in M.joint_trainable(Wt)}
    assert max(worst.values()) < 5e-2, sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    # the optimizer keeps the bf16 operand copies equal to the rounded fp32 master weights
    model.compile(1e-4)
    model.train_on_batch(inputs)
    st = model.caption_model.store
    assert torch.equal(st.flat_bf16, st.flat[:st.flat_bf16.numel()].to(torch.bfloat16))


def test_joint_model_training_reduces_loss_and_round_trips_weights(gpu, tmp_path, conv_math):
    S, V, T, blocks = 128, 24, 5, 1
    model, cfg, Wt = make_joint(S, V, T, blocks)
    back = model.get_weights_dict()
    for k, v in Wt.items():
        assert np.array_equal(back[k], np.asarray(v, np.float32)), k
    inputs = joint_inputs(S, V, T)
    model.compile(3e-5)
    first = model.train_on_batch(inputs)
    assert len(first) == 4 and abs(first[0] - (first[1] + first[2] + first[3] + model.last_losses['reg_loss'])) < 1e-5
    for _ in range(12):
        last = model.train_on_batch(inputs)
    assert np.isfinite(last).all() and last[1] + last[2] < first[1] + first[2]       # the RPN targets are fixed: its losses fall
    path = str(tmp_path / "joint.npz")
    model.save_weights(path)
    other, _, _ = make_joint(S, V, T, blocks)
    other.load_weights(path, by_name=True)
    a, b = model.get_weights_dict(), other.get_weights_dict()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    # caption_only freezes FPN/RPN/head: their weights stay put, the decoder moves
    other.set_trainable(other.LAYER_REGEX["caption_only"])
    other.compile(1e-3)
    other.train_on_batch(inputs)
    c = other.get_weights_dict()
    assert np.array_equal(b['fpn_p3/kernel'], c['fpn_p3/kernel']) and np.array_equal(b['mrcnn_class_conv1/kernel'], c['mrcnn_class_conv1/kernel'])
    assert not np.array_equal(b['imgcap_lstm_d2/kernel'], c['imgcap_lstm_d2/kernel'])


def test_joint_model_trains_with_the_reference_recurrent_dropout_by_default(gpu):
    """dense_img_cap/dense_model.py:769-770 builds both LSTMs with recurrent_dropout=0.2: without an explicit opt-out the joint
    model's train steps draw fresh masks every step (seeded: two models with one seed agree), validation never does."""
    S, V, T, blocks = 128, 24, 5, 1
    _, cfg, Wt = make_joint(S, V, T, blocks)
    del type(cfg).RECURRENT_DROPOUT                      # back to Config's default
    assert cfg.RECURRENT_DROPOUT == 0.2
    from image_captioning_amd.dense_model import DenseImageCapRCNN
    inputs = joint_inputs(S, V, T)
    runs = []
    for _ in range(2):
        model = DenseImageCapRCNN("training", cfg, "logs", stage4_blocks=blocks)
        model.set_weights(Wt)
        model.compile(1e-5)
        assert model.caption_model.recurrent_dropout == 0.2
        v0 = model.test_on_batch(inputs)
        assert model.caption_model.last_rec_masks is None and np.allclose(v0, model.test_on_batch(inputs), rtol=1e-9)   # learning phase 0: no masks (RPN loss sums are float atomics)
        l1 = model.train_on_batch(inputs)
        m1 = model.caption_model.last_rec_masks
        l2 = model.train_on_batch(inputs)
        m2 = model.caption_model.last_rec_masks
        assert np.isfinite(l1).all() and np.isfinite(l2).all()
        assert m1[0].shape == (4, cfg.TRAIN_ROIS_PER_IMAGE, 512) and set(np.unique(m1[0])) <= {0.0, np.float32(1.25)}
        assert not np.array_equal(m1[0], m2[0]) and not np.array_equal(m1[0], m1[1])                    # per step, per LSTM
        runs.append((l1, l2, m2))
    assert np.allclose(runs[0][0], runs[1][0], rtol=1e-5) and np.allclose(runs[0][1], runs[1][1], rtol=1e-4)      # (RPN loss sums are float atomics)
    assert np.array_equal(runs[0][2][1], runs[1][2][1])                                                           # same seed, same step: same masks
    # DROPOUT_ROWS='prefix': masks per (RoI, prefix) row, the LSTMs over the padded prefixes (Keras' TimeDistributed graph)
    type(cfg).DROPOUT_ROWS = "prefix"
    try:
        model = DenseImageCapRCNN("training", cfg, "logs", stage4_blocks=blocks)
        model.set_weights(Wt)
        model.compile(1e-5)
        v1 = model.test_on_batch(inputs)
        assert np.allclose(v1, v0, rtol=1e-5)                                   # evaluation does not depend on the switch
        lp = model.train_on_batch(inputs)
        mp = model.caption_model.last_rec_masks
        assert np.isfinite(lp).all() and mp[0].shape == (4, T * cfg.TRAIN_ROIS_PER_IMAGE, 512)
        assert np.allclose(lp[1:3], runs[0][0][1:3], rtol=1e-5)                 # the RPN losses do not see the decoder's masks
    finally:
        del type(cfg).DROPOUT_ROWS


def test_joint_model_validation_is_forward_only(gpu):
    """test_on_batch (what train() validates with, like Keras): no gradient, the gradient bucket and the training run's
    detection-target generator untouched, and the same four losses as the training graph's forward on the same RoI sample."""
    S, V, T, blocks = 128, 24, 5, 1
    model, cfg, Wt = make_joint(S, V, T, blocks)
    inputs = joint_inputs(S, V, T)
    model.forward_backward(inputs)                                   # some gradient in the bucket
    g0 = model.store.flat_grad.clone()
    state = (model._dt_step, model.caption_model._drop_step)          # positions of the training run's sampling / dropout streams
    out = model.test_on_batch(inputs)
    assert len(out) == 4 and np.isfinite(out).all()
    assert torch.equal(model.store.flat_grad, g0)
    assert state == (model._dt_step, model.caption_model._drop_step) and model._dt_val_step == 1
    fwd = model._loss_list(model.forward_backward(inputs, shuffle=None, backward=False))
    full = model._loss_list(model.forward_backward(inputs, shuffle=None))
    for k in full:
        assert abs(fwd[k] - full[k]) < 1e-6 * max(1.0, abs(full[k])), (k, fwd[k], full[k])


def test_joint_model_inference_captions(gpu, conv_math):
    """Inference graph of the joint model: proposals -> RoI features -> greedy decoder -> GenerationMatchLayer.  The decoded
    ids of the surviving boxes against the oracle's greedy decoder run on the oracle's features of the same proposals."""
    from image_captioning_amd import synth
    from image_captioning_amd.dense_model import DenseImageCapRCNN
    S, V, T, blocks = 128, 24, 5, 1
    _, cfg, Wt = make_joint(S, V, T, blocks)
    cfg.POST_NMS_ROIS_INFERENCE = 40
    cfg.DETECTION_MAX_INSTANCES = 10
    model = DenseImageCapRCNN("inference", cfg, "logs", stage4_blocks=blocks)
    model.set_weights(Wt)
    img = synth.images(7, 1, S, S)
    res = model.generate_captions([img[0]])[0]
    K = res["rois"].shape[0]
    assert 0 < K <= 10 and res["captions"].shape == (K, T, V) and res["ids"].shape == (K, T)
    assert np.all(res["rois"][:, 2] > res["rois"][:, 0]) and res["rois"].min() >= 0 and res["rois"].max() <= S
    np.testing.assert_allclose(res["captions"].sum(-1), 1.0, atol=1e-5)
    assert np.array_equal(res["captions"].argmax(-1), res["ids"])
    props = model.last_proposals.cpu().numpy()
    x = O.mold_image(img, MEAN)
    _, C2, C3, C4, C5 = M.resnet_graph(x, Wt, blocks)
    maps = M.fpn_graph(C2, C3, C4, C5, Wt)[:4]
    feats = O.pyramid_roi_align(props, list(maps), (S, S, 3), 7)[0]
    want_probs, want_ids = M.v1_greedy_decode(Wt, feats, T)
    # match the survivors back to their proposals through the decoded probabilities
    hits = 0
    for k in range(K):
        d = np.abs(want_probs - res["captions"][k][None]).reshape(len(want_probs), -1).max(1)
        j = int(d.argmin())
        assert d[j] < 1e-3
        hits += int(np.array_equal(want_ids[j], res["ids"][k]))
    assert hits == K
    light = model.generate_captions([img[0]], return_probabilities=False)[0]
    assert "captions" not in light and np.array_equal(light["ids"], res["ids"]) and np.array_equal(light["rois"], res["rois"])


@pytest.mark.parametrize("math,tol", [("f32", 2e-4), ("bf16x3", 2e-4), ("bf16x2", 1e-3), ("bf16", 3e-2)])
def test_encoder_conv_math_modes(gpu, math, tol):
    """The three conv arithmetic modes through the whole ResNet+FPN+RoIAlign stack against the float64 oracle: exact fp32
    products and the 3-piece bf16 split are held to 2e-4 of the feature scale, the 2-piece split (2^-16 products) to the
    1e-3 north-star tolerance (measured ~1e-5: see DESIGN.md)."""
    from image_captioning_amd import synth
    from image_captioning_amd.config import Config
    from image_captioning_amd.modified_dense_model import DenseImageCapRCNN

    class Cfg(Config):
        IMAGES_PER_GPU = 1
        IMAGE_MIN_DIM = 256
        IMAGE_MAX_DIM = 256
    Wt = synth.encoder_weights(0, stage4_blocks=3)
    img = synth.images(5, 1, 256, 256)
    rois = synth.rois(6, 1, 16, 256, 256, lo=16, hi=256)
    want = M.encoder_features(img, rois, Wt, MEAN, stage4_blocks=3)
    model = DenseImageCapRCNN("inference", Cfg(), "logs", stage4_blocks=3, conv_math=math)
    model.set_weights(Wt)
    got = model.extract_features(img, rois).cpu().numpy()
    err = rel_err(got, want)
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/math_mode_errors.txt", "a") as f:
        f.write("%s %.3e\n" % (math, err))
    assert err < tol, (math, err)


def test_joint_model_degenerate_batches(gpu, conv_math):
    """No positive RoI (no GT box overlaps a proposal), no RPN anchors selected: the caption and RPN losses are exactly 0,
    the gradient is the L2 regulariser's alone, nothing is NaN, and an optimizer step still works."""
    S, V, T, blocks = 128, 24, 5, 1
    model, cfg, Wt = make_joint(S, V, T, blocks)
    inputs = joint_inputs(S, V, T)
    inputs[2] = np.zeros_like(inputs[2])                       # every anchor neutral
    inputs[5] = np.zeros_like(inputs[5])
    inputs[5][0, 0] = [0, 0, 1, 1]                             # a 1-pixel GT box: IoU < 0.5 with every proposal
    losses = model._loss_list(model.forward_backward(inputs, shuffle=None))
    assert model.last_targets['npos'] == 0 and model.last_targets['nneg'] == 0
    assert losses['imgcap_loss'] == 0.0 and losses['rpn_class_loss'] == 0.0 and losses['rpn_bbox_loss'] == 0.0
    assert losses['reg_loss'] > 0 and np.isfinite(losses['loss'])
    got = joint_grads_as_reference(model)
    wd = cfg.WEIGHT_DECAY
    for k in ('fpn_p2/kernel', 'rpn_class_raw/kernel', 'imgcap_lstm_d2/kernel', 'mrcnn_class_conv1/bias'):
        want = 2.0 * wd * np.asarray(Wt[k], np.float64) / np.asarray(Wt[k]).size
        assert rel_err(got[k], want) < 1e-5, k
    assert float(np.abs(got['mrcnn_class_bn1/gamma']).max()) == 0.0          # BN gamma/beta are not regularised
    model.compile(1e-4)
    out = model.train_on_batch(inputs)
    assert np.isfinite(out).all()


def test_joint_model_train_loop_checkpoints_and_resumes(gpu, tmp_path):
    """train(): data_generator -> train_on_batch x STEPS_PER_EPOCH -> validation -> checkpoint per epoch -> find_last /
    load_weights resume (the flow of train_dense_captions.py:170-203 on a toy dataset)."""
    from image_captioning_amd import utils
    from image_captioning_amd.dense_model import DenseImageCapRCNN
    S, V, T, blocks = 128, 24, 5, 1
    _, cfg, Wt = make_joint(S, V, T, blocks)
    cfg.STEPS_PER_EPOCH, cfg.MAX_GT_INSTANCES = 2, 6

    class Toy(utils.Dataset):
        def load_image(self, image_id):
            return np.random.RandomState(image_id).randint(0, 255, (S, S, 3)).astype(np.uint8)

        def load_captions_and_rois(self, image_id):
            r = np.random.RandomState(100 + image_id)
            y, x = r.randint(0, 60, 4), r.randint(0, 60, 4)
            boxes = np.stack([y, x, y + r.randint(20, 60, 4), x + r.randint(20, 60, 4)], axis=1)
            caps = np.zeros((4, T), np.float32)
            caps[:, 0], caps[:, 1:3], caps[:, 3] = 1, r.randint(3, V, (4, 2)), 2
            return boxes, caps
    train, val = Toy(), Toy()
    for ds, ids in ((train, range(3)), (val, range(3, 5))):
        for i in ids:
            ds.add_image("toy", image_id=i, path=None)
        ds.prepare()
    model = DenseImageCapRCNN("training", cfg, str(tmp_path / "logs"), stage4_blocks=blocks)
    model.set_weights(Wt)
    history = model.train(train, val, learning_rate=1e-5, epochs=2, layers="no_backbone")
    assert len(history) == 2 and all(np.isfinite(v) for h in history for v in h.values())
    assert set(history[0]) >= {"loss", "rpn_class_loss", "rpn_bbox_loss", "imgcap_loss", "val_loss"}
    folder, last = model.find_last()
    name = cfg.NAME.lower()
    assert os.path.basename(last) == "img_cap_%s_0002.npz" % name and model.epoch == 2
    assert os.path.basename(folder).startswith(name) and os.path.dirname(folder) == str(tmp_path / "logs")
    resumed = DenseImageCapRCNN("training", cfg, str(tmp_path / "logs"), stage4_blocks=blocks)
    assert resumed.epoch == 0
    resumed.load_weights(last, by_name=True)
    assert resumed.epoch == 2 and resumed.log_dir == folder           # the checkpoint's name carries epoch and run directory
    a, b = model.get_weights_dict(), resumed.get_weights_dict()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert not np.array_equal(a['fpn_p2/kernel'], np.asarray(Wt['fpn_p2/kernel'], np.float32))      # it did train
    # train(epochs=3) after the resume runs exactly ONE more epoch (epoch index 2) and writes checkpoint 0003 only
    before = set(os.listdir(folder))
    more = resumed.train(train, val, learning_rate=1e-5, epochs=3, layers="no_backbone")
    assert len(more) == 1 and resumed.epoch == 3
    assert set(os.listdir(folder)) - before == {"img_cap_%s_0003.npz" % name}
    # the next call widens the trainable set to ResNet stage 5 (the reference's schedule: train_dense_captions.py trains the heads first):
    # the bucket is rebuilt around the current weights, one more epoch runs, its checkpoint carries the trained stage-5 weights
    w3 = resumed.get_weights_dict()
    last5 = resumed.train(train, val, learning_rate=1e-5, epochs=4, layers="5+")
    assert len(last5) == 1 and resumed.epoch == 4 and resumed.backbone_from == 5
    w4 = load_ckpt = __import__("image_captioning_amd.modified_dense_model", fromlist=["load_weight_file"]).load_weight_file(
        os.path.join(folder, "img_cap_%s_0004.npz" % name))
    assert not np.array_equal(w4["res5c_branch2c/kernel"], w3["res5c_branch2c/kernel"]) and not np.array_equal(w4["bn5a_branch1/gamma"], w3["bn5a_branch1/gamma"])
    assert np.array_equal(w4["res4a_branch2a/kernel"], w3["res4a_branch2a/kernel"]) and np.array_equal(w4["bn5a_branch1/moving_mean"], w3["bn5a_branch1/moving_mean"])
    assert all(np.isfinite(v).all() for v in w4.values())


def test_vgg16_plan_matches_oracle(gpu):
    """The alternative backbone of the configs[2] label (13 convs + 2x2 pools + RoIAlign on block5_conv3) against the float64
    oracle's conv / ReLU / crop_and_resize on a 64x64 image; eager, graph capture, graph replay."""
    from image_captioning_amd import synth
    from image_captioning_amd.encoder import Vgg16Plan
    from image_captioning_amd.layers import vgg16_convs
    B, S, R = 2, 64, 6
    Wt = synth.vgg16_weights(3)
    img = synth.images(4, B, S, S)
    rois = synth.rois(5, B, R, S, S, lo=8, hi=64)
    x = O.mold_image(img, MEAN)
    specs = vgg16_convs()
    for i, s in enumerate(specs):
        x = O.relu(O.conv2d_nhwc(x, Wt[s.name + "/kernel"], Wt[s.name + "/bias"], 1, "same"))
        last_of_block = i + 1 == len(specs) or specs[i + 1].name.split("_")[0] != s.name.split("_")[0]
        if last_of_block and not s.name.startswith("block5"):
            n, h, w, c = x.shape
            x = x.reshape(n, h // 2, 2, w // 2, 2, c).max(axis=(2, 4))
    assert x.shape == (B, S // 16, S // 16, 512)
    boxes = O.normalize_boxes(rois, S, S)
    want = np.stack([O.crop_and_resize(x, boxes[b], np.full(R, b), (7, 7)) for b in range(B)])
    plan = Vgg16Plan(Wt, B, S, S, "cuda")
    assert abs(plan.flops / B - sum(2.0 * S * S / 4 ** (int(s.name[5]) - 1) * 9 * s.cin * s.cout for s in specs)) < 1
    for rep in range(3):
        plan.forward(torch.as_tensor(img))
        assert rel_err(plan.C[0].cpu().numpy(), x) < 2e-4, rep
        got = plan.roi_features(rois).cpu().numpy()
        assert got.shape == (B, R, 7, 7, 512)
        assert rel_err(got, want) < 2e-4, rep


def test_joint_model_inference_in_bf16(gpu):
    """The inference graph with configs[4]'s arithmetic (bf16 storage between the convolutions, bf16 head / decoder): valid boxes,
    probabilities that sum to one, and word distributions close to the exact-fp32 model's on the boxes both models keep."""
    from image_captioning_amd import synth
    from image_captioning_amd.dense_model import DenseImageCapRCNN
    S, V, T, blocks = 128, 24, 5, 1
    _, cfg, Wt = make_joint(S, V, T, blocks)
    cfg.POST_NMS_ROIS_INFERENCE = 40
    cfg.DETECTION_MAX_INSTANCES = 10
    img = synth.images(7, 1, S, S)
    out = {}
    for name, kw in (("f32", {}), ("bf16", dict(compute_dtype="bf16", conv_math="bf16"))):
        model = DenseImageCapRCNN("inference", cfg, "logs", stage4_blocks=blocks, **kw)
        model.set_weights(Wt)
        out[name] = model.generate_captions([img[0]])[0]
        if name == "bf16":
            assert model.plan().fast_bf16
    res = out["bf16"]
    K = res["rois"].shape[0]
    assert 0 < K <= 10 and res["captions"].shape == (K, T, V) and np.isfinite(res["captions"]).all()
    assert np.all(res["rois"][:, 2] > res["rois"][:, 0]) and res["rois"].min() >= 0 and res["rois"].max() <= S
    np.testing.assert_allclose(res["captions"].sum(-1), 1.0, atol=1e-3)
    common = 0
    for k in range(K):                                     # boxes within a pixel of an fp32 box: same region, compare the words
        d = np.abs(out["f32"]["rois"].astype(np.float64) - res["rois"][k]).max(1)
        if d.size and d.min() <= 1:
            j = int(d.argmin())
            assert np.abs(out["f32"]["captions"][j] - res["captions"][k]).max() < 5e-2
            common += 1
    assert common >= max(1, K // 2)


# ---------------------------------------------------------------------------------------------
# the v2 training script's flow on the device (text_generation_model_v2.py:224-346) and the pipeline route behind it
# ---------------------------------------------------------------------------------------------

def _toy_vg(S, n_images, R, L, V):
    """A VisualGenomeDataset with in-memory pixels: n_images images, R regions each, captions of exactly L in-vocabulary words."""
    from image_captioning_amd.text_generation_model_v2 import VisualGenomeDataset
    w2i = {"<unk>": 0, "<start>": 1, "<end>": 2}
    w2i.update({"w%d" % i: i for i in range(3, V)})
    ds = VisualGenomeDataset(w2i, L + 2)
    rng = np.random.RandomState(5)
    for i in range(n_images):
        y, x = rng.randint(0, S - 40, R), rng.randint(0, S - 40, R)
        rois = [[int(a), int(b), int(a + rng.randint(24, 40)), int(b + rng.randint(24, 40))] for a, b in zip(y, x)]
        caps = [[" ".join("w%d" % rng.randint(3, V) for _ in range(L))] for _ in range(R)]
        ds.add_image("VisualGenome", image_id=1000 + i, path="none", width=S, height=S, rois=rois, captions=caps,
                     pixels=rng.randint(0, 255, (S, S, 3)).astype(np.uint8))
    ds.prepare()
    return ds


def test_v2_script_flow_device_resident_generator_and_train_on_dataset(gpu, tmp_path):
    """The reference's training script flow on a toy Visual Genome: feature model -> load_sequences -> data_generator ->
    fit_generator (background enqueuer, ModelCheckpoint + CSVLogger per epoch) -> load_weights -> greedy decode.  Then the same
    training (a) from the device-resident generator (features stay on the GPU, sparse targets): bit-identical weights; (b) through
    train_on_dataset (two-stream pipeline, every caption once through the word LSTM): the same weights up to fp32 summation order,
    because here a batch of the as-written generator is exactly the captions of whole images."""
    from image_captioning_amd import synth
    from image_captioning_amd.config import Config
    from image_captioning_amd.keras_like import ModelCheckpoint, CSVLogger
    from image_captioning_amd.modified_dense_model import DenseImageCapRCNN
    from image_captioning_amd.text_generation_model_v2 import (DenseCapConfig, build_model, Adam, data_generator, load_sequences,
                                                                train_on_dataset)
    S, V, R, L, n_img, k, blocks = 128, 40, 4, 3, 4, 2, 1

    class FCfg(Config):
        NAME = "toy"
        IMAGES_PER_GPU = 1
        IMAGE_MIN_DIM = S
        IMAGE_MAX_DIM = S
    feats = DenseImageCapRCNN("inference", FCfg(), str(tmp_path / "logs"), stage4_blocks=blocks)
    feats.set_weights(synth.encoder_weights(0, blocks))
    ds = _toy_vg(S, n_img, R, L, V)
    ds.add_sequences(load_sequences(ds))
    assert len(ds.sequences) == n_img * R * L
    cfg = DenseCapConfig(V, synth.embedding_matrix(3, V))
    cfg.PADDING_SIZE = L + 2
    batch, steps, epochs = k * R * L, n_img // k, 2

    def fresh():
        m = build_model((7, 7, 256), (cfg.PADDING_SIZE,), cfg, 256, inject=True, seed=7)
        m.compile(optimizer=Adam(amsgrad=True), loss="categorical_crossentropy")
        return m
    # ---- as written
    ref = fresh()
    ckpt = str(tmp_path / "ck" / "weights-{epoch:02d}.npz")
    hist = ref.fit_generator(data_generator(ds, feats, cfg, batch), epochs=epochs, steps_per_epoch=steps, verbose=0, workers=1, max_queue_size=10,
                             callbacks=[ModelCheckpoint(ckpt, verbose=0, save_weights_only=True, mode='min'), CSVLogger(str(tmp_path / "log.csv"))])
    assert len(hist) == epochs and hist[1]["loss"] < hist[0]["loss"]
    assert open(str(tmp_path / "log.csv")).read().splitlines()[0] == "epoch,loss" and os.path.exists(ckpt.format(epoch=2))
    want = ref.get_weights_dict()
    again = fresh()
    again.load_weights(ckpt.format(epoch=2))
    assert all(np.array_equal(want[n], v) for n, v in again.get_weights_dict().items())
    f0 = feats.generate_captions([ds.load_image(0)], ds.load_captions_and_rois(0)[0][None])[0]["features"]
    ids, probs = again.greedy_decode(f0[0])
    assert len(ids) == cfg.PADDING_SIZE - 1 and np.isfinite(probs).all()
    # ---- (a) device-resident generator: the same samples, features never on the host, sparse targets
    g = data_generator(ds, feats, cfg, batch, device_resident=True)
    (fd, wd), td = next(g)
    (fh, wh), th = next(data_generator(ds, feats, cfg, batch))
    assert isinstance(fd, torch.Tensor) and fd.is_cuda and np.array_equal(fd.cpu().numpy(), fh) and np.array_equal(wd, wh)
    assert td.dtype == np.int32 and np.array_equal(td, th.argmax(1))
    a = fresh()
    a.fit_generator(data_generator(ds, feats, cfg, batch, device_resident=True), epochs=epochs, steps_per_epoch=steps, verbose=0)
    assert all(np.array_equal(want[n], v) for n, v in a.get_weights_dict().items())
    # ---- (b) the measured pipeline behind the same objects
    b = fresh()
    hb = train_on_dataset(b, feats, ds, images_per_step=k, rois_per_image=R, epochs=epochs, steps_per_epoch=steps, verbose=0)
    got = b.get_weights_dict()
    # (b1) against the same single-pass steps run one after the other on one stream: the two-stream pipeline (encoder of step
    # i + 1 beside the decoder of step i, double-buffered features, events) changes NOTHING -- bit-identical weights and losses
    c = fresh()
    hc = []
    for _ in range(epochs):
        losses = []
        for st in range(steps):
            imgs = np.stack([feats.mold_inputs([ds.load_image(i)])[0][0] for i in range(st * k, (st + 1) * k)])
            boxes = np.stack([ds.load_captions_and_rois(i)[0][:R] for i in range(st * k, (st + 1) * k)])
            caps = [[int(np.argmax(w)) for w in cap] for i in range(st * k, (st + 1) * k) for cap in ds.load_captions_and_rois(i)[1][:R]]
            f = feats.extract_features(imgs, boxes)
            losses.append(float(c.train_on_captions(f.reshape(-1, 7, 7, 256), caps).item()))
        hc.append(float(np.mean(np.float32(losses))))
    assert all(np.array_equal(got[n], v) for n, v in c.get_weights_dict().items())
    assert [abs(h["loss"] - l) < 1e-6 * l for h, l in zip(hb, hc)] == [True] * epochs
    # (b2) against the as-written path: the first epoch's loss (one update in) agrees to fp32 summation order.  Later weights
    # are the same training run but not the same bits: the single pass and the expanded batch sum the same gradient in
    # different orders (test_v2_single_pass_equals_expanded_batch: 2e-4), and AMSGrad's first steps are +-lr whatever |g| is.
    assert abs(hb[0]["loss"] - hist[0]["loss"]) < 1e-4 * abs(hist[0]["loss"]), (hb, hist)
    assert abs(hb[1]["loss"] - hist[1]["loss"]) < 0.05 * abs(hist[1]["loss"]) and hb[1]["loss"] < hb[0]["loss"], (hb, hist)
    start = fresh().get_weights_dict()
    for n in want:
        d = np.abs(got[n] - want[n])
        assert d.max() <= 1e-3 * epochs * steps + 1e-6 and d.mean() < 0.25 * np.abs(want[n] - start[n]).mean() + 1e-7, (n, d.mean(), d.max())


def test_joint_train_step_as_a_captured_graph_equals_the_eager_step(gpu):
    """train_on_batch on one GPU: after two eager steps everything behind the encoder pass -- proposals, DetectionTargetLayer on the
    device, RoIAlign, head + decoder forward / backward, RPN / FPN backward, regulariser, clip + AMSGrad -- is ONE captured hipGraph
    whose per-step inputs (RPN selection and counts, GT boxes / captions, lr_t, dropout and sampling stream positions) are device
    words refreshed by one asynchronous copy.  Seven steps on inputs that CHANGE every step (other GT boxes, other selected
    anchors, other counts) with the reference's recurrent_dropout = 0.2: weights and losses bit-equal to the same model stepping
    eagerly (DCAP_STEP_GRAPH=0's path), and the RoI sample differs from step to step."""
    S, V, T, blocks = 128, 24, 5, 1
    _, cfg, Wt = make_joint(S, V, T, blocks)
    del type(cfg).RECURRENT_DROPOUT
    assert cfg.RECURRENT_DROPOUT == 0.2
    from image_captioning_amd.dense_model import DenseImageCapRCNN
    steps = [joint_inputs(S, V, T, seed=8 + (k % 3)) for k in range(7)]
    for k, inp in enumerate(steps):                           # other GT boxes / captions and another number of positive anchors per step
        inp[5] = inp[5].copy()
        inp[5][0, :3] += np.float32(2 * (k % 3))
        inp[2] = inp[2].copy()
        pos = np.nonzero(inp[2][0, :, 0] == 1)[0]
        inp[2][0, pos[:k % 4], 0] = 0
    runs = {}
    for mode in ("graph", "eager"):
        model = DenseImageCapRCNN("training", cfg, "logs", stage4_blocks=blocks)
        model.set_weights(Wt)
        model.compile(1e-4)
        model.use_step_graph = mode == "graph"
        dev_inputs = []
        losses, samples = [], []
        for inp in steps:
            inp = list(inp)
            inp[0] = torch.tensor(inp[0], device="cuda")     # device-resident uint8 image: the step then has no blocking copy at all
            dev_inputs.append(inp)
            losses.append(model.train_on_batch(inp))
            samples.append(model.last_targets["rois"].copy())
        if mode == "graph":
            assert any(k[0] == "train" for k in model._graphs) and model.use_step_graph, "the step was not captured"
            assert model.optimizer.iterations == len(steps) and model.caption_model._drop_step == len(steps)
        runs[mode] = (np.array(losses), model.store.flat.cpu().numpy(), samples)
    np.testing.assert_array_equal(runs["graph"][0], runs["eager"][0])
    np.testing.assert_array_equal(runs["graph"][1], runs["eager"][1])
    for a, b in zip(runs["graph"][2], runs["eager"][2]):
        np.testing.assert_array_equal(a, b)
    assert not np.array_equal(runs["graph"][2][3], runs["graph"][2][6])      # same inputs (seed 8 + 0), another step: another sample


def test_joint_step_graph_is_dropped_and_recaptured_when_what_it_baked_changes(gpu):
    """A captured step bakes buffer addresses, the trainable subset (regulariser mask), the optimizer state and the capacity of the
    packed step inputs.  set_trainable() + compile() between steps, and a step whose RPN selection outgrows the packed buffer, must
    drop the graph, step eagerly and capture again -- the weights stay bit-equal to a model that never uses a graph."""
    S, V, T, blocks = 128, 24, 5, 1
    _, cfg, Wt = make_joint(S, V, T, blocks)
    from image_captioning_amd.dense_model import DenseImageCapRCNN
    base = joint_inputs(S, V, T)
    base[0] = torch.tensor(base[0], device="cuda")
    big = list(base)
    big[2] = base[2].copy()
    rng = np.random.default_rng(3)
    n_anchor = big[2].shape[1]
    pick = rng.choice(n_anchor, 300, replace=False)             # 300 selected anchors > RPN_TRAIN_ANCHORS_PER_IMAGE = 256
    big[2][0, :, 0] = 0
    big[2][0, pick, 0] = np.where(rng.random(300) < 0.15, 1, -1)
    big[3] = rng.standard_normal((1, 64, 4))
    out = {}
    for mode in ("graph", "eager"):
        model = DenseImageCapRCNN("training", cfg, "logs", stage4_blocks=blocks)
        model.set_weights(Wt)
        model.compile(1e-4)
        model.use_step_graph = mode == "graph"
        seen = []
        for _ in range(4):
            model.train_on_batch(base)
        seen.append(any(k[0] == "train" for k in model._graphs))
        model.set_trainable(model.LAYER_REGEX["caption_only"])
        seen.append(any(k[0] == "train" for k in model._graphs))
        model.compile(1e-4)
        for _ in range(4):
            model.train_on_batch(base)
        seen.append(any(k[0] == "train" for k in model._graphs))
        cap0 = model._step_in.cap
        model.train_on_batch(big)                               # the packed inputs grow: new buffer, graph dropped
        seen.append((any(k[0] == "train" for k in model._graphs), model._step_in.cap > cap0))
        for _ in range(3):
            model.train_on_batch(big)
        seen.append(any(k[0] == "train" for k in model._graphs))
        if mode == "graph":
            assert seen == [True, False, True, (False, True), True], seen
        out[mode] = model.store.flat.cpu().numpy()
    np.testing.assert_array_equal(out["graph"], out["eager"])


def test_joint_model_two_images_per_gpu_small(gpu, tmp_path):
    """IMAGES_PER_GPU = 2 at 128 px (the 512-px oracle comparison is tests/test_gpu_oracle_fullsize.py): losses and gradients pooled over
    the batch against M.joint_loss_and_grads_batch with images whose counts differ; the captured optimizer step equals the eager one
    bit for bit; train() runs the batched generator end to end; batched inference returns per image what a one-image model returns."""
    from image_captioning_amd import synth, utils
    from image_captioning_amd.dense_model import DenseImageCapRCNN
    S, V, T, blocks = 128, 24, 5, 1
    _, cfg, Wt = make_joint(S, V, T, blocks)
    cfg.IMAGES_PER_GPU, cfg.BATCH_SIZE = 2, 2
    one = joint_inputs(S, V, T, seed=8)
    two = joint_inputs(S, V, T, seed=9)                      # image 1: the same picture (its proposals meet the GT boxes), but two GT boxes
    two[5][0, 0] = 0                                          # instead of three, other captions, another anchor selection and target rows
    two[4][0, 0] = 0
    two[4][0, 1] = [1, 7, 2, 0, 0][:T]
    two[4][0, 2] = [1, 9, 11, 13, 2][:T]
    inputs = [np.concatenate([a, b]) for a, b in zip(one, two)]
    model = DenseImageCapRCNN("training", cfg, "logs", stage4_blocks=blocks)
    model.set_weights(Wt)
    for rep in range(2):
        losses = model._loss_list(model.forward_backward(inputs, shuffle=None))
    tg = model.last_targets
    assert tg['rois'].shape[0] == 2 and np.all(tg['npos'] > 0), tg['npos']
    oc = dict(mean_pixel=MEAN, scales=cfg.RPN_ANCHOR_SCALES, ratios=cfg.RPN_ANCHOR_RATIOS, strides=cfg.BACKBONE_STRIDES,
              proposal_count=cfg.POST_NMS_ROIS_TRAINING, nms=cfg.RPN_NMS_THRESHOLD, train_rois=cfg.TRAIN_ROIS_PER_IMAGE,
              positive_ratio=cfg.ROI_POSITIVE_RATIO, weight_decay=cfg.WEIGHT_DECAY, T=cfg.PADDING_SIZE)
    want, G, auxes = M.joint_loss_and_grads_batch({k: np.asarray(v, np.float64) for k, v in Wt.items()}, inputs[0], inputs[2][:, :, 0], inputs[3],
                                                  inputs[4], inputs[5], oc, (tg['rois'], tg['caps']), stage4_blocks=blocks)
    for k in ('imgcap_loss', 'rpn_class_loss', 'rpn_bbox_loss', 'reg_loss', 'loss'):
        assert abs(losses[k] - want[k]) < 1e-4 * max(1.0, abs(want[k])), (k, losses[k], want[k])
    got = joint_grads_as_reference(model)
    worst = {k: rel_err(got[k], G[k]) for k in M.joint_trainable(Wt)}
    assert max(worst.values()) < 5e-4, sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    # eager and captured optimizer steps on the batch: bit-equal weights
    flats = {}
    for mode in ("eager", "graph"):
        m = DenseImageCapRCNN("training", cfg, "logs", stage4_blocks=blocks)
        m.set_weights(Wt)
        m.compile(1e-4)
        m.use_step_graph = mode == "graph"
        for _ in range(5):
            m.train_on_batch(inputs)
        if mode == "graph":
            assert any(k[0] == "train" for k in m._graphs)
        flats[mode] = m.store.flat.clone()
    assert torch.equal(flats["eager"], flats["graph"])
    # batched inference == two one-image passes
    cfg.POST_NMS_ROIS_INFERENCE, cfg.DETECTION_MAX_INSTANCES = 40, 10
    inf2 = DenseImageCapRCNN("inference", cfg, "logs", stage4_blocks=blocks)
    inf2.set_weights(Wt)
    res2 = inf2.generate_captions([inputs[0][0], inputs[0][1]], return_probabilities=False)
    _, cfg1, _ = make_joint(S, V, T, blocks)
    cfg1.POST_NMS_ROIS_INFERENCE, cfg1.DETECTION_MAX_INSTANCES = 40, 10
    inf1 = DenseImageCapRCNN("inference", cfg1, "logs", stage4_blocks=blocks)
    inf1.set_weights(Wt)
    assert len(res2) == 2
    for b in range(2):                                        # (a batch of two may pick other conv tiles than a batch of one: last-bit differences)
        r1 = inf1.generate_captions([inputs[0][b]], return_probabilities=False)[0]
        assert abs(len(r1["rois"]) - len(res2[b]["rois"])) <= 1 and len(r1["rois"]) > 0
        same = sum(1 for box, ids in zip(r1["rois"], r1["ids"])
                   if any(np.abs(box - bx).max() <= 1 and np.array_equal(ids, ix) for bx, ix in zip(res2[b]["rois"], res2[b]["ids"])))
        assert same >= 0.8 * len(r1["rois"]), (same, len(r1["rois"]))
    # train(): the generator's batches of two images through fit's loop, one checkpoint
    cfg.STEPS_PER_EPOCH, cfg.MAX_GT_INSTANCES = 2, 6

    class Toy(utils.Dataset):
        def load_image(self, image_id):
            return np.random.RandomState(image_id).randint(0, 255, (S, S, 3)).astype(np.uint8)

        def load_captions_and_rois(self, image_id):
            r = np.random.RandomState(100 + image_id)
            n = 2 + image_id % 3
            y, x = r.randint(0, 60, n), r.randint(0, 60, n)
            boxes = np.stack([y, x, y + r.randint(20, 60, n), x + r.randint(20, 60, n)], axis=1)
            caps = np.zeros((n, T), np.float32)
            caps[:, 0], caps[:, 1:3], caps[:, 3] = 1, r.randint(3, V, (n, 2)), 2
            return boxes, caps
    train, val = Toy(), Toy()
    for ds, ids in ((train, range(4)), (val, range(4, 6))):
        for i in ids:
            ds.add_image("toy", image_id=i, path=None)
        ds.prepare()
    tm = DenseImageCapRCNN("training", cfg, str(tmp_path / "logs"), stage4_blocks=blocks)
    tm.set_weights(Wt)
    np.random.seed(5)                                         # (the generator's shuffle and the RPN-target sampling draw from np.random)
    hist = tm.train(train, val, learning_rate=1e-5, epochs=1, layers="no_backbone")
    assert len(hist) == 1 and all(np.isfinite(v) for v in hist[0].values())
    assert not np.array_equal(tm.get_weights_dict()['fpn_p2/kernel'], np.asarray(Wt['fpn_p2/kernel'], np.float32))
    # train() ran the frozen backbone of batch i + 1 beside the rest of batch i's step (pipeline.JointTrainPipeline, host batches uploaded on
    # their own stream); the serial loop (DCAP_JOINT_PIPELINE=0) gives the same epoch bit for bit: weights, epoch means, validation losses
    assert len(tm._plans) == 2
    ts = DenseImageCapRCNN("training", cfg, str(tmp_path / "logs_serial"), stage4_blocks=blocks)
    ts.set_weights(Wt)
    np.random.seed(5)
    os.environ["DCAP_JOINT_PIPELINE"] = "0"
    try:
        hist_s = ts.train(train, val, learning_rate=1e-5, epochs=1, layers="no_backbone")
    finally:
        del os.environ["DCAP_JOINT_PIPELINE"]
    assert len(ts._plans) == 1
    assert hist_s == hist, (hist_s, hist)
    assert torch.equal(ts.store.flat, tm.store.flat)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_joint_train_pipeline_equals_the_serial_steps(gpu, dtype):
    """pipeline.JointTrainPipeline -- the frozen backbone of batch k + 1 on a second stream beside the rest of batch k's step, two encoder
    plans alternating -- against train_on_batch_device called batch after batch: six different batches, every loss term and every
    weight afterwards equal bit for bit (same kernels, same reduction orders, batch k still updated from batch k's features and the
    weights after update k - 1); the losses arrive one call late.  A model with a trainable ResNet stage is refused."""
    from image_captioning_amd import synth
    from image_captioning_amd.pipeline import JointTrainPipeline
    S, V, T, blocks = 128, 24, 5, 1
    batches = []
    for s in range(6):
        inp = joint_inputs(S, V, T, seed=8 + s)
        if s & 1:                                                   # another image (its GT boxes then match no proposal: no caption loss);
            inp[0] = synth.images(20 + s, 1, S, S)                  # the even batches keep the image the GT boxes were made for
        batches.append(inp)
    runs = []
    for piped in (False, True):
        model, cfg, Wt = make_joint(S, V, T, blocks, compute_dtype=dtype, conv_math="bf16" if dtype == "bf16" else None)
        model.compile(1e-5)
        model.use_step_graph = False
        losses = []
        if piped:
            pipe = JointTrainPipeline(model)
            for b in batches:
                l = pipe.step(b)
                if l is not None:
                    losses.append(l.clone())
            losses.append(pipe.flush().clone())
            assert len(model._plans) == 2 and model._plans[0] is not model._plans[1]
        else:
            for b in batches:
                losses.append(model.train_on_batch_device(b).clone())
        torch.cuda.synchronize()
        runs.append((torch.stack(losses).cpu().numpy(), model.store.flat.clone(), model.optimizer.iterations))
    assert runs[0][2] == runs[1][2] == 6
    assert np.array_equal(runs[0][0], runs[1][0]), (runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1])
    assert len(np.unique(runs[0][0], axis=0)) == 6 and (runs[0][0][:, 2] > 0).any()          # six different batches, positive RoIs (a caption loss) in at least one
    trainable_trunk, _, _ = make_joint(S, V, T, blocks)
    trainable_trunk.set_trainable(trainable_trunk.LAYER_REGEX["5+"])
    with pytest.raises(ValueError):
        JointTrainPipeline(trainable_trunk)
