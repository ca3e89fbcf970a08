"""__graft_entry__.smoke(): ONE small train step of the whole hot path on cuda:0 -- encoder
(ResNet-FPN with the reference's 'resnet50' block count on a 256x256 image) -> PyramidROIAlign ->
RoI head -> v2-inject decoder forward/backward -> AMSGrad -- checked against the NumPy oracle."""
import numpy as np


def run():
    import torch
    from oracle import np_models as M                 # the checker (allowed here, never in the product path)
    from . import synth
    from .config import Config
    from .modified_dense_model import DenseImageCapRCNN
    from .text_generation_model_v2 import DenseCapConfig, build_model, Adam

    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    torch.cuda.set_device(0)
    S, R, V, T, blocks = 256, 8, 1000, 6, 2

    class Cfg(Config):
        IMAGES_PER_GPU = 1
        IMAGE_MIN_DIM = S
        IMAGE_MAX_DIM = S

    encW = synth.encoder_weights(0, blocks)
    img = synth.images(1, 1, S, S)
    rois = synth.rois(2, 1, R, S, S, lo=16, hi=S)
    caps = synth.captions_v2(3, R, T, V, full=False, lmin=2)

    enc = DenseImageCapRCNN("inference", Cfg(), "logs", stage4_blocks=blocks)
    enc.set_weights(encW)
    feat = enc.extract_features(img, rois)[0]                                  # [R,7,7,256] on the GPU
    cfg = DenseCapConfig(V, synth.embedding_matrix(4, V))
    cfg.PADDING_SIZE = T
    dec = build_model((7, 7, 256), (T,), cfg, 256, inject=True, seed=5)
    dec.compile(optimizer=Adam(amsgrad=True), loss="categorical_crossentropy")
    Wt = {k: v.astype(np.float64) for k, v in dec.get_weights_dict().items()}
    loss = float(dec.train_on_captions(feat, caps).item())

    want_feat = M.encoder_features(img, rois, encW, [123.7, 116.8, 103.9], stage4_blocks=blocks)[0]
    err = np.abs(feat.cpu().numpy() - want_feat).max() / np.abs(want_feat).max()
    assert err < 2e-4, "RoI features differ from the oracle: %.3e" % err
    roi_idx, words, tgt = M.v2_expand_samples(caps, T)
    want_loss, G, _ = M.v2_loss_and_grads(Wt, want_feat[roi_idx], words, tgt, True)
    assert abs(loss - want_loss) < 1e-3 * max(1.0, abs(want_loss)), (loss, want_loss)
    for k, g in G.items():
        got = dec.store.grad[k].cpu().numpy()
        scale = np.abs(g).max()
        assert scale < 1e-12 or np.abs(got - g).max() / scale < 1e-3, k
    print("smoke ok: loss %.5f (oracle %.5f), feature err %.2e" % (loss, want_loss, err))
