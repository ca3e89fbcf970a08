"""
oracle/np_models.py -- CPU restatement (NumPy float64) of the reference's model graphs *as written*:
ResNet-101 + FPN + PyramidROIAlign encoder, RoI head, caption decoders v2 (inject / merge) and v1
(par-inject, T teacher-forced prefixes), their losses, gradients, Keras AMSGrad and greedy decode.

*** TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED for the model graphs in this file (the Keras/TF arithmetic
cannot run here); the host-side geometry they call is pinned -- see the np_oracle.py header. ***

Weights are a dict keyed '<keras layer name>/<weight name>' with Keras shapes
(kernel HWIO / [in,out], LSTM gate blocks i,f,c,o).
"""
import numpy as np

from . import np_oracle as O

F64 = np.float64


# --------------------------------------------------------------------------------------------
# Encoder: ResNet-101 + FPN (feature_generation/dense_model.py:82-173, :1404-1427;
# dense_img_cap_separate_models/modified_dense_model.py:1410-1433)
# --------------------------------------------------------------------------------------------

def _conv_bn(x, Wt, conv, bn, stride=1, padding='valid', act=True):
    y = O.conv2d_nhwc(x, Wt[conv + '/kernel'], Wt[conv + '/bias'], stride, padding)
    y = O.batchnorm_inference(y, Wt[bn + '/gamma'], Wt[bn + '/beta'],
                              Wt[bn + '/moving_mean'], Wt[bn + '/moving_variance'])
    return O.relu(y) if act else y


def _bottleneck(x, Wt, stage, block, stride, shortcut_conv):
    cn, bn = 'res%d%s_branch' % (stage, block), 'bn%d%s_branch' % (stage, block)
    y = _conv_bn(x, Wt, cn + '2a', bn + '2a', stride=stride)            # stride on the first 1x1
    y = _conv_bn(y, Wt, cn + '2b', bn + '2b', padding='same')
    y = _conv_bn(y, Wt, cn + '2c', bn + '2c', act=False)
    sc = _conv_bn(x, Wt, cn + '1', bn + '1', stride=stride, act=False) if shortcut_conv else x
    return O.relu(y + sc)


def _conv_bn_cached(x, Wt, conv, bn, stride=1, padding='valid'):
    """conv + frozen-statistics BN (no activation) with what the backward needs."""
    a = O.conv2d_nhwc(x, Wt[conv + '/kernel'], Wt[conv + '/bias'], stride, padding)
    rstd = 1.0 / np.sqrt(np.asarray(Wt[bn + '/moving_variance'], F64) + O.BN_EPS)
    xhat = (a - np.asarray(Wt[bn + '/moving_mean'], F64)) * rstd
    y = np.asarray(Wt[bn + '/gamma'], F64) * xhat + np.asarray(Wt[bn + '/beta'], F64)
    return y, dict(x=x, xhat=xhat, rstd=rstd, conv=conv, bn=bn, stride=stride, padding=padding)


def _conv_bn_backward(dy, c, Wt, G):
    """dy = d(loss)/d(BN output).  Adds the gradients of kernel / bias / gamma / beta to G, returns d(loss)/d(conv input)."""
    G[c['bn'] + '/gamma'] = (dy * c['xhat']).sum(axis=(0, 1, 2))
    G[c['bn'] + '/beta'] = dy.sum(axis=(0, 1, 2))
    da = dy * (np.asarray(Wt[c['bn'] + '/gamma'], F64) * c['rstd'])
    dx, dw, db = O.conv2d_nhwc_backward(c['x'], Wt[c['conv'] + '/kernel'], da, c['stride'], c['padding'])
    G[c['conv'] + '/kernel'], G[c['conv'] + '/bias'] = dw, db
    return dx


def _bottleneck_cached(x, Wt, stage, block, stride, shortcut_conv):
    cn, bn = 'res%d%s_branch' % (stage, block), 'bn%d%s_branch' % (stage, block)
    ya, ca = _conv_bn_cached(x, Wt, cn + '2a', bn + '2a', stride)
    m1 = O.relu(ya)
    yb, cb = _conv_bn_cached(m1, Wt, cn + '2b', bn + '2b', 1, 'same')
    m2 = O.relu(yb)
    yc, cc = _conv_bn_cached(m2, Wt, cn + '2c', bn + '2c')
    c1 = None
    if shortcut_conv:
        sc, c1 = _conv_bn_cached(x, Wt, cn + '1', bn + '1', stride)
    else:
        sc = x
    out = O.relu(yc + sc)
    return out, dict(a=ca, b=cb, c=cc, s=c1, m1=m1, m2=m2, out=out)


def _bottleneck_backward(d_out, k, Wt, G):
    ds = d_out * (k['out'] > 0)
    dm2 = _conv_bn_backward(ds, k['c'], Wt, G)
    dm1 = _conv_bn_backward(dm2 * (k['m2'] > 0), k['b'], Wt, G)
    dx = _conv_bn_backward(dm1 * (k['m1'] > 0), k['a'], Wt, G)
    return dx + (ds if k['s'] is None else _conv_bn_backward(ds, k['s'], Wt, G))


def resnet_graph_cached(image, Wt, stage4_blocks=22):
    """resnet_graph with the per-block caches the backward of trainable stages needs (train(layers="3+" ...))."""
    x = np.pad(np.asarray(image, F64), ((0, 0), (3, 3), (3, 3), (0, 0)))
    y1, c_stem = _conv_bn_cached(x, Wt, 'conv1', 'bn_conv1', 2)
    r1 = O.relu(y1)
    x = C1 = O.maxpool3x3s2_same(r1)
    caches = {1: dict(stem=c_stem, r1=r1, pooled=C1), 2: [], 3: [], 4: [], 5: []}
    plan = [(2, 'abc', 1), (3, 'abcd', 2), (4, ['a'] + [chr(98 + i) for i in range(stage4_blocks)], 2), (5, 'abc', 2)]
    Cs = {1: C1}
    for stage, blocks, stride in plan:
        for i, blk in enumerate(blocks):
            x, k = _bottleneck_cached(x, Wt, stage, blk, stride if i == 0 else 1, i == 0)
            caches[stage].append(k)
        Cs[stage] = x
    return Cs, caches


def maxpool3x3s2_same_backward(x, y, dy):
    """Gradient of MaxPooling2D((3,3), strides 2, 'same'): each window's dy goes to its first (row-major) maximum."""
    x, y, dy = np.asarray(x, F64), np.asarray(y, F64), np.asarray(dy, F64)
    N, H, W, C = x.shape
    Ho, Wo = y.shape[1:3]
    pt, pl = O.same_pad(H, 3, 2)[0], O.same_pad(W, 3, 2)[0]
    dx = np.zeros_like(x)
    taken = np.zeros(y.shape, bool)
    for ky in range(3):
        for kx in range(3):
            for oy in range(Ho):
                iy = oy * 2 - pt + ky
                if not 0 <= iy < H:
                    continue
                ox = np.arange(Wo)
                ix = ox * 2 - pl + kx
                ok = (ix >= 0) & (ix < W)
                ox, ix = ox[ok], ix[ok]
                hit = (x[:, iy, ix, :] == y[:, oy, ox, :]) & ~taken[:, oy, ox, :]
                dx[:, iy, ix, :] += np.where(hit, dy[:, oy, ox, :], 0.0)
                taken[:, oy, ox, :] |= hit
    return dx


def resnet_backward(dC, caches, Wt, backbone_from):
    """dC: {stage: gradient w.r.t. that stage's output C_stage} (the FPN laterals' data gradients).  Walks the trainable stages
    top-down; returns the gradients of their kernels, biases, gammas and betas."""
    G = {}
    d = None
    for stage in (5, 4, 3, 2):
        if stage < max(backbone_from, 2):
            break
        d = dC[stage] if d is None else dC[stage] + d
        for k in reversed(caches[stage]):
            d = _bottleneck_backward(d, k, Wt, G)
    if backbone_from == 1:
        st = caches[1]
        d1 = maxpool3x3s2_same_backward(st['r1'], st['pooled'], d)
        _conv_bn_backward(d1 * (st['r1'] > 0), st['stem'], Wt, G)
    return G


def backbone_trainable(Wt, backbone_from, stage4_blocks=22):
    """Trunk weights train(layers="<backbone_from>+") adds to joint_trainable: kernels, biases, BN gammas and betas of the
    ResNet stages >= backbone_from (1 = 'all': the stem too); moving statistics stay frozen."""
    import re
    keys = []
    for k in Wt:
        layer, wname = k.split('/')
        m = re.match(r'(?:res|bn)(\d)[a-z]+_branch', layer)
        stage = 1 if layer in ('conv1', 'bn_conv1') else (int(m.group(1)) if m else None)
        if stage is not None and stage >= backbone_from and wname in ('kernel', 'bias', 'gamma', 'beta'):
            keys.append(k)
    return keys


def resnet_graph(image, Wt, stage4_blocks=22):
    """resnet_graph(input_image, 'resnet101', stage5=True) (dense_model.py:143-173).
    stage4_blocks=22 is ResNet-101 (5 = the file's 'resnet50' option, used for fast tests)."""
    x = np.pad(np.asarray(image, F64), ((0, 0), (3, 3), (3, 3), (0, 0)))   # ZeroPadding2D((3,3))
    x = _conv_bn(x, Wt, 'conv1', 'bn_conv1', stride=2)
    x = C1 = O.maxpool3x3s2_same(x)
    x = _bottleneck(x, Wt, 2, 'a', 1, True)
    x = _bottleneck(x, Wt, 2, 'b', 1, False)
    x = C2 = _bottleneck(x, Wt, 2, 'c', 1, False)
    x = _bottleneck(x, Wt, 3, 'a', 2, True)
    for blk in 'bcd':
        x = _bottleneck(x, Wt, 3, blk, 1, False)
    C3 = x
    x = _bottleneck(x, Wt, 4, 'a', 2, True)
    for i in range(stage4_blocks):
        x = _bottleneck(x, Wt, 4, chr(98 + i), 1, False)
    C4 = x
    x = _bottleneck(x, Wt, 5, 'a', 2, True)
    x = _bottleneck(x, Wt, 5, 'b', 1, False)
    x = C5 = _bottleneck(x, Wt, 5, 'c', 1, False)
    return C1, C2, C3, C4, C5


def fpn_graph(C2, C3, C4, C5, Wt):
    """Top-down pathway (dense_model.py:1406-1423). Returns P2,P3,P4,P5,P6."""
    def conv(x, name, padding='valid'):
        return O.conv2d_nhwc(x, Wt[name + '/kernel'], Wt[name + '/bias'], 1, padding)
    P5 = conv(C5, 'fpn_c5p5')
    P4 = O.upsample2x(P5) + conv(C4, 'fpn_c4p4')
    P3 = O.upsample2x(P4) + conv(C3, 'fpn_c3p3')
    P2 = O.upsample2x(P3) + conv(C2, 'fpn_c2p2')
    P2 = conv(P2, 'fpn_p2', 'same')
    P3 = conv(P3, 'fpn_p3', 'same')
    P4 = conv(P4, 'fpn_p4', 'same')
    P5 = conv(P5, 'fpn_p5', 'same')
    P6 = O.subsample2(P5)
    return P2, P3, P4, P5, P6


def encoder_features(images_u8, rois_px, Wt, mean_pixel, stage4_blocks=22, return_maps=False):
    """generate_captions -> keras_model.predict for the GT-RoI variant
    (modified_dense_model.py:1886-1923, :1522-1527): mold (no resize: the image is already the
    model's size), ResNet+FPN, PyramidROIAlign on rois/[h,w,h,w].  images_u8 [B,H,W,3],
    rois_px [B,R,4] (y1,x1,y2,x2).  Returns [B,R,7,7,256]."""
    x = O.mold_image(images_u8, mean_pixel)
    B, H, W, _ = x.shape
    _, C2, C3, C4, C5 = resnet_graph(x, Wt, stage4_blocks)
    P2, P3, P4, P5, _ = fpn_graph(C2, C3, C4, C5, Wt)
    boxes = O.normalize_boxes(rois_px, H, W)
    feats = O.pyramid_roi_align(boxes, [P2, P3, P4, P5], (H, W, 3), 7)
    if return_maps:
        return feats, (C2, C3, C4, C5, P2, P3, P4, P5)
    return feats


def image_level_features(roi_feats):
    """feature_generation/generate_roi_features.py:60-75: mean over RoIs, flattened 12 544."""
    return np.asarray(roi_feats, F64).mean(axis=0).reshape(-1)


# --------------------------------------------------------------------------------------------
# RoI head  mrcnn_class_conv1/bn1/conv2/bn2 (text_generation_model.py:250-262; _v2.py:141-150)
# --------------------------------------------------------------------------------------------

def roi_head_forward(feat, Wt):
    """[R,7,7,256] -> conv7x7 valid -> BN -> ReLU -> conv1x1 -> BN -> ReLU -> squeeze -> [R,1024]."""
    R = feat.shape[0]
    x = np.asarray(feat, F64).reshape(R, -1)                       # (h,w,c) row-major == HWIO flatten
    K1 = np.asarray(Wt['mrcnn_class_conv1/kernel'], F64).reshape(-1, 1024)
    K2 = np.asarray(Wt['mrcnn_class_conv2/kernel'], F64).reshape(1024, 1024)
    bn = lambda y, n: (np.asarray(Wt[n + '/gamma'], F64), np.asarray(Wt[n + '/beta'], F64),
                       np.asarray(Wt[n + '/moving_mean'], F64),
                       np.sqrt(np.asarray(Wt[n + '/moving_variance'], F64) + O.BN_EPS))
    y1 = x @ K1 + np.asarray(Wt['mrcnn_class_conv1/bias'], F64)
    g1, b1, m1, s1 = bn(y1, 'mrcnn_class_bn1')
    n1 = (y1 - m1) / s1
    a1 = O.relu(g1 * n1 + b1)
    y2 = a1 @ K2 + np.asarray(Wt['mrcnn_class_conv2/bias'], F64)
    g2, b2, m2, s2 = bn(y2, 'mrcnn_class_bn2')
    n2 = (y2 - m2) / s2
    f = O.relu(g2 * n2 + b2)
    cache = dict(x=x, K1=K1, K2=K2, n1=n1, a1=a1, n2=n2, f=f, g1=g1, s1=s1, g2=g2, s2=s2)
    return f, cache


def roi_head_backward(df, cache):
    """Gradients of the trainable head weights (v1 / joint model: kernels, biases, BN gamma/beta)."""
    c = cache
    dz2 = df * (c['f'] > 0)
    G = {'mrcnn_class_bn2/gamma': (dz2 * c['n2']).sum(0), 'mrcnn_class_bn2/beta': dz2.sum(0)}
    dy2 = dz2 * c['g2'] / c['s2']
    G['mrcnn_class_conv2/kernel'] = (c['a1'].T @ dy2).reshape(1, 1, 1024, 1024)
    G['mrcnn_class_conv2/bias'] = dy2.sum(0)
    dz1 = (dy2 @ c['K2'].T) * (c['a1'] > 0)
    G['mrcnn_class_bn1/gamma'] = (dz1 * c['n1']).sum(0)
    G['mrcnn_class_bn1/beta'] = dz1.sum(0)
    dy1 = dz1 * c['g1'] / c['s1']
    G['mrcnn_class_conv1/kernel'] = (c['x'].T @ dy1).reshape(7, 7, 256, 1024)
    G['mrcnn_class_conv1/bias'] = dy1.sum(0)
    G['_dx'] = dy1 @ c['K1'].T                        # gradient w.r.t. the flattened RoI features [R,12544]
    return G


# --------------------------------------------------------------------------------------------
# v2 decoder (Model 1 inject / Model 2 merge): text_generation_model_v2.py:140-166
# --------------------------------------------------------------------------------------------

def pad_sequences_pre(seqs, maxlen):
    """keras pad_sequences defaults: padding='pre', truncating='pre', value 0, int32 (_v2.py:183)."""
    out = np.zeros((len(seqs), maxlen), np.int32)
    for i, s in enumerate(seqs):
        s = list(s)[-maxlen:]
        if s:
            out[i, maxlen - len(s):] = s
    return out


def v2_expand_samples(captions, window):
    """load_sequences + data_generator (_v2.py:128-137, :169-205): a caption of L word ids yields L
    samples; sample j has prefix ids[:j] (the first one [0]) pre-padded/pre-truncated to `window`
    and target ids[j].  Returns roi_index [N], words [N,window] int32, targets [N] int32."""
    roi, seqs, tgt = [], [], []
    for r, cap in enumerate(captions):
        cap = [int(c) for c in cap]
        for j in range(len(cap)):
            roi.append(r)
            seqs.append(cap[:j] if j > 0 else [0])
            tgt.append(cap[j])
    return np.array(roi, np.int32), pad_sequences_pre(seqs, window), np.array(tgt, np.int32)


V2_WORD_LSTM = 'lstm_1'   # the unnamed KL.LSTM(1024) gets Keras' auto name (_v2.py:157)


def v2_forward(Wt, feat, words, inject=True):
    """build_model(...).predict([feat, words]) -> probs [B,V]; feat [B,7,7,256], words [B,Tw]."""
    f, hc = roi_head_forward(feat, Wt)
    emb, mask = O.embedding(words, Wt['imgcap_embedding_layer/embeddings'])
    H, lc = O.lstm_forward(emb, mask, Wt[V2_WORD_LSTM + '/kernel'], Wt[V2_WORD_LSTM + '/recurrent_kernel'],
                           Wt[V2_WORD_LSTM + '/bias'])
    word = H[:, -1]                      # return_sequences=False: last output (carried through masks)
    cat = np.concatenate([f, word], axis=1)          # Concatenate()([features, word]) -> [B,2048]
    cache = dict(head=hc, lstm=lc, cat=cat)
    if inject:
        H2, ic = O.lstm_forward(cat[:, None, :], None, Wt['imgcap_lstm/kernel'],
                                Wt['imgcap_lstm/recurrent_kernel'], Wt['imgcap_lstm/bias'])
        top = H2[:, 0]
        cache['inj'] = ic
    else:
        top = cat
    logits = top @ np.asarray(Wt['imgcap_d1/kernel'], F64) + np.asarray(Wt['imgcap_d1/bias'], F64)
    probs = O.softmax(logits)
    cache.update(top=top, logits=logits, probs=probs)
    return probs, cache


def v2_loss_and_grads(Wt, feat, words, targets, inject=True):
    """model.train_on_batch loss (mean over the batch of K.categorical_crossentropy, _v2.py:266-267)
    and gradients of the trainable weights (word LSTM, imgcap_lstm, imgcap_d1; head and
    embedding are trainable=False, _v2.py:142-156)."""
    probs, c = v2_forward(Wt, feat, words, inject)
    B = probs.shape[0]
    loss = O.categorical_crossentropy(targets, probs).mean()
    dlog = O.softmax_ce_grad_logits(targets, probs, np.full(B, 1.0 / B))
    G = {'imgcap_d1/kernel': c['top'].T @ dlog, 'imgcap_d1/bias': dlog.sum(0)}
    dtop = dlog @ np.asarray(Wt['imgcap_d1/kernel'], F64).T
    if inject:
        dcat3, dW, dU, db = O.lstm_backward(None, c['inj'], dh_last=dtop)
        G['imgcap_lstm/kernel'], G['imgcap_lstm/recurrent_kernel'], G['imgcap_lstm/bias'] = dW, dU, db
        dcat = dcat3[:, 0]
    else:
        dcat = dtop
    dword = dcat[:, 1024:]
    _, dW, dU, db = O.lstm_backward(None, c['lstm'], dh_last=dword)
    G[V2_WORD_LSTM + '/kernel'], G[V2_WORD_LSTM + '/recurrent_kernel'], G[V2_WORD_LSTM + '/bias'] = dW, dU, db
    return loss, G, probs


def v2_greedy_decode(Wt, feat_one, window, steps):
    """_v2.py:328-346 per RoI: prev=[zeros]; repeat PADDING_SIZE-1 times: predict on
    pad_sequences([[argmax(p) for p in prev]]), append the probability row.  Returns the token ids
    argmax'ed from every appended row (ids fed back) and the probability rows."""
    ids, rows = [0], []
    for _ in range(steps):
        words = pad_sequences_pre([ids], window)
        p, _ = v2_forward(Wt, feat_one[None], words)
        rows.append(p[0])
        ids.append(int(np.argmax(p[0])))
    return np.array(ids[1:], np.int32), np.array(rows)


# --------------------------------------------------------------------------------------------
# v1 decoder (Model 3, par-inject): text_generation_model.py:130-294
# --------------------------------------------------------------------------------------------

def v1_word_model_forward(Wt, f, prefix, rec_masks=None):
    """word_generation_model (text_generation_model.py:130-156): f [B,1024], prefix [B,T] float
    token ids (0 = pad) -> probs [B,V].  rec_masks = (masks of imgcap_lstm1, masks of imgcap_lstm2), each [4,B,512]: the
    training phase of recurrent_dropout=0.2 (:141-142) with those masks; None = dropout off."""
    emb, mask = O.embedding(prefix, Wt['imgcap_embedding_layer/embeddings'])
    T = prefix.shape[1]
    x = np.concatenate([emb, np.repeat(f[:, None, :], T, axis=1)], axis=2)      # [emb(300) | f(1024)]
    m1, m2 = (None, None) if rec_masks is None else rec_masks
    H1, c1 = O.lstm_forward(x, mask, Wt['imgcap_lstm1/kernel'], Wt['imgcap_lstm1/recurrent_kernel'],
                            Wt['imgcap_lstm1/bias'], rec_masks=m1)
    H2, c2 = O.lstm_forward(H1, mask, Wt['imgcap_lstm2/kernel'], Wt['imgcap_lstm2/recurrent_kernel'],
                            Wt['imgcap_lstm2/bias'], rec_masks=m2)
    cat = np.concatenate([H2[:, -1], f], axis=1)                                 # [lstm2(512) | f(1024)]
    z1 = cat @ np.asarray(Wt['imgcap_lstm_d1/kernel'], F64) + np.asarray(Wt['imgcap_lstm_d1/bias'], F64)
    a1 = O.relu(z1)
    logits = a1 @ np.asarray(Wt['imgcap_lstm_d2/kernel'], F64) + np.asarray(Wt['imgcap_lstm_d2/bias'], F64)
    probs = O.softmax(logits)
    return probs, dict(c1=c1, c2=c2, cat=cat, a1=a1, probs=probs, T=T)


def v1_word_model_backward(Wt, dlog, c):
    """Returns (df [B,1024], grads dict) for one word_model call."""
    G = {'imgcap_lstm_d2/kernel': c['a1'].T @ dlog, 'imgcap_lstm_d2/bias': dlog.sum(0)}
    dz1 = (dlog @ np.asarray(Wt['imgcap_lstm_d2/kernel'], F64).T) * (c['a1'] > 0)
    G['imgcap_lstm_d1/kernel'] = c['cat'].T @ dz1
    G['imgcap_lstm_d1/bias'] = dz1.sum(0)
    dcat = dz1 @ np.asarray(Wt['imgcap_lstm_d1/kernel'], F64).T
    dh2, df = dcat[:, :512].copy(), dcat[:, 512:].copy()
    dH1, dW, dU, db = O.lstm_backward(None, c['c2'], dh_last=dh2)
    G['imgcap_lstm2/kernel'], G['imgcap_lstm2/recurrent_kernel'], G['imgcap_lstm2/bias'] = dW, dU, db
    dx, dW, dU, db = O.lstm_backward(dH1, c['c1'])
    G['imgcap_lstm1/kernel'], G['imgcap_lstm1/recurrent_kernel'], G['imgcap_lstm1/bias'] = dW, dU, db
    df += dx[:, :, 300:].sum(1)
    return df, G


def v1_prefixes(caps):
    """build_roi_caption_model_training's Lambda (text_generation_model.py:180-185):
    row j (1..T) = [c_0..c_{j-1}, 0, ...]."""
    caps = np.asarray(caps, F64)
    B, T = caps.shape
    out = np.zeros((B, T, T))
    for j in range(1, T + 1):
        out[:, j - 1, :j] = caps[:, :j]
    return out


def v1_targets(caps):
    """data_generator target ids (text_generation_model.py:352-357): caption shifted left by one,
    last = 0; pads become class 0 (and *do* count in the loss)."""
    caps = np.asarray(caps)
    return np.concatenate([caps[:, 1:], np.zeros((caps.shape[0], 1))], axis=1).astype(np.int32)


def v1_training_forward(Wt, feat, caps, rec_masks=None):
    """build_lstm_model(..., 'training').predict([feat, caps]) -> [B,T,V].  rec_masks: see v1_word_model_forward.  Masks of
    shape [4,B,512] are applied to every prefix of a RoI (the product's single-pass form); masks of shape [T,4,B,512] give prefix
    j its own set rec_masks[.][j] -- what Keras draws: TimeDistributed(word_model) (text_generation_model.py:187) applies the
    LSTM cells once per prefix and each application makes its own K.dropout masks."""
    f, hc = roi_head_forward(feat, Wt)
    P = v1_prefixes(caps)
    T = P.shape[1]
    outs, caches = [], []
    for j in range(T):
        rm = rec_masks if rec_masks is None or np.ndim(rec_masks[0]) == 3 else (rec_masks[0][j], rec_masks[1][j])
        p, c = v1_word_model_forward(Wt, f, P[:, j], rm)
        outs.append(p)
        caches.append(c)
    return np.stack(outs, axis=1), dict(head=hc, f=f, caches=caches)


def v1_loss_and_grads(Wt, feat, caps, rec_masks=None):
    """roi_caption_loss (text_generation_model.py:286-294): every row has sum(y_true)=1>0, so the
    loss is the mean CE over all B*T rows.  Gradients of all trainable weights (decoder + head)."""
    probs, c = v1_training_forward(Wt, feat, caps, rec_masks)
    tg = v1_targets(caps)
    B, T, V = probs.shape
    loss = O.categorical_crossentropy(tg, probs).mean()
    G, df = {}, np.zeros_like(c['f'])
    for j in range(T):
        dlog = O.softmax_ce_grad_logits(tg[:, j], probs[:, j], np.full(B, 1.0 / (B * T)))
        dfj, Gj = v1_word_model_backward(Wt, dlog, c['caches'][j])
        df += dfj
        for k, v in Gj.items():
            G[k] = G.get(k, 0.0) + v
    G.update(roi_head_backward(df, c['head']))
    G.pop('_dx')
    return loss, G, probs


def v1_backward_from_dlogits(Wt, cache, dlog):
    """Backward of v1_training_forward for an arbitrary gradient w.r.t. the softmax LOGITS [B,T,V] (the joint
    model's masked sparse-CE).  Returns (grads dict incl. '_dx' = d RoI features [B,12544])."""
    G, df = {}, np.zeros_like(cache['f'])
    for j in range(dlog.shape[1]):
        dfj, Gj = v1_word_model_backward(Wt, dlog[:, j], cache['caches'][j])
        df += dfj
        for k, v in Gj.items():
            G[k] = G.get(k, 0.0) + v
    G.update(roi_head_backward(df, cache['head']))
    return G


def v1_greedy_decode(Wt, feat, T):
    """ROICaptionInferenceLayer (text_generation_model.py:192-232): start token 1; step j feeds
    [prev..., 0...] and appends float(argmax).  T = config.PADDING_SIZE.
    Returns probs [B,T,V] and ids [B,T]."""
    f, _ = roi_head_forward(feat, Wt)
    B = f.shape[0]
    prev = np.ones((B, 1))
    rows = []
    for j in range(T):
        ctx = np.concatenate([prev, np.zeros((B, T - j - 1))], axis=1)
        p, _ = v1_word_model_forward(Wt, f, ctx)
        rows.append(p)
        prev = np.concatenate([prev, O.argmax_rows(p)[:, None].astype(F64)], axis=1)
    return np.stack(rows, axis=1), prev[:, 1:].astype(np.int32)


# --------------------------------------------------------------------------------------------
# Optimizer over a weight dict, data-parallel averaging (parallel_model.py:58-102)
# --------------------------------------------------------------------------------------------

class AMSGrad:
    """keras.optimizers.Adam(lr, amsgrad=True[, clipnorm]) over a dict of arrays."""

    def __init__(self, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, clipnorm=None):
        self.lr, self.b1, self.b2, self.eps, self.clipnorm = lr, beta_1, beta_2, epsilon, clipnorm
        self.t, self.state = 0, {}

    def step(self, Wt, G):
        self.t += 1
        keys = sorted(G)
        grads = [np.asarray(G[k], F64) for k in keys]
        if self.clipnorm is not None:
            grads, _ = O.clip_by_global_norm(grads, self.clipnorm)
        for k, g in zip(keys, grads):
            m, v, vh = self.state.get(k, (0.0, 0.0, 0.0))
            p, m, v, vh = O.amsgrad_step(Wt[k], g, m, v, vh, self.t, self.lr, self.b1, self.b2, self.eps)
            Wt[k] = p
            self.state[k] = (m, v, vh)


def data_parallel_mean(per_rank):
    """ParallelModel semantics (parallel_model.py:93-101): mean over towers of the per-tower mean
    loss => gradients are the plain average of the per-rank gradients."""
    keys = per_rank[0].keys()
    return {k: sum(np.asarray(g[k], F64) for g in per_rank) / len(per_rank) for k in keys}


# --------------------------------------------------------------------------------------------
# RPN + proposals: the image-level variant named by north_star
# (feature_generation/dense_model.py:684-725 rpn_graph, :1425-1455, :221-305; generate_roi_features.py:60-75)
# --------------------------------------------------------------------------------------------

def rpn_forward(P, Wt):
    """rpn_graph on one feature map [B,H,W,256] -> (probs [B,H*W*A,2], bbox [B,H*W*A,4])."""
    shared = O.relu(O.conv2d_nhwc(P, Wt['rpn_conv_shared/kernel'], Wt['rpn_conv_shared/bias'], 1, 'same'))
    cls = O.conv2d_nhwc(shared, Wt['rpn_class_raw/kernel'], Wt['rpn_class_raw/bias'])
    box = O.conv2d_nhwc(shared, Wt['rpn_bbox_pred/kernel'], Wt['rpn_bbox_pred/bias'])
    B = P.shape[0]
    return O.softmax(cls.reshape(B, -1, 2)), box.reshape(B, -1, 4)


def rpn_proposals(maps5, Wt, config_like, image_hw):
    """maps5 = [P2..P6]; returns (proposals [B,count,4] normalised float32, scores [B,A], deltas [B,A,4])."""
    outs = [rpn_forward(p, Wt) for p in maps5]
    probs = np.concatenate([o[0] for o in outs], axis=1)
    bbox = np.concatenate([o[1] for o in outs], axis=1)
    H, W = image_hw
    shapes = [[-(-H // s), -(-W // s)] for s in config_like['strides']]
    anchors = O.generate_pyramid_anchors(config_like['scales'], config_like['ratios'], shapes, config_like['strides'], 1)
    props = [O.proposal_layer(probs[b, :, 1], bbox[b], anchors, image_hw, config_like['count'], config_like['nms'])[0]
             for b in range(probs.shape[0])]
    return np.stack(props), probs[:, :, 1], bbox, anchors


def image_level_encoder_features(images_u8, Wt, mean_pixel, config_like, stage4_blocks=22):
    """feature_generation variant: RoIs from the RPN; returns ([B,count,7,7,256], proposals)."""
    x = O.mold_image(images_u8, mean_pixel)
    B, H, W, _ = x.shape
    _, C2, C3, C4, C5 = resnet_graph(x, Wt, stage4_blocks)
    P2, P3, P4, P5, P6 = fpn_graph(C2, C3, C4, C5, Wt)
    props, scores, bbox, anchors = rpn_proposals([P2, P3, P4, P5, P6], Wt, config_like, (H, W))
    feats = O.pyramid_roi_align(props, [P2, P3, P4, P5], (H, W, 3), 7)
    return feats, props, (scores, bbox, anchors)


# --------------------------------------------------------------------------------------------
# Joint model (configs[4]): dense_img_cap/dense_model.py build('training') :1429-1600, losses :877-946,
# compile :1694-1730, train(layers="no_backbone") :1829-1831
# --------------------------------------------------------------------------------------------

JOINT_TRAINABLE_PREFIXES = ('imgcap_', 'rpn_', 'fpn_', 'mrcnn_')


def joint_trainable(Wt):
    """layers="no_backbone": imgcap_*, rpn_*, fpn_*, mrcnn_*; the embedding layer is trainable=False and BN moving
    statistics are not variables of the optimizer."""
    return [k for k in Wt if k.startswith(JOINT_TRAINABLE_PREFIXES) and not k.startswith('imgcap_embedding') and 'moving_' not in k]


def joint_loss_and_grads(Wt, image_u8, rpn_match, rpn_bbox_target, gt_captions, gt_boxes_px, cfg, shuffle=None, stage4_blocks=22,
                         targets_override=None, backbone_from=None, term_weights=None, trunk_cache=None):
    """One training step's losses and gradients for ONE image (IMAGES_PER_GPU = 1, train_dense_captions.py:27).
    term_weights = dict(imgcap_loss=, rpn_class_loss=, rpn_bbox_loss=, reg_loss=): every loss term AND its gradient is scaled by its
    weight -- how joint_loss_and_grads_batch pools a batch (below).  trunk_cache: a dict that keeps the frozen ResNet's C2..C5 of this
    image between calls (they depend on the image and the frozen weights only; a second device model checked against the same image
    does not pay for the float64 trunk again).
    cfg: dict(mean_pixel, scales, ratios, strides, proposal_count, nms, train_rois, positive_ratio, weight_decay, T).
    targets_override = (rois [R,4] normalised, caps [R,T]) replaces the DetectionTargetLayer's sample (which carries no
    gradient): parity tests hand in the sample the device drew, since near-tied proposal scores may order differently.
    Returns (losses dict, grads dict over joint_trainable(Wt), aux dict)."""
    x = O.mold_image(image_u8[None], cfg['mean_pixel'])
    _, H, W, _ = x.shape
    tw = dict(imgcap_loss=1.0, rpn_class_loss=1.0, rpn_bbox_loss=1.0, reg_loss=1.0)
    tw.update(term_weights or {})
    if backbone_from is None:
        if trunk_cache is not None and 'Cs' in trunk_cache:
            Cs = trunk_cache['Cs']
        else:
            _, C2, C3, C4, C5 = resnet_graph(x, Wt, stage4_blocks)
            Cs = {2: C2, 3: C3, 4: C4, 5: C5}
            if trunk_cache is not None:
                trunk_cache['Cs'] = Cs
    else:                                                           # train(layers="3+" | "4+" | "5+" | "all"): ResNet stages train too
        Cs, trunk_caches = resnet_graph_cached(x, Wt, stage4_blocks)
    conv = lambda t, n, pad='valid': O.conv2d_nhwc(t, Wt[n + '/kernel'], Wt[n + '/bias'], 1, pad)
    C5 = Cs[5]
    pre = {5: conv(C5, 'fpn_c5p5')}
    for k in (4, 3, 2):
        pre[k] = O.upsample2x(pre[k + 1]) + conv(Cs[k], 'fpn_c%dp%d' % (k, k))
    P = {k: conv(pre[k], 'fpn_p%d' % k, 'same') for k in (2, 3, 4, 5)}
    P[6] = O.subsample2(P[5])
    # RPN
    shared, cls, box = {}, {}, {}
    for k in (2, 3, 4, 5, 6):
        shared[k] = O.relu(conv(P[k], 'rpn_conv_shared', 'same'))
        cls[k] = conv(shared[k], 'rpn_class_raw')
        box[k] = conv(shared[k], 'rpn_bbox_pred')
    logits = np.concatenate([cls[k].reshape(1, -1, 2) for k in (2, 3, 4, 5, 6)], axis=1)[0]
    bbox = np.concatenate([box[k].reshape(1, -1, 4) for k in (2, 3, 4, 5, 6)], axis=1)[0]
    probs = O.softmax(logits)
    shapes = [[-(-H // s_), -(-W // s_)] for s_ in cfg['strides']]
    anchors = O.generate_pyramid_anchors(cfg['scales'], cfg['ratios'], shapes, cfg['strides'], 1)
    proposals, _, _ = O.proposal_layer(probs[:, 1], bbox, anchors, (H, W), cfg['proposal_count'], cfg['nms'])
    gt_norm = (np.asarray(gt_boxes_px, np.float32) / np.array([H, W, H, W], np.float32)).astype(np.float32)
    rois, caps, npos, nneg = O.detection_targets(proposals, gt_captions, gt_norm, cfg['train_rois'], cfg['positive_ratio'], shuffle)
    if targets_override is not None:
        rois, caps = np.asarray(targets_override[0], np.float32), np.asarray(targets_override[1])
    maps = [P[2], P[3], P[4], P[5]]
    feats = O.pyramid_roi_align(rois[None], maps, (H, W, 3), 7)[0]
    cap_probs, cache = v1_training_forward(Wt, feats, caps.astype(np.float64))
    tg = v1_targets(caps)                                             # remove <start>, append 0
    w = (tg > 0).astype(F64)
    cnt = w.sum()
    rows_loss, dlog = O.sparse_cce_keras_with_grad(tg, cap_probs, w / max(cnt, 1.0))
    losses = {'imgcap_loss': tw['imgcap_loss'] * (float(rows_loss.sum()) if cnt > 0 else 0.0)}
    dlog = dlog * tw['imgcap_loss']
    l_cls, dlogits = O.rpn_class_loss(rpn_match, logits)
    l_box, dbbox = O.rpn_bbox_loss(rpn_bbox_target, rpn_match, bbox)
    losses['rpn_class_loss'], losses['rpn_bbox_loss'] = tw['rpn_class_loss'] * l_cls, tw['rpn_bbox_loss'] * l_box
    dlogits, dbbox = dlogits * tw['rpn_class_loss'], dbbox * tw['rpn_bbox_loss']
    train = joint_trainable(Wt) + (backbone_trainable(Wt, backbone_from, stage4_blocks) if backbone_from is not None else [])
    reg_keys = [k for k in train if 'gamma' not in k and 'beta' not in k]
    losses['reg_loss'] = tw['reg_loss'] * float(sum(cfg['weight_decay'] * (np.asarray(Wt[k], F64) ** 2).sum() / np.asarray(Wt[k]).size for k in reg_keys))
    losses['loss'] = sum(losses[k] for k in ('imgcap_loss', 'rpn_class_loss', 'rpn_bbox_loss', 'reg_loss'))

    # ---- backward
    G = v1_backward_from_dlogits(Wt, cache, dlog if cnt > 0 else np.zeros_like(dlog))
    dfeat = G.pop('_dx').reshape(feats.shape)
    dP = O.pyramid_roi_align_backward(rois[None], [m.shape for m in maps], (H, W, 3), dfeat[None])
    dP = {k: dP[i] for i, k in enumerate((2, 3, 4, 5))}
    dP[6] = np.zeros_like(P[6])
    off = 0
    acc = lambda name, val: G.__setitem__(name, G.get(name, 0.0) + val)
    for k in (2, 3, 4, 5, 6):
        n = cls[k].shape[1] * cls[k].shape[2] * len(cfg['ratios'])
        dcls = dlogits[off:off + n].reshape(cls[k].shape)
        dbox = dbbox[off:off + n].reshape(box[k].shape)
        off += n
        ds1, dw, db = O.conv2d_nhwc_backward(shared[k], Wt['rpn_class_raw/kernel'], dcls)
        acc('rpn_class_raw/kernel', dw); acc('rpn_class_raw/bias', db)
        ds2, dw, db = O.conv2d_nhwc_backward(shared[k], Wt['rpn_bbox_pred/kernel'], dbox)
        acc('rpn_bbox_pred/kernel', dw); acc('rpn_bbox_pred/bias', db)
        dsh = (ds1 + ds2) * (shared[k] > 0)
        dpk, dw, db = O.conv2d_nhwc_backward(P[k], Wt['rpn_conv_shared/kernel'], dsh, 1, 'same')
        acc('rpn_conv_shared/kernel', dw); acc('rpn_conv_shared/bias', db)
        dP[k] = dP[k] + dpk
    dP[5][:, ::2, ::2, :] += dP[6]                                    # MaxPooling2D(1, strides 2) backward
    dpre = {}
    for k in (2, 3, 4, 5):
        dpre[k], dw, db = O.conv2d_nhwc_backward(pre[k], Wt['fpn_p%d/kernel' % k], dP[k], 1, 'same')
        G['fpn_p%d/kernel' % k], G['fpn_p%d/bias' % k] = dw, db
    for k in (2, 3, 4):                                               # top-down: pre[k] = up(pre[k+1]) + lateral
        g = dpre[k]
        N_, h_, w_, c_ = g.shape
        dpre[k + 1] = dpre[k + 1] + g.reshape(N_, h_ // 2, 2, w_ // 2, 2, c_).sum(axis=(2, 4))
    dC = {}
    for k in (2, 3, 4, 5):
        dC[k], dw, db = O.conv2d_nhwc_backward(Cs[k], Wt['fpn_c%dp%d/kernel' % (k, k)], dpre[k])
        G['fpn_c%dp%d/kernel' % (k, k)], G['fpn_c%dp%d/bias' % (k, k)] = dw, db
    if backbone_from is not None:
        G.update(resnet_backward(dC, trunk_caches, Wt, backbone_from))
    for k in reg_keys:
        G[k] = G[k] + tw['reg_loss'] * 2.0 * cfg['weight_decay'] * np.asarray(Wt[k], F64) / np.asarray(Wt[k]).size
    aux = dict(proposals=proposals, rois=rois, caps=caps, npos=npos, nneg=nneg, count=cnt, cap_probs=cap_probs, anchors=anchors)
    return losses, {k: G[k] for k in train}, aux


def joint_loss_and_grads_batch(Wt, images_u8, rpn_match, rpn_bbox_target, gt_captions, gt_boxes_px, cfg, targets_override, stage4_blocks=22,
                               trunk_caches=None):
    """The reference's BATCHED training graph (IMAGES_PER_GPU = B, dense_img_cap/config.py:35): DetectionTargetLayer slices the batch
    (utils.batch_slice, dense_model.py:531-572), every loss gathers over ALL images before its mean --
      rpn_class_loss   mean over the non-neutral anchors of the batch               (:877-900)
      rpn_bbox_loss    mean over the positive anchors of the batch (x 4 coordinates) (:903-933, batch_pack_graph)
      imgcap_loss      mean over the caption positions with target > 0 of the batch  (:936-946)
    -- so the batch loss is NOT the mean of the per-image losses when the images' counts differ: image b's per-image mean enters
    with the weight count_b / sum(count).  The weight regulariser is counted once.  Arguments carry the image axis first;
    targets_override = (rois [B,R,4], caps [B,R,T]) as in joint_loss_and_grads.  Returns (losses, grads, [aux per image])."""
    B = len(images_u8)
    rpn_match = [np.asarray(m).reshape(-1) for m in rpn_match]
    n_sel = np.array([(m != 0).sum() for m in rpn_match], F64)
    n_pos = np.array([(m == 1).sum() for m in rpn_match], F64)
    caps = [np.asarray(targets_override[1][b]) for b in range(B)]
    n_tok = np.array([(v1_targets(c) > 0).sum() for c in caps], F64)
    share = lambda n: n / max(n.sum(), 1.0)
    w_cls, w_box, w_cap = share(n_sel), share(n_pos), share(n_tok)
    losses, grads, auxes = {}, {}, []
    for b in range(B):
        tw = dict(imgcap_loss=w_cap[b], rpn_class_loss=w_cls[b], rpn_bbox_loss=w_box[b], reg_loss=1.0 if b == 0 else 0.0)
        l, g, aux = joint_loss_and_grads(Wt, images_u8[b], rpn_match[b], rpn_bbox_target[b], gt_captions[b], gt_boxes_px[b], cfg,
                                         stage4_blocks=stage4_blocks, targets_override=(targets_override[0][b], caps[b]), term_weights=tw,
                                         trunk_cache=None if trunk_caches is None else trunk_caches[b])
        for k, v in l.items():
            losses[k] = losses.get(k, 0.0) + v
        for k, v in g.items():
            grads[k] = grads.get(k, 0.0) + v
        auxes.append(aux)
    return losses, grads, auxes
