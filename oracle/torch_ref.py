"""
oracle/torch_ref.py -- independent torch-CPU restatement of the reference's graphs *as written*
(autograd for gradients).  Two uses:
  * float64: cross-checks oracle/np_models.py (forward and hand-written backward);
  * float32, all host threads: the "cpu_baseline" leg of bench.py (kind "port"), i.e. the
    reference-as-written algorithm (per-prefix samples that each recompute the RoI head and the
    word LSTM; batch-1 ResNet-101+FPN(+RPN convs) per image) timed on the GPU box's host cores.

*** TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED (see np_oracle.py header). ***
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
KERAS_EPS = 1e-7


def to_t(Wt, dtype=torch.float64, requires_grad=()):
    out = {}
    for k, v in Wt.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        if k in requires_grad:
            t.requires_grad_(True)
        out[k] = t
    return out


def hard_sigmoid(z):
    return torch.clamp(0.2 * z + 0.5, 0.0, 1.0)


def lstm(x, mask, W, U, b):
    """Keras LSTM, gates i,f,c,o, hard-sigmoid, mask carry; returns outputs per step [B,T,U]."""
    B, T, _ = x.shape
    Uh = U.shape[0]
    h = x.new_zeros(B, Uh)
    c = x.new_zeros(B, Uh)
    outs = []
    for t in range(T):
        z = x[:, t] @ W + h @ U + b
        i, f, g, o = hard_sigmoid(z[:, :Uh]), hard_sigmoid(z[:, Uh:2 * Uh]), torch.tanh(z[:, 2 * Uh:3 * Uh]), \
            hard_sigmoid(z[:, 3 * Uh:])
        cn = f * c + i * g
        hn = o * torch.tanh(cn)
        if mask is not None:
            m = mask[:, t].unsqueeze(1)
            h = torch.where(m, hn, h)
            c = torch.where(m, cn, c)
        else:
            h, c = hn, cn
        outs.append(h)
    return torch.stack(outs, 1)


def keras_cce(target_ids, probs):
    p = probs / probs.sum(-1, keepdim=True)
    p = torch.clamp(p, KERAS_EPS, 1.0 - KERAS_EPS)
    return -torch.log(p.gather(-1, target_ids.long().unsqueeze(-1)).squeeze(-1))


def bn(x, Wt, name):
    return Wt[name + '/gamma'] * (x - Wt[name + '/moving_mean']) / torch.sqrt(Wt[name + '/moving_variance'] + BN_EPS) \
        + Wt[name + '/beta']


def roi_head(feat, Wt):
    R = feat.shape[0]
    x = feat.reshape(R, -1)
    y = x @ Wt['mrcnn_class_conv1/kernel'].reshape(-1, 1024) + Wt['mrcnn_class_conv1/bias']
    y = torch.relu(bn(y, Wt, 'mrcnn_class_bn1'))
    y = y @ Wt['mrcnn_class_conv2/kernel'].reshape(1024, 1024) + Wt['mrcnn_class_conv2/bias']
    return torch.relu(bn(y, Wt, 'mrcnn_class_bn2'))


def v2_forward(Wt, feat, words, inject=True, word_lstm='lstm_1'):
    f = roi_head(feat, Wt)
    ids = words.long()
    emb = Wt['imgcap_embedding_layer/embeddings'][ids]
    H = lstm(emb, ids != 0, Wt[word_lstm + '/kernel'], Wt[word_lstm + '/recurrent_kernel'], Wt[word_lstm + '/bias'])
    cat = torch.cat([f, H[:, -1]], 1)
    if inject:
        top = lstm(cat.unsqueeze(1), None, Wt['imgcap_lstm/kernel'], Wt['imgcap_lstm/recurrent_kernel'],
                   Wt['imgcap_lstm/bias'])[:, 0]
    else:
        top = cat
    return torch.softmax(top @ Wt['imgcap_d1/kernel'] + Wt['imgcap_d1/bias'], -1)


def v2_loss(Wt, feat, words, targets, inject=True):
    return keras_cce(targets, v2_forward(Wt, feat, words, inject)).mean()


def v1_word_model(Wt, f, prefix):
    ids = prefix.long()
    emb = Wt['imgcap_embedding_layer/embeddings'][ids]
    mask = ids != 0
    T = prefix.shape[1]
    x = torch.cat([emb, f.unsqueeze(1).expand(-1, T, -1)], 2)
    H1 = lstm(x, mask, Wt['imgcap_lstm1/kernel'], Wt['imgcap_lstm1/recurrent_kernel'], Wt['imgcap_lstm1/bias'])
    H2 = lstm(H1, mask, Wt['imgcap_lstm2/kernel'], Wt['imgcap_lstm2/recurrent_kernel'], Wt['imgcap_lstm2/bias'])
    cat = torch.cat([H2[:, -1], f], 1)
    a1 = torch.relu(cat @ Wt['imgcap_lstm_d1/kernel'] + Wt['imgcap_lstm_d1/bias'])
    return torch.softmax(a1 @ Wt['imgcap_lstm_d2/kernel'] + Wt['imgcap_lstm_d2/bias'], -1)


def v1_training_forward(Wt, feat, caps):
    f = roi_head(feat, Wt)
    B, T = caps.shape
    outs = []
    for j in range(1, T + 1):
        prefix = torch.cat([caps[:, :j], caps.new_zeros(B, T - j)], 1)
        outs.append(v1_word_model(Wt, f, prefix))
    return torch.stack(outs, 1)


def v1_loss(Wt, feat, caps):
    probs = v1_training_forward(Wt, feat, caps)
    tg = torch.cat([caps[:, 1:], caps.new_zeros(caps.shape[0], 1)], 1)
    return keras_cce(tg, probs).mean()


# ------------------------------------------------------------------ encoder (NCHW inside torch)

def _same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def conv(x, Wt, name, stride=1, padding='valid'):
    w = Wt[name + '/kernel'].permute(3, 2, 0, 1)        # HWIO -> OIHW
    if padding == 'same':
        k = w.shape[-1]
        pt, pb = _same_pad(x.shape[2], k, stride)
        pl, pr = _same_pad(x.shape[3], k, stride)
        x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w, Wt[name + '/bias'], stride=stride)


def bn4(x, Wt, name):
    sh = (1, -1, 1, 1)
    return Wt[name + '/gamma'].view(sh) * (x - Wt[name + '/moving_mean'].view(sh)) / \
        torch.sqrt(Wt[name + '/moving_variance'].view(sh) + BN_EPS) + Wt[name + '/beta'].view(sh)


def bottleneck(x, Wt, s, blk, stride, shortcut):
    cn, bnn = 'res%d%s_branch' % (s, blk), 'bn%d%s_branch' % (s, blk)
    y = torch.relu(bn4(conv(x, Wt, cn + '2a', stride), Wt, bnn + '2a'))
    y = torch.relu(bn4(conv(y, Wt, cn + '2b', 1, 'same'), Wt, bnn + '2b'))
    y = bn4(conv(y, Wt, cn + '2c'), Wt, bnn + '2c')
    sc = bn4(conv(x, Wt, cn + '1', stride), Wt, bnn + '1') if shortcut else x
    return torch.relu(y + sc)


def resnet_fpn(x_nhwc, Wt, stage4_blocks=22):
    x = x_nhwc.permute(0, 3, 1, 2)
    x = F.pad(x, (3, 3, 3, 3))
    x = torch.relu(bn4(conv(x, Wt, 'conv1', 2), Wt, 'bn_conv1'))
    x = F.pad(x, (0, 1, 0, 1), value=float('-inf'))     # TF SAME for 3x3/s2 on even sizes
    x = F.max_pool2d(x, 3, 2)
    x = bottleneck(x, Wt, 2, 'a', 1, True)
    x = bottleneck(x, Wt, 2, 'b', 1, False)
    C2 = x = bottleneck(x, Wt, 2, 'c', 1, False)
    x = bottleneck(x, Wt, 3, 'a', 2, True)
    for b in 'bcd':
        x = bottleneck(x, Wt, 3, b, 1, False)
    C3 = x
    x = bottleneck(x, Wt, 4, 'a', 2, True)
    for i in range(stage4_blocks):
        x = bottleneck(x, Wt, 4, chr(98 + i), 1, False)
    C4 = x
    x = bottleneck(x, Wt, 5, 'a', 2, True)
    x = bottleneck(x, Wt, 5, 'b', 1, False)
    C5 = x = bottleneck(x, Wt, 5, 'c', 1, False)
    up = lambda t: F.interpolate(t, scale_factor=2, mode='nearest')
    P5 = conv(C5, Wt, 'fpn_c5p5')
    P4 = up(P5) + conv(C4, Wt, 'fpn_c4p4')
    P3 = up(P4) + conv(C3, Wt, 'fpn_c3p3')
    P2 = up(P3) + conv(C2, Wt, 'fpn_c2p2')
    outs = [conv(P2, Wt, 'fpn_p2', 1, 'same'), conv(P3, Wt, 'fpn_p3', 1, 'same'),
            conv(P4, Wt, 'fpn_p4', 1, 'same'), conv(P5, Wt, 'fpn_p5', 1, 'same')]
    return [o.permute(0, 2, 3, 1) for o in outs]


def crop_and_resize(fm, boxes, pool=7):
    """Vectorised tf.image.crop_and_resize for one image; fm [H,W,C], boxes [N,4] float32 normalised."""
    H, W, C = fm.shape
    b = boxes.to(torch.float32)
    y1, x1, y2, x2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    i = torch.arange(pool, dtype=torch.float32)
    hs = (y2 - y1) * float(H - 1) / float(pool - 1)
    ws = (x2 - x1) * float(W - 1) / float(pool - 1)
    in_y = (y1 * float(H - 1)).unsqueeze(1) + i.unsqueeze(0) * hs.unsqueeze(1)      # [N,p]
    in_x = (x1 * float(W - 1)).unsqueeze(1) + i.unsqueeze(0) * ws.unsqueeze(1)
    vy = (in_y >= 0) & (in_y <= H - 1)
    vx = (in_x >= 0) & (in_x <= W - 1)
    iy = in_y.clamp(0, H - 1)
    ix = in_x.clamp(0, W - 1)
    t, bt = iy.floor().long(), iy.ceil().long()
    l, r = ix.floor().long(), ix.ceil().long()
    ly = (iy - t.float()).to(fm.dtype)[:, :, None, None]
    lx = (ix - l.float()).to(fm.dtype)[:, None, :, None]
    g = lambda yy, xx: fm[yy[:, :, None], xx[:, None, :]]                             # [N,p,p,C]
    top = g(t, l) + (g(t, r) - g(t, l)) * lx
    bot = g(bt, l) + (g(bt, r) - g(bt, l)) * lx
    out = top + (bot - top) * ly
    return out * (vy[:, :, None, None] & vx[:, None, :, None]).to(fm.dtype)


def roi_levels(boxes, area):
    b = boxes.to(torch.float32)
    h, w = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    lvl = torch.log(torch.sqrt(h * w) / (224.0 / math.sqrt(area))) / math.log(2.0)
    return torch.clamp(4 + torch.round(lvl).to(torch.int64), 2, 5)


def pyramid_roi_align(boxes, maps, image_hw):
    """boxes [R,4] normalised (one image); maps = [P2..P5] each [1,H,W,C]."""
    lv = roi_levels(boxes, image_hw[0] * image_hw[1])
    out = maps[0].new_zeros(boxes.shape[0], 7, 7, maps[0].shape[-1])
    for i, level in enumerate(range(2, 6)):
        sel = torch.nonzero(lv == level).squeeze(1)
        if sel.numel():
            out[sel] = crop_and_resize(maps[i][0], boxes[sel])
    return out


# ------------------------------------------------------------------ CPU baseline (as written)

def rpn_dead_work(P, Wt_rpn):
    """The reference always evaluates the RPN on P2..P6 even on the GT-RoI path
    (modified_dense_model.py build: rpn_graph over rpn_feature_maps); results are unused."""
    outs = []
    for p in P:
        x = p.permute(0, 3, 1, 2)
        s = torch.relu(F.conv2d(x, Wt_rpn['shared'], padding=1))
        outs.append((F.conv2d(s, Wt_rpn['cls']), F.conv2d(s, Wt_rpn['bbox'])))
    return outs


def cpu_baseline_step(enc_W, dec_W, image_u8, rois_px, captions, mean_pixel, window, vocab,
                      opt_state, stage4_blocks=22, with_rpn=True, lr=1e-3):
    """One reference-as-written train step for ONE image (float32): batch-1 encoder (+dead RPN),
    RoIAlign, then one Keras batch holding every (prefix -> next word) sample of the image's
    captions, each sample recomputing the RoI head and the word LSTM over its padded prefix;
    Keras Adam(amsgrad=True).  Returns the number of captions processed."""
    from .np_models import v2_expand_samples
    with torch.no_grad():
        x = torch.tensor(image_u8[None].astype(np.float32)) - torch.tensor(np.asarray(mean_pixel, np.float32))
        P = resnet_fpn(x, enc_W, stage4_blocks)
        if with_rpn:
            P6 = P[3][:, ::2, ::2, :]
            rpn_dead_work(P + [P6], enc_W['_rpn'])
        H, W = image_u8.shape[:2]
        boxes = torch.tensor(np.asarray(rois_px, np.float32) / np.array([H, W, H, W], np.float32))
        feats = pyramid_roi_align(boxes, P, (H, W))
    roi_idx, words, targets = v2_expand_samples(captions, window)
    feat_b = feats[torch.tensor(roi_idx).long()]
    loss = v2_loss(dec_W, feat_b, torch.tensor(words), torch.tensor(targets), True)
    train = [k for k, v in dec_W.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [dec_W[k] for k in train])
    opt_state['t'] = opt_state.get('t', 0) + 1
    t = opt_state['t']
    lr_t = lr * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
    with torch.no_grad():
        for k, g in zip(train, grads):
            m, v, vh = opt_state.get(k, (torch.zeros_like(g),) * 3)
            m = 0.9 * m + 0.1 * g
            v = 0.999 * v + 0.001 * g * g
            vh = torch.maximum(vh, v)
            dec_W[k] -= lr_t * m / (vh.sqrt() + 1e-7)
            opt_state[k] = (m, v, vh)
    return len(captions), float(loss.detach())


# ------------------------------------------------------------------ joint model (autograd cross-check)

def joint_loss(Wt, image_u8, rpn_match, rpn_bbox_target, rois, caps, mean_pixel, ratios, weight_decay, stage4_blocks=22, term_weights=None):
    """Total loss of the joint model for one image given the (non-differentiable) detection targets `rois`/`caps`:
    imgcap sparse-CE (target > 0) + RPN class + RPN bbox + L2/size regulariser.  float64 tensors in Wt.
    term_weights (dict by loss name): the total is the weighted sum -- one image's share of a pooled batch loss."""
    x = torch.tensor(image_u8[None].astype(np.float64)) - torch.tensor(np.asarray(mean_pixel, np.float64))
    H, W = image_u8.shape[:2]
    xin = x.permute(0, 3, 1, 2)
    xin = F.pad(xin, (3, 3, 3, 3))
    t = torch.relu(bn4(conv(xin, Wt, 'conv1', 2), Wt, 'bn_conv1'))
    t = F.max_pool2d(F.pad(t, (0, 1, 0, 1), value=float('-inf')), 3, 2)
    t = bottleneck(t, Wt, 2, 'a', 1, True); t = bottleneck(t, Wt, 2, 'b', 1, False); C2 = t = bottleneck(t, Wt, 2, 'c', 1, False)
    t = bottleneck(t, Wt, 3, 'a', 2, True)
    for b in 'bcd':
        t = bottleneck(t, Wt, 3, b, 1, False)
    C3 = t
    t = bottleneck(t, Wt, 4, 'a', 2, True)
    for i in range(stage4_blocks):
        t = bottleneck(t, Wt, 4, chr(98 + i), 1, False)
    C4 = t
    t = bottleneck(t, Wt, 5, 'a', 2, True); t = bottleneck(t, Wt, 5, 'b', 1, False); C5 = bottleneck(t, Wt, 5, 'c', 1, False)
    up = lambda q: F.interpolate(q, scale_factor=2, mode='nearest')
    p5 = conv(C5, Wt, 'fpn_c5p5'); p4 = up(p5) + conv(C4, Wt, 'fpn_c4p4'); p3 = up(p4) + conv(C3, Wt, 'fpn_c3p3'); p2 = up(p3) + conv(C2, Wt, 'fpn_c2p2')
    P = [conv(p2, Wt, 'fpn_p2', 1, 'same'), conv(p3, Wt, 'fpn_p3', 1, 'same'), conv(p4, Wt, 'fpn_p4', 1, 'same'), conv(p5, Wt, 'fpn_p5', 1, 'same')]
    P6 = P[3][:, :, ::2, ::2]
    logits, bbox = [], []
    for p in P + [P6]:
        sh = torch.relu(conv(p, Wt, 'rpn_conv_shared', 1, 'same'))
        logits.append(conv(sh, Wt, 'rpn_class_raw').permute(0, 2, 3, 1).reshape(-1, 2))
        bbox.append(conv(sh, Wt, 'rpn_bbox_pred').permute(0, 2, 3, 1).reshape(-1, 4))
    logits, bbox = torch.cat(logits), torch.cat(bbox)
    m = torch.tensor(np.asarray(rpn_match).reshape(-1))
    idx = torch.nonzero(m != 0).squeeze(1)
    l_cls = F.cross_entropy(logits[idx], (m[idx] == 1).long()) if idx.numel() else logits.sum() * 0
    pidx = torch.nonzero(m == 1).squeeze(1)
    if pidx.numel():
        diff = (torch.tensor(np.asarray(rpn_bbox_target, np.float64))[:pidx.numel()] - bbox[pidx]).abs()
        l_box = torch.where(diff < 1.0, 0.5 * diff ** 2, diff - 0.5).mean()
    else:
        l_box = bbox.sum() * 0
    maps = [q.permute(0, 2, 3, 1) for q in P]
    feats = pyramid_roi_align(torch.tensor(np.asarray(rois, np.float32)), maps, (H, W))
    capt = torch.tensor(np.asarray(caps, np.float64))
    probs = v1_training_forward(Wt, feats, capt)
    tg = torch.cat([capt[:, 1:], capt.new_zeros(capt.shape[0], 1)], 1).long()
    q = torch.clamp(probs, KERAS_EPS, 1 - KERAS_EPS)
    rows = -torch.log(q.gather(-1, tg.unsqueeze(-1)).squeeze(-1)) + torch.log(q.sum(-1))
    w = (tg > 0).double()
    l_cap = (rows * w).sum() / w.sum().clamp(min=1.0)
    reg = 0
    for k, v in Wt.items():
        if v.requires_grad and 'gamma' not in k and 'beta' not in k:
            reg = reg + weight_decay * (v ** 2).sum() / v.numel()
    tw = dict(imgcap_loss=1.0, rpn_class_loss=1.0, rpn_bbox_loss=1.0, reg_loss=1.0)
    tw.update(term_weights or {})
    total = tw['imgcap_loss'] * l_cap + tw['rpn_class_loss'] * l_cls + tw['rpn_bbox_loss'] * l_box + tw['reg_loss'] * reg
    return total, dict(imgcap_loss=float(l_cap), rpn_class_loss=float(l_cls), rpn_bbox_loss=float(l_box), reg_loss=float(reg))
