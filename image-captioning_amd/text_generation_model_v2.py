"""Models 1 (inject) and 2 (merge) of the reference, behind its own module interface
(dense_img_cap_separate_models/text_generation_model_v2.py: DenseCapConfig :25-50,
VisualGenomeDataset :53-125, load_sequences :128-137, build_model :140-166, data_generator :169-205).

build_model() returns a Keras-like object whose predict / train_on_batch take exactly the reference's
batch layout ([feat f32[B,7,7,256], words int[B,T]], one-hot f64[B,V]) and run it AS WRITTEN (each
sample recomputes the frozen RoI head and the masked word LSTM).  The same engine also has the
algorithmic form the benchmark uses: train_on_captions() teacher-forces every caption ONCE through
the word LSTM and reads the prefix states h_0..h_{L-1} off that single pass -- identical loss and
gradients to the expanded per-prefix batch (tests/test_gpu_models.py) at 1/7 of the FLOPs.
"""
import json
import os

import numpy as np
import torch

from . import ops, step_graph, synth, utils
from .config import Config
from .keras_like import KerasLikeModel, ModelCheckpoint, CSVLogger  # noqa: F401  (re-exported for scripts)
from .layers import V2_WORD_LSTM
from .packing import fold_bn
from .params import ParamStore, Adam  # noqa: F401
from .utils import Dataset


class DenseCapConfig(Config):
    NAME = "dense image captioning"
    GPU_COUNT = 1
    IMAGES_PER_GPU = 1
    BATCH_SIZE = 64
    STEPS_PER_EPOCH = 500
    VALIDATION_STEPS = 50
    PADDING_SIZE = 10

    def __init__(self, vocab_size, embedding_weights):
        super(DenseCapConfig, self).__init__()
        self.VOCABULARY_SIZE = vocab_size
        self.EMBEDDING_WEIGHTS = embedding_weights
        self.EMBEDDING_SIZE = embedding_weights.shape[1]


def pad_sequences(sequences, maxlen, dtype='int32', padding='pre', truncating='pre', value=0):
    """keras.preprocessing.sequence.pad_sequences (defaults as used at _v2.py:183)."""
    out = np.full((len(sequences), maxlen), value, dtype=dtype)
    for i, s in enumerate(sequences):
        s = list(s)
        if not s:
            continue
        s = s[-maxlen:] if truncating == 'pre' else s[:maxlen]
        if padding == 'pre':
            out[i, maxlen - len(s):] = s
        else:
            out[i, :len(s)] = s
    return out


class VisualGenomeDataset(Dataset):
    def __init__(self, words_to_ids, padding_size):
        super(VisualGenomeDataset, self).__init__()
        self.word_to_id = words_to_ids
        self.padding_size = padding_size

    def load_visual_genome(self, data_dir, image_ids, image_meta_file, data_file):
        with open(data_file, 'r', encoding='utf-8') as doc:
            regions = {x['id']: x['regions'] for x in json.load(doc)}
        with open(image_meta_file, 'r', encoding='utf-8') as doc:
            meta = {x['image_id']: x for x in json.load(doc)}
        for i in image_ids:
            self.add_image("VisualGenome", image_id=i, path=os.path.join(data_dir, '{}.jpg'.format(i)),
                           width=meta[i]['width'], height=meta[i]['height'],
                           rois=[[d['y'], d['x'], d['y'] + d['height'], d['x'] + d['width']] for d in regions[i]],
                           captions=[[d['phrase']] for d in regions[i]])

    def add_sequences(self, sequences):
        self.sequences = sequences

    def image_reference(self, image_id):
        return "https://cs.stanford.edu/people/rak248/VG_100K/{}.jpg".format(self.image_info[image_id]["id"])

    def load_captions_and_rois(self, image_id):
        """rois [N,4] (y1,x1,y2,x2) pixels; captions = list of one-hot [L,V] arrays; regions whose
        caption encodes to nothing are dropped."""
        info = self.image_info[image_id]
        rois, caps = [], []
        for roi, caption in zip(info['rois'], info['captions']):
            cap = self.encode_region_caption(caption[0])
            if cap.size != 0:
                rois.append(roi)
                caps.append(cap)
        return np.array(rois), np.array(caps, dtype=object) if len({c.shape for c in caps}) > 1 else np.array(caps)

    def load_caption_ids_and_rois(self, image_id):
        """The regions of load_captions_and_rois with their captions as word-id lists (the argmax of the one-hot rows, without
        building them: [L, V] float64 per caption is 1.2 MB at V = 10 000); train_on_dataset reads this form when a dataset has it."""
        from .preprocess import encode_caption
        info = self.image_info[image_id]
        rois, caps = [], []
        for roi, caption in zip(info['rois'], info['captions']):
            ids = encode_caption(caption[0], self.word_to_id)
            if ids.size != 0:
                rois.append(roi)
                caps.append([int(i) for i in ids])
        return np.array(rois), caps

    def load_original_captions_and_rois(self, image_id):
        info = self.image_info[image_id]
        return np.array(info['rois']), info['captions']

    def encode_region_caption(self, caption):
        from .preprocess import encode_caption_v2
        return encode_caption_v2(caption, self.word_to_id)


def load_sequences(dataset):
    """Every caption of L words -> L samples (image_id, roi, prefix ids, next id); first prefix [0]."""
    sequences = []
    for image_id in dataset._image_ids:
        _, captions = dataset.load_captions_and_rois(image_id)
        for i in range(len(captions)):
            ids = [int(np.argmax(c)) for c in captions[i]]
            sequences.append((image_id, i, [0], ids[0]))
            for j in range(1, len(ids)):
                sequences.append((image_id, i, ids[:j], ids[j]))
    return sequences


def data_generator(dataset, features_model, config, batch_size, shuffle=False, device_resident=False):
    """Infinite generator of ([feat f32[B,7,7,256], words int32[B,T]], onehot f64[B,V]) batches -- the reference's layout
    (text_generation_model_v2.py:169-205).  device_resident=True yields the SAME samples in the form the device wants: the RoI
    features as a torch tensor that never left the GPU (rows gathered from the feature model's output) and the next-word
    targets as int32 ids [B] instead of float64 one-hot rows (B x V x 8 bytes per batch: 5 MB at V = 10 000); train_on_batch /
    fit_generator take both forms."""
    if device_resident:
        for batch in _data_generator_device(dataset, features_model, config, batch_size, shuffle):
            yield batch
    from .generate_one_roi_features import generate_features
    b = 0
    sequence_index = -1
    sequence_ids = np.arange(len(dataset.sequences))
    prev_im_id, prev_img_features = -1, None
    while True:
        sequence_index = (sequence_index + 1) % len(sequence_ids)
        if shuffle and sequence_index == 0:
            np.random.shuffle(sequence_ids)
        sequence_id = sequence_ids[sequence_index]
        try:
            image_id, roi_id, prev_words, next_word = dataset.sequences[sequence_id]
            prev_word_features = pad_sequences([prev_words], config.PADDING_SIZE)[0]
            if prev_im_id != image_id:
                prev_img_features = generate_features(dataset, image_id, features_model)
            roi_features = prev_img_features[roi_id]
            prev_im_id = image_id
            next_word_feature = np.zeros(config.VOCABULARY_SIZE)
            next_word_feature[next_word] = 1
            if b == 0:
                batch_image_features = np.zeros((batch_size,) + roi_features.shape, dtype=roi_features.dtype)
                batch_prev_words = np.zeros((batch_size,) + prev_word_features.shape, dtype=prev_word_features.dtype)
                batch_next_word = np.zeros((batch_size,) + next_word_feature.shape, dtype=next_word_feature.dtype)
            batch_image_features[b] = roi_features
            batch_prev_words[b] = prev_word_features
            batch_next_word[b] = next_word_feature
            b += 1
        except Exception:
            raise Exception('An error occurred while processing sequence ' + str(sequence_id))
        if b >= batch_size:
            yield [batch_image_features, batch_prev_words], batch_next_word
            b = 0


def _data_generator_device(dataset, features_model, config, batch_size, shuffle):
    """data_generator(device_resident=True): same sequence order, same shuffling stream, same batches."""
    from .generate_one_roi_features import generate_features
    sequence_index = -1
    sequence_ids = np.arange(len(dataset.sequences))
    prev_im_id, prev_img_features = -1, None
    rows, words, targets = [], [], []
    while True:
        sequence_index = (sequence_index + 1) % len(sequence_ids)
        if shuffle and sequence_index == 0:
            np.random.shuffle(sequence_ids)
        sequence_id = sequence_ids[sequence_index]
        image_id, roi_id, prev_words, next_word = dataset.sequences[sequence_id]
        if prev_im_id != image_id:
            prev_img_features = generate_features(dataset, image_id, features_model, device_features=True)
        prev_im_id = image_id
        rows.append(prev_img_features[roi_id])
        words.append(pad_sequences([prev_words], config.PADDING_SIZE)[0])
        targets.append(next_word)
        if len(rows) >= batch_size:
            yield [torch.stack(rows), np.stack(words)], np.asarray(targets, np.int32)
            rows, words, targets = [], [], []


def train_on_dataset(model, features_model, dataset, images_per_step, rois_per_image, epochs=1, steps_per_epoch=None, shuffle=False,
                     max_queue_size=4, callbacks=None, verbose=1):
    """The measured pipeline (bench.py) behind the training script's objects: trains `model` (build_model(...), compiled) on
    `dataset` (a VisualGenomeDataset-like utils.Dataset: load_image, load_captions_and_rois) with the feature model's encoder
    plan and the decoder on TWO HIP streams (pipeline.CaptionTrainPipeline): the encoder of step i + 1 runs while the decoder
    of step i trains, RoI features never leave the GPU, every caption goes through the word LSTM once
    (train_on_captions: the same loss and gradients as the reference's (prefix -> next word) samples of those captions).
    One step = `images_per_step` images x their first `rois_per_image` regions (images with fewer regions are skipped);
    images are resized like the feature model does (mold_inputs) on a background thread, `max_queue_size` steps ahead.
    Differences to fit_generator(data_generator(...)): a batch is whole captions of whole images instead of 64 consecutive
    prefix samples, so batch boundaries fall differently unless images_per_step x rois_per_image x caption length equals the
    batch size (tests/test_gpu_models.py::test_train_on_dataset_equals_fit_generator builds exactly that case).
    Returns the per-epoch logs like fit_generator."""
    from .keras_like import GeneratorEnqueuer
    from .pipeline import CaptionTrainPipeline
    cfg = features_model.config
    H = W = cfg.IMAGE_MAX_DIM
    plan = features_model.plan(images_per_step, H, W)
    inner = getattr(model, "inner_model", model)
    pipe = CaptionTrainPipeline(plan, inner, rois_per_image)
    if steps_per_epoch is None:
        steps_per_epoch = max(1, len(dataset.image_ids) // images_per_step)

    dev = inner.device
    s_copy = torch.cuda.Stream(device=dev)          # uploads run on the producer thread's own stream, beside both compute streams

    def batches():
        ids = np.array(dataset.image_ids)
        pos = 0
        stage = np.empty((images_per_step, H, W, 3), np.uint8)      # molded images are written here one by one: no np.stack copy
        while True:
            imgs, boxes, caps = [], [], []
            while len(imgs) < images_per_step:
                if pos == 0 and shuffle:
                    np.random.shuffle(ids)
                image_id = ids[pos]
                pos = (pos + 1) % len(ids)
                if hasattr(dataset, "load_caption_ids_and_rois"):
                    rois, words = dataset.load_caption_ids_and_rois(image_id)
                else:                                            # the reference's form: one-hot rows per word
                    rois, captions = dataset.load_captions_and_rois(image_id)
                    words = [[int(np.argmax(w)) for w in c] for c in captions[:rois_per_image]]
                if len(rois) < rois_per_image:
                    continue
                molded = utils.resize_image(dataset.load_image(image_id), min_dim=cfg.IMAGE_MIN_DIM, max_dim=cfg.IMAGE_MAX_DIM,
                                            padding=cfg.IMAGE_PADDING)[0]       # mold_inputs() without its stack / meta (mean pixel: on the GPU)
                stage[len(imgs)] = molded
                imgs.append(image_id)
                boxes.append(np.asarray(rois[:rois_per_image], np.float32))
                caps += [[int(t) for t in c] for c in words[:rois_per_image]]
            # everything the step needs goes to the GPU here, on the producer thread (blocking copies on its stream: complete when
            # the batch is queued), so the training loop below only enqueues kernels
            with torch.cuda.stream(s_copy):
                images_dev = torch.as_tensor(stage).to(dev)              # blocking copy: `stage` is free again when it returns
                boxes_dev = plan.normalize_boxes(np.stack(boxes))
                tables = SampleTables.from_captions(caps, dev)
                s_copy.synchronize()
            yield images_dev, boxes_dev, tables

    enq = GeneratorEnqueuer(batches(), workers=1, max_queue_size=max_queue_size)
    feed = enq.get()
    history = []
    try:
        for epoch in range(epochs):
            acc, n = None, 0
            for _ in range(steps_per_epoch):
                batch = next(feed)                               # (the pipeline keeps the batch referenced until the GPU is done with it)
                loss = pipe.step(*batch)
                if loss is not None:
                    with torch.cuda.stream(pipe.s_dec):          # the loss buffer belongs to the decoder's stream (and is reused by the next step)
                        acc, n = (loss.clone() if acc is None else acc.add_(loss)), n + 1
            last = pipe.flush()                                  # the epoch's last decoder pass; joins both streams
            acc, n = (last.clone() if acc is None else acc.add_(last)), n + 1
            logs = {"loss": float((acc / n).item())}             # the epoch's one host synchronisation
            if verbose:
                print("Epoch %d/%d - loss: %.4f" % (epoch + 1, epochs, logs["loss"]))
            for cb in callbacks or []:
                cb.on_epoch_end(inner, epoch, logs)
            history.append(logs)
    finally:
        enq.stop()
    return history


# --------------------------------------------------------------------------------------------------
# Sample tables: which RoI feature and which word-LSTM state feed each (prefix -> next word) sample
# --------------------------------------------------------------------------------------------------

class SampleTables(object):
    """Host-built int32 index tables, ONE packed device tensor per batch (one upload; a captured train step copies it into its
    persistent buffer with one device-to-device copy):
    ids_tm [T*Bw] time-major tokens of the Bw word sequences, mask (ids != 0),
    roi_idx [N] row of the RoI feature, hrow_idx [N] row of h_seq (-1: the empty prefix -> zeros),
    inv_hrow [T*Bw] sample fed by each h row (-1: none), targets [N]."""

    @staticmethod
    def layout(T, Bw, N):
        """[(part, words)] of the packed tensor; every part starts 16-byte aligned (step_graph.PackedInputs' rule)."""
        return [("ids_tm", T * Bw), ("roi_idx", N), ("hrow_idx", N), ("inv_hrow", T * Bw), ("targets", N), ("mask", (T * Bw + 3) // 4)]

    def __init__(self, ids_tm, Bw, T, roi_idx, hrow_idx, targets, device):
        ids_tm = np.ascontiguousarray(ids_tm, np.int32)
        self.Bw, self.T, self.N = Bw, T, len(targets)
        inv = np.full(T * Bw, -1, np.int32)
        hr = np.asarray(hrow_idx, np.int32)
        used = hr >= 0
        if len(np.unique(hr[used])) != int(used.sum()):
            raise ValueError("an LSTM state row may feed at most one sample")
        inv[hr[used]] = np.nonzero(used)[0].astype(np.int32)
        parts = {"ids_tm": ids_tm, "roi_idx": np.asarray(roi_idx, np.int32), "hrow_idx": hr, "inv_hrow": inv,
                 "targets": np.asarray(targets, np.int32), "mask": (ids_tm != 0).astype(np.uint8)}
        off, pos = {}, 0
        for k, n in self.layout(T, Bw, self.N):
            off[k] = (pos, n)
            pos += (n + 3) // 4 * 4
        host = np.zeros(max(pos, 4), np.int32)
        for k, a in parts.items():
            o, n = off[k]
            if a.dtype == np.uint8:
                host[o:o + n].view(np.uint8)[:a.size] = a
            else:
                host[o:o + a.size] = a
        self._bind(torch.tensor(host, dtype=torch.int32, device=device), off)

    def _bind(self, packed, off):
        self.packed, self._off = packed, off
        cut = lambda k: packed[off[k][0]:off[k][0] + off[k][1]]
        self.ids_tm, self.roi_idx, self.hrow_idx, self.inv_hrow, self.targets = (cut(k) for k in ("ids_tm", "roi_idx", "hrow_idx", "inv_hrow", "targets"))
        self.mask = cut("mask").view(torch.uint8)[:self.T * self.Bw]

    def like(self, packed):
        """The same tables read from another packed tensor of this layout (a captured step's persistent copy)."""
        tb = object.__new__(SampleTables)
        tb.Bw, tb.T, tb.N = self.Bw, self.T, self.N
        tb._bind(packed, self._off)
        return tb

    @staticmethod
    def from_samples(words, targets, device):
        """The reference's batch: words [B,Tw] (pre-padded prefixes); every sample owns its feature row."""
        words = np.asarray(words).astype(np.int32)
        B, Tw = words.shape
        return SampleTables(words.T.reshape(-1), B, Tw, np.arange(B), (Tw - 1) * B + np.arange(B), targets, device)

    @staticmethod
    def from_captions(captions, device):
        """Single pass: caption r = [w_0..w_{L-1}]; the word LSTM consumes w_0..w_{L-2} (time-major,
        zero-padded => masked); sample (r,j) reads h after j tokens and predicts w_j."""
        R = len(captions)
        Lmax = max(len(c) for c in captions)
        T = max(Lmax - 1, 1)
        ids = np.zeros((T, R), np.int32)
        roi, hrow, tgt = [], [], []
        for r, c in enumerate(captions):
            c = np.asarray(c, np.int32)
            ids[:len(c) - 1, r] = c[:-1]
            for j in range(len(c)):
                roi.append(r)
                hrow.append(-1 if j == 0 else (j - 1) * R + r)
                tgt.append(int(c[j]))
        return SampleTables(ids.reshape(-1), R, T, roi, hrow, tgt, device)


# --------------------------------------------------------------------------------------------------
# The model
# --------------------------------------------------------------------------------------------------

def build_model(features_shape, word_shape, config, units, inject=True, device=None, seed=0):
    return CaptionModelV2(features_shape, word_shape, config, units, inject, device, seed)


class CaptionModelV2(KerasLikeModel):
    WORD_UNITS = 1024
    FEAT = 1024

    def __init__(self, features_shape, word_shape, config, units, inject=True, device=None, seed=0):
        self.features_shape, self.word_shape = tuple(features_shape), tuple(word_shape)
        self.config, self.units, self.inject = config, units, inject
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.V, self.E = int(config.VOCABULARY_SIZE), int(config.EMBEDDING_SIZE)
        pool, cin = self.features_shape[0], self.features_shape[2]
        if self.E % 4 or self.V % 4:
            raise ValueError("EMBEDDING_SIZE and VOCABULARY_SIZE must be multiples of 4 (16-byte rows); pad the vocabulary")
        W = dict(synth.head_weights(seed + 1, pool, cin, self.FEAT))
        W.update(synth.v2_weights(seed + 2, self.V, self.E, self.WORD_UNITS, units, inject, self.FEAT))
        W['imgcap_embedding_layer/embeddings'] = np.asarray(config.EMBEDDING_WEIGHTS, np.float32)
        st = ParamStore(self.device)
        trainable_layers = (V2_WORD_LSTM, 'imgcap_lstm', 'imgcap_d1')
        for k in sorted(W):
            st.add(k, W[k], k.split('/')[0] in trainable_layers)
        self.store = st.finalize()
        self.grad_sync = None            # set by ParallelModel: called between backward and the update
        self._bufs = {}
        self._weights_changed()

    # frozen head: fold BN once (float64 on the host)
    def _weights_changed(self):
        self._invalidate_graphs()                 # (the folded head below lives in new tensors: captured steps hold the old ones)
        self.store.refresh_shadow()
        w = {k: v.detach().cpu().numpy() for k, v in self.store.w.items() if k.startswith('mrcnn_class')}
        dev = self.device
        self._head = []
        for conv, bn in (('mrcnn_class_conv1', 'mrcnn_class_bn1'), ('mrcnn_class_conv2', 'mrcnn_class_bn2')):
            sc, sh = fold_bn(w[bn + '/gamma'], w[bn + '/beta'], w[bn + '/moving_mean'], w[bn + '/moving_variance'],
                             w[conv + '/bias'])
            k = self.store.w[conv + '/kernel']
            self._head.append((k.view(-1, k.shape[-1]), torch.tensor(sc, device=dev), torch.tensor(sh, device=dev)))

    def _buf(self, key, shape, dtype=torch.float32):
        b = self._bufs.get(key)
        if b is None or tuple(b.shape) != tuple(shape):
            b = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[key] = b
        return b

    # ---------------------------------------------------------------------------------- engine
    def _forward(self, feat, tb, want_probs=False, want_grad=False):
        """feat [R,7,7,256] device tensor; tb SampleTables.  Returns (loss_rows or None, probs or None)."""
        w = self.store.w
        R = feat.shape[0]
        N, T, Bw, U = tb.N, tb.T, tb.Bw, self.WORD_UNITS
        X = feat.reshape(R, -1)
        (K1, s1, h1), (K2, s2, h2) = self._head
        a1 = ops.gemm(X, K1, out=self._buf('a1', (R, self.FEAT)), scale=s1, shift=h1, relu=True)
        f = ops.gemm(a1, K2, out=self._buf('f', (R, self.FEAT)), scale=s2, shift=h2, relu=True)
        zx = ops.gemm(w['imgcap_embedding_layer/embeddings'], w[V2_WORD_LSTM + '/kernel'], gather=tb.ids_tm,
                      shift=w[V2_WORD_LSTM + '/bias'], out=self._buf('zx', (T * Bw, 4 * U)))
        h_seq, c_seq = ops.lstm_seq_fwd(zx, w[V2_WORD_LSTM + '/recurrent_kernel'], tb.mask, Bw, T,
                                        self._buf('h_seq', (T * Bw, U)), self._buf('c_seq', (T * Bw, U)))
        cat = self._buf('cat', (N, self.FEAT + U))
        ops.gather_rows(f, tb.roi_idx, cat[:, :self.FEAT])
        ops.gather_rows(h_seq, tb.hrow_idx, cat[:, self.FEAT:])
        if self.inject:
            u = self.units
            z2 = ops.gemm(cat, w['imgcap_lstm/kernel'], shift=w['imgcap_lstm/bias'], out=self._buf('z2', (N, 4 * u)))
            top, c2 = ops.lstm_seq_fwd(z2, w['imgcap_lstm/recurrent_kernel'], None, N, 1,
                                       self._buf('h2', (N, u)), self._buf('c2', (N, u)))
        else:
            top = cat
        Wv, bv = w['imgcap_d1/kernel'], w['imgcap_d1/bias']
        Vp = (self.V + 3) // 4 * 4
        loss_rows = self._buf('loss_rows', (N,)) if tb.targets is not None else None
        probs = dlog = None
        if not want_probs and tb.targets is not None and ops.vocab_ce_supported(top, Wv):
            # training / evaluation: Dense(V) + softmax + categorical cross-entropy fused into the GEMM (dc_vocab_ce); the
            # [N,V] logits are never written, only d(loss)/d(logits) (and its column sums = the bias gradient) when asked for
            if want_grad:
                dlog = self._buf('dlogits', (N, Vp))[:, :self.V]
            ops.vocab_ce(top, Wv, bv, tb.targets, loss_rows=loss_rows, dlogits=dlog, dbias=self.store.grad['imgcap_d1/bias'] if want_grad else None,
                         grad_scale=1.0 / N)
        else:                                     # predict(): the probabilities themselves are the output
            logits = ops.gemm(top, Wv, shift=bv, out=self._buf('dlogits', (N, Vp))[:, :self.V])
            probs = self._buf('probs', (N, Vp))[:, :self.V] if want_probs else None
            dlog = logits if want_grad else None
            ops.softmax_ce(logits, tb.targets, probs, loss_rows, dlog, grad_scale=1.0 / N)
            if want_grad:
                ops.colsum(dlog, out=self.store.grad['imgcap_d1/bias'])
        self._ctx = (tb, zx, h_seq, c_seq, cat, top, dlog)
        return loss_rows, probs

    def _backward(self):
        """Gradients of the trainable weights into the flat gradient bucket (every view is fully
        overwritten, so the bucket needs no zeroing)."""
        w, g = self.store.w, self.store.grad
        tb, zx, h_seq, c_seq, cat, top, dlogits = self._ctx
        N, T, Bw, U = tb.N, tb.T, tb.Bw, self.WORD_UNITS
        ops.gemm(top, dlogits, a_trans=True, out=g['imgcap_d1/kernel'])          # (the bias gradient came with the forward)
        self._grads_ready('imgcap_d1')
        dtop = ops.gemm(dlogits, w['imgcap_d1/kernel'], b_trans=True, out=self._buf('dtop', (N, top.shape[1])))
        if self.inject:
            u = self.units
            z2, h2, c2 = self._bufs['z2'], self._bufs['h2'], self._bufs['c2']
            dz2, _ = ops.lstm_seq_bwd(z2, w['imgcap_lstm/recurrent_kernel'], None, h2, c2, N, 1, dh_last=dtop,
                                      dz=self._buf('dz2', (N, 4 * u)), dU=g['imgcap_lstm/recurrent_kernel'])
            ops.gemm(cat, dz2, a_trans=True, out=g['imgcap_lstm/kernel'])
            ops.colsum(dz2, out=g['imgcap_lstm/bias'])
            self._grads_ready('imgcap_lstm')
            dword = ops.gemm(dz2, w['imgcap_lstm/kernel'][self.FEAT:], b_trans=True, out=self._buf('dword', (N, U)))
        else:
            dword = dtop[:, self.FEAT:]
        dh_seq = ops.gather_rows(dword, tb.inv_hrow, self._buf('dh_seq', (T * Bw, U)))
        dz, _ = ops.lstm_seq_bwd(zx, w[V2_WORD_LSTM + '/recurrent_kernel'], tb.mask, h_seq, c_seq, Bw, T, dh_seq=dh_seq,
                                 dz=self._buf('dz', (T * Bw, 4 * U)), dU=g[V2_WORD_LSTM + '/recurrent_kernel'])
        ops.gemm(w['imgcap_embedding_layer/embeddings'], dz, a_trans=True, gather=tb.ids_tm, out=g[V2_WORD_LSTM + '/kernel'])
        ops.colsum(dz, out=g[V2_WORD_LSTM + '/bias'])

    def _grads_ready(self, layer):
        """Data parallel: this layer's gradients are final -- start their all-reduce while the backward goes on."""
        if self.grad_sync is not None and hasattr(self.grad_sync, 'ready'):
            lo, hi = self.store.layer_range(layer)
            self.grad_sync.ready(self.store.flat_grad, lo, hi)

    MAX_STEP_GRAPHS = 4        # batch shapes kept as captured graphs; further shapes run eagerly
    # Replaying the step from a captured hipGraph is OPT-IN for this model: at the reference's batch (64 samples, 1024 units) every kernel
    # runs 10 us or longer, the eager step is already GPU-bound (0.70 ms) and the replay's three input copies make it 0.73 ms
    # (bench.py other_configs.configs1_gpu reports both).  The v1 decoder at B = 8 is launch-bound and replays by default.
    use_step_graph = False

    def train_step(self, feat, tb):
        """forward + backward + (all-reduce) + AMSGrad; returns the loss as a DEVICE scalar (no sync).
        use_step_graph (one GPU): the whole step is replayed from a hipGraph captured on the third call with the same batch shape
        (step_graph.py); the batch reaches it through two device-to-device copies (features, packed tables) and one word (lr_t)."""
        if self.optimizer is None:
            raise RuntimeError("compile(optimizer, loss) first")
        world = 1 if self.grad_sync is None else getattr(self.grad_sync, "world", None)
        key = (tuple(feat.shape), tb.N, tb.T, tb.Bw, self.optimizer.baked_key())
        steps = self._steps
        cs = steps.get(key)
        if world != 1 or not self.use_step_graph or not step_graph.enabled() or (cs is None and len(steps) >= self.MAX_STEP_GRAPHS):
            return self._train_step_eager(feat, tb)
        opt = self.optimizer
        if cs is None:
            cs = steps[key] = step_graph.CapturedStep()
            cs.feat = torch.empty(tuple(feat.shape), dtype=torch.float32, device=self.device)
            cs.tb = tb.like(torch.empty_like(tb.packed))
            cs.scalars = step_graph.PackedInputs(self.device, [("lr_t", 1)])
        cs.feat.copy_(self._dev_feat(feat))
        cs.tb.packed.copy_(tb.packed)
        cs.scalars.upload({"lr_t": step_graph.lr_word(opt)})
        lr_dev = cs.scalars.view("lr_t", torch.float32)

        def body():
            loss_rows, _ = self._forward(cs.feat, cs.tb, want_grad=True)
            loss = ops.mean(loss_rows, out=self._buf('loss', (1,)))
            self._backward()
            opt.apply(self.store, grad_scale=1.0, lr_t_dev=lr_dev)
            return loss

        def bump():
            opt.iterations += 1

        own, self._bufs = self._bufs, cs.bufs                # this shape's private scratch buffers (see CapturedStep)
        try:
            return cs.run(body, lambda: opt.iterations, lambda v: setattr(opt, "iterations", v), bump)
        finally:
            self._bufs = own

    def _train_step_eager(self, feat, tb):
        loss_rows, _ = self._forward(feat, tb, want_grad=True)
        loss = ops.mean(loss_rows, out=self._buf('loss', (1,)))
        self._backward()
        scale = self.grad_sync(self.store.flat_grad) if self.grad_sync is not None else 1.0
        self.optimizer.apply(self.store, grad_scale=scale)
        return loss

    # ---------------------------------------------------------------------------------- Keras surface
    def _dev_feat(self, feat):
        if isinstance(feat, torch.Tensor):
            return feat.to(self.device, torch.float32).contiguous()
        return torch.tensor(np.ascontiguousarray(feat, np.float32), device=self.device)

    @staticmethod
    def _target_ids(y):
        y = np.asarray(y)
        if y.ndim == 1:
            return y.astype(np.int32)
        ids = y.argmax(-1)
        if not (np.all(y.max(-1) == 1) and np.all(y.sum(-1) == 1)):
            raise ValueError("targets must be one-hot rows (as the reference's data_generator yields)")
        return ids.astype(np.int32)

    def predict(self, inputs, verbose=0):
        feat, words = inputs
        words = np.asarray(words)
        tb = SampleTables.from_samples(words, np.zeros(words.shape[0], np.int32), self.device)
        _, probs = self._forward(self._dev_feat(feat), tb, want_probs=True)
        return probs.cpu().numpy()

    def train_on_batch_device(self, inputs, targets):
        """train_on_batch without the host round trip: the loss as a float32 device tensor [1] (see keras_like)."""
        feat, words = inputs
        tb = SampleTables.from_samples(words, self._target_ids(targets), self.device)
        return self.train_step(self._dev_feat(feat), tb)

    def train_on_batch(self, inputs, targets):
        return float(self.train_on_batch_device(inputs, targets).item())

    def test_on_batch_device(self, inputs, targets):
        feat, words = inputs
        tb = SampleTables.from_samples(words, self._target_ids(targets), self.device)
        loss_rows, _ = self._forward(self._dev_feat(feat), tb)
        return ops.mean(loss_rows)

    def test_on_batch(self, inputs, targets):
        return float(self.test_on_batch_device(inputs, targets).item())

    def train_on_captions(self, feat, captions_or_tables):
        """Algorithmic (single teacher-forced pass) train step over R RoIs and their captions."""
        tb = captions_or_tables if isinstance(captions_or_tables, SampleTables) else \
            SampleTables.from_captions(captions_or_tables, self.device)
        return self.train_step(self._dev_feat(feat), tb)

    def greedy_decode(self, feat_one, steps=None):
        """The reference's test loop (_v2.py:328-346) for one RoI: feed back argmax ids, re-running the
        model on the growing, pre-padded prefix.  Returns (ids [steps], probs [steps,V])."""
        Tw = self.word_shape[0]
        steps = Tw - 1 if steps is None else steps
        feat = self._dev_feat(feat_one)[None]
        ids, rows = [0], []
        for _ in range(steps):
            words = pad_sequences([ids], Tw)
            tb = SampleTables.from_samples(words, np.zeros(1, np.int32), self.device)
            _, probs = self._forward(feat, tb, want_probs=True)
            nxt = int(ops.argmax_rows(probs).item())
            rows.append(probs[0].cpu().numpy())
            ids.append(nxt)
        return np.array(ids[1:], np.int32), np.array(rows)
