"""Model 3 of the reference (par-inject: 2 x LSTM-512 + Dense-1024 + Dense-V on top of a TRAINABLE
RoI head), behind its own module interface (dense_img_cap_separate_models/text_generation_model.py:
DenseCapConfig :23-49, VisualGenomeDataset :52-127, word_generation_model :130-156,
build_roi_caption_model_training :159-189, ROICaptionInferenceLayer :192-232, build_lstm_model :235-283,
roi_caption_loss :286-294, create_roi_info :322-329, data_generator :332-371).

The reference expands every caption into T zero-padded prefixes and runs the whole 2-layer LSTM over
each of them (T^2 LSTM steps per RoI).  With Keras' mask carry a post-padded prefix ends in exactly the
state the full caption has after that prefix, so ONE masked pass over the caption yields all T outputs:
predict / train_on_batch here take the reference's batch layout ([feat f32[B,7,7,256], caps f32[B,T]],
one-hot f64[B,T,V]) and return what the T-prefix graph returns (parity: tests/test_gpu_models.py against
the as-written oracle), at 1/T of the LSTM work.  recurrent_dropout=0.2 of the reference is applied in the training
phase (device-side masks; `recurrent_dropout`, `dropout_rows` below); parity runs set it to 0 or replay the masks in the oracle.
"""
import json
import os

import numpy as np
import torch

from . import ops, step_graph, synth
from .config import Config
from .keras_like import KerasLikeModel, ModelCheckpoint, CSVLogger  # noqa: F401
from .params import ParamStore, Adam  # noqa: F401
from .text_generation_model_v2 import pad_sequences
from .utils import Dataset

roi_caption_loss = "roi_caption_loss"     # marker accepted by compile(); the loss is built into the model


class DenseCapConfig(Config):
    NAME = "dense image captioning"
    GPU_COUNT = 1
    IMAGES_PER_GPU = 1
    BATCH_SIZE = 10
    STEPS_PER_EPOCH = 500
    VALIDATION_STEPS = 50
    PADDING_SIZE = 10

    def __init__(self, vocab_size, embedding_weights, batch_size):
        super(DenseCapConfig, self).__init__()
        self.VOCABULARY_SIZE = vocab_size
        self.EMBEDDING_WEIGHTS = embedding_weights
        self.EMBEDDING_SIZE = embedding_weights.shape[1]
        self.BATCH_SIZE = batch_size


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

    def add_rois(self, rois):
        self.rois = rois

    def load_captions_and_rois(self, image_id):
        """rois [N,4]; captions float32 [N,T] = [1(<start>), ids..., 2(<end>), 0-pad], truncated to T."""
        info = self.image_info[image_id]
        T = self.padding_size
        rois, caps = [], []
        for roi, caption in zip(info['rois'], info['captions']):
            cap = self.encode_region_caption(caption[0])
            if cap.size != 0:
                rois.append(roi)
                body = cap if len(cap) < (T - 2) else cap[:(T - 2)]
                caps.append(np.hstack((np.array(1), body, np.array(2))))
        captions = pad_sequences(caps, maxlen=T, padding='post', dtype='float').astype(np.float32)
        return np.array(rois), captions

    def load_original_captions_and_rois(self, image_id):
        info = self.image_info[image_id]
        return np.array(info['rois']), info['captions']

    def encode_region_caption(self, caption):
        from .preprocess import encode_caption
        return encode_caption(caption, self.word_to_id)


def create_roi_info(dataset):
    roi = []
    for image_id in dataset._image_ids:
        _, captions = dataset.load_captions_and_rois(image_id)
        for i in range(captions.shape[0]):
            roi.append((image_id, i, captions[i]))
    return roi


def caption_targets(caps, vocab=None):
    """Target ids of data_generator (:352-357): the caption shifted left by one, last = 0; pads are class 0.
    With vocab given, returns the reference's one-hot float64 [B,T,V]."""
    caps = np.asarray(caps)
    ids = np.concatenate([caps[:, 1:], np.zeros((caps.shape[0], 1), caps.dtype)], axis=1).astype(np.int32)
    return ids if vocab is None else np.eye(vocab)[ids]


def data_generator(dataset, features_model, config, batch_size, shuffle=False):
    """Infinite generator of ([feat f32[B,7,7,256], caps f32[B,T]], onehot f64[B,T,V])."""
    from .generate_one_roi_features import generate_features
    b = 0
    roi_index = -1
    roi_ids = np.arange(len(dataset.rois))
    prev_im_id, prev_img_features = -1, None
    while True:
        roi_index = (roi_index + 1) % len(roi_ids)
        if shuffle and roi_index == 0:
            np.random.shuffle(roi_ids)
        roi_id = roi_ids[roi_index]
        try:
            image_id, img_roi_id, cap = dataset.rois[roi_id]
            if prev_im_id != image_id:
                prev_img_features = generate_features(dataset, image_id, features_model)
            roi_features = prev_img_features[img_roi_id]
            prev_im_id = image_id
            output_words = caption_targets(cap[None], config.VOCABULARY_SIZE)[0]
            if b == 0:
                batch_image_features = np.zeros((batch_size,) + roi_features.shape, dtype=roi_features.dtype)
                batch_input_words = np.zeros((batch_size,) + cap.shape, dtype=cap.dtype)
                batch_output_words = np.zeros((batch_size,) + output_words.shape, dtype=output_words.dtype)
            batch_image_features[b] = roi_features
            batch_input_words[b] = cap
            batch_output_words[b] = output_words
            b += 1
        except Exception:
            raise Exception('An error occurred while processing roi ' + str(roi_id))
        if b >= batch_size:
            yield [batch_image_features, batch_input_words], batch_output_words
            b = 0


def build_lstm_model(features_input, config, units, mode, device=None, seed=0):
    assert mode in ['training', 'inference']
    return CaptionModelV1(features_input, config, units, mode, device, seed)


class _Act(object):
    """An activation (or gradient) matrix with its lazily made bf16 copy (the operand the bf16 GEMMs read)."""
    __slots__ = ("f", "_b", "_model", "_key")

    def __init__(self, model, key, f, b=None):
        self.f, self._b, self._model, self._key = f, b, model, key

    @property
    def b(self):
        if self._b is None:
            self._b = ops.to_bf16(self.f, out=self._model._buf(self._key + ':bf16', tuple(self.f.shape), torch.bfloat16))
        return self._b


class CaptionModelV1(KerasLikeModel):
    FEAT = 1024
    D1 = 1024
    HEAD = (("mrcnn_class_conv1", "mrcnn_class_bn1"), ("mrcnn_class_conv2", "mrcnn_class_bn2"))
    overlap_sync = True        # data parallel: all-reduce a layer group's gradients as soon as its backward is enqueued
    before_sync = None         # optional hook(lo, hi): last touch of a gradient range before its all-reduce starts
    # Keras recurrent_dropout of imgcap_lstm1 / imgcap_lstm2 (:141-142; dense_img_cap/dense_model.py:769-770).  Default = the
    # reference's 0.2: every train step (train_on_batch / fit_generator / train()) draws four inverted-dropout masks [B, units]
    # per LSTM (one per gate i,f,c,o), fixed over the timesteps, from a seeded counter-based generator ON THE DEVICE
    # (dc_dropout_mask_f32; stream = (seed, step, lstm)).  Parity tests and bench.py set 0.0 explicitly (the deterministic graph
    # the oracle restates).  One mask set per RoI serves all T prefixes of its caption: Keras draws one per (RoI, prefix) row of
    # the TimeDistributed batch -- the same marginal distribution per row, but rows of one RoI are correlated here; per-RoI
    # masks are what keeps the T-prefix graph equal to the single masked pass ("recurrent dropout").  Never
    # applied in predict / test_on_batch / generate (Keras' learning phase 0).
    recurrent_dropout = 0.2
    # Which rows get their own mask.  'roi' (default): one mask set per RoI, shared by its T prefixes -- the single masked pass.
    # 'prefix': one per (RoI, prefix) row of the TimeDistributed batch, which is what the reference's graph draws
    # (text_generation_model.py:179-187: TimeDistributed(word_model) over the T prefixes; every application of the LSTM cell
    # makes its own K.dropout mask): the LSTMs then really run over the B*T padded prefixes (T x the LSTM work, as the
    # reference does), everything above them is unchanged.  Only matters when training with recurrent_dropout > 0.
    dropout_rows = "roi"

    def __init__(self, features_input, config, units, mode, device=None, seed=0, extra_params=(), compute_dtype="f32"):
        """extra_params: (name, array, trainable) entries that share this model's flat parameter bucket (the joint
        model's FPN/RPN weights: one optimizer launch and one gradient all-reduce cover everything).
        compute_dtype: 'f32' (exact fp32 MFMA products, the separate-models configs) or 'bf16' (BASELINE configs[4]:
        RoI head, decoder and vocabulary GEMMs on the bf16 matrix pipe from bf16 copies of weights and activations; fp32
        master weights, fp32 accumulation, fp32 LSTM recurrence / gates / BN / softmax statistics)."""
        if compute_dtype not in ("f32", "bf16"):
            raise ValueError("compute_dtype must be 'f32' or 'bf16'")
        self.features_input, self.config, self.units, self.mode = list(features_input), config, units, mode
        self.compute_dtype = compute_dtype
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.V, self.E, self.T = int(config.VOCABULARY_SIZE), int(config.EMBEDDING_SIZE), int(config.PADDING_SIZE)
        if self.E % 4 or self.V % 4:
            raise ValueError("EMBEDDING_SIZE and VOCABULARY_SIZE must be multiples of 4 (16-byte rows); pad the vocabulary")
        pool, cin = self.features_input[0], self.features_input[2]
        W = dict(synth.head_weights(seed + 1, pool, cin, self.FEAT))
        W.update(synth.v1_weights(seed + 2, self.V, self.E, units, self.FEAT))
        W['imgcap_embedding_layer/embeddings'] = np.asarray(config.EMBEDDING_WEIGHTS, np.float32)
        st = ParamStore(self.device)
        own = []
        for k in sorted(W):
            frozen = k.startswith('imgcap_embedding') or 'moving_' in k
            st.add(k, W[k], not frozen)
            if not frozen and k.endswith('kernel') and 'recurrent' not in k:
                own.append(k)
        for name, array, trainable in extra_params:
            st.add(name, array, trainable)
        self.store = st.finalize()
        if compute_dtype == "bf16":
            st.enable_bf16_shadow(own)                    # GEMM kernels only: biases, BN parameters and the recurrences stay fp32
        self.grad_sync = None
        self._bufs = {}
        self._steps = {}                          # captured train steps by batch shape (step_graph.CapturedStep)
        self._drop_seed, self._drop_step = (seed + 77) & 0xFFFFFFFF, 0
        self._drop_offset_dev = None
        self._rec_masks = (None, None)           # device masks of the current train step (lstm1, lstm2) or None

    @property
    def last_rec_masks(self):
        """Host copies of the current train step's masks ([4, B, units] per LSTM; tests feed them to the oracle) or None."""
        if self._rec_masks[0] is None:
            return None
        return [m.cpu().numpy() for m in self._rec_masks]

    def compile(self, optimizer, loss=None):
        self.optimizer, self.loss = optimizer, loss
        self._invalidate_graphs()              # captured train steps hold the previous optimizer's state tensors

    def _prefix_rows(self, training):
        if self.dropout_rows not in ("roi", "prefix"):
            raise ValueError("dropout_rows must be 'roi' or 'prefix'")
        return bool(training) and self.dropout_rows == "prefix" and float(self.recurrent_dropout or 0.0) > 0.0

    def _draw_rec_masks(self, B, training):
        """Keras LSTMCell._generate_recurrent_dropout_mask for both LSTMs: K.dropout(ones, rate) x 4 in the training phase.
        B = rows the LSTMs run over (RoIs, or RoIs x prefixes with dropout_rows='prefix': row j*B_roi + b = prefix j of RoI b)."""
        rate = float(self.recurrent_dropout or 0.0)
        if not training or rate <= 0.0:
            self._rec_masks = (None, None)
            return
        self._drop_step += 1
        if self._drop_offset_dev is not None:
            # the stream position comes from a device word (= 2 * _drop_step, written by the caller before the step): the launch is the
            # same every step, so a captured hipGraph draws fresh masks on every replay
            self._rec_masks = tuple(ops.dropout_mask(self._buf('rec_mask%d' % l, (4, B, self.units)), rate, self._drop_seed, l,
                                                     offset_dev=self._drop_offset_dev) for l in range(2))
            return
        self._rec_masks = tuple(ops.dropout_mask(self._buf('rec_mask%d' % l, (4, B, self.units)), rate, self._drop_seed, 2 * self._drop_step + l)
                                for l in range(2))

    def _buf(self, key, shape, dtype=torch.float32):
        b = self._bufs.get(key)
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != dtype:
            b = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[key] = b
        return b

    def _grads_ready(self, *layers):
        """Data parallel: these layers' gradients are final -- start their all-reduce while the backward goes on."""
        if self.overlap_sync and self.grad_sync is not None and hasattr(self.grad_sync, 'ready'):
            for layer in layers:
                lo, hi = self.store.layer_range(layer)
                if self.before_sync is not None:        # the joint model adds its regulariser's gradient to the range first
                    self.before_sync(lo, hi)
                self.grad_sync.ready(self.store.flat_grad, lo, hi)

    # ---------------------------------------------------------------------------------- engine
    def _act(self, key, f):
        return _Act(self, key, f)

    def _wview(self, name, rows=None):
        """(fp32 view, bf16 view or None) of a 2-D weight (conv kernels flattened to [k*k*cin, cout]), optionally a row range."""
        w = self.store.w[name]
        w = w.view(-1, w.shape[-1])
        wb = self.store.wb.get(name)
        if wb is not None:
            wb = wb.view(-1, wb.shape[-1])
        if rows is not None:
            w = w[rows[0]:rows[1]]
            wb = None if wb is None else wb[rows[0]:rows[1]]
        return w, wb

    def _mm(self, A, W, key=None, out=None, a_trans=False, b_trans=False, want_b=False, **ep):
        """op(A) @ op(W) with the fused epilogue `ep`.  A: _Act; W: (fp32, bf16-or-None) weight views or an _Act.  Runs on
        the bf16 matrix pipe when this model computes in bf16 and the shapes fit dc_gemm_bf16 (16-byte chunks along K),
        else on the fp32 MFMA path.  Returns an _Act of the fp32 result (with its bf16 copy when want_b)."""
        wf, wb = (W.f, None) if isinstance(W, _Act) else W
        Af = A.f
        M = Af.shape[1] if a_trans else Af.shape[0]
        K = Af.shape[0] if a_trans else Af.shape[1]
        N = wf.shape[0] if b_trans else wf.shape[1]
        if out is None:
            out = self._buf(key, (M, N))
        use_b = (self.compute_dtype == "bf16" and K % 8 == 0 and Af.stride(0) % 8 == 0 and wf.stride(0) % 8 == 0 and
                 (not a_trans or M % 8 == 0) and (b_trans or N % 8 == 0) and (isinstance(W, _Act) or wb is not None))
        if use_b:
            if isinstance(W, _Act):
                wb = W.b
            ob = self._buf((key or 'mm') + ':outb', (M, N), torch.bfloat16) if want_b else None
            ops.gemm_bf16(A.b, wb, out=out, out_bf16=ob, a_trans=a_trans, b_trans=b_trans, **ep)
            return _Act(self, key or 'mm', out, ob)
        ops.gemm(Af, wf, out=out, a_trans=a_trans, b_trans=b_trans, **ep)
        return _Act(self, key or 'mm', out)

    def _head_forward(self, X):
        w = self.store.w
        R = X.shape[0]
        x = self._act('X', X)
        self._head_in = []
        for li, (conv, bn) in enumerate(self.HEAD):
            self._head_in.append(x)
            acc = self._mm(x, self._wview(conv + '/kernel'), key='acc%d' % li).f
            y = ops.bn_relu_fwd(acc, w[conv + '/bias'], w[bn + '/gamma'], w[bn + '/beta'], w[bn + '/moving_mean'],
                                w[bn + '/moving_variance'], self._buf('hact%d' % li, (R, self.FEAT)))
            x = self._act('hact%d' % li, y)
        return x

    def _hidden(self, f, ids_tm, mask, B, T, Bl=None):
        """word_generation_model up to the Dense-1024 layer, over time-major token ids: a1 [T*B, 1024] (row t*B+b = the
        state after step t, i.e. for the prefix c_0..c_t).  Bl = rows the LSTMs run over: B (one masked pass per RoI serves
        all its prefixes) or T*B (dropout_rows='prefix': row j*B+b = prefix j of RoI b, zero-padded; Keras' mask carry leaves
        the state after the prefix in the LAST step's rows, which are laid out exactly like the single pass's [T*B] rows)."""
        w, u = self.store.w, self.units
        Bl = B if Bl is None else Bl
        zf = self._mm(f, self._wview('imgcap_lstm1/kernel', (self.E, self.E + self.FEAT)), key='zf').f        # per-RoI half of x.W
        emb_b = self._emb_bf16()
        if emb_b is not None:
            # bf16 model: the embedding half of x.W on the bf16 pipe like the other half.  The table's bf16 copy is zero-padded to a multiple of
            # 8 columns (E = 300 -> 304): the kernel rows 300 .. 303 it then also reads (the feature half's first rows) meet zeros
            z1 = ops.gemm_bf16(emb_b, self.store.wb['imgcap_lstm1/kernel'][:emb_b.shape[1]], gather=ids_tm, shift=w['imgcap_lstm1/bias'],
                               residual=zf, res_rows=B, out=self._buf('z1', (T * Bl, 4 * u)))
        else:
            z1 = ops.gemm(w['imgcap_embedding_layer/embeddings'], w['imgcap_lstm1/kernel'][:self.E], gather=ids_tm, shift=w['imgcap_lstm1/bias'],
                          residual=zf, res_rows=B, out=self._buf('z1', (T * Bl, 4 * u)))
        h1, c1 = ops.lstm_seq_fwd(z1, w['imgcap_lstm1/recurrent_kernel'], mask, Bl, T, self._buf('h1', (T * Bl, u)),
                                  self._buf('c1', (T * Bl, u)), rec_masks=self._rec_masks[0])
        self._h1 = self._act('h1', h1)
        z2 = self._mm(self._h1, self._wview('imgcap_lstm2/kernel'), key='z2', shift=w['imgcap_lstm2/bias']).f
        h2, c2 = ops.lstm_seq_fwd(z2, w['imgcap_lstm2/recurrent_kernel'], mask, Bl, T, self._buf('h2', (T * Bl, u)),
                                  self._buf('c2', (T * Bl, u)), rec_masks=self._rec_masks[1])
        self._h2 = self._act('h2', h2)
        self._h2_out = self._h2 if Bl == B else self._act('h2_out', h2[(T - 1) * Bl:])        # [T*B, u] either way
        zdf = self._mm(f, self._wview('imgcap_lstm_d1/kernel', (u, u + self.FEAT)), key='zdf').f
        return self._mm(self._h2_out, self._wview('imgcap_lstm_d1/kernel', (0, u)), key='a1', shift=w['imgcap_lstm_d1/bias'], residual=zdf,
                        res_rows=B, relu=True, want_b=True)

    def _emb_bf16(self):
        """bf16 copy of the (frozen) embedding table, zero-padded to a multiple of 8 columns -- the bf16 GEMMs' operand granularity -- or None
        when this model computes in fp32.  Made once per table content (the tensor's version counter), outside the timed / captured steps."""
        wb1 = self.store.wb.get('imgcap_lstm1/kernel')
        if self.compute_dtype != "bf16" or wb1 is None:
            return None
        emb = self.store.w['imgcap_embedding_layer/embeddings']
        Ep = (self.E + 7) // 8 * 8
        if Ep > wb1.shape[0]:
            return None
        key = (emb.data_ptr(), emb._version)
        cached = getattr(self, '_emb_b', None)
        if cached is None or cached[0] != key:
            t = torch.zeros((emb.shape[0], Ep), dtype=torch.bfloat16, device=emb.device)
            t[:, :self.E] = emb.to(torch.bfloat16)           # (round to nearest even, as the library's cast)
            self._emb_b = cached = (key, t)
        return cached[1]

    def _word_model(self, f, ids_tm, mask, B, T):
        """Logits [T*B, V] of the whole word model (inference / predict path; training never materialises them)."""
        a1 = self._hidden(f, ids_tm, mask, B, T)
        w = self.store.w
        Vp = (self.V + 3) // 4 * 4
        return ops.gemm(a1.f, w['imgcap_lstm_d2/kernel'], shift=w['imgcap_lstm_d2/bias'], out=self._buf('logits', (T * B, Vp))[:, :self.V])

    def _tables(self, caps, prefix_rows=False):
        """Time-major token ids and Keras masks of the rows the LSTMs run over.  prefix_rows: the T zero-padded prefixes of
        every caption (build_roi_caption_model_training's Lambda, :180-185), row j*B+b = [c_0..c_j, 0...] of RoI b."""
        caps = np.asarray(caps)
        B, T = caps.shape
        ids = caps.astype(np.int32)                        # Embedding casts float ids to int32
        if prefix_rows:
            ids = (ids[None, :, :] * (np.arange(T)[None, None, :] <= np.arange(T)[:, None, None])).reshape(T * B, T)
        up = lambda a, dt: torch.tensor(np.ascontiguousarray(a), dtype=dt, device=self.device)
        return up(ids.T.reshape(-1), torch.int32), up((ids != 0).T.reshape(-1), torch.uint8), B, T

    def _forward_train(self, feat, caps, targets=None, want_probs=False, want_grad=False, row_weights=None, keras_sparse=False,
                       device_tables=None):
        """row_weights [B,T] + keras_sparse: the joint model's masked K.sparse_categorical_crossentropy
        (dense_img_cap/dense_model.py:936-946); loss rows and dlogits are then weighted per row instead of 1/N.
        device_tables = (ids_tm, mask, targets_tm, row_weights_tm, B, T): the index tables already on the device
        (ops.caption_tables from device-resident captions: the joint model's step never builds them on the host); caps / targets /
        row_weights are then ignored."""
        pr = self._prefix_rows(want_grad)
        if device_tables is not None:
            if pr:
                raise NotImplementedError("dropout_rows='prefix' builds its prefix tables on the host: pass captions, not device_tables")
            ids_tm, mask, tg_dev, rw_dev, B, T = device_tables
        else:
            ids_tm, mask, B, T = self._tables(caps, prefix_rows=pr)
        Bl = T * B if pr else B
        self._draw_rec_masks(Bl, training=want_grad)
        X = feat.reshape(B, -1)
        f = self._head_forward(X)
        a1 = self._hidden(f, ids_tm, mask, B, T, Bl)
        w, g = self.store.w, self.store.grad
        N = T * B
        tg = loss_rows = None
        rw = None
        if device_tables is not None:
            tg, rw, loss_rows = tg_dev, rw_dev, self._buf('loss_rows', (N,))
        elif targets is not None:
            tg = torch.tensor(np.ascontiguousarray(np.asarray(targets, np.int32).T.reshape(-1)), device=self.device)
            loss_rows = self._buf('loss_rows', (N,))
        if device_tables is None and row_weights is not None:
            rw = torch.tensor(np.ascontiguousarray(np.asarray(row_weights, np.float32).T.reshape(-1)), device=self.device)
        gscale = 1.0 if rw is not None else 1.0 / N
        Wv, Wvb = self._wview('imgcap_lstm_d2/kernel')
        probs = dl = None
        bf = Wvb is not None and a1._b is not None and N % 8 == 0 and ops.vocab_ce_supported(a1._b, Wvb)
        if not want_probs and tg is not None and (bf or ops.vocab_ce_supported(a1.f, Wv)):
            # Dense(V) + softmax + cross-entropy fused into the GEMM: no [N,V] logits; d(loss)/d(logits) in the compute dtype
            if want_grad:
                if bf:
                    dl = _Act(self, 'dlogits', None, self._buf('dlogits:bf16', (N, (self.V + 7) // 8 * 8), torch.bfloat16)[:, :self.V])
                else:
                    dl = self._act('dlogits', self._buf('dlogits', (N, (self.V + 3) // 4 * 4))[:, :self.V])
            ops.vocab_ce(a1._b if bf else a1.f, Wvb if bf else Wv, w['imgcap_lstm_d2/bias'], tg, loss_rows=loss_rows,
                         dlogits=None if dl is None else (dl._b if bf else dl.f), dbias=g['imgcap_lstm_d2/bias'] if want_grad else None,
                         grad_scale=gscale, row_weights=rw, keras_sparse=keras_sparse)
        else:
            Vp = (self.V + 3) // 4 * 4
            logits = ops.gemm(a1.f, Wv, shift=w['imgcap_lstm_d2/bias'], out=self._buf('logits', (N, Vp))[:, :self.V])
            probs = self._buf('probs', (N, Vp))[:, :self.V] if want_probs else None
            ops.softmax_ce(logits, tg, probs, loss_rows, logits if want_grad else None, grad_scale=gscale, row_weights=rw, keras_sparse=keras_sparse)
            if want_grad:
                dl = self._act('logits', logits)
                ops.colsum(logits, out=g['imgcap_lstm_d2/bias'])
        self._ctx = (X, f, a1, dl, ids_tm, mask, B, T, Bl)
        return loss_rows, probs

    def _backward(self, want_dx=False):
        """Gradients of every trainable weight into the flat bucket; with want_dx also returns the gradient w.r.t.
        the flattened RoI features [B, pool*pool*C] (the joint model backpropagates it through RoIAlign)."""
        w, g, u = self.store.w, self.store.grad, self.units
        X, f, a1, dl, ids_tm, mask, B, T, Bl = self._ctx
        bf = self._bufs
        h1, h2 = self._h1, self._h2_out
        N = T * B                                          # rows of the layers above the LSTMs
        NL = T * Bl                                        # rows of the LSTM sequences (= N unless dropout_rows='prefix')
        if dl.f is None:                                   # bf16 gradient of the fused loss: both products on the bf16 pipe
            ops.gemm_bf16(a1.b, dl.b, a_trans=True, out=g['imgcap_lstm_d2/kernel'])
            da1 = ops.gemm_bf16(dl.b, self.store.wb['imgcap_lstm_d2/kernel'], b_trans=True, out=self._buf('da1', (N, self.D1)))
        else:
            ops.gemm(a1.f, dl.f, a_trans=True, out=g['imgcap_lstm_d2/kernel'])
            da1 = ops.gemm(dl.f, w['imgcap_lstm_d2/kernel'], b_trans=True, out=self._buf('da1', (N, self.D1)))
        self._grads_ready('imgcap_lstm_d2')
        dz_d1 = self._act('dz_d1', ops.relu_bwd(da1, a1.f, da1))
        gWd1 = g['imgcap_lstm_d1/kernel']
        self._mm(h2, dz_d1, a_trans=True, out=gWd1[:u])
        ops.colsum(dz_d1.f, out=g['imgcap_lstm_d1/bias'])
        dzd_f = self._act('dzd_f', ops.fold_time(dz_d1.f, T, B, self._buf('dzd_f', (B, self.D1))))
        self._mm(f, dzd_f, a_trans=True, out=gWd1[u:])
        self._grads_ready('imgcap_lstm_d1')
        df = self._mm(dzd_f, self._wview('imgcap_lstm_d1/kernel', (u, u + self.FEAT)), key='df', b_trans=True).f
        dh2 = self._mm(dz_d1, self._wview('imgcap_lstm_d1/kernel', (0, u)), key='dh2', b_trans=True).f
        # lstm2
        # bf16 model without recurrent-dropout masks: the recurrent kernels' gradients dU = h[0 .. T-1)^T dz[1 .. T) on the bf16 pipe from the
        # bf16 copies, like every other weight gradient of this model (fp32: 62 us each at 3000 x 512 x 2048 on the step's critical chain)
        du_b = (dl.f is None and self.compute_dtype == "bf16" and T > 1 and self._rec_masks[0] is None and self._rec_masks[1] is None
                and ((T - 1) * Bl) % 8 == 0 and u % 8 == 0)

        def dU_bf16(h, dz, name):
            ops.gemm_bf16(h.b[:(T - 1) * Bl], dz.b[Bl:], a_trans=True, out=g[name])
        # (prefix rows: only the last -- carried -- state of each padded prefix feeds the dense layers: Keras' lstm2 without return_sequences)
        dz2, _ = ops.lstm_seq_bwd(bf['z2'], w['imgcap_lstm2/recurrent_kernel'], mask, self._h2.f, bf['c2'], Bl, T,
                                  dh_seq=dh2 if Bl == B else None, dh_last=None if Bl == B else dh2,
                                  dz=self._buf('dz2', (NL, 4 * u)), dU=False if du_b else g['imgcap_lstm2/recurrent_kernel'], rec_masks=self._rec_masks[1])
        dz2 = self._act('dz2', dz2)
        if du_b:
            dU_bf16(self._h2, dz2, 'imgcap_lstm2/recurrent_kernel')
        self._mm(h1, dz2, a_trans=True, out=g['imgcap_lstm2/kernel'])
        ops.colsum(dz2.f, out=g['imgcap_lstm2/bias'])
        self._grads_ready('imgcap_lstm2')
        dh1 = self._mm(dz2, self._wview('imgcap_lstm2/kernel'), key='dh1', b_trans=True).f
        # lstm1
        dz1, _ = ops.lstm_seq_bwd(bf['z1'], w['imgcap_lstm1/recurrent_kernel'], mask, h1.f, bf['c1'], Bl, T, dh_seq=dh1,
                                  dz=self._buf('dz1', (NL, 4 * u)), dU=False if du_b else g['imgcap_lstm1/recurrent_kernel'], rec_masks=self._rec_masks[0])
        emb_b = self._emb_bf16() if dl.f is None else None
        dz1a = self._act('dz1', dz1) if (du_b or emb_b is not None) else None
        if du_b:
            dU_bf16(h1, dz1a, 'imgcap_lstm1/recurrent_kernel')
        gW1 = g['imgcap_lstm1/kernel']
        if emb_b is not None and NL % 8 == 0:
            # (rows E .. Ep of the result are the padded columns' zeros; the feature half's gradient below writes those rows)
            ops.gemm_bf16(emb_b, dz1a.b, a_trans=True, gather=ids_tm, out=gW1[:emb_b.shape[1]])
        else:
            ops.gemm(w['imgcap_embedding_layer/embeddings'], dz1, a_trans=True, gather=ids_tm, out=gW1[:self.E])
        ops.colsum(dz1, out=g['imgcap_lstm1/bias'])
        dzf = self._act('dzf', ops.fold_time(dz1, NL // B, B, self._buf('dzf', (B, 4 * u))))      # rows (t*Bl + j*B + b) -> RoI b
        self._mm(f, dzf, a_trans=True, out=gW1[self.E:])
        self._grads_ready('imgcap_lstm1')
        self._mm(dzf, self._wview('imgcap_lstm1/kernel', (self.E, self.E + self.FEAT)), out=df, b_trans=True, accumulate=True)
        # trainable head (kernels, biases, BN gamma/beta; statistics frozen)
        dy = df
        for li in (1, 0):
            conv, bn = self.HEAD[li]
            dacc = ops.bn_relu_bwd(bf['acc%d' % li], w[conv + '/bias'], w[bn + '/gamma'], w[bn + '/beta'], w[bn + '/moving_mean'],
                                   w[bn + '/moving_variance'], dy, self._buf('dacc%d' % li, (B, self.FEAT)),
                                   g[bn + '/gamma'], g[bn + '/beta'], g[conv + '/bias'])
            dacc = self._act('dacc%d' % li, dacc)
            gk = g[conv + '/kernel']
            self._mm(self._head_in[li], dacc, a_trans=True, out=gk.view(-1, gk.shape[-1]))
            self._grads_ready(conv, bn)
            if li == 1:
                dy = self._mm(dacc, self._wview(conv + '/kernel'), key='dhact0', b_trans=True).f
            elif want_dx:
                return self._mm(dacc, self._wview(conv + '/kernel'), key='dX', b_trans=True).f
        return None

    MAX_STEP_GRAPHS = 4        # batch shapes kept as captured graphs; further shapes run eagerly

    def train_step(self, feat, caps, targets):
        """forward + roi_caption_loss + backward + (all-reduce) + AMSGrad; the loss as a DEVICE scalar (no sync).
        One GPU, one mask set per RoI: the whole step is replayed from a hipGraph captured on the third call with the same batch
        shape (step_graph.py); captions, targets, lr_t and the dropout stream position travel in ONE upload, the features in one copy."""
        if self.optimizer is None:
            raise RuntimeError("compile(optimizer, loss) first")
        world = 1 if self.grad_sync is None else getattr(self.grad_sync, "world", None)
        caps = np.asarray(caps)
        B, T = caps.shape
        # what the captured launches bake: the batch shape and whether (and at which rate) the mask kernels are part of the step
        key = (tuple(feat.shape), B, T, float(self.recurrent_dropout or 0.0), self.optimizer.baked_key())
        steps = self._steps
        cs = steps.get(key)
        if world != 1 or not step_graph.enabled() or self._prefix_rows(True) or (cs is None and len(steps) >= self.MAX_STEP_GRAPHS):
            return self._train_step_eager(feat, caps, targets)
        opt = self.optimizer
        N = T * B
        if cs is None:
            cs = steps[key] = step_graph.CapturedStep()
            cs.feat = torch.empty(tuple(feat.shape), dtype=torch.float32, device=self.device)
            cs.inputs = step_graph.PackedInputs(self.device, [("ids_tm", N), ("targets", N), ("mask", (N + 3) // 4), ("scalars", 4)])
        ids = caps.astype(np.int32)                        # Embedding casts float ids to int32 (_tables)
        scal = np.zeros(4, np.int32)
        scal[0:1] = step_graph.lr_word(opt).view(np.int32)
        scal[1:2] = np.array([(2 * (self._drop_step + 1)) & 0xFFFFFFFF], np.uint32).view(np.int32)      # this step's masks (lstm l: + l)
        cs.inputs.upload({"ids_tm": ids.T, "targets": np.asarray(targets, np.int32).T, "mask": (ids != 0).T.astype(np.uint8), "scalars": scal})
        cs.feat.copy_(self._dev_feat(feat))
        sc = cs.inputs.view("scalars")
        lr_dev, drop_dev = sc[0:1].view(torch.float32), sc[1:2]
        tables = (cs.inputs.view("ids_tm"), cs.inputs.bytes_view("mask", N), cs.inputs.view("targets"), None, B, T)
        dropout = float(self.recurrent_dropout or 0.0) > 0.0

        def body():
            self._drop_offset_dev = drop_dev if dropout else None
            try:
                loss_rows, _ = self._forward_train(cs.feat, None, want_grad=True, device_tables=tables)
            finally:
                self._drop_offset_dev = None              # a later eager step draws from the host counter again
            loss = ops.mean(loss_rows, out=self._buf('loss', (1,)))
            self._backward()
            opt.apply(self.store, grad_scale=1.0, lr_t_dev=lr_dev)
            cs.rec_masks = self._rec_masks
            return loss

        def restore(v):
            opt.iterations, self._drop_step = v

        def bump():                                          # what the captured Python did once: the host-side counters
            opt.iterations += 1
            self._rec_masks = cs.rec_masks                   # (last_rec_masks: this shape's mask buffers, redrawn by the replay)
            if dropout:
                self._drop_step += 1

        own, self._bufs = self._bufs, cs.bufs                # this shape's private scratch buffers (see CapturedStep)
        try:
            return cs.run(body, lambda: (opt.iterations, self._drop_step), restore, bump)
        finally:
            self._bufs = own

    def _train_step_eager(self, feat, caps, targets):
        loss_rows, _ = self._forward_train(feat, caps, targets, want_grad=True)
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
        if y.ndim == 2:
            return y.astype(np.int32)
        if not (np.all(y.max(-1) == 1) and np.all(y.sum(-1) == 1)):
            raise ValueError("targets must be one-hot rows (as the reference's data_generator yields)")
        return y.argmax(-1).astype(np.int32)

    def _unpack(self, probs_tm, B, T):
        return probs_tm.view(T, B, self.V).permute(1, 0, 2).contiguous().cpu().numpy()

    def predict(self, inputs, verbose=0):
        if self.mode == 'inference':
            return self.generate(inputs)[0]
        feat, caps = inputs
        _, probs = self._forward_train(self._dev_feat(feat), caps, want_probs=True)
        B, T = np.asarray(caps).shape
        return self._unpack(probs, B, T)

    def train_on_batch_device(self, inputs, targets):
        """train_on_batch without the host round trip: the loss as a float32 device tensor [1] (see keras_like)."""
        feat, caps = inputs
        return self.train_step(self._dev_feat(feat), caps, self._target_ids(targets))

    def train_on_batch(self, inputs, targets):
        return float(self.train_on_batch_device(inputs, targets).item())

    def test_on_batch_device(self, inputs, targets):
        feat, caps = inputs
        loss_rows, _ = self._forward_train(self._dev_feat(feat), caps, self._target_ids(targets))
        return ops.mean(loss_rows)

    def test_on_batch(self, inputs, targets):
        return float(self.test_on_batch_device(inputs, targets).item())

    def generate(self, feat, return_probabilities=None):
        """ROICaptionInferenceLayer (:192-232): start token 1; step j feeds [prev..., 0...] through the word
        model and appends float(argmax).  Returns (probs [B,T,V], ids [B,T]); with return_probabilities given (the joint
        model) returns (probs or None, ids, word_scores [B,T] = the probability of each chosen word)."""
        feat = self._dev_feat(feat)
        B, T = feat.shape[0], self.T
        self._draw_rec_masks(B, training=False)
        f = self._head_forward(feat.reshape(B, -1))
        prefix = np.zeros((B, T), np.float32)
        prefix[:, 0] = 1
        want_probs = return_probabilities is None or return_probabilities
        rows, ids, best = [], np.zeros((B, T), np.int32), np.zeros((B, T), np.float32)
        rng = torch.arange(B, device=self.device)
        for j in range(T):
            ids_tm, mask, _, _ = self._tables(prefix)
            logits = self._word_model(f, ids_tm, mask, B, T)
            last = logits[(T - 1) * B:]                        # LSTM-2's last (carried) state = state after step j
            probs = self._buf('gprobs', (B, (self.V + 3) // 4 * 4))[:, :self.V]
            ops.softmax_ce(last, None, probs, None, None)
            nxt_d = ops.argmax_rows(probs)
            nxt = nxt_d.cpu().numpy()
            if want_probs:
                rows.append(probs.cpu().numpy())
            if return_probabilities is not None:
                best[:, j] = probs[rng, nxt_d.long()].cpu().numpy()
            ids[:, j] = nxt
            if j + 1 < T:
                prefix[:, j + 1] = nxt
        all_probs = np.stack(rows, axis=1) if want_probs else None
        if return_probabilities is None:
            return all_probs, ids
        return all_probs, ids, best
