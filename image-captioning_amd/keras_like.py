"""The slice of the Keras Model / callbacks surface the reference's training scripts touch
(text_generation_model_v2.py:262-313, text_generation_model.py:424-472): compile, predict,
train_on_batch, test_on_batch, fit_generator, load_weights(by_name, skip_mismatch), save_weights,
trainable_weights, summary; ModelCheckpoint(save_weights_only) and CSVLogger."""
import csv
import os

import numpy as np

from .modified_dense_model import load_weight_file, save_weight_file


class ModelCheckpoint(object):
    """keras.callbacks.ModelCheckpoint(filepath, verbose, save_weights_only=True, mode='min'): the
    reference saves every epoch (save_best_only is left False), filename formatted with epoch/logs."""

    def __init__(self, filepath, verbose=0, save_weights_only=True, mode='min'):
        self.filepath, self.verbose = filepath, verbose

    def on_epoch_end(self, model, epoch, logs):
        path = self.filepath.format(epoch=epoch + 1, **logs)      # '.h5' names get Keras-layout HDF5 files, anything else .npz
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        model.save_weights(path)
        if self.verbose:
            print("Epoch %05d: saving model to %s" % (epoch + 1, path))


class CSVLogger(object):
    """keras.callbacks.CSVLogger(filename): epoch,<sorted log keys> rows."""

    def __init__(self, filename):
        self.filename, self._keys = filename, None

    def on_epoch_end(self, model, epoch, logs):
        os.makedirs(os.path.dirname(self.filename) or ".", exist_ok=True)
        new = self._keys is None
        if new:
            self._keys = sorted(logs)
        with open(self.filename, "w" if new else "a", newline="") as f:
            w = csv.writer(f)
            if new:
                w.writerow(["epoch"] + self._keys)
            w.writerow([epoch] + [logs[k] for k in self._keys])


class KerasLikeModel(object):
    """Sub-classes provide: self.store (ParamStore), _forward_loss(inputs, targets, train) and predict()."""

    optimizer = None
    loss = None

    def compile(self, optimizer, loss=None):
        self.optimizer, self.loss = optimizer, loss
        self._invalidate_graphs()              # captured train steps hold the previous optimizer's state tensors

    def _invalidate_graphs(self):
        """Drop the captured train steps (step_graph.CapturedStep per batch shape): something they baked has moved."""
        self._steps = {}

    @property
    def trainable_weights(self):
        return list(self.store.trainable_names)

    @property
    def non_trainable_weights(self):
        return list(self.store.frozen_names)

    def get_weights_dict(self):
        return self.store.to_numpy()

    def save_weights(self, path):
        save_weight_file(path, self.get_weights_dict())

    def load_weights(self, filepath, by_name=False, skip_mismatch=False):
        loaded = load_weight_file(filepath) if isinstance(filepath, str) else dict(filepath)
        for k, v in loaded.items():
            if k not in self.store.w:
                if by_name:
                    continue
                raise KeyError("weight %s is not part of this model" % k)
            if tuple(np.shape(v)) != tuple(self.store.w[k].shape):
                if skip_mismatch:
                    continue
                raise ValueError("shape mismatch for %s" % k)
            self.store.assign(k, v, refresh=False)
        self._weights_changed()

    def _weights_changed(self):
        """Master weights were rewritten from outside the optimizer (load_weights, ParallelModel's broadcast): re-cast the bf16
        operand copies the bf16 GEMMs read (compute_dtype='bf16'); sub-classes also re-derive what they fold from weights."""
        self._invalidate_graphs()
        self.store.refresh_shadow()

    def summary(self):
        lines = ["%-40s %-22s %s" % ("weight", "shape", "trainable")]
        total = train = 0
        for names, tr in ((self.store.trainable_names, True), (self.store.frozen_names, False)):
            for n in names:
                shp = tuple(self.store.w[n].shape)
                k = int(np.prod(shp))
                total += k
                train += k if tr else 0
                lines.append("%-40s %-22s %s" % (n, shp, tr))
        lines.append("Total params: %d  Trainable: %d  Non-trainable: %d" % (total, train, total - train))
        return "\n".join(lines)

    # ---- losses without a host round trip --------------------------------------------------------------------------------
    # train_on_batch / test_on_batch return Python floats like Keras and therefore wait for the step.  Training loops
    # (fit_generator, ParallelModel) use the *_device forms instead: a float32 DEVICE tensor of the step's loss terms, nothing
    # synchronised; _losses_to_api turns its host copy into what the Keras call returns.
    def train_on_batch_device(self, inputs, targets):
        raise NotImplementedError

    @staticmethod
    def _losses_to_api(v):
        return float(v[0])

    def fit_generator(self, generator, epochs=1, steps_per_epoch=None, callbacks=None, validation_data=None,
                      verbose=1, max_queue_size=10, workers=1, use_multiprocessing=False, initial_epoch=0):
        """keras.Model.fit_generator as the reference calls it (text_generation_model_v2.py:300-313: workers=1,
        max_queue_size=10; text_generation_model.py:458-472: max_queue_size=100).  Keras' GeneratorEnqueuer semantics:
        workers >= 1 runs next(generator) on ONE background thread per worker filling a queue of at most max_queue_size batches
        (threads, not processes: use_multiprocessing=True with a plain generator is refused like Keras warns against; more than
        one worker needs a thread-safe generator exactly as in Keras); workers=0 runs the generator on the calling thread.
        The loop itself never waits for the GPU: step losses stay on the device and are read once per epoch."""
        if use_multiprocessing:
            raise NotImplementedError("use_multiprocessing=True duplicates a plain generator in every process (Keras warns about it); "
                                      "the reference uses threads (use_multiprocessing=False)")
        if steps_per_epoch is None:
            raise ValueError("steps_per_epoch is required for a generator")
        outer = getattr(self, "_outer", None) or self          # under ParallelModel the loop feeds GLOBAL batches through the wrapper
        enq = GeneratorEnqueuer(generator, workers, max_queue_size) if workers and workers > 0 else None
        batches = enq.get() if enq is not None else generator
        history = []
        try:
            for epoch in range(initial_epoch, epochs):
                acc = None
                for _ in range(steps_per_epoch):
                    inputs, targets = next(batches)
                    step = outer.train_on_batch_device(inputs, targets)      # device tensor; its buffer is reused by the next step
                    acc = step.clone() if acc is None else acc.add_(step)    # (bookkeeping on the step's stream, not model arithmetic)
                logs = {"loss": self._losses_to_api((acc / steps_per_epoch).cpu().numpy())}       # the epoch's ONE host sync
                if validation_data is not None:
                    logs["val_loss"] = float(outer.test_on_batch(validation_data[0], validation_data[1]))
                if verbose and getattr(self, "is_chief", True):
                    print("Epoch %d/%d - " % (epoch + 1, epochs) + " - ".join("%s: %.4f" % kv for kv in sorted(logs.items())))
                if getattr(self, "is_chief", True):
                    for cb in callbacks or []:
                        cb.on_epoch_end(self, epoch, logs)
                history.append(logs)
        finally:
            if enq is not None:
                enq.stop()
        return history


class GeneratorEnqueuer(object):
    """keras.utils.GeneratorEnqueuer(generator, use_multiprocessing=False): `workers` daemon threads call next(generator)
    (under a lock: a Python generator is not re-entrant; with one worker -- the reference's setting -- the lock is free) and put
    batches into a bounded queue; get() yields them in arrival order.  A generator that raises ends the stream with that error;
    StopIteration ends it cleanly."""

    _END = object()

    def __init__(self, generator, workers=1, max_queue_size=10):
        import queue
        import threading
        self._gen, self._q = generator, queue.Queue(maxsize=max(1, int(max_queue_size)))
        self._stop, self._lock = threading.Event(), threading.Lock()
        self._threads = [threading.Thread(target=self._work, daemon=True) for _ in range(max(1, int(workers)))]
        for t in self._threads:
            t.start()

    def _put(self, item):
        import queue
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.05)
                return
            except queue.Full:
                continue

    def _work(self):
        while not self._stop.is_set():
            try:
                with self._lock:
                    item = next(self._gen)
            except StopIteration:
                self._put(self._END)
                return
            except BaseException as e:          # hand the error to the consumer
                self._put(e)
                return
            self._put(item)

    def get(self):
        while True:
            item = self._q.get()
            if item is self._END:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    def stop(self):
        self._stop.set()
        for t in self._threads:
            t.join(timeout=5)
