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
            self.store.assign(k, v)
        self._weights_changed()

    def _weights_changed(self):
        pass

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

    def fit_generator(self, generator, epochs=1, steps_per_epoch=None, callbacks=None, validation_data=None,
                      verbose=1, max_queue_size=10, workers=1, use_multiprocessing=False, initial_epoch=0):
        """Runs the generator on the calling thread (the reference lets Keras run it on one
        background thread; the synthetic/benchmark path feeds device-resident batches instead)."""
        history = []
        for epoch in range(initial_epoch, epochs):
            losses = []
            for _ in range(steps_per_epoch):
                inputs, targets = next(generator)
                losses.append(self.train_on_batch(inputs, targets))
            logs = {"loss": float(np.mean(losses))}
            if validation_data is not None:
                logs["val_loss"] = float(self.test_on_batch(validation_data[0], validation_data[1]))
            if verbose:
                print("Epoch %d/%d - " % (epoch + 1, epochs) + " - ".join("%s: %.4f" % kv for kv in sorted(logs.items())))
            for cb in callbacks or []:
                cb.on_epoch_end(self, epoch, logs)
            history.append(logs)
        return history
