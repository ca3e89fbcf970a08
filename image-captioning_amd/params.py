"""Flat parameter storage and the Keras-compatible AMSGrad optimizer.

All trainable weights of a model live in ONE contiguous fp32 buffer (and so do their gradients and
the Adam moments): the optimizer is a single fused kernel launch over the bucket, and data-parallel
training all-reduces the gradient bucket directly (RCCL) with no flatten/unflatten copies.
Per-weight tensors are views into the buckets; every offset is 16-byte aligned.
"""
import math

import numpy as np
import torch

from . import ops


class ParamStore:
    def __init__(self, device):
        self.device = torch.device(device)
        self._train, self._frozen = [], []           # (name, ndarray)
        self.w, self.grad = {}, {}                    # name -> view
        self.flat = self.flat_grad = None
        self.trainable_names, self.frozen_names = [], []

    def add(self, name, array, trainable):
        (self._train if trainable else self._frozen).append((name, np.ascontiguousarray(array, np.float32)))

    def finalize(self):
        off, layout = 0, []
        for name, a in self._train:
            layout.append((name, off, a.shape))
            off += (a.size + 3) // 4 * 4
        self.n_train = off
        flat = np.zeros(max(off, 4), np.float32)
        for (name, o, shape), (_, a) in zip(layout, self._train):
            flat[o:o + a.size] = a.reshape(-1)
        self.flat = torch.tensor(flat, device=self.device)
        self.flat_grad = torch.zeros_like(self.flat)
        self._ranges, covered = {}, {}
        for name, o, shape in layout:
            n = int(np.prod(shape))
            layer = name.split('/')[0]
            lo, hi = self._ranges.get(layer, (o, o))
            self._ranges[layer] = (min(lo, o), max(hi, o + (n + 3) // 4 * 4))
            covered[layer] = covered.get(layer, 0) + (n + 3) // 4 * 4
            self.w[name] = self.flat[o:o + n].view(*shape)
            self.grad[name] = self.flat_grad[o:o + n].view(*shape)
            self.trainable_names.append(name)
        # a layer whose weights are not adjacent in the bucket has no range of its own
        self._ranges = {k: r for k, r in self._ranges.items() if r[1] - r[0] == covered[k]}
        for name, a in self._frozen:
            self.w[name] = torch.tensor(a, device=self.device)
            self.frozen_names.append(name)
        self._train = self._frozen = None
        self.flat_bf16, self.wb = None, {}
        return self

    def enable_bf16_shadow(self, names=None):
        """bf16 copies of the trainable weights `names` (default: all) for the bf16 GEMMs: ONE bf16 buffer mirroring the
        head of the flat bucket up to the last of those weights (fp32 stays the master the optimizer updates; the AMSGrad
        kernel refreshes the mirror in the same pass).  self.wb[name] are views shaped like self.w[name]."""
        names = list(self.trainable_names if names is None else names)
        base = self.flat.data_ptr()
        end = max((self.w[n].data_ptr() - base) // 4 + (self.w[n].numel() + 3) // 4 * 4 for n in names)
        self.flat_bf16 = torch.empty(end, dtype=torch.bfloat16, device=self.device)
        for n in names:
            o = (self.w[n].data_ptr() - base) // 4
            self.wb[n] = self.flat_bf16[o:o + self.w[n].numel()].view(*self.w[n].shape)
        self.refresh_shadow()
        return self

    def refresh_shadow(self):
        if self.flat_bf16 is not None:
            ops.to_bf16(self.flat[:self.flat_bf16.numel()], out=self.flat_bf16)

    def layer_range(self, layer):
        """[lo, hi) of the flat buckets holding every trainable weight of `layer` (a layer's weights are adjacent: the
        bucket is laid out in name order).  The unit of the bucketed gradient all-reduce."""
        if layer not in self._ranges:
            raise KeyError("layer %r has no contiguous range in the parameter bucket" % layer)
        return self._ranges[layer]

    def to_numpy(self):
        return {k: v.detach().cpu().numpy() for k, v in self.w.items()}

    def assign(self, name, array, refresh=True):
        """refresh=False: the caller assigns many weights and calls refresh_shadow() once afterwards (load_weights)."""
        t = self.w[name]
        a = np.asarray(array, np.float32)
        if tuple(a.shape) != tuple(t.shape):
            raise ValueError("shape mismatch for %s: %s vs %s" % (name, a.shape, tuple(t.shape)))
        t.copy_(torch.tensor(a, device=self.device))
        if refresh and self.flat_bf16 is not None and name in self.wb:
            self.refresh_shadow()


class Adam:
    """keras.optimizers.Adam(lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None -> K.epsilon()=1e-7,
    decay=0., amsgrad=False, clipnorm=None) -- the reference uses amsgrad=True everywhere
    (text_generation_model.py:425; _v2.py:266; dense_img_cap/dense_model.py:1699 adds clipnorm=0.5).
    Semantics: t = iterations+1; lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m, v as usual;
    vhat = max(vhat, v); p -= lr_t*m/(sqrt(vhat)+eps)."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0.0, amsgrad=False, clipnorm=None):
        if not amsgrad:
            raise NotImplementedError("only the amsgrad=True variant the reference trains with is implemented")
        if decay:
            raise NotImplementedError("lr decay is not used by the reference")
        self.lr, self.beta_1, self.beta_2 = lr, beta_1, beta_2
        self.epsilon = 1e-7 if epsilon is None else epsilon
        self.clipnorm = clipnorm
        self.iterations = 0
        self._state = None

    def baked_key(self):
        """What a captured train step bakes of this optimizer as kernel arguments (only lr reaches a replay, through the lr_t device
        word): part of every step-graph key, so that changing any of them after a capture takes a fresh capture instead of being
        silently ignored."""
        return (id(self), float(self.beta_1), float(self.beta_2), float(self.epsilon), float(self.clipnorm or 0.0))

    def _init(self, store):
        z = lambda: torch.zeros_like(store.flat)
        self._state = (z(), z(), z())
        self._gnorm = torch.zeros(1, dtype=torch.float32, device=store.device)

    def lr_t(self):
        t = self.iterations
        return self.lr * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)

    def apply(self, store, grad_scale=1.0, lr_t_dev=None, reg=None, reg_loss=None):
        """One update of every trainable weight from store.flat_grad (scaled by grad_scale, e.g.
        1/world_size after a summing all-reduce).  lr_t_dev: a float32 device word holding this step's lr_t (the caller wrote
        lr * sqrt(1 - b2^t) / (1 - b1^t) for t = iterations + 1 there): the launch then carries no per-step host value and can be
        replayed from a captured hipGraph.
        reg (ops.RegSegmentTable): the joint model's L2 regulariser and trainable mask are applied INSIDE the update -- store.flat_grad
        holds the plain loss gradient and is left alone; one read-only pass over (weights, gradient) gives the clip norm of the
        regularised gradient and the regulariser's loss term (-> reg_loss[0]) -- instead of a pass that rewrites the gradient bucket
        (three reads + one write) followed by a norm pass (one more read)."""
        if self._state is None:
            self._init(store)
        self.iterations += 1
        m, v, vh = self._state
        gn = None
        if reg is not None:
            if self.clipnorm or reg_loss is not None:
                ops.reg_sumsq(store.flat, store.flat_grad, reg, loss=reg_loss, gnorm_sq=self._gnorm if self.clipnorm else None)
            gn = self._gnorm if self.clipnorm else None
        elif self.clipnorm:
            gn = ops.sumsq(store.flat_grad, out=self._gnorm)
        ops.amsgrad_step(store.flat, store.flat_grad, m, v, vh, self.lr_t(), self.beta_1, self.beta_2, self.epsilon,
                         grad_scale=grad_scale, gnorm_sq=gn, clipnorm=self.clipnorm or 0.0, p_bf16=getattr(store, "flat_bf16", None),
                         lr_t_dev=lr_t_dev, reg=reg)
