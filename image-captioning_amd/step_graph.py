"""A train step as ONE replayed hipGraph (the train_on_batch path of text_generation_model.py:425-438 and
text_generation_model_v2.py:262-287; the joint model's step has its own driver in dense_model.py).

The decoder-only configurations (BASELINE configs[0] / configs[1]) are chains of 60 - 150 kernels of 5 - 20 us each: issued one by one
from Python the host is the bottleneck (10 us per launch through ctypes), not the GPU.  Here the step is enqueued once into a hipGraph
-- forward, loss, backward, AMSGrad -- and replayed; what changes from step to step reaches the kernels through persistent device
buffers:

  * PackedInputs: the host's per-step words (token ids, masks, targets, index tables, Keras' lr_t, the dropout stream position) packed
    into ONE buffer and moved with ONE asynchronous copy from a small ring of page-locked buffers;
  * the RoI features: copied into a persistent device tensor (device -> device, or one upload when the caller holds them on the host).

Values that Python computed while the capture ran (optimizer.iterations, the dropout step counter) are advanced by hand on every replay.
Replays are bit-identical to the eager step: the same launches with the same arguments (tests/test_gpu_models.py).
"""
import math
import os
import warnings

import numpy as np
import torch

from . import ops


def enabled():
    """DCAP_STEP_GRAPH=0 issues every step eagerly (the same launches, one by one)."""
    return os.environ.get("DCAP_STEP_GRAPH", "1") != "0"


class PackedInputs(object):
    """Per-step host inputs as 4-byte words in ONE persistent device buffer, filled by ONE asynchronous copy per step.
    sizes: [(key, n_words)]; every part starts 16-byte aligned.  The device side has fixed addresses (a captured hipGraph replays
    them); the host side is a ring of page-locked buffers, each guarded by the event of its last copy, so the host never waits for the
    device and never rewrites a buffer whose copy is still queued."""
    SLOTS = 4

    def __init__(self, device, sizes):
        self.off, pos = {}, 0
        for k, n in sizes:
            self.off[k] = (pos, int(n))
            pos += (int(n) + 3) // 4 * 4
        self.words = max(pos, 4)
        self.dev = torch.zeros(self.words, dtype=torch.int32, device=device)
        self.pins = [torch.zeros(self.words, dtype=torch.int32, pin_memory=True) for _ in range(self.SLOTS)]
        self.events = [None] * self.SLOTS
        self.k = 0

    def view(self, key, dtype=torch.int32):
        o, n = self.off[key]
        v = self.dev[o:o + n]
        return v if dtype == torch.int32 else v.view(dtype)

    def bytes_view(self, key, nbytes):
        """The first nbytes of a part as uint8 (Keras masks travel as bytes)."""
        o, n = self.off[key]
        if nbytes > 4 * n:
            raise ValueError("%s: %d bytes do not fit the %d reserved" % (key, nbytes, 4 * n))
        return self.dev[o:o + n].view(torch.uint8)[:nbytes]

    def upload(self, parts):
        """parts: {key: numpy array of int32 / float32 / uint32 words, or uint8 bytes}; missing keys are zero."""
        k = self.k
        self.k = (k + 1) % self.SLOTS
        if self.events[k] is not None:
            self.events[k].synchronize()                    # (SLOTS steps old: complete long ago unless the host runs far ahead)
        host = self.pins[k].numpy()
        host[:] = 0
        for key, a in parts.items():
            o, n = self.off[key]
            a = np.ascontiguousarray(a).reshape(-1)
            if a.dtype == np.uint8 or a.dtype == np.bool_:
                if a.size > 4 * n:
                    raise ValueError("%s: %d bytes do not fit the %d reserved" % (key, a.size, 4 * n))
                host[o:o + n].view(np.uint8)[:a.size] = a.view(np.uint8)
                continue
            if a.size > n:
                raise ValueError("%s: %d words do not fit the %d reserved" % (key, a.size, n))
            if a.dtype.itemsize != 4:
                raise TypeError("%s: 4-byte words expected, got %s" % (key, a.dtype))
            host[o:o + a.size] = a if a.dtype == np.int32 else a.view(np.int32)
        self.dev.copy_(self.pins[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev


class CapturedStep(object):
    """One shape of one model's train step: two eager runs (they size every buffer and workspace), then a capture, then replays.
    `bufs` is the scratch-buffer dictionary the step's kernels use -- private to this shape, so that another batch shape or a
    predict() call in between can never free a buffer whose address the graph has baked."""
    WARM = 2

    def __init__(self):
        self.graph = None
        self.out = None
        self.warm = 0
        self.bufs = {}
        self.failed = None            # the error text when the capture failed: this shape then stays eager
        self._keep = None             # the split-K workspace the captured launches point into

    def run(self, body, counters_get, counters_set, on_replay):
        """body(): enqueue the step, return its output tensor(s).  counters_get() / counters_set(v): the host-side counters body()
        advances (restored when a capture fails, since the eager retry advances them again); on_replay(): advance them by one step."""
        if self.graph is not None:
            self.graph.replay()
            on_replay()
            return self.out
        if self.failed is not None or self.warm < self.WARM:
            self.warm += 1
            return body()
        saved = counters_get()
        try:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with ops.no_gc_during_capture(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = body()
                keep = ops.WORKSPACE.current()
        except RuntimeError as e:                            # a capture error (torch raises RuntimeError): this shape stays eager
            warnings.warn("train step: hipGraph capture failed (%s); running eagerly" % (repr(e)[:200],))
            self.failed = repr(e)[:200]
            counters_set(saved)
            torch.cuda.synchronize()
            return body()
        self.graph, self.out, self._keep = graph, out, keep
        graph.replay()                                       # (a capture records, it does not run: this is the step itself)
        return out


def lr_word(opt):
    """Keras' lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) of the update the NEXT step ends with (t = iterations + 1), as the float32 word
    the eager launch would carry as its argument."""
    t = opt.iterations + 1
    return np.array([opt.lr * math.sqrt(1.0 - opt.beta_2 ** t) / (1.0 - opt.beta_1 ** t)], np.float32)
