"""Data parallelism with the reference's ParallelModel semantics, MI355X-style.

The reference (parallel_model.py:22-102) replicates the Keras graph on `gpu_count` towers inside one
process, tf.split()s every input on axis 0, and reduces the towers' [1,1]-reshaped losses with
tf.reduce_mean => the update uses the MEAN over towers of the per-tower gradients.
Here each tower is one process on one GPU (torchrun / torch.distributed, backend "nccl" == RCCL over
xGMI); the only exchange per step is one all-reduce(sum) of the flat gradient bucket
(params.ParamStore.flat_grad), scaled by 1/world inside the fused AMSGrad kernel.  Weights and
optimizer state are replicated; nothing else is communicated.
"""
import os

import torch
import torch.distributed as dist


def init_process_group_from_env(backend=None):
    """torchrun contract: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT.  One process per GPU:
    the device is bound (cuda:LOCAL_RANK) BEFORE the RCCL communicator is created.  DCAP_DIST_BACKEND=gloo
    rehearses the multi-process path on a box with fewer GPUs than ranks (ranks then share devices)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = os.environ.get("DCAP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        n = torch.cuda.device_count()
        if backend == "nccl" and local_rank >= n:
            raise RuntimeError("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, n))
        local_rank = local_rank % max(n, 1)
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
    return rank, world, local_rank


def reserve_cus_for_collectives(world, cus=None):
    """world > 1: the encoder's persistent Winograd grids (one block per CU, held for the whole launch) leave 8 of the 256 CUs --
    one per XCD -- to the RCCL kernels of the gradient all-reduce, which otherwise could only start between encoder launches.
    DCAP_WINO_CUS overrides; call before the encoder plan captures its hipGraph (a captured graph keeps its grid)."""
    if world <= 1 or not torch.cuda.is_available():
        return None
    from . import ops
    want = int(os.environ.get("DCAP_WINO_CUS", cus if cus is not None else 248))
    return ops.set_persistent_cus(want)


def shard(x, rank, world):
    """tf.split(x, gpu_count) on axis 0, keeping this rank's slice (parallel_model.py:60-62)."""
    n = x.shape[0] if hasattr(x, "shape") else len(x)
    if n % world:
        raise ValueError("batch of %d does not split evenly over %d towers" % (n, world))
    per = n // world
    return x[rank * per:(rank + 1) * per]


class GradAllReduce(object):
    """Callable installed as model.grad_sync: sums the gradient bucket over ranks (in `buckets` chunks so
    RCCL can pipeline them over all 7 xGMI links) and returns the scale (1/world) the optimizer applies.

    dtype='bf16' (opt-in; SURVEY section 5 allows configs[4]'s exchange in bf16): every range is rounded into a persistent bf16
    bucket as it becomes final, the bf16 slices are all-reduced -- half the bytes per link: ~150 MB instead of ~300 MB for the joint
    model's 77 M parameters -- and the summed bf16 values go back into the fp32 gradient bucket before the clip norm and AMSGrad read
    it (fp32 master weights, fp32 optimizer state; only the wire format changes).  Every rank receives the same bits from the
    collective, so replicas stay bit-identical; the update differs from the fp32 exchange by bf16 rounding of the gradient
    (tests/test_dp_gloo.py, tests/test_gpu_multirank.py).  The default wire is fp32 (the reference's tower mean is fp32):
    ParallelModel(grad_dtype='bf16') or DCAP_GRAD_DTYPE=bf16 selects this one."""

    def __init__(self, group=None, bucket_bytes=64 << 20, dtype="f32"):
        if dtype not in ("f32", "bf16"):
            raise ValueError("GradAllReduce dtype must be 'f32' or 'bf16'")
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.dtype = dtype
        self.bucket_elems = max(1, bucket_bytes // (2 if dtype == "bf16" else 4))
        self._wire = None               # bf16: the persistent bf16 twin of the gradient bucket

        self._pending = []              # (work handle, lo, hi) of the ranges already on the wire this step
        self.ranks_seen = None          # world size RCCL reported after a real all-reduce (bench.py prints it)
        # timing=True: a HIP event pair on the CURRENT stream around the step's waits -- the time the compute stream stood still for
        # the exchange (the EXPOSED part of the all-reduce; what overlapped the backward does not show) -- plus the host's own wait
        self.timing = False
        self._marks, self._host_wait = [], 0.0

    def ready(self, flat_grad, lo, hi):
        """A contiguous range [lo, hi) of the gradient bucket is final (its layer group's backward has been enqueued):
        start its all-reduce now, asynchronously, so it travels over xGMI while the rest of the backward still runs
        (bucketed against backward).  Ranges are cut into bucket_elems pieces."""
        if self.world == 1 or hi <= lo:
            return
        src = flat_grad
        if self.dtype == "bf16":
            if self._wire is None or self._wire.numel() != flat_grad.numel() or self._wire.device != flat_grad.device:
                self._wire = torch.empty(flat_grad.numel(), dtype=torch.bfloat16, device=flat_grad.device)
            self._round(flat_grad[lo:hi], self._wire[lo:hi])
            src = self._wire
        for o in range(lo, hi, self.bucket_elems):
            e = min(hi, o + self.bucket_elems)
            self._pending.append((dist.all_reduce(src[o:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True), o, e))

    @staticmethod
    def _round(src, dst):
        """fp32 -> bf16 (round to nearest even) of one range.  On the GPU the library's cast kernel; host tensors (the gloo tests of
        the bucket logic run without a GPU) take torch's conversion, which rounds the same way."""
        if src.is_cuda:
            from . import ops
            ops.to_bf16(src, out=dst)
        else:
            dst.copy_(src)

    @staticmethod
    def _widen(src, dst):
        if src.is_cuda:
            from . import ops
            ops.from_bf16(src, dst)
        else:
            dst.copy_(src)

    def __call__(self, flat_grad):
        """Finish the step's exchange: reduce whatever ready() has not covered, wait for everything in flight (the
        current stream then waits for the collectives), return the scale the optimizer applies (mean over towers)."""
        if self.world == 1:
            return 1.0
        n = flat_grad.numel()
        done = sorted((lo, hi) for _, lo, hi in self._pending)
        pos = 0
        for lo, hi in done + [(n, n)]:
            if lo > pos:
                self.ready(flat_grad, pos, lo)
            pos = max(pos, hi)
        if self.timing and flat_grad.is_cuda:
            import time
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            t0 = time.perf_counter()
        for work, _, _ in self._pending:
            work.wait()
        if self.timing and flat_grad.is_cuda:
            e1.record()
            self._host_wait += time.perf_counter() - t0
            self._marks.append((e0, e1))
        if self.dtype == "bf16":                             # the summed bf16 values back into the fp32 bucket (exact)
            for lo, hi in self._merged(sorted((lo, hi) for _, lo, hi in self._pending)):
                self._widen(self._wire[lo:hi], flat_grad[lo:hi])
        self._pending = []
        return 1.0 / self.world

    @staticmethod
    def _merged(ranges):
        out = []
        for lo, hi in ranges:
            if out and lo <= out[-1][1]:
                out[-1][1] = max(out[-1][1], hi)
            else:
                out.append([lo, hi])
        return out

    def exposed_ms(self, reset=True):
        """(mean exposed all-reduce time per step on the compute stream in ms, mean host wait per step in ms, steps) since the last
        reset; synchronises the device."""
        if not self._marks:
            return None
        torch.cuda.synchronize()
        n = len(self._marks)
        dev_ms = sum(a.elapsed_time(b) for a, b in self._marks) / n
        host_ms = 1e3 * self._host_wait / n
        if reset:
            self._marks, self._host_wait = [], 0.0
        return dev_ms, host_ms, n

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)

    def check_ranks(self, device):
        """An actual all-reduce of ones: the number of ranks the backend really connected."""
        t = torch.ones(1, dtype=torch.float32, device=device)
        if self.world > 1:
            dist.all_reduce(t, group=self.group)
        self.ranks_seen = int(round(float(t.item())))
        return self.ranks_seen


def default_grad_dtype(keras_model=None):
    """The gradient exchange's wire format when the caller names none: DCAP_GRAD_DTYPE if set, else fp32 for EVERY model -- the
    reference's towers are averaged in fp32 (parallel_model.py:88-102), and a bf16 wire rounds each tower's gradient before the
    collective and sums in bf16 inside RCCL, an error that grows with the rank count (tests/test_dp_gloo.py measures it at 8 ranks).
    The bf16 wire (half the bytes per xGMI link) is opt-in: ParallelModel(grad_dtype='bf16') or DCAP_GRAD_DTYPE=bf16."""
    return os.environ.get("DCAP_GRAD_DTYPE") or "f32"


class ParallelModel(object):
    """ParallelModel(keras_model, gpu_count): same constructor as the reference.  gpu_count must equal
    the torch.distributed world size (one process per GPU).  Attribute access falls through to the
    wrapped model (the reference's __getattribute__ trick, parallel_model.py:41-46)."""

    def __init__(self, keras_model, gpu_count, grad_dtype=None):
        """grad_dtype: 'f32' or 'bf16' = the gradient exchange's wire format (GradAllReduce); default: default_grad_dtype(keras_model)."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        if gpu_count != world:
            raise ValueError("gpu_count=%d but %d process(es) are running: launch one process per GPU "
                             "(python -m torch.distributed.run --nproc-per-node %d ...)" % (gpu_count, world, gpu_count))
        self.inner_model = keras_model
        self.gpu_count = gpu_count
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        reserve_cus_for_collectives(world)
        keras_model.grad_sync = GradAllReduce(dtype=grad_dtype or default_grad_dtype(keras_model))
        keras_model.is_chief = self.rank == 0          # one rank prints and writes checkpoints (the others barrier)
        keras_model._outer = self                      # the wrapped model's train() loop feeds global batches through this wrapper
        self.broadcast_weights()

    def broadcast_weights(self):
        """Towers share variables in the reference; here rank 0's weights seed every replica."""
        if self.gpu_count > 1:
            dist.broadcast(self.inner_model.store.flat, src=0)
            for k in self.inner_model.store.frozen_names:
                dist.broadcast(self.inner_model.store.w[k], src=0)
            self.inner_model._weights_changed()
            # the reference's towers draw INDEPENDENT K.dropout masks for their shards (recurrent_dropout=0.2, text_generation_model.py:
            # 141-142); every rank built its model from the same seed, so fold the rank into the Philox key of the mask stream
            for m in (self.inner_model, getattr(self.inner_model, "caption_model", None)):
                if m is not None and hasattr(m, "_drop_seed"):
                    base = m.__dict__.setdefault("_drop_seed_base", m._drop_seed)          # idempotent: derived from the build seed every time
                    m._drop_seed = (base ^ (self.rank * 0x9E3779B9)) & 0xFFFFFFFF
            # ... and shuffle their proposals independently (tf.random_shuffle inside each tower's DetectionTargetLayer,
            # dense_img_cap/dense_model.py:450-528): the rank enters the key of the joint model's detection-target sort as well
            if hasattr(self.inner_model, "_dt_rank"):
                self.inner_model._dt_rank = self.rank

    def __getattr__(self, name):
        return getattr(self.inner_model, name)

    def shard_inputs(self, inputs):
        return [shard(x, self.rank, self.gpu_count) for x in inputs]

    def mean_over_towers(self, raw):
        """tf.reduce_mean over the towers' loss terms (parallel_model.py:88-102): all-reduce(sum) of a small float32 device
        tensor, scaled by 1/gpu_count -- enqueued like the gradient exchange, the host does not wait (RCCL; the gloo rehearsal
        backend stages through the host by nature).  Returns a new device tensor."""
        if self.gpu_count == 1:
            return raw
        t = raw.to(torch.float32).clone()
        dist.all_reduce(t)
        return t.mul_(1.0 / self.gpu_count)

    def _run_device(self, fn, inputs, targets):
        mine = None if targets is None or len(targets) == 0 else shard(targets, self.rank, self.gpu_count)
        return self.mean_over_towers(fn(self.shard_inputs(inputs), mine))

    def train_on_batch_device(self, inputs, targets=None):
        """Global batch in, split like tf.split; the mean over towers of the towers' loss terms as a DEVICE tensor -- no host
        synchronisation anywhere in the step, so a training loop (fit_generator, train()) keeps enqueueing and the two-stream
        overlap survives; inner_model._losses_to_api(host copy) gives what train_on_batch returns."""
        return self._run_device(self.inner_model.train_on_batch_device, inputs, targets)

    def train_on_batch(self, inputs, targets=None):
        """Keras return value (a float, or the joint model's [loss, rpn_class_loss, rpn_bbox_loss, imgcap_loss] list): ONE
        device->host copy at the end of the step, as on a single GPU."""
        return self.inner_model._losses_to_api(self.train_on_batch_device(inputs, targets).cpu().numpy())

    def test_on_batch(self, inputs, targets=None):
        return self.inner_model._losses_to_api(self._run_device(self.inner_model.test_on_batch_device, inputs, targets).cpu().numpy())
