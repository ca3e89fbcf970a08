"""Two-stream training pipeline: encoder(i+1) overlaps decoder(i).

The encoder (frozen ResNet-101+FPN+RoIAlign) does not depend on the trainable parameters, so the feature
extraction of batch i+1 can run while batch i is still in its decoder forward/backward, gradient
all-reduce and AMSGrad update.  The encoder's ~100 large conv launches keep the CUs full; the decoder's
many small, latency-bound launches (skinny LSTM GEMMs, reductions, the RCCL all-reduce) fill in beside
them instead of serialising behind them.  Semantics are unchanged: batch i's update uses exactly batch
i's features and the weights after update i-1 (no staleness); the two RoI-feature buffers are handed over
with HIP events.
"""
import os

import torch


class CaptionTrainPipeline(object):
    def __init__(self, plan, decoder, rois_per_image):
        # (round 6, tried and removed: starting the decoder step only once the encoder pass has got behind stage 2 / stage 3 -- the
        # encoder pass as two graphs with an event between them -- to keep the decoder's small kernels away from the bandwidth-bound early
        # layers, which lose 0.3 ms per pass beside them: 6.18 / 6.15 ms against 6.18 / 6.14 without, profiles/r06_decoder_behind.txt)
        self.plan, self.dec = plan, decoder
        dev = plan.device
        # equal stream priorities (round 6): the decoder's short kernels fill what the encoder's launches leave; with the decoder stream at
        # high priority (rounds 2 - 5) they were served first and cost the encoder pass -- the critical path -- 0.6 % (10 882 / 10 967 against
        # 10 953 / 11 037 captions/s, alternating runs on one box; encoder high / decoder normal: 10 870 / 10 860)
        self.s_enc = torch.cuda.Stream(device=dev)
        self.s_dec = torch.cuda.Stream(device=dev)
        B = plan.B
        self.fc = fc = getattr(plan, "feat_channels", 256)
        self.feat = [torch.empty((B, rois_per_image, 7, 7, fc), dtype=torch.float32, device=dev) for _ in range(2)]
        self.ev_feat = [torch.cuda.Event() for _ in range(2)]       # features of slot ready
        self.ev_free = [torch.cuda.Event() for _ in range(2)]       # decoder done reading slot
        self.pending = None                                          # (slot, tables) awaiting its decoder pass
        self.n = 0
        self.late_wait = os.environ.get("DCAP_PIPE_LATE_WAIT", "1") != "0"      # the slot event gates the RoIAlign launch only (see _encode)
        # inputs of the steps in flight, each with the event recorded behind its decoder pass (which itself waits for its encoder
        # pass): their memory -- allocated on the PRODUCER's stream -- must not go back to the allocator, and from there into the next
        # upload, while a kernel that reads them is still queued.  Lifetime follows the GPU, not a host step count: an entry is
        # dropped once its event has completed, and the host blocks on the oldest one when more than `max_in_flight` are pending
        # (which also bounds how far the host runs ahead of the device).
        self._hold = []
        self.max_in_flight = 4
        cur = torch.cuda.current_stream(dev)
        self.s_enc.wait_stream(cur)
        self.s_dec.wait_stream(cur)

    def _encode(self, slot, images, boxes):
        with torch.cuda.stream(self.s_enc):
            if self.n >= 2 and not self.late_wait:
                self.s_enc.wait_event(self.ev_free[slot])
            self.plan.forward(images)
            # Only the RoIAlign launch writes the slot the decoder pass of two batches ago read: the ~100 convolution launches in front of it
            # wait for nothing but the previous encoder pass.  (Rounds 2 - 6 waited HERE, in front of the whole pass: the encoder stream then
            # stood still from the end of each pass until the decoder step running beside it -- starved of CUs by the pass's persistent grids --
            # had finished: 0.69 ms of every 5.87 ms step in profiles/r06_headline_timeline.tsv.)
            if self.n >= 2 and self.late_wait:
                self.s_enc.wait_event(self.ev_free[slot])
            self.plan.roi_features(boxes_norm=boxes, out=self.feat[slot])
            self.ev_feat[slot].record(self.s_enc)

    def _decode(self, slot, tables):
        with torch.cuda.stream(self.s_dec):
            self.s_dec.wait_event(self.ev_feat[slot])
            f = self.feat[slot]
            loss = self.dec.train_step(f.view(-1, 7, 7, self.fc), tables)
            self.ev_free[slot].record(self.s_dec)
            for h in self._hold:                        # this batch's inputs may go once the pass just enqueued has run
                if h[0] is tables and h[2] is None:
                    h[2] = torch.cuda.Event()
                    h[2].record(self.s_dec)
                    break                               # (the oldest pending entry: the one this pass belongs to)
        return loss

    def _release(self):
        hold = self._hold
        while hold and hold[0][2] is not None and hold[0][2].query():
            hold.pop(0)
        while len(hold) > self.max_in_flight and hold[0][2] is not None:
            hold[0][2].synchronize()                    # the host is more than max_in_flight batches ahead of the GPU: wait for the oldest
            hold.pop(0)

    def step(self, images, boxes, tables):
        """Enqueue the encoder of this batch and the decoder of the previous one.  images: uint8 [B,H,W,3]
        device tensor or None (already in plan.images); boxes: normalised [B,R,4]; tables: SampleTables.
        Returns the previous batch's loss (device scalar) or None on the first call."""
        slot = self.n & 1
        self._release()
        self._hold.append([tables, (images, boxes), None])
        self._encode(slot, images, boxes)
        loss = None
        if self.pending is not None:
            loss = self._decode(*self.pending)
        self.pending = (slot, tables)
        self.n += 1
        return loss

    def flush(self):
        """Run the decoder of the last enqueued batch and join both streams into the current stream."""
        loss = None
        if self.pending is not None:
            loss = self._decode(*self.pending)
            self.pending = None
        cur = torch.cuda.current_stream(self.plan.device)
        cur.wait_stream(self.s_enc)
        cur.wait_stream(self.s_dec)
        return loss


class JointTrainPipeline(object):
    """The joint model's train step (dense_model.DenseImageCapRCNN, BASELINE configs[4]) with the frozen backbone of batch i + 1 running
    beside the rest of batch i's step.

    Behind the backbone the step is a long chain of small, latency-bound launches (proposal selection, the single-wave NMS scan,
    detection targets, two LSTMs step by step, ~170 launches of under 10 us) that leave most of the chip idle; the backbone pass
    (ResNet-101, frozen: it reads nothing a train step changes) is ~100 convolution launches that fill it.  Two encoder plans with their
    own activation buffers alternate: step(batch k) enqueues plan[k % 2].forward_trunk() on a second stream and, on the caller's stream,
    the rest of batch k - 1's step (FPN, RPN, proposals ... AMSGrad) on the other plan.  Semantics are unchanged -- batch k's update uses
    batch k's C2..C5 and the weights after update k - 1; every kernel and every reduction order is the serial step's, so the weights after
    N steps are bit-equal to N calls of train_on_batch_device (tests/test_gpu_models.py) -- the result is just returned one call late,
    like CaptionTrainPipeline's.  Only when no ResNet stage is trainable (layers 'heads'-like sets; a trainable stage makes the backbone
    pass depend on the previous update and the model falls back to the serial step)."""

    def __init__(self, model):
        inner = getattr(model, "inner_model", model)
        if inner.backbone_from is not None:
            raise ValueError("JointTrainPipeline: ResNet stages are trainable (backbone_from = %r): the backbone pass of the next batch depends on "
                             "this batch's update" % (inner.backbone_from,))
        self.model, self.inner = model, inner
        self.plans = inner.plan_pair()
        dev = inner.device
        # (tried: the rest of the step on a HIGH-priority stream of its own, the backbone pass at normal priority -- 11.9 - 12.3 ms per step
        # against 6.89 - 6.93 with both at the default priority, same box, alternating runs; the backbone stream restricted to 30 / 28 / 24 / 20 of
        # the 32 CUs of every XCD (hipExtStreamCreateWithCUMask), to leave the chain's small kernels CUs of their own: 23.3 / 23.8 / 24.1 / 25.0 ms
        # against 6.75 -- profiles/r06_joint_pipeline_cumask.txt.  Queues that are not plain streams lose their concurrency on this runtime.)
        # (tried: the backbone pass enqueued LATER in the other batch's step instead of at its start -- behind the FPN / RPN forward 7.00 - 7.06 ms,
        # behind the decoder's forward 7.15 - 7.19, behind its backward or behind the FPN backward 7.73 - 7.77 = the serial step; at the start
        # 6.75 - 6.83: profiles/r06_joint_pipeline_phase.txt)
        # (tried: the backbone pass waiting only for an event behind the FPN backward of the step that last ran on its plan instead of for the
        # whole caller's stream, so that its first layers run beside that step's HBM-bound regulariser / optimizer passes: 6.75 - 6.78 ms
        # against 6.75 - 6.83, i.e. nothing -- the backbone pass has a whole step to finish in, where it starts inside it moves the contention,
        # it does not remove it; the only stream priority this runtime offers besides the default is "high": 13.7 ms with it on this stream.
        # The host is 5 steps ahead of the GPU after 20: 4.95 ms of enqueueing per 6.80 ms step, tools/joint_host_time.py)
        self.s_trunk = torch.cuda.Stream(device=dev)
        self.s_copy = torch.cuda.Stream(device=dev)                  # host images: uploaded on a stream that never waits for a step
        self.ev_trunk = [torch.cuda.Event(), torch.cuda.Event()]     # C2..C5 of plan j are complete
        self.pending = None                                          # (inputs, plan index) awaiting the rest of its step
        self.n = 0
        self._hold = []                                              # image tensors of the batches in flight (allocated on the caller's stream)
        inner.use_step_graph = False                                 # the rest of the step alternates between two plans' buffers: eager launches

    def _trunk(self, j, images):
        dev = self.inner.device
        cur = torch.cuda.current_stream(dev)
        u8 = self.inner._images_u8(images)
        if not u8.is_cuda:
            # A host batch (the data generator's): a copy from pageable memory blocks the host until the stream it is issued on has drained.
            # On the backbone stream that would be the end of batch k - 2's step -- the host could never run ahead of the GPU.  The copy goes
            # to a device tensor of its own on a stream with nothing else in it (the host waits for the 3 MB per image only); the backbone
            # stream then waits for that stream.  The tensor is kept in _hold until two more batches have been enqueued.
            with torch.cuda.stream(self.s_copy):
                u8 = u8.to(dev)
            self.s_trunk.wait_stream(self.s_copy)
            u8.record_stream(self.s_trunk)                 # (allocated on the copy stream, read on the backbone stream)
            self._hold.append(u8)
        # everything the caller's stream holds so far -- the rest of batch k - 2's step, the last reader of plan j's buffers, and whatever
        # produced `images` -- comes first
        self.s_trunk.wait_stream(cur)
        with torch.cuda.stream(self.s_trunk):
            self.plans[j].forward_trunk(u8)
            self.ev_trunk[j].record(self.s_trunk)

    def _rest(self, inputs, j):
        torch.cuda.current_stream(self.inner.device).wait_event(self.ev_trunk[j])
        self.inner.use_plan(j)
        losses = self.inner.train_on_batch_device(inputs, trunk_done=True)
        return self.model.mean_over_towers(losses) if self.model is not self.inner else losses      # (ParallelModel: the towers' mean)

    def step(self, inputs):
        """Enqueue the backbone pass of `inputs` and the rest of the previous batch's step.  Returns the previous batch's raw loss terms
        (device tensor [4]) or None on the first call."""
        j = self.n & 1
        if self.model is not self.inner:
            inputs = self.model.shard_inputs(inputs)                 # ParallelModel: this rank's share of the global batch (tf.split)
        self._hold = self._hold[-4:] + [inputs[0]]                   # (a host batch adds its device copy in _trunk: two entries per batch)
        self._trunk(j, inputs[0])
        losses = None
        if self.pending is not None:
            losses = self._rest(*self.pending)
        self.pending = (inputs, j)
        self.n += 1
        return losses

    def flush(self):
        """The rest of the last enqueued batch's step; joins the backbone stream into the caller's."""
        losses = None
        if self.pending is not None:
            losses = self._rest(*self.pending)
            self.pending = None
        torch.cuda.current_stream(self.inner.device).wait_stream(self.s_trunk)
        return losses
