"""Two-stream training pipeline: encoder(i+1) overlaps decoder(i).

The encoder (frozen ResNet-101+FPN+RoIAlign) does not depend on the trainable parameters, so the feature
extraction of batch i+1 can run while batch i is still in its decoder forward/backward, gradient
all-reduce and AMSGrad update.  The encoder's ~100 large conv launches keep the CUs full; the decoder's
many small, latency-bound launches (skinny LSTM GEMMs, reductions, the RCCL all-reduce) fill in beside
them instead of serialising behind them.  Semantics are unchanged: batch i's update uses exactly batch
i's features and the weights after update i-1 (no staleness); the two RoI-feature buffers are handed over
with HIP events.
"""
import torch


class CaptionTrainPipeline(object):
    def __init__(self, plan, decoder, rois_per_image):
        self.plan, self.dec = plan, decoder
        dev = plan.device
        self.s_enc = torch.cuda.Stream(device=dev)
        self.s_dec = torch.cuda.Stream(device=dev, priority=-1)      # its short kernels slot in between the convs
        B = plan.B
        self.fc = fc = getattr(plan, "feat_channels", 256)
        self.feat = [torch.empty((B, rois_per_image, 7, 7, fc), dtype=torch.float32, device=dev) for _ in range(2)]
        self.ev_feat = [torch.cuda.Event() for _ in range(2)]       # features of slot ready
        self.ev_free = [torch.cuda.Event() for _ in range(2)]       # decoder done reading slot
        self.pending = None                                          # (slot, tables) awaiting its decoder pass
        self.n = 0
        self._hold = []        # inputs of the last steps: their memory must not go back to the allocator (and be handed to the next
                               # step's uploads, made on another stream) while kernels that read them are still queued
        cur = torch.cuda.current_stream(dev)
        self.s_enc.wait_stream(cur)
        self.s_dec.wait_stream(cur)

    def _encode(self, slot, images, boxes):
        with torch.cuda.stream(self.s_enc):
            if self.n >= 2:
                self.s_enc.wait_event(self.ev_free[slot])
            self.plan.forward(images)
            self.plan.roi_features(boxes_norm=boxes, out=self.feat[slot])
            self.ev_feat[slot].record(self.s_enc)

    def _decode(self, slot, tables):
        with torch.cuda.stream(self.s_dec):
            self.s_dec.wait_event(self.ev_feat[slot])
            f = self.feat[slot]
            loss = self.dec.train_step(f.view(-1, 7, 7, self.fc), tables)
            self.ev_free[slot].record(self.s_dec)
        return loss

    def step(self, images, boxes, tables):
        """Enqueue the encoder of this batch and the decoder of the previous one.  images: uint8 [B,H,W,3]
        device tensor or None (already in plan.images); boxes: normalised [B,R,4]; tables: SampleTables.
        Returns the previous batch's loss (device scalar) or None on the first call."""
        slot = self.n & 1
        self._hold = (self._hold + [(images, boxes, tables)])[-3:]
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
