"""Diagnostics without provoking a fault: which device pointers does the captured joint step bake that are NOT backed by live memory
afterwards?  Every Tensor.data_ptr() call made while the step is captured is recorded with its Python stack; after the capture the
allocator's snapshot says, per pointer, whether its block is still allocated and which pool it belongs to.  A pointer into a FREED block
of the default pool is a use-after-free waiting for the next replay.  The graph is never replayed here."""
import gc
import os
import sys
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import test_gpu_fullsize as F


def main():
    dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
    model, cfg, inputs = F._full_size_joint(dt)
    inputs[0] = torch.tensor(inputs[0], device="cuda")
    model.use_step_graph = False
    for k in range(3):
        out = model.train_on_batch(inputs)
    torch.cuda.synchronize()
    print("eager steps ok", out, flush=True)
    # ---- capture by hand, recording pointers
    images, _meta, rpn_match, rpn_bbox, gt_caps, gt_boxes = inputs[:6]
    p = model.plan()
    gt_norm = (np.asarray(gt_boxes[0], np.float32) / np.array([p.H, p.W, p.H, p.W], np.float32)).astype(np.float32)
    rpn_up = model._step_uploads(p, rpn_match, rpn_bbox, gt_norm, gt_caps[0], True)
    p.forward(model._images_u8(images))
    torch.cuda.synchronize()
    rec = []
    orig = torch.Tensor.data_ptr

    def spy(self):
        ptr = orig(self)
        if self.is_cuda:
            rec.append((ptr, self.numel() * self.element_size(), "".join(traceback.format_stack(limit=5)[:-1])))
        return ptr
    g = torch.cuda.CUDAGraph()
    torch.Tensor.data_ptr = spy
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            losses = model._after_encoder(p, rpn_up, "rng", True, gt_caps[0], gt_norm)
            model.optimizer.apply(model.store, grad_scale=1.0, lr_t_dev=rpn_up["lr_t"])
    finally:
        torch.Tensor.data_ptr = orig
    gc.collect()
    print("captured: %d pointer uses" % len(rec), flush=True)
    snap = torch.cuda.memory_snapshot()
    blocks = []
    for seg in snap:
        pool = tuple(seg.get("segment_pool_id", (0, 0)))
        a = seg["address"]
        for b in seg["blocks"]:
            blocks.append((a, a + b["size"], b["state"], pool))
            a += b["size"]
    blocks.sort()
    import bisect
    starts = [b[0] for b in blocks]
    bad, unknown = {}, 0
    for ptr, nbytes, stack in rec:
        i = bisect.bisect_right(starts, ptr) - 1
        if i < 0 or not (blocks[i][0] <= ptr < blocks[i][1]):
            unknown += 1
            continue
        lo, hi, state, pool = blocks[i]
        if state != "active_allocated" and pool == (0, 0):
            bad.setdefault(stack, []).append((hex(ptr), nbytes, state))
        elif ptr + nbytes > hi and state == "active_allocated":
            pass
    print("pointers outside the caching allocator:", unknown)
    print("DANGLING (freed block of the default pool): %d distinct call sites" % len(bad))
    for stack, items in list(bad.items())[:12]:
        print("---- %d uses, e.g. %s" % (len(items), items[0]))
        print(stack)
    pools = {}
    for lo, hi, state, pool in blocks:
        pools.setdefault((pool, state), 0)
        pools[(pool, state)] += hi - lo
    for k, v in sorted(pools.items()):
        print(k, "%.1f MB" % (v / 1e6))


if __name__ == "__main__":
    main()
