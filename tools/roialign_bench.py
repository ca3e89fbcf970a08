"""dc_roi_align_pyramid_f32 bandwidth at the benchmark size (64 RoIs), at 16 images x 32 RoIs and at 16 x 128 -- the launches ROTATE over
independent input sets (pyramid maps + boxes + output) whose touched bytes exceed the 256 MB Infinity Cache, like bench.py's
hbm_kernels leg: what a launch reads is not in any cache.  tools/roialign_profile.sh runs this under rocprofv3 (kernel trace, FETCH_SIZE,
WRITE_SIZE) and divides the counters by the launches."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops, synth  # noqa: E402

dev = torch.device("cuda")
ROUNDS = 4                                          # every set is used ROUNDS times (+ one warm-up round)
for B, R in ((2, 32), (16, 32), (16, 128)):
    S = 1024
    alg = B * R * 250880.0          # <= 250 880 B per RoI (4 corner rows read + 1 row written per bin)
    nsets = int(min(24, max(3, math.ceil(320e6 / (0.8 * alg)) + 1)))
    sets = []
    for i in range(nsets):
        maps = [torch.randn(B, S // s, S // s, 256, device=dev) for s in (4, 8, 16, 32)]
        boxes = torch.tensor(synth.rois(1 + i, B, R, S, S) / np.array([S, S, S, S], np.float32), device=dev)
        sets.append((maps, boxes, torch.empty(B, R, 7, 7, 256, device=dev)))
    for m, bx, o in sets:
        ops.roi_align_pyramid(m, bx, S * S, 7, out=o)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ROUNDS):
        for m, bx, o in sets:
            ops.roi_align_pyramid(m, bx, S * S, 7, out=o)
    e1.record()
    torch.cuda.synchronize()
    n = ROUNDS * nsets
    us = e0.elapsed_time(e1) / n * 1e3
    print("B=%2d R=%3d sets=%2d launches=%3d: %7.1f us per launch incl. launch gaps  %6.2f TB/s algorithmic (%.0f%% of 8 TB/s)"
          % (B, R, nsets, n + nsets, us, alg / us / 1e6, 100 * alg / us / 1e6 / 8))
    del sets
    torch.cuda.empty_cache()
