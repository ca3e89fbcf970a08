"""dc_roi_align_pyramid_f32 bandwidth at the benchmark size (64 RoIs) and at 16 images x 32 RoIs."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops, synth

dev = torch.device("cuda")
for B, R in ((2, 32), (16, 32), (16, 128)):
    S = 1024
    maps = [torch.randn(B, S // s, S // s, 256, device=dev) for s in (4, 8, 16, 32)]
    rois = synth.rois(1, B, R, S, S)
    boxes = torch.tensor(rois / np.array([S, S, S, S], np.float32), device=dev)
    out = torch.empty(B, R, 7, 7, 256, device=dev)
    for _ in range(3):
        ops.roi_align_pyramid(maps, boxes, S * S, 7, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.roi_align_pyramid(maps, boxes, S * S, 7, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    alg = B * R * 250880.0          # <= 250 880 B per RoI (4 corner rows read + 1 row written per bin)
    print("B=%2d R=%3d: %7.1f us  %6.2f TB/s algorithmic (%.0f%% of 8 TB/s)" % (B, R, us, alg / us / 1e6, 100 * alg / us / 1e6 / 8))
