#!/usr/bin/env python
"""Time of the Winograd 3x3 kernel against Cin at a fixed map (2 x 128 x 128, Cout 256: 1024 work items = 4 per CU): the fit
t = items * (a + b * pairs) separates the per-item cost (prologue, output transform) from the per-32-channel cost."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops  # noqa: E402

dev = torch.device("cuda")
B3 = "--b3" in sys.argv            # the split-bf16 products (w_wino_b3): MFMA floor per 32-channel pair 1.28 us instead of 3.41
FLOOR = 1.28 if B3 else 3.41
rows = []
for Cin in (32, 64, 128, 256, 512, 1024):
    H, Cout, B = 128, 256, 2
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, 9 * Cin, device=dev) / np.sqrt(9 * Cin)
    u = ops.winograd_pack_b3(w, Cin, Cout) if B3 else ops.winograd_pack(w, Cin, Cout)
    y = torch.empty(B, H, H, Cout, device=dev)
    run = (lambda: ops.conv2d(x, w, 3, 3, 1, 1, 1, H, H, None, None, None, 0, False, out=y, w_wino_b3=u)) if B3 else \
          (lambda: ops.conv2d(x, w, 3, 3, 1, 1, 1, H, H, None, None, None, 0, False, out=y, w_wino=u))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 30
    rows.append((Cin // 32, us))
    print("Cin %4d  pairs %2d  %7.1f us   per item %.2f us   MFMA floor per item %.2f us" % (Cin, Cin // 32, us, us / 4, FLOOR * Cin / 32))
A = np.array([[1.0, p] for p, _ in rows[1:]])
t = np.array([us / 4 for _, us in rows[1:]])
(a, b), *_ = np.linalg.lstsq(A, t, rcond=None)
print("fit per item: a = %.2f us, b = %.2f us per pair (MFMA floor %.2f)" % (a, b, FLOOR))
