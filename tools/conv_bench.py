#!/usr/bin/env python
"""Micro-benchmark of dc_conv2d_nhwc_f32 on the distinct ResNet-101/FPN layer shapes of the benchmark
(2 images @ 1024x1024).  Prints one line per shape: time, TFLOP/s, GB/s of
compulsory traffic.  Usage: python tools/conv_bench.py [--reps 20] [--filter substr]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops  # noqa: E402

# name, H(in), Cin, Cout, k, stride, res_mode, relu
SHAPES = [
    ("res2_2a 1x1 64>64", 256, 64, 64, 1, 1, 0, 1),
    ("res2_2b 3x3 64", 256, 64, 64, 3, 1, 0, 1),
    ("res2_2c 1x1 64>256 +res", 256, 64, 256, 1, 1, 1, 1),
    ("res2_2a 1x1 256>64", 256, 256, 64, 1, 1, 0, 1),
    ("res3_2b 3x3 128", 128, 128, 128, 3, 1, 0, 1),
    ("res3_2c 1x1 128>512 +res", 128, 128, 512, 1, 1, 1, 1),
    ("res3_2a 1x1 512>128", 128, 512, 128, 1, 1, 0, 1),
    ("res4_2b 3x3 256", 64, 256, 256, 3, 1, 0, 1),
    ("res4_2c 1x1 256>1024 +res", 64, 256, 1024, 1, 1, 1, 1),
    ("res4_2a 1x1 1024>256", 64, 1024, 256, 1, 1, 0, 1),
    ("res5_2b 3x3 512", 32, 512, 512, 3, 1, 0, 1),
    ("res5_2c 1x1 512>2048 +res", 32, 512, 2048, 1, 1, 1, 1),
    ("res5_2a 1x1 2048>512", 32, 2048, 512, 1, 1, 0, 1),
    ("res4a_1 1x1s2 512>1024", 128, 512, 1024, 1, 2, 0, 0),
    ("res2a_1 1x1 64>256", 256, 64, 256, 1, 1, 0, 0),
    ("res3a_1 1x1s2 256>512", 256, 256, 512, 1, 2, 0, 0),
    ("res3a_2a 1x1s2 256>128", 256, 256, 128, 1, 2, 0, 1),
    ("res4a_2a 1x1s2 512>256", 128, 512, 256, 1, 2, 0, 1),
    ("res5a_1 1x1s2 1024>2048", 64, 1024, 2048, 1, 2, 0, 0),
    ("res5a_2a 1x1s2 1024>512", 64, 1024, 512, 1, 2, 0, 1),
    ("fpn_c2p2 1x1 256>256 +up", 256, 256, 256, 1, 1, 2, 0),
    ("fpn_c3p3 1x1 512>256 +up", 128, 512, 256, 1, 1, 2, 0),
    ("fpn_c4p4 1x1 1024>256 +up", 64, 1024, 256, 1, 1, 2, 0),
    ("fpn_c5p5 1x1 2048>256", 32, 2048, 256, 1, 1, 0, 0),
    ("fpn_p2 3x3 256", 256, 256, 256, 3, 1, 0, 0),
    ("fpn_p3 3x3 256", 128, 256, 256, 3, 1, 0, 0),
    ("fpn_p4 3x3 256", 64, 256, 256, 3, 1, 0, 0),
    ("fpn_p5 3x3 256", 32, 256, 256, 3, 1, 0, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--filter", default="")
    ap.add_argument("--math", type=int, default=0, help="0 = fp32 MFMA, 1 = split-bf16 (3 pieces, 6 products), 2 = 2 pieces, 3 products")
    ap.add_argument("--winograd", action="store_true", help="math 0: 3x3 layers in the Winograd F(2x2,3x3) form (pre-transformed weights)")
    ap.add_argument("--b3", action="store_true", help="with --winograd: the products on the bf16 matrix pipe in split arithmetic (w_wino_b3)")
    args = ap.parse_args()
    dev = torch.device("cuda")
    B = args.batch
    for name, H, Cin, Cout, k, stride, res_mode, relu in SHAPES:
        if args.filter not in name:
            continue
        Ho = H // stride
        x = torch.randn(B, H, H, Cin, device=dev)
        w = torch.randn(Cout, k * k * Cin, device=dev) / np.sqrt(k * k * Cin)
        sc = torch.rand(Cout, device=dev) + 0.5
        sh = torch.randn(Cout, device=dev)
        res = None
        if res_mode == 1:
            res = torch.randn(B, Ho, Ho, Cout, device=dev)
        elif res_mode == 2:
            res = torch.randn(B, Ho // 2, Ho // 2, Cout, device=dev)
        y = torch.empty(B, Ho, Ho, Cout, device=dev)
        pad = (k - 1) // 2
        wino = ops.winograd_pack(w, Cin, Cout) if (args.winograd and k == 3 and stride == 1 and args.math == 0) else None
        wb3 = ops.winograd_pack_b3(w, Cin, Cout) if (wino is not None and args.b3) else None
        run = lambda: ops.conv2d(x, w, k, k, stride, pad, pad, Ho, Ho, sc, sh, res, res_mode, bool(relu), out=y, math=args.math,
                                 w_wino=wino, w_wino_b3=wb3)
        kname = ops.conv2d_kernel_name(x, w, k, k, stride, pad, pad, Ho, Ho, sc, sh, res, res_mode, bool(relu), math=args.math,
                                       w_wino=wino, w_wino_b3=wb3)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / args.reps
        fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin
        by = 4.0 * (x.numel() / (stride * stride if k == 1 else 1) + y.numel() + (res.numel() if res is not None else 0) + w.numel())
        gain = 2.25
        tag = "  [%s: %.1f TF/s executed]" % (kname, fl / gain / us / 1e6) if wino is not None else ""
        print("%-28s %8.1f us  %6.1f TF/s  %6.0f GB/s%s" % (name, us, fl / us / 1e6, by / us / 1e3, tag))


if __name__ == "__main__":
    main()
