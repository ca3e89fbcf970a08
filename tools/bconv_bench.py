"""dc_conv2d_bf16 on the joint model's layer shapes (1 image @ 1024x1024): time, TFLOP/s against the 2.5 PF bf16 dense peak.
Usage: python tools/bconv_bench.py [--reps 30]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops  # noqa: E402

# name, H(in), Cin, Cout, k, stride
SHAPES = [
    ("res2_2b 3x3 64", 256, 64, 64, 3, 1), ("res2_2c 1x1 64>256", 256, 64, 256, 1, 1), ("res2_2a 1x1 256>64", 256, 256, 64, 1, 1),
    ("res3_2b 3x3 128", 128, 128, 128, 3, 1), ("res3_2c 1x1 128>512", 128, 128, 512, 1, 1),
    ("res4_2b 3x3 256", 64, 256, 256, 3, 1), ("res4_2c 1x1 256>1024", 64, 256, 1024, 1, 1), ("res4_2a 1x1 1024>256", 64, 1024, 256, 1, 1),
    ("res5_2b 3x3 512", 32, 512, 512, 3, 1), ("fpn_p2 3x3 256", 256, 256, 256, 3, 1), ("fpn_p3 3x3 256", 128, 256, 256, 3, 1),
    ("rpn_shared P2 3x3 256>512", 256, 256, 512, 3, 1), ("rpn dgrad P2 3x3 512>256", 256, 512, 256, 3, 1),
    ("res5_2a 1x1 2048>512", 32, 2048, 512, 1, 1), ("res5_2c 1x1 512>2048", 32, 512, 2048, 1, 1), ("res3_2a 1x1 512>128", 128, 512, 128, 1, 1),
    ("rpn_shared P4 3x3 256>512", 64, 256, 512, 3, 1), ("rpn dgrad P4 3x3 512>256", 64, 512, 256, 3, 1), ("fpn_p4 3x3 256", 64, 256, 256, 3, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--tile", type=int, default=0, help="0 = the library's choice; 64 | 128 | 256 forces a block tile")
    a = ap.parse_args()
    dev = torch.device("cuda")
    for name, H, Cin, Cout, k, stride in SHAPES:
        x = torch.randn(a.batch, H, H, Cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(Cout, k * k * Cin, device=dev) / (k * k * Cin) ** 0.5).to(torch.bfloat16)
        Ho = H // stride
        pad = (k - 1) // 2
        yb = torch.empty(a.batch, Ho, Ho, Cout, device=dev, dtype=torch.bfloat16)
        yf = torch.empty(a.batch, Ho, Ho, Cout, device=dev)
        res = []
        for out, outb in ((None, yb), (yf, yb)):
            info = {}
            f = lambda: ops.conv2d_bf16(x, w, k, k, stride, pad, pad, Ho, Ho, relu=True, out=out, out_bf16=outb, want_f32=False, tile=a.tile, info=info)
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / a.reps)
        fl = 2.0 * a.batch * Ho * Ho * Cout * k * k * Cin
        print("%-28s tile %3d split %2d  bf16 out %7.1f us %6.1f TF/s   f32+bf16 out %7.1f us %6.1f TF/s"
              % (name, info["tile"], info["split_k"], res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6), flush=True)


if __name__ == "__main__":
    main()
