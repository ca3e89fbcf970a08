#!/usr/bin/env python
"""dc_pw_chain_f32 alone on the chip: a bottleneck's `2c` (+ shortcut + ReLU) and the next block's `2a` in one launch, at the
benchmark's shapes (2 images @ 1024 x 1024), split-bf16 and fp32 products, against the two separate launches.  Time = a captured
hipGraph of one launch per rotating buffer set / number of sets.  Usage: python tools/chain_bench.py [--sets 6]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops  # noqa: E402

# name, pixels, K1, N1, N2
SHAPES = [("stage 4 (64x64 x 2)", 8192, 256, 1024, 256), ("stage 3 (128x128 x 2)", 32768, 128, 512, 128), ("stage 2 (256x256 x 2)", 131072, 64, 256, 64)]


def graph_time(fns, reps=10):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.no_gc_during_capture(), torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / len(fns)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", type=int, default=6)
    ap.add_argument("--filter", default="")
    args = ap.parse_args()
    dev = torch.device("cuda")
    for name, M, K1, N1, N2 in SHAPES:
        if args.filter not in name or not ops.pw_chain_supported(K1, N1, N2):
            continue
        w1 = torch.randn(N1, K1, device=dev) / np.sqrt(K1)
        w2 = torch.randn(N2, N1, device=dev) / np.sqrt(N1)
        h1, h2 = torch.randn(N1, device=dev), torch.randn(N2, device=dev)
        s1, s2 = torch.rand(N1, device=dev) + 0.5, torch.rand(N2, device=dev) + 0.5
        sets = [(torch.randn(M, K1, device=dev), torch.randn(M, N1, device=dev), torch.empty(M, N1, device=dev), torch.empty(M, N2, device=dev))
                for _ in range(args.sets)]
        gf = 2.0 * M * (K1 * N1 + N1 * N2) / 1e9
        line = "%-24s %.2f GF" % (name, gf)
        for tag, pack in (("b3", ops.pw_chain_pack_b3), ("f32", ops.pw_chain_pack)):
            if tag == "b3" and N1 == 256:                   # the stage-2 seam runs in the streaming form with fp32 products only
                continue
            w1f, w2f = pack(w1), pack(w2)
            fns = [lambda x=x, r=r, y=y, z=z: ops.pw_chain(x, w1f, h1, w2f, h2, scale1=s1, scale2=s2, residual=r, y=y, z=z) for x, r, y, z in sets]
            us = graph_time(fns)
            line += "   chain %s %6.1f us (%5.1f TF)" % (tag, us, gf / us * 1e3)
        fns = []
        for x, r, y, z in sets:
            def two(x=x, r=r, y=y, z=z):
                ops.conv2d(x.view(1, 1, M, K1), w1, 1, 1, 1, 0, 0, 1, M, s1, h1, r.view(1, 1, M, N1), 1, True, out=y.view(1, 1, M, N1))
                ops.conv2d(y.view(1, 1, M, N1), w2, 1, 1, 1, 0, 0, 1, M, s2, h2, None, 0, True, out=z.view(1, 1, M, N2))
            fns.append(two)
        us = graph_time(fns)
        line += "   two launches %6.1f us" % us
        print(line, flush=True)


if __name__ == "__main__":
    main()
