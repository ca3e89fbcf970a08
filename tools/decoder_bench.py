"""BASELINE configs[1] alone: v2-inject decoder train step on precomputed RoI features (64 samples as written, or --captions R for
the single-pass form of R 15-token captions).  Usage: python tools/decoder_bench.py [--captions 64]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--captions", type=int, default=0)
    ap.add_argument("--steps", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda")
    if not a.captions:
        print(bench.gpu_configs1(dev, a.steps))
        return
    from image_captioning_amd import synth
    from image_captioning_amd.text_generation_model_v2 import DenseCapConfig, build_model, Adam, SampleTables
    V, T, R = 10000, 15, a.captions
    cfg = DenseCapConfig(V, synth.embedding_matrix(3, V))
    cfg.PADDING_SIZE = T
    dec = build_model((7, 7, 256), (T,), cfg, 256, inject=True, device=dev, seed=0)
    dec.compile(optimizer=Adam(amsgrad=True), loss="categorical_crossentropy")
    feat = torch.randn(R, 7, 7, 256, device=dev)
    tb = SampleTables.from_captions(synth.captions_v2(5, R, T, V, full=True), dev)
    for _ in range(3):
        dec.train_step(feat, tb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        dec.train_step(feat, tb)
    torch.cuda.synchronize()
    print("single-pass %d captions: %.3f ms/step" % (R, 1e3 * (time.perf_counter() - t0) / a.steps))


if __name__ == "__main__":
    main()
