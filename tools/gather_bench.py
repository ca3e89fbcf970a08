"""Embedding / row gathers of the decoder: achieved bytes per second (HIP events, random ids).
  * dc_gather_rows_f32 (the concat operands: RoI-feature rows and word-LSTM state rows into [N, 2048]) at the benchmark's size
    (960 rows x 1024 floats) and at a size that is no longer launch-bound (131072 rows);
  * the embedding lookup, which is the A-operand loader of the x-projection GEMM (gather = token ids): table [V, 300] -> [T*B, 4096]
    at the benchmark's size (V = 10 000, 960 tokens) -- a GEMM, priced by its MFMA work, with the gathered bytes beside it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops


def timed(fn, reps=50):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    for rows, src_rows in ((960, 64), (960, 960), (131072, 131072)):
        src = torch.randn((src_rows, 1024), device=dev, generator=g)
        idx = torch.randint(0, src_rows, (rows,), device=dev, generator=g, dtype=torch.int32)
        out = torch.empty((rows, 2048), device=dev)
        us = timed(lambda: ops.gather_rows(src, idx, out[:, :1024]))
        b = rows * 1024 * 4 * 2.0
        print("gather_rows  %6d rows x 1024 f32 from a %6d-row table: %7.1f us  %6.2f TB/s (read + write, %.0f%% of 8 TB/s)"
              % (rows, src_rows, us, b / us / 1e6, 100 * b / us / 1e6 / 8), flush=True)
    V, E, N, U4 = 10000, 300, 960, 4096
    table = torch.randn((V, E), device=dev, generator=g)
    W = torch.randn((E, U4), device=dev, generator=g)
    ids = torch.randint(0, V, (N,), device=dev, generator=g, dtype=torch.int32)
    out = torch.empty((N, U4), device=dev)
    us = timed(lambda: ops.gemm(table, W, gather=ids, out=out))
    print("embedding gather + x-projection GEMM (ids -> [%d, %d], K = %d): %7.1f us  %5.1f TFLOP/s; gathered %.2f MB + written %.2f MB = %5.2f TB/s"
          % (N, U4, E, us, 2.0 * N * E * U4 / us / 1e6, N * E * 4 / 1e6, N * U4 * 4 / 1e6, (N * E * 4 + N * U4 * 4 + E * U4 * 4) / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
