"""Micro-benchmark of dc_gemm_bf16 on the joint model's (configs[4]) GEMM shapes: TFLOP/s on random data, interleaved rounds.
Usage: python tools/bgemm_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from image_captioning_amd import ops

SHAPES = [  # name, M, N, K, a_trans, b_trans
    ("vocab fwd  NN  a1[3000,1024] x W[1024,50000]", 3000, 50000, 1024, 0, 0),
    ("vocab dX   NT  dl[3000,50000] x W^T", 3000, 1024, 50000, 0, 1),
    ("vocab dW   TN  a1^T x dl[3000,50000]", 1024, 50000, 3000, 1, 0),
    ("head fwd   NN  X[200,12544] x K1[12544,1024]", 200, 1024, 12544, 0, 0),
    ("head wgrad TN  X^T x dacc[200,1024]", 12544, 1024, 200, 1, 0),
    ("head dgrad NT  dacc x K1^T", 200, 12544, 1024, 0, 1),
    ("lstm proj  NN  h[3000,512] x W[512,2048]", 3000, 2048, 512, 0, 0),
    ("square 4096 NT", 4096, 4096, 4096, 0, 1),
    ("square 4096 NN", 4096, 4096, 4096, 0, 0),
    ("square 4096 TN", 4096, 4096, 4096, 1, 0),
]


def main():
    dev = torch.device("cuda")
    for name, M, N, K, ta, tb in SHAPES:
        a = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
        b = torch.randn((N, K) if tb else (K, N), device=dev).to(torch.bfloat16)
        out = torch.empty((M, N), device=dev)
        for _ in range(3):
            ops.gemm_bf16(a, b, out=out, a_trans=bool(ta), b_trans=bool(tb))
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm_bf16(a, b, out=out, a_trans=bool(ta), b_trans=bool(tb))
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        print("%-50s %8.1f us  %7.1f TFLOP/s" % (name, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12), flush=True)


if __name__ == "__main__":
    main()
