"""Run-to-run determinism of dc_gemm_bf16 (round 6: a one-in-fifteen-calls flicker of 16 entries was seen in dc_vocab_ce's 128-tile passes):
the same product repeated, every output compared bit for bit with the first call's, per layout (NN reads B through the K-major LDS image
and ds_read_b64_tr_b16, NT through the K-contiguous image) and per tile.  Usage: python tools/determinism_gemm_bf16.py [reps]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from image_captioning_amd import ops

BF = torch.bfloat16


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    for label, M, N, K in (("128-tile", 130, 50000, 1024), ("256-tile", 3000, 50000, 1024), ("128-tile small", 200, 2048, 512)):
        A = torch.randn((M, K), device=dev, generator=g).to(BF)
        Bn = torch.randn((K, N), device=dev, generator=g).to(BF)          # NN: B [K][N]
        Bt = Bn.t().contiguous()                                          # NT: B [N][K]
        for lay, fn in (("NN", lambda: ops.gemm_bf16(A, Bn)), ("NT", lambda: ops.gemm_bf16(A, Bt, b_trans=True))):
            ref = fn().clone()
            bad, worst, pattern = 0, 0.0, None
            for it in range(reps):
                out = fn()
                if not torch.equal(out, ref):
                    bad += 1
                    d = (out - ref).abs()
                    worst = max(worst, float(d.max()))
                    if pattern is None:
                        nz = torch.nonzero(out != ref).cpu().numpy()
                        pattern = (sorted(set(nz[:, 0].tolist()))[:6], sorted(set(nz[:, 1].tolist()))[:20], len(nz))
            torch.cuda.synchronize()
            print("%-15s %s  M=%d N=%d K=%d: %d of %d calls differ from the first (max |diff| %.3e)%s"
                  % (label, lay, M, N, K, bad, reps, worst, "" if pattern is None else "  first pattern: rows %s cols %s (%d entries)" % pattern), flush=True)


if __name__ == "__main__":
    main()
