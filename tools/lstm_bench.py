"""Times dc_lstm_seq_fwd_f32 / dc_lstm_seq_bwd_f32 alone (whole sequences, per-timestep figures derived).
Usage: python tools/lstm_bench.py [--shapes 64x15x512,32x15x512,200x15x512] [--reps 50]
DCAP_LSTM_BWD=steps selects the three-launch backward timestep for comparison."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="64x15x512,32x15x512,200x15x512,64x15x1024")
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--dropout", action="store_true", help="with Keras recurrent_dropout masks (rate 0.2) on the recurrence")
    a = ap.parse_args()
    from image_captioning_amd import ops
    dev = torch.device("cuda")
    for shp in a.shapes.split(","):
        B, T, U = (int(v) for v in shp.split("x"))
        g = torch.Generator(device=dev).manual_seed(0)
        z0 = torch.randn(T * B, 4 * U, device=dev, generator=g)
        Ur = torch.randn(U, 4 * U, device=dev, generator=g) / U ** 0.5
        dh = torch.randn(T * B, U, device=dev, generator=g)
        z = z0.clone()
        rm = ops.dropout_mask(torch.empty(4, B, U, device=dev), 0.2, 7, 1) if a.dropout else None
        h, c = ops.lstm_seq_fwd(z, Ur, None, B, T, rec_masks=rm)
        dz = torch.empty_like(z)
        dU = torch.empty_like(Ur)

        def timed(fn):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / a.reps

        zz = z0.clone()
        t_f = timed(lambda: ops.lstm_seq_fwd(zz, Ur, None, B, T, h_seq=h, c_seq=c, rec_masks=rm))
        t_b = timed(lambda: ops.lstm_seq_bwd(z, Ur, None, h, c, B, T, dh_seq=dh, dz=dz, dU=dU, rec_masks=rm))
        print("B=%d T=%d U=%d  fwd %.1f us (%.1f/step)  bwd %.1f us (%.1f/step incl. dU)" % (B, T, U, t_f, t_f / T, t_b, t_b / T), flush=True)


if __name__ == "__main__":
    main()
