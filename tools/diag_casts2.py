"""Diagnostics: does the joint step get faster by the time the kernel trace charges to cast_bf16_kernel when the Python-level casts are
skipped?  (Results are wrong with the casts skipped -- stale bf16 copies -- only the step time matters.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from image_captioning_amd import ops


def timed(model, inputs, n=15):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model.train_on_batch(inputs)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


sys.argv = [sys.argv[0], "--config", "joint"]
args = bench.parse()
_, model, inputs, cfg = bench.build_joint(args, torch.device("cuda"))
assert model.conv_math_name == "bf16"
model.use_step_graph = False
for _ in range(3):
    model.train_on_batch(inputs)
print("eager, casts on : %.3f ms/step" % timed(model, inputs), flush=True)
orig = ops.to_bf16


def skip(x, out=None, pad_cols=None):
    if out is not None and pad_cols is None:
        return out
    return orig(x, out=out, pad_cols=pad_cols)
ops.to_bf16 = skip
print("eager, casts off: %.3f ms/step" % timed(model, inputs), flush=True)
ops.to_bf16 = orig
print("eager, casts on : %.3f ms/step" % timed(model, inputs), flush=True)
