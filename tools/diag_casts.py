"""Diagnostics: which fp32 -> bf16 casts does one joint train step make (sizes, callers), and how long does dc_cast_f32_bf16 take alone."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_fullsize as F
from image_captioning_amd import ops

for n in (1 << 20, 4_600_000, 16_777_216, 67_108_864):
    x = torch.randn(n, device="cuda")
    o = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        ops.to_bf16(x, out=o)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.to_bf16(x, out=o)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("to_bf16 %10d elements: %7.1f us  %.2f TB/s" % (n, us, 6.0 * n / us / 1e6), flush=True)

model, cfg, inputs = F._full_size_joint("bf16")
inputs[0] = torch.tensor(inputs[0], device="cuda")
model.use_step_graph = False
for _ in range(2):
    model.train_on_batch(inputs)
calls = collections.Counter()
orig = ops.to_bf16


def spy(x, out=None, pad_cols=None):
    st = traceback.extract_stack(limit=4)
    where = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in st[:-1][::-1][:2])
    calls[(tuple(x.shape), where)] += 1
    return orig(x, out=out, pad_cols=pad_cols)
ops.to_bf16 = spy
model.train_on_batch(inputs)
ops.to_bf16 = orig
tot = 0
for (shape, where), c in sorted(calls.items(), key=lambda kv: -kv[1] * int(torch.tensor(kv[0][0]).prod())):
    n = 1
    for d in shape:
        n *= d
    tot += n * c
    print("%3d x %-28s %10d elements  %s" % (c, shape, n, where))
print("total elements cast per step (python-level calls only): %.1f M" % (tot / 1e6))
plan = model.plan()
pc = [(tuple(op[1].shape), 1) for op in plan._ops if op[0] == "cast"]
print("casts inside the encoder plan's graph: %d, %.1f M elements" % (len(pc), sum(torch.tensor(s).prod().item() for s, _ in pc) / 1e6))
for s, _ in pc:
    print("   plan cast", s)
