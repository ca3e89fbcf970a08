"""Diagnostic: which Python lines of one joint train step cause host<->device copies (torch.tensor(..., device=), .cpu(), .to(),
.item(), copy_ from host).  Usage: python tools/count_copies.py   (GPU box)"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
counts = collections.Counter()
sizes = collections.Counter()
ON = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if fr.filename.startswith(ROOT) and "count_copies" not in fr.filename:
            return "%s:%d" % (os.path.relpath(fr.filename, ROOT), fr.lineno)
    return "?"


def wrap_fn(mod, name, label, pred=lambda a, k: True):
    orig = getattr(mod, name)

    def f(*a, **k):
        if ON[0] and pred(a, k):
            counts[(label, site())] += 1
            if isinstance(a[0], torch.Tensor):
                sizes[(label, site())] += a[0].numel() * a[0].element_size()
        return orig(*a, **k)
    setattr(mod, name, f)


def dev_kw(a, k):
    d = k.get("device")
    return d is not None and "cuda" in str(d)


for n in ("tensor", "as_tensor", "zeros", "empty", "full", "ones", "arange"):
    wrap_fn(torch, n, "torch." + n, dev_kw if n in ("tensor", "as_tensor") else lambda a, k: False)
wrap_fn(torch.Tensor, "cpu", "Tensor.cpu", lambda a, k: a[0].is_cuda)
wrap_fn(torch.Tensor, "item", "Tensor.item", lambda a, k: a[0].is_cuda)
wrap_fn(torch.Tensor, "tolist", "Tensor.tolist", lambda a, k: a[0].is_cuda)
wrap_fn(torch.Tensor, "to", "Tensor.to", lambda a, k: True)
wrap_fn(torch.Tensor, "copy_", "Tensor.copy_", lambda a, k: a[0].is_cuda != a[1].is_cuda)
wrap_fn(torch.Tensor, "copy_", "Tensor.copy_ d2d", lambda a, k: a[0].is_cuda and a[1].is_cuda)
wrap_fn(torch.Tensor, "clone", "Tensor.clone", lambda a, k: a[0].is_cuda)
wrap_fn(torch.Tensor, "zero_", "Tensor.zero_", lambda a, k: a[0].is_cuda)
wrap_fn(torch.Tensor, "fill_", "Tensor.fill_", lambda a, k: a[0].is_cuda)
wrap_fn(torch.Tensor, "__float__", "Tensor.__float__", lambda a, k: a[0].is_cuda)
wrap_fn(torch.Tensor, "__int__", "Tensor.__int__", lambda a, k: a[0].is_cuda)

import bench  # noqa: E402


def main():
    dev = torch.device("cuda")
    sys.argv = [sys.argv[0], "--config", "joint"]
    args = bench.parse()
    args.vocab = args.vocab or 50000
    _, model, inputs, _cfg = bench.build_joint(args, dev)
    for _ in range(2):
        model.train_on_batch(inputs)
    torch.cuda.synchronize()
    ON[0] = True
    model.train_on_batch(inputs)
    torch.cuda.synchronize()
    ON[0] = False
    for (label, where), n in sorted(counts.items(), key=lambda kv: -kv[1]):
        print("%3d  %-18s %-60s %10.1f KB" % (n, label, where, sizes[(label, where)] / 1024))
    print("total", sum(counts.values()))


if __name__ == "__main__":
    main()
