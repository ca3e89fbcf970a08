"""Run-to-run determinism of the two benchmark steps at the benchmark's sizes:
  headline   two fresh copies of bench.py's configs[2]/[3] model (same seeds), N train steps each through the two-stream pipeline:
             the last step's loss and every trainable weight afterwards must be equal bit for bit;
  joint      configs[4] at 1024 x 1024: N forward_backward calls on the same weights and batch (no optimizer step): the four losses
             and the whole gradient bucket equal to the first call's, bit for bit.
Usage: python tools/determinism_steps.py [headline-steps] [joint-calls]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench


def args_for(argv):
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        a = bench.parse()
    finally:
        sys.argv = old
    if a.vocab is None:
        a.vocab = 50000 if a.config == "joint" else 10000
    return a


def headline(steps):
    dev = torch.device("cuda:0")
    a = args_for([])
    runs = []
    for rep in range(2):
        e = bench.E2E(a, dev, 0, 1, a.images_per_gpu)
        for _ in range(steps):                             # as the bench drives it: nothing read back between steps
            e.step()
        last = e.flush()
        torch.cuda.synchronize()
        runs.append(([], None if last is None else float(torch.as_tensor(last).float().reshape(-1)[0]), e.inner.store.flat.clone()))
        del e
        torch.cuda.empty_cache()
    same_loss = runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1]
    diff = int((runs[0][2] != runs[1][2]).sum())
    print("headline: %d steps twice: losses %s, %d of %d weights differ (last loss %r / %r)"
          % (steps, "identical" if same_loss else "DIFFER", diff, runs[0][2].numel(), runs[0][1], runs[1][1]), flush=True)
    return same_loss and diff == 0


def joint(calls):
    dev = torch.device("cuda:0")
    a = args_for(["--config", "joint"])
    model, inner, inputs, cfg = bench.build_joint(a, dev)
    first, bad = None, 0
    for call in range(calls):
        losses = inner.forward_backward(inputs, shuffle=None)
        got = (losses.clone() if torch.is_tensor(losses) else torch.as_tensor(np.asarray(losses)), inner.store.flat_grad.clone())
        if first is None:
            first = got
            continue
        if not torch.equal(got[0], first[0]) or not torch.equal(got[1], first[1]):
            bad += 1
            d = got[1] != first[1]
            print("  call %d: losses %s, %d gradient entries differ (first at %d)"
                  % (call, "equal" if torch.equal(got[0], first[0]) else "differ", int(d.sum()), int(torch.nonzero(d)[0]) if bool(d.any()) else -1), flush=True)
    print("joint: %d of %d calls differ from the first (%d gradient entries each)" % (bad, calls - 1, first[1].numel()), flush=True)
    return bad == 0


if __name__ == "__main__":
    n_head = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n_joint = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    ok = True
    if n_head:
        ok = headline(n_head) and ok
    if n_joint:
        ok = joint(n_joint) and ok
    sys.exit(0 if ok else 1)
