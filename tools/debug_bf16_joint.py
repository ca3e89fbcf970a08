"""Diagnostic: per-tensor gradient error of the small joint model vs the float64 oracle, fp32 and bf16 compute."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import test_gpu_models as T
from oracle import np_models as M


def run(dt, rois, Tn, variant=""):
    S, V, blocks = 128, 24, 1
    model, cfg, Wt = T.make_joint(S, V, Tn, blocks, rois=rois, compute_dtype=dt)
    cm = model.caption_model
    if variant == "nofuse":
        import image_captioning_amd.ops as ops
        ops_vs = ops.vocab_ce_supported
        ops.vocab_ce_supported = lambda X, W: False
    inputs = T.joint_inputs(S, V, Tn)
    losses = model._loss_list(model.forward_backward(inputs, shuffle=None))
    tg = model.last_targets
    want, G, aux = T.joint_oracle(Wt, cfg, inputs, (tg['rois'], tg['caps']), blocks)
    got = T.joint_grads_as_reference(model)
    worst = {k: (T.rel_err(got[k], G[k]), float(np.linalg.norm(np.asarray(got[k], np.float64) - G[k]) / max(1e-30, np.linalg.norm(G[k])))) for k in M.joint_trainable(Wt)}
    print("==", dt, variant, "rois", rois, "T", Tn, "npos", tg['npos'], {k: (round(losses[k], 5), round(want[k], 5)) for k in ('imgcap_loss', 'loss')})
    for k, v in sorted(worst.items(), key=lambda kv: -kv[1][1])[:12]:
        print("   %-36s max-rel %.4f   l2-rel %.4f" % (k, v[0], v[1]))
    if variant == "nofuse":
        ops.vocab_ce_supported = ops_vs


if __name__ == "__main__":
    run("bf16", 16, 8)
    run("bf16", 64, 8)
