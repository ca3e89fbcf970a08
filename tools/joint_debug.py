"""Debug aid: per-weight gradient error of the joint model against the oracle, and a short loss trajectory."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_gpu_models as TM
from oracle import np_models as M

S, V, T, blocks = 128, 24, 5, 1
model, cfg, Wt = TM.make_joint(S, V, T, blocks)
inputs = TM.joint_inputs(S, V, T)
losses = model._loss_list(model.forward_backward(inputs, shuffle=None))
tg = model.last_targets
print("npos", tg['npos'], "nneg", tg['nneg'])
want, G, aux = TM.joint_oracle(Wt, cfg, inputs, (tg['rois'], tg['caps']), blocks)
print({k: (losses[k], want[k]) for k in want})
got = TM.joint_grads_as_reference(model)
for k in M.joint_trainable(Wt):
    print("%-40s relerr %.3e   |want|max %.3e" % (k, TM.rel_err(got[k], G[k]), np.abs(G[k]).max()))
for lr in (1e-4, 3e-5):
    model, cfg, Wt = TM.make_joint(S, V, T, blocks)
    model.compile(lr)
    for i in range(13):
        print(lr, i, model.train_on_batch(inputs), model.last_losses['reg_loss'])
