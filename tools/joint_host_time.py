"""How far ahead of the GPU is the host in the pipelined joint step?  Enqueue time of K steps (no synchronisation) against their wall time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from image_captioning_amd.pipeline import JointTrainPipeline

sys.argv = [sys.argv[0], "--config", "joint"] + sys.argv[1:]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model, inner, inputs, cfg = bench.build_joint(args, dev)
for _ in range(8):
    inner.train_on_batch(inputs)
K = 20
for piped in (False, True):
    if piped:
        pipe = JointTrainPipeline(inner)
        for _ in range(6):
            pipe.step(inputs)
        pipe.flush()
    else:
        inner.use_step_graph = False
        for _ in range(3):
            inner.train_on_batch(inputs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        if piped:
            pipe.step(inputs)
        else:
            inner.train_on_batch_device(inputs)
    t1 = time.perf_counter()
    if piped:
        pipe.flush()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host enqueue %.3f ms per step, wall %.3f ms per step" % ("pipelined" if piped else "serial eager", 1e3 * (t1 - t0) / K, 1e3 * (t2 - t0) / K))
