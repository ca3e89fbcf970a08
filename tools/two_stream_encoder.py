"""Experiment: one batch-2 encoder plan vs two batch-1 plans on two HIP streams."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import synth
from image_captioning_amd.encoder import EncoderPlan

dev = torch.device("cuda")
W = synth.encoder_weights(0, 22)
p2 = EncoderPlan(W, 2, 1024, 1024, dev)
pa = EncoderPlan(W, 1, 1024, 1024, dev)
pb = EncoderPlan(W, 1, 1024, 1024, dev)
img = torch.tensor(synth.images(0, 2), device=dev)
p2.images.copy_(img); pa.images.copy_(img[:1]); pb.images.copy_(img[1:])
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(3):
    p2.forward(None)
    with torch.cuda.stream(sa): pa.forward(None)
    with torch.cuda.stream(sb): pb.forward(None)
torch.cuda.synchronize()
def timeit(f, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def two():
    with torch.cuda.stream(sa): pa.forward(None)
    with torch.cuda.stream(sb): pb.forward(None)
def seq():
    pa.forward(None); pb.forward(None)
print("batch-2 plan          : %.3f ms" % timeit(lambda: p2.forward(None)))
print("2 x batch-1, 2 streams: %.3f ms" % timeit(two))
print("2 x batch-1, 1 stream : %.3f ms" % timeit(seq))
