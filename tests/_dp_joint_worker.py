"""Worker of tests/test_gpu_multirank.py (joint model): one rank of a 2-rank data-parallel run of the REAL joint model
(dense_img_cap/dense_model.py) under ParallelModel, one image per rank and step."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

S, V, T, BLOCKS = 128, 24, 5, 1


def build():
    import test_gpu_models as TM
    model, cfg, Wt = TM.make_joint(S, V, T, BLOCKS)
    model.compile(1e-3)
    return model


def global_inputs(world):
    import test_gpu_models as TM
    per = [TM.joint_inputs(S, V, T, seed=8 + r) for r in range(world)]
    for r, p in enumerate(per):                         # different images too
        from image_captioning_amd import synth
        p[0] = synth.images(7 + r, 1, S, S)
    return [np.concatenate([p[i] for p in per], axis=0) for i in range(6)], per


def main():
    out_dir, steps = sys.argv[1], int(sys.argv[2])
    from image_captioning_amd.parallel_model import ParallelModel, init_process_group_from_env
    import torch.distributed as dist
    rank, world, _ = init_process_group_from_env()
    model = build()
    pm = ParallelModel(model, world)
    assert model.grad_sync.dtype == os.environ.get("DCAP_GRAD_DTYPE", "f32")        # the wire format the test asked for
    inputs, _ = global_inputs(world)
    mode = sys.argv[3] if len(sys.argv) > 3 else "serial"
    tag = "" if mode == "serial" else "_" + mode
    if mode == "pipeline":
        # pipeline.JointTrainPipeline over the ParallelModel: GLOBAL batches in (tf.split inside step()), the towers' mean losses one call late
        from image_captioning_amd.pipeline import JointTrainPipeline
        model.use_step_graph = False
        pipe = JointTrainPipeline(pm)
        raw = [pipe.step(inputs) for _ in range(steps)][1:] + [pipe.flush()]
        losses = [model._losses_to_api(l.cpu().numpy()) for l in raw]
    else:
        if mode == "serial_eager":
            model.use_step_graph = False
        losses = [pm.train_on_batch(inputs) for _ in range(steps)]
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "joint%s_rank%d.npz" % (tag, rank)), flat=model.store.flat.cpu().numpy(), losses=np.array(losses))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
