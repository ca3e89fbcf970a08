"""Data parallelism with the REAL model on real devices (-m gpu): two ranks (RCCL when the box has two GPUs, otherwise the
gloo backend with both ranks sharing the one GPU) wrap CaptionModelV2 in ParallelModel and train on a global batch; the
weights after two steps must equal a single-rank run on the concatenated batch (the reference's ParallelModel semantics,
parallel_model.py:58-102: tf.split the batch, mean over towers of the per-tower mean-loss gradients)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_parallel_model_two_ranks_match_single_rank_on_the_concatenated_batch(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, HERE)
    import _dp_worker as W
    V, T, B, steps, world = 1000, 6, 16, 2, 2
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DCAP_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dp_worker.py"), str(tmp_path), str(V), str(T), str(B), str(steps)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(world)]
    assert int(r[0]["seen"][0]) == world and str(r[0]["backend"][0]) == backend
    np.testing.assert_array_equal(r[0]["flat"], r[1]["flat"])            # replicas stay bit-identical
    np.testing.assert_allclose(r[0]["losses"], r[1]["losses"], rtol=0, atol=0)
    # single rank, whole batch, same initial weights (rank 0's seed)
    model = W.build(V, T, seed=0)
    feat, words, onehot = W.batch(V, T, B)
    losses = [model.train_on_batch([feat, words], onehot) for _ in range(steps)]
    single = model.store.flat.cpu().numpy()
    np.testing.assert_allclose(r[0]["losses"], losses, rtol=2e-5)        # mean over towers of tower means == batch mean (equal shards)
    scale = np.abs(single).max()
    assert np.abs(r[0]["flat"] - single).max() < 2e-5 * scale


def test_configs3_shaped_step_two_ranks_at_the_full_per_rank_size(tmp_path):
    """BASELINE configs[3]'s per-rank shard (2 images x 32 RoIs = 64 captions x 15 tokens, V = 10 000) on two ranks, the way
    bench.py steps it (every rank its own shard, per-layer-group gradient all-reduce, mean over towers in the AMSGrad kernel):
    the replicas stay bit-identical over the steps and equal ONE process training on the 128 captions of both shards."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, HERE)
    import _dp_worker as W
    V, T, R, steps, world = 10000, 15, 64, 3, 2
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DCAP_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dp_worker.py"), str(tmp_path), str(V), str(T), str(R), str(steps), "captions"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(world)]
    assert int(r[0]["seen"][0]) == world
    np.testing.assert_array_equal(r[0]["flat"], r[1]["flat"])            # replicas bit-identical after 3 steps at full size
    model = W.build(V, T, seed=0)
    shards = [W.caption_shard(V, T, R, k) for k in range(world)]
    feat = np.concatenate([s[0] for s in shards])
    caps = [c for s in shards for c in s[1]]
    losses = [float(model.train_on_captions(feat, caps).item()) for _ in range(steps)]
    np.testing.assert_allclose((r[0]["losses"] + r[1]["losses"]) / 2, losses, rtol=2e-5)       # equal shards: mean of tower means == batch mean
    single = model.store.flat.cpu().numpy()
    assert np.abs(r[0]["flat"] - single).max() < 2e-5 * np.abs(single).max()


@pytest.mark.parametrize("grad_dtype", ["f32", "bf16"])
def test_joint_model_two_ranks_average_the_tower_gradients(tmp_path, grad_dtype):
    """grad_dtype = 'bf16': the same step with the gradient exchange in bf16 buckets (DCAP_GRAD_DTYPE / GradAllReduce(dtype='bf16'),
    SURVEY section 5: configs[4]'s wire format): replicas still bit-identical; the weights differ from the fp32 exchange's by the bf16
    rounding of the summed gradient only (a small fraction of one AMSGrad step).
    The joint model (configs[4]'s model at a small size) under ParallelModel, one image per rank: after a step every replica
    holds the same weights, and they are the weights a single process gets from the MEAN of the two images' gradient buckets
    (the reference's mean-of-tower-means, parallel_model.py:58-102) through the same clip + AMSGrad update."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, HERE)
    import _dp_joint_worker as W
    world, steps = 2, 1
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DCAP_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0", DCAP_GRAD_DTYPE=grad_dtype)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dp_joint_worker.py"), str(tmp_path), str(steps)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
    r = [np.load(tmp_path / ("joint_rank%d.npz" % k)) for k in range(world)]
    np.testing.assert_array_equal(r[0]["flat"], r[1]["flat"])            # replicas stay bit-identical
    np.testing.assert_allclose(r[0]["losses"], r[1]["losses"], rtol=0, atol=0)
    # single process: each image's gradient bucket from a fresh model (same weights, same sampling stream), averaged, one update
    _, per = W.global_inputs(world)
    grads, tower_losses = [], []
    for k in range(world):
        m = W.build()
        m._dt_rank = k                                   # the tower's own shuffle stream (ParallelModel folds the rank into the key)
        losses = m._loss_list(m.forward_backward(per[k]))
        grads.append(m.store.flat_grad.clone())
        tower_losses.append([losses["loss"], losses["rpn_class_loss"], losses["rpn_bbox_loss"], losses["imgcap_loss"]])
    ref = W.build()
    ref.store.flat_grad.copy_((grads[0] + grads[1]) / 2)
    ref.optimizer.apply(ref.store, grad_scale=1.0)
    single = ref.store.flat.cpu().numpy()
    np.testing.assert_allclose(r[0]["losses"][0], np.mean(tower_losses, axis=0), rtol=1e-5)
    start = W.build().store.flat.cpu().numpy()
    moved = np.abs(single - start).max()
    assert moved > 0
    diff = np.abs(r[0]["flat"] - single).max()
    print("joint 2-rank vs averaged single process: max |dw| = %.3e of a %.3e update" % (diff, moved))
    if grad_dtype == "bf16":
        # the same update up to the wire's rounding, not the fp32 bits: in the L2 norm (AMSGrad's first step is lr * g / (|g| + eps), so an
        # element whose gradient is of the order of eps = 1e-7 moves by a large fraction of a step when its gradient is rounded to 8 bits:
        # single elements may differ by tenths of a step, the update as a whole by a fraction of a percent)
        upd = (single - start).astype(np.float64)
        dl2 = float(np.linalg.norm(r[0]["flat"].astype(np.float64) - single) / np.linalg.norm(upd))
        print("bf16 wire: relative L2 difference of the update %.3e" % dl2)
        assert moved > 0 and dl2 < 2e-2 and diff < 0.5 * moved, (dl2, diff, moved)
        return
    assert moved > 0 and diff == 0.0      # bit-equal (round 2: 2e-3 of the update -- RoIAlign's backward was an atomic scatter then; it is a fixed-order gather now)


def test_joint_train_pipeline_under_parallel_model_two_ranks(tmp_path):
    """pipeline.JointTrainPipeline wrapped around the ParallelModel (global batches in, tf.split inside step(), the towers' mean losses
    one call late): three steps on two ranks leave both replicas with the same weights, bit for bit the weights and losses of three
    serial ParallelModel.train_on_batch calls on the same global batches."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world, steps = 2, 3
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    for mode in ("serial_eager", "pipeline"):
        port = _free_port()
        procs = []
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       DCAP_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0", DCAP_GRAD_DTYPE="f32")
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dp_joint_worker.py"), str(tmp_path), str(steps), mode],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=600)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
    ser = [np.load(tmp_path / ("joint_serial_eager_rank%d.npz" % k)) for k in range(world)]
    pip = [np.load(tmp_path / ("joint_pipeline_rank%d.npz" % k)) for k in range(world)]
    np.testing.assert_array_equal(pip[0]["flat"], pip[1]["flat"])            # replicas stay bit-identical
    np.testing.assert_array_equal(pip[0]["losses"], pip[1]["losses"])
    assert pip[0]["losses"].shape == ser[0]["losses"].shape == (steps, 4)
    np.testing.assert_array_equal(pip[0]["losses"], ser[0]["losses"])
    np.testing.assert_array_equal(pip[0]["flat"], ser[0]["flat"])
