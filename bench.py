#!/usr/bin/env python
"""bench.py -- captions/sec of one dense-captioning TRAIN STEP on synthetic Visual-Genome-shaped data.

Default workload (BASELINE.json configs[2], and configs[3] at --gpus 8): per GPU `images_per_gpu` (2) synthetic
1024x1024 images x 32 ground-truth RoIs x 15-token captions; one step =
  frozen ResNet-101 + FPN forward -> PyramidROIAlign -> frozen RoI head -> v2-inject caption decoder
  (word-LSTM-1024, inject-LSTM-256, fused Dense-V softmax / cross-entropy, V = 10 000) forward + backward -> [RCCL gradient
  all-reduce, bucketed against the backward] -> Keras AMSGrad update.
Arithmetic: fp32-grade throughout -- fp32 storage, fp32 accumulation; since round 5 most convolutions of the default plan form every fp32
product as SIX bf16 MFMA products in split arithmetic (three bf16 pieces per operand) on the bf16 matrix pipe, the rest and the decoder
GEMMs as exact fp32 MFMA products; measured against the float64 oracle 1.0 - 1.3e-6 at full depth (tests/test_gpu_oracle_fullsize.py).
The line's `dtype` says so.  Inputs are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either under python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ..., or plainly:
   without WORLD_SIZE in the environment this process starts the N ranks itself as child processes, see self_launch)
  python bench.py --config joint      BASELINE configs[4]: the joint model (bf16 decoder / head / vocabulary GEMMs), one
                                      1024x1024 image per GPU and step, 2000 proposals -> 200 RoIs, V = 50 000.  The timed schedule is the one
                                      DenseImageCapRCNN.train() runs with a frozen ResNet: the backbone pass of batch i + 1 on a second stream
                                      beside the rest of batch i's step (pipeline.JointTrainPipeline; K step() calls + flush() inside the timed
                                      region, bit-equal to K serial steps); `config.serial_ms_per_step` = the serial step in the same process,
                                      --no-pipeline times that one alone

Rank 0 prints ONE JSON line (see the driver contract) with extra objects:
  roofline      -- the dominant kernel (the conv instantiation with the largest time share): the FLOPs the matrix pipe EXECUTES per
                   launch / mean launch time against the peak OF THE PIPE THE KERNEL RUNS ON (`peak`: 2500 TFLOP/s dense bf16 for the
                   split-bf16 kernels wino32b (wino64b when forced) / pw_chain<..., true> / igemm_bs, 157.3 TFLOP/s for the fp32-MFMA kernels);
                   `frac` <= 1 by construction.  For the Winograd F(2x2,3x3) kernels executed = direct-form FLOPs / 2.25
                   (x 6 when split), and the direct-form rate is reported beside it as `achieved_direct_form` with
                   `frac_algorithmic` = direct-form rate / the same pipe's peak (SURVEY 8d's definition).  `target` states
                   north_star's 0.40 and how far the line is from it.  `algorithmic_bytes` = compulsory HBM bytes per launch (input + output +
                   weights), `traffic` = measured HBM bytes per launch (committed rocprofv3 PMC passes), `traffic_ratio` their
                   quotient.  The time is measured in this process with HIP events on the launch stream while a decoder step runs
                   beside the encoder on its own stream, as in the timed pipeline -- the condition a rocprofv3 kernel trace of this
                   command sees (profiles/r04_bench_kernel_stats.csv); `isolated` holds the same quantities with the chip to itself;
  cpu_baseline  -- the reference-as-written algorithm (oracle/torch_ref.py, float32, all host threads) timed on a bounded
                   sample of the same workload (N = 1 only);
  other_configs -- BASELINE configs[2] proper (ONE image per step), configs[1] (v2-inject decoder on precomputed RoI
                   features, batch 64) on the GPU, and the CPU companions of configs[0] / configs[1] (N = 1 only).
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WINOGRAD_GAIN = 2.25            # direct 3x3: 36 products per 2x2 output tile and channel pair; Winograd F(2x2,3x3): 16


def winograd_gain(kernel):
    """Direct-form FLOPs / FLOPs the matrix pipe executes for the layers that run on `kernel` (dc_conv2d_kernel_name's spelling).
    wino64b / wino32b (round 5): the 16 products per tile as SIX bf16 MFMA products each (split arithmetic): 6 / 2.25 of the direct
    form's count, on the bf16 pipe."""
    if kernel.startswith("wino"):
        return WINOGRAD_GAIN / 6.0 if split_bf16_kernel(kernel) else WINOGRAD_GAIN
    if kernel.startswith("igemm_bs_kernel") and os.environ.get("DCAP_CONV_MATH", "f32") == "bf16x2":
        return 1.0 / 3.0                                        # the two-piece split: three products
    return 1.0 / 6.0 if split_bf16_kernel(kernel) else 1.0      # pw_chain_kernel<.., true>, igemm_bs_kernel: six bf16 products per fp32 product


def split_bf16_kernel(kernel):
    """Kernels whose products run on the bf16 matrix pipe in split arithmetic (fp32-grade): wino64b / wino32b, the chained pointwise
    kernel's `true` instantiations and the direct kernels of DC_MATH_BF16X3 (igemm_bs_kernel)."""
    if kernel.startswith("wino"):
        return kernel.split("_")[0].endswith("b")
    if kernel.startswith("igemm_bs_kernel"):               # the direct kernels in DC_MATH_BF16X3 (split in the loop): six bf16 products per fp32 product
        return True
    return kernel.startswith("pw_chain_kernel<") and kernel.rstrip(">").rstrip().endswith("true")


def pipe_peak(kernel):
    """The matrix-pipe peak a conv kernel's executed FLOPs are priced against (TFLOP/s)."""
    return PEAK_BF16_MFMA_TFLOPS if split_bf16_kernel(kernel) else PEAK_F32_MFMA_TFLOPS
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="e2e", choices=["e2e", "joint"], help="e2e = configs[2]/[3] (the headline); joint = configs[4]")
    ap.add_argument("--images-per-gpu", type=int, default=2)
    ap.add_argument("--rois", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=15)
    ap.add_argument("--vocab", type=int, default=None, help="default 10000 (e2e) / 50000 (joint)")
    ap.add_argument("--image-size", type=int, default=1024)
    ap.add_argument("--stage4-blocks", type=int, default=22, help="22 = ResNet-101 (the benchmark config)")
    ap.add_argument("--backbone", default="resnet101", choices=["resnet101", "vgg16"],
                    help="vgg16 = the 13-conv alternative backbone of the configs[2] label (no counterpart in the reference's dense-captioning paths)")
    ap.add_argument("--no-pipeline", action="store_true", help="run encoder and decoder back to back on one stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-alt-math", action="store_true", help="skip the extra timing of the split-bf16 conv arithmetic")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the configs[0]/[1]/[2]-proper companion legs")
    ap.add_argument("--cpu-baseline-steps", type=int, default=5)
    ap.add_argument("--layer-table", default=None, help="write the per-layer conv timing table (TSV) to this path")
    ap.add_argument("--joint-dtype", default="bf16", choices=["bf16", "f32"], help="joint leg: decoder/head/vocabulary arithmetic")
    ap.add_argument("--joint-dropout", type=float, default=0.0, help="joint leg: recurrent_dropout of the two LSTMs (the reference trains with "
                    "0.2 = this package's default; the benchmark opts out so that runs are comparable: the masks cost one small kernel per LSTM)")
    ap.add_argument("--joint-images-per-gpu", type=int, default=1, help="joint leg: IMAGES_PER_GPU (the reference's script trains with 1; its graph is batched)")
    ap.add_argument("--joint-host-images", action="store_true", help="joint leg: hand the image over as a host array every step")
    ap.add_argument("--joint-conv-math", default=None, help="joint leg: conv arithmetic (f32 | bf16x3 | bf16x2 | bf16); default bf16 with "
                    "--joint-dtype bf16 (forward convolutions and data gradients; weight gradients accumulate fp32 products)")
    a = ap.parse_args()
    if a.vocab is None:
        a.vocab = 50000 if a.config == "joint" else 10000
    return a


def host_cores():
    # the GPU box hands one GPU a share of 16 host cores; os.cpu_count() reports the whole machine
    return max(1, min(len(os.sched_getaffinity(0)), 16))


def _median_steps(fn, warm, steps, budget_s):
    """`warm` untimed calls, then up to `steps` timed ones inside a wall-clock budget; returns (median seconds, n timed)."""
    t_start = time.perf_counter()
    for _ in range(warm):
        fn()
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s:
            break
    return float(np.median(times)), len(times)


def cpu_baseline(args):
    """Reference-as-written train step on the host (oracle/torch_ref.py), float32, all threads:
    batch-1 ResNet-101+FPN (+ the RPN convs the reference always evaluates) per image, RoIAlign, then
    every (prefix -> next word) sample recomputing RoI head + word LSTM; Keras AMSGrad.
    BASELINE.md section 3 protocol: 2 warm-up steps, median of >= 5 timed steps (bounded to ~45 s)."""
    from oracle import torch_ref as TR
    from image_captioning_amd import synth
    cores = host_cores()
    torch.set_num_threads(cores)
    S, V, T, R = args.image_size, args.vocab, args.tokens, args.rois
    encW = TR.to_t(synth.encoder_weights(0, args.stage4_blocks), torch.float32)
    g = torch.Generator().manual_seed(0)
    encW['_rpn'] = {'shared': 0.01 * torch.randn(512, 256, 3, 3, generator=g), 'cls': 0.01 * torch.randn(6, 512, 1, 1, generator=g),
                    'bbox': 0.01 * torch.randn(12, 512, 1, 1, generator=g)}
    W = dict(synth.head_weights(1), **synth.v2_weights(2, V))
    W['imgcap_embedding_layer/embeddings'] = synth.embedding_matrix(3, V)
    train = [k for k in W if k.split('/')[0] in ('lstm_1', 'imgcap_lstm', 'imgcap_d1')]
    decW = TR.to_t(W, torch.float32, requires_grad=train)
    imgs = synth.images(99, 1, S, S)
    rois = synth.rois(98, 1, R, S, S)[0]
    caps = synth.captions_v2(97, R, T, V, full=True)
    state = {}
    med, n = _median_steps(lambda: TR.cpu_baseline_step(encW, decW, imgs[0], rois, caps, [123.7, 116.8, 103.9], T, V, state,
                                                        args.stage4_blocks), 2, max(1, args.cpu_baseline_steps), 45.0)
    return {"value": R / med, "unit": "captions/s", "cores": cores, "kind": "port",
            "sample": "median of %d timed train steps of 1 image x %d RoI x %d tok (%.2f s/step) after 2 warm-ups; "
                      "reference-as-written algorithm incl. dead RPN convs, torch-CPU fp32" % (n, R, T, med)}


def cpu_companions():
    """CPU companions BASELINE.md section 3 promises: configs[0] (v1 decoder as written: T zero-padded prefixes x full 2-layer
    LSTM, batch 8, V = 1000, T = 10 -- the reference's own CPU-runnable case) and configs[1] (v2-inject as written: batch of
    64 (prefix -> next word) samples, each recomputing RoI head + 10-step word LSTM, V = 10 000): forward + backward
    (autograd) + Keras AMSGrad, torch-CPU fp32, 2 warm-ups, median of 5."""
    import math
    from oracle import torch_ref as TR
    from image_captioning_amd import synth
    cores = host_cores()
    torch.set_num_threads(cores)

    def amsgrad(Wd, train, grads, st):
        st['t'] = st.get('t', 0) + 1
        lr_t = 1e-3 * math.sqrt(1 - 0.999 ** st['t']) / (1 - 0.9 ** st['t'])
        with torch.no_grad():
            for k, g in zip(train, grads):
                m, v, vh = st.get(k, (torch.zeros_like(g),) * 3)
                m, v = 0.9 * m + 0.1 * g, 0.999 * v + 0.001 * g * g
                vh = torch.maximum(vh, v)
                Wd[k] -= lr_t * m / (vh.sqrt() + 1e-7)
                st[k] = (m, v, vh)

    out = {}
    rng = np.random.default_rng(5)
    # configs[0]
    V, T, B = 1000, 10, 8
    W = dict(synth.head_weights(1), **synth.v1_weights(2, V))
    W['imgcap_embedding_layer/embeddings'] = synth.embedding_matrix(3, V)
    train = [k for k in W if not k.startswith('imgcap_embedding') and 'moving_' not in k]
    Wd = TR.to_t(W, torch.float32, requires_grad=train)
    feat = torch.tensor(rng.standard_normal((B, 7, 7, 256)).astype(np.float32))
    caps = torch.tensor(synth.captions_v1(6, B, T, V))
    st = {}

    def step0():
        loss = TR.v1_loss(Wd, feat, caps)
        amsgrad(Wd, train, torch.autograd.grad(loss, [Wd[k] for k in train]), st)
    med, n = _median_steps(step0, 2, 5, 20.0)
    out["configs0_cpu"] = {"workload": "BASELINE configs[0]: text_generation_model.py decoder as written (T^2 prefix graph), RoI features "
                                       "[8,7,7,256], V=1000, T=10, trainable head, train step", "value": B / med, "unit": "captions/s",
                           "ms_per_step": 1e3 * med, "cores": cores, "kind": "port", "timed_steps": n}
    # configs[1]
    V, T, B = 10000, 10, 64
    W = dict(synth.head_weights(1), **synth.v2_weights(2, V))
    W['imgcap_embedding_layer/embeddings'] = synth.embedding_matrix(3, V)
    train = [k for k in W if k.split('/')[0] in ('lstm_1', 'imgcap_lstm', 'imgcap_d1')]
    Wd = TR.to_t(W, torch.float32, requires_grad=train)
    feat = torch.tensor(rng.standard_normal((B, 7, 7, 256)).astype(np.float32))
    words = torch.tensor(_prefix_batch(rng, B, T, V))
    tgt = torch.tensor(rng.integers(3, V, B))
    st = {}

    def step1():
        loss = TR.v2_loss(Wd, feat, words, tgt, True)
        amsgrad(Wd, train, torch.autograd.grad(loss, [Wd[k] for k in train]), st)
    med, n = _median_steps(step1, 2, 5, 20.0)
    out["configs1_cpu"] = {"workload": "BASELINE configs[1]: text_generation_model_v2.py inject decoder as written, 64 (prefix -> next word) "
                                       "samples on precomputed RoI features, V=10000, window 10, train step", "value": B / med,
                           "unit": "samples/s", "ms_per_step": 1e3 * med, "cores": cores, "kind": "port", "timed_steps": n}
    return out


def _prefix_batch(rng, B, T, V):
    """B pre-padded prefixes of random length 0..T (the reference's data_generator layout, _v2.py:183)."""
    words = np.zeros((B, T), np.int64)
    for b in range(B):
        L = int(rng.integers(0, T + 1))
        if L:
            words[b, T - L:] = rng.integers(3, V, L)
    return words


def gpu_configs1(dev, steps=30):
    """BASELINE configs[1] on the GPU: v2-inject decoder, as-written batch of 64 samples on precomputed RoI features."""
    from image_captioning_amd import synth
    from image_captioning_amd.text_generation_model_v2 import DenseCapConfig, build_model, Adam, SampleTables
    V, T, B = 10000, 10, 64
    cfg = DenseCapConfig(V, synth.embedding_matrix(3, V))
    cfg.PADDING_SIZE = T
    dec = build_model((7, 7, 256), (T,), cfg, 256, inject=True, device=dev, seed=0)
    dec.compile(optimizer=Adam(amsgrad=True), loss="categorical_crossentropy")
    rng = np.random.default_rng(5)
    feat = torch.tensor(rng.standard_normal((B, 7, 7, 256)).astype(np.float32), device=dev)
    tb = SampleTables.from_samples(_prefix_batch(rng, B, T, V), rng.integers(3, V, B), dev)
    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, out
    dt, loss = timed(lambda: dec.train_step(feat, tb), steps)                # the model's default: eager (GPU-bound at this batch)
    final = float(loss.item())
    dec.use_step_graph = True
    dt_graph, _ = timed(lambda: dec.train_step(feat, tb), steps)
    captured = any(cs.graph is not None for cs in dec._steps.values())
    U, E, u = dec.WORD_UNITS, dec.E, 256
    fwd_frozen = 2.0 * B * (7 * 7 * 256 * 1024 + 1024 * 1024)                                   # RoI head (frozen: forward only)
    fwd_train = 2.0 * B * (T * (E + U) * 4 * U + (1024 + U + u) * 4 * u + u * V)                # word LSTM, inject LSTM, Dense(V)
    gf = (fwd_frozen + 3.0 * fwd_train) / 1e9
    return {"workload": "BASELINE configs[1]: text_generation_model_v2.py inject decoder as written, 64 (prefix -> next word) samples on "
                        "precomputed RoI features, V=10000, window 10, train step, fp32", "value": B / dt, "unit": "samples/s",
            "ms_per_step": 1e3 * dt, "steps": steps, "final_loss": final,
            "step_path": "eager (the default of CaptionModelV2: GPU-bound at this batch)",
            "graph_replay_ms_per_step": 1e3 * dt_graph if captured else None,
            "roofline": {"bound": "mfma", "gflop_per_step": gf, "achieved": gf / dt / 1e3, "peak": 157.3, "unit": "TFLOP/s", "frac": gf / dt / 1e3 / 157.3,
                         "note": "a dependent chain of ~65 kernels of 5-20 us (10 + 1 LSTM steps forward, as many backward, each a launch): "
                                 "latency-bound, neither the MFMA pipe nor HBM is near a limit; every kernel runs >= 10 us, so replaying the step from a hipGraph (opt-in, use_step_graph) gains nothing"}}


def gpu_configs0(dev, steps=30):
    """BASELINE configs[0] on the GPU (the reference's own CPU-runnable case, timed beside configs0_cpu): text_generation_model.py
    Model 3 -- trainable RoI head + 2 x LSTM-512 + Dense-1024 + Dense-V, batch 8, V = 1000, T = 10 -- one train_on_batch-equivalent
    step (forward, roi_caption_loss, backward incl. the head, AMSGrad) through the single masked pass, with the reference's
    recurrent_dropout = 0.2 (device-side Philox masks) and with dropout off."""
    from image_captioning_amd import synth
    from image_captioning_amd.text_generation_model import DenseCapConfig, build_lstm_model, Adam, roi_caption_loss, caption_targets
    V, T, B = 1000, 10, 8
    out = {"workload": "BASELINE configs[0]: text_generation_model.py decoder (Model 3), RoI features [8,7,7,256], V=1000, T=10, trainable head, "
                       "train step, fp32; single masked pass == the reference's T-prefix graph (tests/test_gpu_models.py)", "unit": "captions/s",
           "steps": steps}
    rng = np.random.default_rng(5)
    feat = torch.tensor(rng.standard_normal((B, 7, 7, 256)).astype(np.float32), device=dev)
    caps = synth.captions_v1(6, B, T, V)
    tgt = caption_targets(caps).astype(np.int32)
    for rate in (0.2, 0.0):
        cfg = DenseCapConfig(V, synth.embedding_matrix(3, V), B)
        cfg.PADDING_SIZE = T
        model = build_lstm_model([7, 7, 256], cfg, 512, 'training', device=dev, seed=0)
        model.recurrent_dropout = rate
        model.compile(optimizer=Adam(amsgrad=True), loss=roi_caption_loss)
        for _ in range(3):
            model.train_step(feat, caps, tgt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = model.train_step(feat, caps, tgt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        captured = any(cs.graph is not None for cs in model._steps.values())
        final = float(loss.item())
        for _ in range(3):
            model._train_step_eager(feat, caps, tgt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model._train_step_eager(feat, caps, tgt)
        torch.cuda.synchronize()
        dt_eager = (time.perf_counter() - t0) / steps
        out["recurrent_dropout_%.1f" % rate] = {"value": B / dt, "ms_per_step": 1e3 * dt, "final_loss": final,
                                                "step_path": "captured hipGraph replay (step_graph.py)" if captured else "eager",
                                                "eager_ms_per_step": 1e3 * dt_eager}
    u, E = 512, model.E
    fwd = 2.0 * B * (7 * 7 * 256 * 1024 + 1024 * 1024) + 2.0 * B * T * ((E + 1024 + u) * 4 * u + 2 * u * 4 * u + u * 1024 + 1024 * V)
    gf = 3.0 * fwd / 1e9                                                                          # trainable head: everything trains
    dt0 = out["recurrent_dropout_0.2"]["ms_per_step"] / 1e3
    out["roofline"] = {"bound": "mfma", "gflop_per_step": gf, "achieved": gf / dt0 / 1e3, "peak": 157.3, "unit": "TFLOP/s", "frac": gf / dt0 / 1e3 / 157.3,
                       "note": "B = 8: a dependent chain of ~130 small kernels (2 x 10 LSTM steps each way, head, vocabulary layer): latency-bound; "
                               "the replayed graph removes the host's launch cost, what is left is kernel latency"}
    out["value"] = out["recurrent_dropout_0.2"]["value"]              # the reference's training default
    out["ms_per_step"] = out["recurrent_dropout_0.2"]["ms_per_step"]
    return out


def hbm_kernels_leg(dev, images_per_gpu, rois_per_image, S=1024):
    """The HBM-bound kernels north_star names, against the 8 TB/s peak (MI355X_MICROARCH.md), measured live: PyramidROIAlign at the
    benchmark's per-GPU size (2 images x 32 RoIs: a ~5 us kernel that moves 16 MB -- two dependent memory latencies, it cannot reach a
    bandwidth figure) and at a size where the 60 % target is a property of the kernel (16 images x 32 RoIs = configs[3]'s global
    batch on one GPU), and the decoder's embedding-row gather.
    Round 5: the launches ROTATE over independent input sets (pyramid maps + boxes + output) whose touched bytes
    add up to more than the 256 MB Infinity Cache, so a replay finds nothing of its input in a cache: the figure is an HBM figure, not
    an upper bound from cache-resident maps.  Kernel time = a captured hipGraph of one launch per set / number of sets (HIP events
    around the replay: no launch gaps).  Algorithmic bytes: SURVEY 8(d), 250 880 B per RoI (4 corner rows read, 1 row written per bin);
    `counter_bytes` = 2 x FETCH_SIZE + WRITE_SIZE per launch from the committed rocprofv3 PMC run of the same kernel
    (profiles/r0N_roialign_hbm.json, tools/roialign_profile.sh), when present."""
    from image_captioning_amd import ops, synth
    out = {"peak": 8000.0, "unit": "GB/s", "infinity_cache_mb": 256,
           "note": "algorithmic bytes (SURVEY 8d) / kernel time over rotating input sets larger than the Infinity Cache; the 60 % north-star bar "
                   "applies where the kernel is bandwidth-bound (16 images); at the benchmark's 64 RoIs it runs ~5 us and is latency-bound"}
    counters, counters_src = {}, None
    for cname in ("r06_roialign_hbm.json", "r05_roialign_hbm.json"):
        try:
            with open(os.path.join(ROOT, "profiles", cname)) as f:
                counters, counters_src = json.load(f), "profiles/" + cname
            break
        except (OSError, ValueError):
            pass

    def timed(fns):
        for fn in fns[:3]:
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with ops.no_gc_during_capture(), torch.cuda.graph(g):
            for fn in fns:
                fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / len(fns) * 1e3         # us per launch
    for label, B, R in (("roialign_bench_size", images_per_gpu, rois_per_image), ("roialign_16_images", 16, rois_per_image)):
        alg = B * R * 250880.0
        nsets = int(min(24, max(3, math.ceil(320e6 / (0.8 * alg)) + 1)))
        sets = []
        for i in range(nsets):
            maps = [torch.randn(B, S // st, S // st, 256, device=dev) for st in (4, 8, 16, 32)]
            boxes = torch.tensor(synth.rois(1 + i, B, R, S, S) / np.array([S, S, S, S], np.float32), device=dev)
            sets.append((maps, boxes, torch.empty(B, R, 7, 7, 256, device=dev)))
        us = timed([(lambda m=m, bx=bx, o=o: ops.roi_align_pyramid(m, bx, float(S * S), 7, out=o)) for m, bx, o in sets])
        row = {"images": B, "rois": B * R, "input_sets": nsets, "touched_mb_per_rotation": round(nsets * alg / 1e6, 1), "kernel_us": round(us, 2),
               "algorithmic_bytes": alg, "achieved": round(alg / us / 1e3, 1), "frac": round(alg / us / 1e3 / 8000.0, 3),
               "frac_label": "%.3f of 8 TB/s at %d RoIs (%d images x %d)%s" % (alg / us / 1e3 / 8000.0, B * R, B, R,
                                                                                 "" if B * R >= 256 else ": latency-bound at this size, the 60 % bar is not met here")}
        cb = counters.get(label, {}).get("counter_bytes_per_launch")
        if cb:
            row.update({"counter_bytes": cb, "achieved_counter_bytes": round(cb / us / 1e3, 1), "frac_counter_bytes": round(cb / us / 1e3 / 8000.0, 3),
                        "counter_source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % counters_src})
        out[label] = row
        del sets
        torch.cuda.empty_cache()
    nrows = 131072
    tabs = [(torch.randn(nrows, 1024, device=dev), torch.randint(0, nrows, (nrows,), dtype=torch.int32, device=dev), torch.empty(nrows, 1024, device=dev))
            for _ in range(2)]                               # 2 x (512 MB table + 512 MB output): nothing survives in the cache
    us = timed([(lambda t=t, i=i, o=o: ops.gather_rows(t, i, o)) for t, i, o in tabs] * 2)
    b = 2.0 * nrows * 1024 * 4
    out["gather_rows_131072x1024"] = {"kernel_us": round(us, 2), "algorithmic_bytes": b, "achieved": round(b / us / 1e3, 1), "frac": round(b / us / 1e3 / 8000.0, 3)}
    return out


def dataset_pipeline_leg(args, dev, steps=120):
    """The headline workload driven through the training script's objects instead of bench.py's resident buffers:
    text_generation_model_v2.train_on_dataset on a synthetic in-memory Dataset (8 images of the benchmark's size, `rois` regions with
    `tokens`-word captions each): images are molded and uploaded, box and sample tables built and uploaded per step by the
    producer thread; encoder and decoder run on the two-stream pipeline.  Captions/s over `steps` steps after a warm-up call."""
    from image_captioning_amd import synth, utils
    from image_captioning_amd.config import Config
    from image_captioning_amd.modified_dense_model import DenseImageCapRCNN
    from image_captioning_amd.text_generation_model_v2 import DenseCapConfig, build_model, Adam, train_on_dataset
    S, V, T, R, B = args.image_size, args.vocab, args.tokens, args.rois, args.images_per_gpu

    class EncCfg(Config):
        NAME = "bench"
        IMAGES_PER_GPU = 1
        IMAGE_MIN_DIM = S
        IMAGE_MAX_DIM = S

    class Synth(utils.Dataset):
        def __init__(self, n):
            super(Synth, self).__init__()
            self._px = synth.images(7, n, S, S)
            self._rois = synth.rois(8, n, R, S, S)
            self._caps = synth.captions_v2(9, n * R, T, V, full=True)
            for i in range(n):
                self.add_image("synth", image_id=i, path=None)
            self.prepare()

        def load_image(self, image_id):
            return self._px[image_id]

        def load_caption_ids_and_rois(self, image_id):
            return self._rois[image_id], [list(c) for c in self._caps[image_id * R:(image_id + 1) * R]]
    feats = DenseImageCapRCNN("inference", EncCfg(), "logs", device=dev, stage4_blocks=args.stage4_blocks)
    feats.set_weights(synth.encoder_weights(0, args.stage4_blocks))
    cfg = DenseCapConfig(V, synth.embedding_matrix(3, V))
    cfg.PADDING_SIZE = T
    dec = build_model((7, 7, 256), (T,), cfg, 256, inject=True, device=dev, seed=0)
    dec.compile(optimizer=Adam(amsgrad=True), loss="categorical_crossentropy")
    ds = Synth(8)
    train_on_dataset(dec, feats, ds, B, R, epochs=1, steps_per_epoch=4, verbose=0)          # warm-up: plan graph capture, allocations
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hist = train_on_dataset(dec, feats, ds, B, R, epochs=1, steps_per_epoch=steps, verbose=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": "the headline model trained by text_generation_model_v2.train_on_dataset from a synthetic in-memory Dataset "
                        "(%d images x %d RoI x %d tok per step; host molding, uploads and table building on the producer thread, inside the timed region)"
                        % (B, R, T), "value": B * R * steps / dt, "unit": "captions/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "final_loss": hist[-1]["loss"]}


class E2E(object):
    """The configs[2]/[3] train step: encoder plan + v2-inject decoder (+ ParallelModel), optionally pipelined on 2 streams."""

    def __init__(self, args, dev, rank, world, B):
        from image_captioning_amd import synth
        from image_captioning_amd.config import Config
        from image_captioning_amd.modified_dense_model import DenseImageCapRCNN
        from image_captioning_amd.parallel_model import ParallelModel
        from image_captioning_amd.pipeline import CaptionTrainPipeline
        from image_captioning_amd.text_generation_model_v2 import DenseCapConfig, build_model, Adam, SampleTables
        S, V, T, R = args.image_size, args.vocab, args.tokens, args.rois

        class EncCfg(Config):
            NAME = "bench"
            IMAGES_PER_GPU = B
            IMAGE_MIN_DIM = S
            IMAGE_MAX_DIM = S

        if args.backbone == "vgg16":
            from image_captioning_amd.encoder import Vgg16Plan
            plan = Vgg16Plan(synth.vgg16_weights(0), B, S, S, dev)
        else:
            enc = DenseImageCapRCNN("inference", EncCfg(), "logs", device=dev, stage4_blocks=args.stage4_blocks)
            enc.set_weights(synth.encoder_weights(0, args.stage4_blocks))
            plan = enc.plan(B, S, S)
        self.fc = fc = plan.feat_channels
        cfg = DenseCapConfig(V, synth.embedding_matrix(3, V))
        cfg.PADDING_SIZE = T
        dec = build_model((7, 7, fc), (T,), cfg, 256, inject=True, device=dev, seed=0)
        dec.compile(optimizer=Adam(amsgrad=True), loss="categorical_crossentropy")
        self.sync = None
        if world > 1:
            dec = ParallelModel(dec, world)
            self.sync = dec.inner_model.grad_sync
        seed = 1234 + rank
        images = torch.tensor(synth.images(seed, B, S, S), device=dev)
        rois = synth.rois(seed + 1, B, R, S, S)
        caps = synth.captions_v2(seed + 2, B * R, T, V, full=True)
        self.tables = SampleTables.from_captions(caps, dev)
        self.plan = plan
        plan.images.copy_(images)
        self.boxes = plan.normalize_boxes(rois)          # device-resident, normalised once
        self.feat = torch.empty((B, R, 7, 7, fc), dtype=torch.float32, device=dev)
        self.inner = dec.inner_model if world > 1 else dec
        self.B, self.R = B, R
        self.pipe = None if args.no_pipeline else CaptionTrainPipeline(plan, self.inner, R)

    def step(self):
        if self.pipe is not None:                     # encoder(i) overlaps decoder(i-1); flushed before the clock stops
            return self.pipe.step(None, self.boxes, self.tables)
        self.plan.forward(None)                       # images already resident in the plan's input buffer
        self.plan.roi_features(boxes_norm=self.boxes, out=self.feat)
        return self.inner.train_step(self.feat.view(self.B * self.R, 7, 7, self.fc), self.tables)

    def flush(self):
        return self.pipe.flush() if self.pipe is not None else None

    def timed(self, warmup, steps, barrier):
        for _ in range(max(warmup, 2)):               # >= 2: the second call captures the encoder hipGraph
            loss = self.step()
        self.flush()
        if self.sync is not None:
            self.sync.exposed_ms()                    # (drop the warm-up's marks)
            self.sync.timing = True
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = self.step()
        last = self.flush()                           # every one of the K steps is complete inside the timed region
        loss = last if last is not None else loss
        barrier()
        return time.perf_counter() - t0, loss

    def dtype_label(self):
        """The arithmetic the timed step computes in: fp32 storage and accumulation everywhere; which products run as split-bf16 pieces
        on the bf16 matrix pipe follows from the plan's kernels (not a precision claim narrower than fp32: tests hold 1e-6 at full depth)."""
        mode = os.environ.get("DCAP_CONV_MATH", "f32")
        keys = [row[-1] for row in self.plan.conv_table()]
        if any(split_bf16_kernel(k) for k in keys):
            n = sum(1 for k in keys if split_bf16_kernel(k))
            if mode == "bf16x2":
                return "f32 (split-bf16x2 products on %d of %d conv layers: 2^-16 products, fp32 accumulate)" % (n, len(keys))
            return "f32 (split-bf16x3 products, fp32 accumulate: %d of %d conv layers on the bf16 pipe, the rest and the decoder exact fp32 MFMA)" % (n, len(keys))
        return "f32"

    def roofline(self, args):
        """Per-instantiation conv timing, in the pipeline's conditions and alone; the dominant kernel's roofline numbers."""
        plan = self.plan
        table = plan.conv_table()
        alg_bytes = plan.conv_algorithmic_bytes()
        self.inner.grad_sync = None        # rank 0 alone runs this leg (after the timed region): its decoder steps must not enter a collective
        beside = None
        if self.pipe is not None:
            s_dec, feat0 = self.pipe.s_dec, self.pipe.feat[0]

            def beside():                              # one decoder train step on the decoder's stream, like the pipeline
                with torch.cuda.stream(s_dec):
                    self.inner.train_step(feat0.view(-1, 7, 7, self.fc), self.tables)
        res = {}
        for label, b in (("pipeline", beside), ("isolated", None)):
            if label == "pipeline" and b is None:
                continue
            times = dict(plan.time_convs(reps=3, beside=b))
            torch.cuda.synchronize()
            groups = {}
            for name, fl, bm, bn, sk, key in table:                 # key: rocprof's spelling of the layer's kernel
                g = groups.setdefault(key, {"flops": 0.0, "alg": 0.0, "ms": 0.0, "launches": 0, "bytes": 0.0})
                g["alg"] += fl                                       # the layer's direct-form FLOPs
                g["flops"] += fl / winograd_gain(key)                # what the matrix pipe executes
                g["ms"] += times[name]
                g["launches"] += 1
                g["bytes"] += alg_bytes[name]                        # compulsory HBM bytes: input + output + weights (+ residual)
            res[label] = (groups, times)
        main = "pipeline" if "pipeline" in res else "isolated"
        groups, times = res[main]
        if args.layer_table:
            with open(args.layer_table, "w") as f:
                f.write("layer\tgflop\tbm\tbn\tsplit_k\tus_%s\ttflops_%s\tus_isolated\tkernel\n" % (main, main))
                iso = res["isolated"][1]
                for name, fl, bm, bn, sk, key in table:
                    f.write("%s\t%.3f\t%d\t%d\t%d\t%.1f\t%.1f\t%.1f\t%s\n" % (name, fl / 1e9, bm, bn, sk, 1e3 * times[name],
                                                                              fl / (times[name] * 1e-3) / 1e12, 1e3 * iso[name], key))
        dom = max(groups, key=lambda k: groups[k]["ms"])
        g = groups[dom]
        achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12
        conv_ms = sum(v["ms"] for v in groups.values())
        # HBM bytes per launch: from the committed rocprofv3 PMC passes of this same command (tools/pmc_traffic.py: separate --pmc
        # passes, FETCH_SIZE / WRITE_SIZE with the guide's gfx950 corrections) -- counters cannot be read from inside the process
        traffic, traffic_src = None, None
        for tname in (("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json") if args.backbone == "resnet101" else ()):
            tpath = os.path.join(ROOT, "profiles", tname)
            if traffic is None and os.path.exists(tpath):
                for name, rec in json.load(open(tpath)).items():
                    if name.startswith("void dcap::" + dom[:40]) or ("::" + dom.split("<")[0] + "(") in name:
                        traffic, traffic_src = rec["hbm_bytes_per_launch_corrected"], "profiles/" + tname
        # `achieved` / `frac`: the FLOPs the matrix pipe EXECUTES for the kernel's launches over their time (<= 1 of the peak by
        # construction).  For the Winograd F(2x2,3x3) kernels that is the layers' direct-form FLOPs / 2.25 -- 16 products per
        # 2x2 output tile and channel pair instead of 36 --; the direct-form rate the layers are credited with is `achieved_direct_form`
        # (it may exceed the peak: that is what the algorithm is for, and it is not a roofline fraction).
        wino = dom.startswith("wino") or split_bf16_kernel(dom)
        alg_b = g["bytes"] / g["launches"]
        peak = pipe_peak(dom)
        pipe_s = sum(v["flops"] / (pipe_peak(k) * 1e12) for k, v in groups.items())      # seconds of matrix-pipe time at each kernel's own peak
        out = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
               "frac": achieved / peak, "traffic": traffic, "algorithmic_bytes": alg_b,
               "traffic_ratio": (traffic / alg_b) if traffic else None,
               "traffic_source": traffic_src,
               "measured": main + " (HIP events on the launch stream"
               + (", one decoder train step running beside every encoder pass on the decoder stream)" if main == "pipeline" else ")"),
               "launches_per_step": g["launches"], "gflop_per_launch": g["flops"] / g["launches"] / 1e9,
               "avg_launch_us": 1e3 * g["ms"] / g["launches"],
               "all_conv": {"gflop_per_step": plan.flops / 1e9, "ms_per_step": conv_ms, "tflops": plan.flops / (conv_ms * 1e-3) / 1e12,
                            "mfma_gflop_per_step": sum(v["flops"] for v in groups.values()) / 1e9,
                            "mfma_frac": pipe_s / (conv_ms * 1e-3),
                            "mfma_frac_note": "matrix-pipe time at each kernel's own peak (fp32 MFMA 157.3 TF; the split-bf16 Winograd kernels' six "
                                              "bf16 products per fp32 product against 2500 TF) / measured time"},
               "kernels": {k: dict({"launches": v["launches"], "ms": round(v["ms"], 4), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                    "frac_of_its_pipe": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / pipe_peak(k), 3)},
                                   **({"tflops_direct_form": round(v["alg"] / (v["ms"] * 1e-3) / 1e12, 2)} if v["alg"] != v["flops"] else {}))
                           for k, v in groups.items()}}
        if split_bf16_kernel(dom):                             # for context: the same launches priced as the fp32 products they stand for
            eq = g["alg"] / (WINOGRAD_GAIN if dom.startswith("wino") else 1.0)
            out["fp32_equivalent"] = {"achieved": eq / (g["ms"] * 1e-3) / 1e12, "frac_of_fp32_mfma_peak": eq / (g["ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                      "note": "fp32 products the kernel's bf16 products replace (one per six), over the same time, against the 157.3 TF fp32 pipe"}
        if wino:
            out["achieved_direct_form"] = g["alg"] / (g["ms"] * 1e-3) / 1e12
            out["gflop_per_launch_direct_form"] = g["alg"] / g["launches"] / 1e9
            if split_bf16_kernel(dom) and not dom.startswith("wino"):
                out["note"] = ("two chained 1x1 convolutions, every fp32 product as six bf16 MFMA products (split arithmetic, fp32 accumulate: "
                               "fp32-grade): executed bf16 MFMA FLOPs = direct-form FLOPs x 6; frac = executed / 2500 TF (dense bf16)")
            elif split_bf16_kernel(dom):
                out["note"] = ("Winograd F(2x2,3x3), fp32 transforms, the 16 products per tile as six bf16 MFMA products each (split arithmetic, fp32 "
                               "accumulate: fp32-grade): executed bf16 MFMA FLOPs = direct-form FLOPs x 6 / 2.25; frac = executed / 2500 TF (dense bf16)")
            else:
                out["note"] = ("Winograd F(2x2,3x3), fp32 transforms and products: executed MFMA FLOPs = direct-form FLOPs / %.2f; frac = executed / peak"
                               % winograd_gain(dom))
        # SURVEY 8(d)'s definition beside the executed fraction: the layers' direct-form (algorithmic) FLOPs over the same time against the
        # peak of the pipe the kernel runs on; and where the line stands against north_star's conv target
        out["frac_algorithmic"] = g["alg"] / (g["ms"] * 1e-3) / 1e12 / peak
        best = max(out["frac"], out["all_conv"]["mfma_frac"])
        out["target"] = {"north_star": ">= 0.40 MFMA utilisation on conv", "met": bool(best >= 0.40),
                         "dominant_kernel_frac": round(out["frac"], 3), "all_conv_mfma_frac": round(out["all_conv"]["mfma_frac"], 3),
                         "short_by": round(max(0.0, 0.40 - best), 3),
                         "note": "executed-FLOP fractions of the pipe each kernel runs on; the split-bf16 plan trades utilisation of the "
                                 "2.5 PF bf16 pipe for throughput (the fp32-pipe plan of round 4 sat at 0.60 of 157 TF and was 10 % slower)"}
        if main == "pipeline":
            gi = res["isolated"][0][dom]
            ai = gi["flops"] / (gi["ms"] * 1e-3) / 1e12
            iso_ms = sum(v["ms"] for v in res["isolated"][0].values())
            out["isolated"] = {"achieved": ai, "frac": ai / peak, "avg_launch_us": 1e3 * gi["ms"] / gi["launches"],
                               "all_conv_ms_per_step": iso_ms, "all_conv_tflops": plan.flops / (iso_ms * 1e-3) / 1e12}
        return out


def build_joint(args, dev, rank=0, world=1):
    """BASELINE configs[4]: the joint model's train step (dense_img_cap/dense_model.py, train_dense_captions.py), one
    1024x1024 image per GPU and step, 2000 proposals -> 200 RoIs (<= 66 positive), 15-token captions, V = 50 000;
    decoder / RoI head / vocabulary layers in bf16 (fp32 master weights, fp32 accumulate)."""
    from image_captioning_amd import synth, utils
    from image_captioning_amd.config import Config
    from image_captioning_amd.dense_model import DenseImageCapRCNN, build_rpn_targets
    from image_captioning_amd.parallel_model import ParallelModel
    S, V, T = args.image_size, args.vocab, args.tokens

    class Cfg(Config):                                  # train_dense_captions.DenseCapConfig's values (:18-41) at the benchmark's size
        NAME = "dense image captioning"
        GPU_COUNT = 1
        IMAGES_PER_GPU = args.joint_images_per_gpu
        IMAGE_MIN_DIM = S
        IMAGE_MAX_DIM = S
        PADDING_SIZE = T
        VOCABULARY_SIZE = V
        EMBEDDING_SIZE = 300
        RECURRENT_DROPOUT = args.joint_dropout
    cfg = Cfg()
    cfg.EMBEDDING_WEIGHTS = synth.embedding_matrix(3, V)
    model = DenseImageCapRCNN("training", cfg, "logs", device=dev, stage4_blocks=args.stage4_blocks, seed=0,
                              conv_math=args.joint_conv_math or ("bf16" if args.joint_dtype == "bf16" else None), compute_dtype=args.joint_dtype)
    # random FPN maps are O(10): keep the RPN / head activations in a trained network's range
    w = model.get_weights_dict()
    model.set_weights({"rpn_conv_shared/kernel": w["rpn_conv_shared/kernel"] * np.float32(0.02),
                       "rpn_bbox_pred/kernel": w["rpn_bbox_pred/kernel"] * np.float32(0.3),
                       "mrcnn_class_conv1/kernel": w["mrcnn_class_conv1/kernel"] * np.float32(0.05)})
    model.compile(1e-5)
    inner = model
    if world > 1:
        model = ParallelModel(model, world)
    seed = 1234 + rank
    rng = np.random.RandomState(seed)
    B = args.joint_images_per_gpu
    img = synth.images(seed, B, S, S)
    # ground truth = 40 of the (random-weight) RPN's own proposals per image, so that DetectionTargetLayer finds its 66 positive RoIs
    plan = inner.plan()
    plan.forward(torch.as_tensor(img))
    props_all = plan.proposals().cpu().numpy().astype(np.float64) * S
    anchors = utils.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, cfg.BACKBONE_SHAPES, cfg.BACKBONE_STRIDES, 1)
    gt_caps = np.zeros((B, cfg.MAX_GT_INSTANCES, T), np.int32)
    gt_boxes = np.zeros((B, cfg.MAX_GT_INSTANCES, 4), np.int32)
    matches, deltas_all = [], []
    for b in range(B):
        props = props_all[b]
        big = props[((props[:, 2] - props[:, 0]) >= 32) & ((props[:, 3] - props[:, 1]) >= 32)]
        boxes = np.rint(big[:40]).astype(np.int32)
        n_gt = boxes.shape[0]
        caps = synth.captions_v1(seed + 2 + 7 * b, n_gt, T, V, lmin=3, lmax=12).astype(np.int32)
        match, deltas = build_rpn_targets(img[b].shape, anchors, caps, boxes, cfg, rng)
        gt_caps[b, :n_gt], gt_boxes[b, :n_gt] = caps, boxes
        matches.append(match[:, None])
        deltas_all.append(deltas)
    # the images are resident in HBM before the timed region (the bench contract); --joint-host-images times the step with the
    # 3 MB per image host->device upload inside (pinned staging + one DMA, encoder.EncoderPlan.forward)
    image_in = img if args.joint_host_images else torch.as_tensor(img).to(dev)
    inputs = [image_in, np.zeros((B, 12)), np.stack(matches), np.stack(deltas_all), gt_caps, gt_boxes]
    return model, inner, inputs, cfg


def run_joint(args, dev, rank, world, barrier):
    """Times args.steps joint train steps (see build_joint); returns (seconds, last losses, RoIs per step, the model)."""
    model, inner, inputs, cfg = build_joint(args, dev, rank, world)
    # warm-up: at least 8 steps on one GPU -- the model's automatic step-path choice needs 3 eager steps, the capture, 2 timed replays and
    # the step that decides (dense_model.DenseImageCapRCNN._choose_step_path); none of that may sit in the timed region
    for _ in range(max(args.warmup, 8 if world == 1 else 2)):
        out = inner.train_on_batch(inputs)
    if world > 1 and getattr(inner, "grad_sync", None) is not None and hasattr(inner.grad_sync, "exposed_ms"):
        inner.grad_sync.exposed_ms()
        inner.grad_sync.timing = True
    # The timed schedule (round 6): the frozen backbone of batch i + 1 on a second stream beside the rest of batch i's step
    # (pipeline.JointTrainPipeline: two encoder plans alternate; bit-equal to the serial steps, tests/test_gpu_models.py).  --no-pipeline, or a
    # trainable ResNet stage: the serial step.
    pipe = None
    if not args.no_pipeline and inner.backbone_from is None:
        from image_captioning_amd.pipeline import JointTrainPipeline
        pipe = JointTrainPipeline(inner)                  # `inputs` is this rank's own batch (weak scaling), as in the serial leg below: the tower, not
        for _ in range(6):                                # the ParallelModel (whose step() would tf.split a global batch); grad_sync lives on the tower
            pipe.step(inputs)
        pipe.flush()
        if world > 1 and getattr(inner, "grad_sync", None) is not None and hasattr(inner.grad_sync, "exposed_ms"):
            inner.grad_sync.exposed_ms()
    barrier()
    t0 = time.perf_counter()
    if pipe is not None:
        for _ in range(args.steps):
            pipe.step(inputs)
        out_dev = pipe.flush()                            # every one of the K steps is complete inside the timed region
    else:
        for _ in range(args.steps):
            # per-rank image: every rank steps its own shard (weak scaling).  The losses stay on the device (train_on_batch_device, what the
            # model's own train() loop calls): no host synchronisation inside the timed region, one read-back after it
            out_dev = inner.train_on_batch_device(inputs)
    barrier()
    dt = time.perf_counter() - t0
    out = inner._losses_to_api(out_dev.cpu().numpy())
    R = cfg.TRAIN_ROIS_PER_IMAGE * args.joint_images_per_gpu          # captions (RoIs with their targets) per step and GPU
    inner.other_path_ms_per_step = None
    inner.pipelined = pipe is not None
    inner.serial_ms_per_step = None
    if world == 1 and pipe is not None:
        # the serial step (eager launches, the path the model's own measurement kept on this pool's boxes) over the same K steps, beside it
        inner.timed_path = "pipelined"
        for _ in range(3):
            inner.train_on_batch(inputs)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            inner.train_on_batch_device(inputs)
        torch.cuda.synchronize()
        inner.serial_ms_per_step = 1e3 * (time.perf_counter() - t1) / args.steps
    elif world == 1:
        # what the data-parallel step gives up (or gains) by issuing its launches from Python (its collectives cannot
        # sit inside a capture): the SAME step with the same two-stream fork on the OTHER path -- eager when the timed region replayed the
        # captured graph, the graph when the model's measured choice was eager -- timed after the headline region, reported beside it
        inner.timed_path = "graph" if (inner.use_step_graph and any(k[0] == "train" for k in inner._graphs)) else "eager"
        inner.use_step_graph = inner.timed_path == "eager"
        for _ in range(4):
            inner.train_on_batch(inputs)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            inner.train_on_batch_device(inputs)
        torch.cuda.synchronize()
        inner.other_path_ms_per_step = 1e3 * (time.perf_counter() - t1) / args.steps
        inner.use_step_graph = inner.timed_path == "graph"
    return dt, out, R, inner


def self_launch(n, argv=None, script=None, timeout_s=None, extra_env=None):
    """`python bench.py --gpus N` without an external launcher: this process touches no GPU; it starts the N ranks as CHILD
    processes (never exec) with the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT on
    127.0.0.1), relays rank 0's stdout (the JSON line) and every rank's stderr, and returns non-zero when a rank fails or the
    job outlives `timeout_s` (the remaining ranks are then killed by PID).  With fewer GPUs than ranks (a one-GPU box) the ranks
    share devices over the gloo backend -- a rehearsal of the multi-process path, flagged as such in the JSON line
    (`dist_backend`)."""
    import socket
    import tempfile
    script = script or os.path.abspath(__file__)
    argv = list(sys.argv[1:] if argv is None else argv)
    timeout_s = float(os.environ.get("DCAP_BENCH_TIMEOUT", 1500) if timeout_s is None else timeout_s)
    # NOTE: this launcher must only ever START CHILD PROCESSES, never exec: device_count() below may initialise the HIP runtime in this
    # process (without amdsmi it falls back to hipGetDeviceCount), and exec-ing from a process that has is forbidden on the GPU pool.
    shared = "DCAP_DIST_BACKEND" not in os.environ and torch.cuda.device_count() < n
    if shared:
        sys.stderr.write("bench.py: %d rank(s) on %d visible GPU(s): ranks share devices, gradient exchange over gloo (rehearsal)\n"
                         % (n, torch.cuda.device_count()))
    deadline = time.monotonic() + timeout_s
    rc, out = 1, ""
    for attempt in range(3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()               # (released before the children bind it: a race another process can win -- hence the retry below)
        env0 = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env0.update(extra_env or {})
        if shared:
            env0["DCAP_DIST_BACKEND"] = "gloo"
        # rank 0's stdout goes to a FILE, not a pipe: a pipe is read only after the ranks have exited, and a JSON line that outgrows the
        # 64 KB pipe buffer would block rank 0 in write() until the timeout
        procs, t_start = [], time.monotonic()
        with tempfile.TemporaryFile(mode="w+") as out0, tempfile.TemporaryFile(mode="w+") as err0:
            for rank in range(n):
                env = dict(env0, RANK=str(rank), LOCAL_RANK=str(rank))
                procs.append(subprocess.Popen([sys.executable, script] + argv, env=env, stdout=out0 if rank == 0 else subprocess.DEVNULL,
                                              stderr=err0 if rank == 0 else None, text=True))
            rc = 0
            try:
                while True:
                    codes = [p.poll() for p in procs]
                    if any(c not in (None, 0) for c in codes):                # a rank died: the others would wait in a collective forever
                        rc = next(c for c in codes if c not in (None, 0))
                        break
                    if all(c == 0 for c in codes):
                        break
                    if time.monotonic() > deadline:
                        sys.stderr.write("bench.py: ranks still running after %.0f s -- killing them\n" % timeout_s)
                        rc = 124
                        break
                    time.sleep(0.2)
            finally:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                for p in procs:
                    p.wait()
            out0.seek(0)
            err0.seek(0)
            out, err = out0.read(), err0.read()
        sys.stderr.write(err)
        # the rendezvous port was taken between its release here and the children's bind: try again on a fresh one
        if rc not in (0, 124) and time.monotonic() - t_start < 60 and ("EADDRINUSE" in err or "Address already in use" in err or "address already in use" in err):
            sys.stderr.write("bench.py: rendezvous port %d was taken, retrying on a fresh port\n" % port)
            continue
        break
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc


def joint_roofline(args, dev, inner):
    """Roofline block of the joint leg (rank 0, after the timed region), against the dense bf16 MFMA peak:
    the bf16 convolutions of the forward plan, timed per launch with HIP events on the launch stream and grouped by kernel
    (bconv256_kernel: the 256 x 256 x 64 tile; bconv64_kernel: the 64 x 64 tile with the whole K loop per block; bconv_kernel: the
    128 x 128 tile incl. its split-K slab reductions) -- `achieved`
    is the group with the largest time share; and the fused vocabulary softmax / cross-entropy at the step's own shape
    (TRAIN_ROIS_PER_IMAGE x T rows, K = 1024, V words; keras_sparse: three GEMM passes), timed as whole calls."""
    from image_captioning_amd import ops
    plan = inner.plan()
    rows = plan.time_bconvs(reps=3)
    groups = {}
    for name, fl, ms, tile, sk in rows:
        g = groups.setdefault({256: "bconv256_kernel", 64: "bconv64_kernel"}.get(tile, "bconv_kernel"), {"flops": 0.0, "ms": 0.0, "launches": 0, "split_k_layers": 0})
        g["flops"] += fl
        g["ms"] += ms
        g["launches"] += 1
        g["split_k_layers"] += int(sk > 1)
    dom = max(groups, key=lambda k: groups[k]["ms"])
    g = groups[dom]
    ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
    tot_fl, tot_ms = sum(v["flops"] for v in groups.values()), sum(v["ms"] for v in groups.values())
    # HBM bytes per launch of the dominant kernel: from the committed rocprofv3 PMC passes of `bench.py --config joint` (tools/collect_profiles.sh
    # -> tools/pmc_traffic.py: separate --pmc passes, FETCH_SIZE / WRITE_SIZE with the guide's gfx950 corrections)
    traffic, traffic_src = None, None
    for tname in ("r06_joint_pmc_traffic.json", "r05_joint_pmc_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", tname)
        if traffic is None and os.path.exists(tpath):       # the newest round's passes that hold the kernel
            for name, rec in json.load(open(tpath)).items():
                if ("::" + dom + "(") in name or name.startswith("dcap::" + dom) or name.startswith("void dcap::" + dom):
                    traffic, traffic_src = rec["hbm_bytes_per_launch_corrected"], "profiles/" + tname
    out = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_MFMA_TFLOPS,
           "traffic": traffic, "traffic_source": traffic_src, "measured": "HIP events around every dc_conv2d_bf16 launch of the forward plan (eager replay, alone on the chip)",
           "launches_per_step": g["launches"], "gflop_per_launch": g["flops"] / g["launches"] / 1e9, "avg_launch_us": 1e3 * g["ms"] / g["launches"],
           "forward_bf16_convs": {"gflop_per_step": tot_fl / 1e9, "ms_per_step": tot_ms, "tflops": tot_fl / (tot_ms * 1e-3) / 1e12},
           "kernels": {k: {"launches": v["launches"], "split_k_layers": v["split_k_layers"], "ms": round(v["ms"], 4),
                           "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)} for k, v in groups.items()}}
    # the vocabulary layer at the step's shape
    M, V, K = inner.config.TRAIN_ROIS_PER_IMAGE * args.tokens, args.vocab, 1024
    gen = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn((M, K), device=dev, generator=gen).to(torch.bfloat16)
    W = (torch.randn((K, V), device=dev, generator=gen) * (2.0 / K ** 0.5)).to(torch.bfloat16)
    b = torch.randn(V, device=dev, generator=gen)
    t = torch.randint(0, V, (M,), device=dev, generator=gen, dtype=torch.int32)
    w = torch.rand(M, device=dev, generator=gen)
    loss, dl, db = torch.empty(M, device=dev), torch.empty((M, V), dtype=torch.bfloat16, device=dev), torch.empty(V, device=dev)

    def timed(call):
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        return best
    gf1 = 2.0 * M * V * K / 1e9                                    # one GEMM pass
    # ordinary logits (+-0.3: random-init weights, the timed step's case): no probability outside [1e-7, 1 - 1e-7] -- at V = 50 000 the mean
    # probability is 2e-5, so a logit 5.3 below the row's log-sum-exp already clips; the second regime has such logits in every row
    Wq, bq = (W.float() * 0.1).to(torch.bfloat16), b * 0.1
    rows_out = {}
    for label, Wx, bx, clipped in (("ordinary_logits", Wq, bq, False), ("every_row_clipped", W, b, True)):
        for flavour, mat in (("materialised_bf16_logits", True), ("recomputed_fp32_logits", False)):
            ms = timed(lambda: ops.vocab_ce(X, Wx, bx, t, loss_rows=loss, dlogits=dl, dbias=db, grad_scale=1.0, row_weights=w, keras_sparse=True,
                                            materialize_bf16=mat))
            passes = 1 if mat else (3 if clipped else 2)
            rows_out[label + "/" + flavour] = {"ms_per_call": round(ms, 4), "gemm_passes": passes, "executed_gflop": round(passes * gf1, 1),
                                               "tflops_executed": round(passes * gf1 / ms, 1), "frac_executed": round(passes * gf1 / ms / PEAK_BF16_MFMA_TFLOPS, 3),
                                               "elementwise_bytes": (4.0 * M * V if mat else 0.0)}
    main_row = rows_out["ordinary_logits/materialised_bf16_logits"]
    ref_row = rows_out["ordinary_logits/recomputed_fp32_logits"]
    out["vocab_ce"] = {"rows": M, "vocab": V, "k": K, "ms_per_call": main_row["ms_per_call"], "gemm_passes": 1, "gflop_per_call": gf1,
                       "tflops": main_row["tflops_executed"], "frac": main_row["frac_executed"],
                       "ms_per_pass_of_the_two_pass_form": round(main_row["ms_per_call"] / 2, 4),
                       "two_pass_form_ms_per_call": ref_row["ms_per_call"], "two_pass_form_frac": ref_row["frac_executed"],
                       "flavours": rows_out,
                       "note": "the timed step runs `ordinary_logits/materialised_bf16_logits` (round 6): ONE GEMM pass that rounds the logits to bf16 and "
                               "parks them in the gradient's buffer + an in-place elementwise gradient pass (4 bytes per logit); `frac` = that one pass's "
                               "FLOPs over the WHOLE call's time against 2500 TF (the call also streams 600 MB).  The recomputing form (rounds 2-5: two GEMM "
                               "passes, three when rows clip) is kept and timed beside it; north_star's 0.40 is met by neither: see DESIGN.md"}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # started plainly: be the launcher (before anything touches the GPU)
        raise SystemExit(self_launch(args.gpus))
    from image_captioning_amd.parallel_model import init_process_group_from_env, GradAllReduce
    rank, world, local_rank = init_process_group_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start it plainly (python bench.py --gpus %d starts its own ranks) or with "
                         "torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            return float(tmax.item())
        return dt

    ranks_seen = GradAllReduce().check_ranks(dev)        # an actual all-reduce of ones: the ranks the backend connected
    backend = dist.get_backend() if world > 1 else None
    if world > 1:
        # one rank per GPU over RCCL whenever the node has a GPU per rank: a silent fall-back to gloo must not produce a scaling number
        if torch.cuda.device_count() >= world and "DCAP_DIST_BACKEND" not in os.environ:
            assert backend == "nccl", "%d GPUs for %d ranks but the process group runs on %r" % (torch.cuda.device_count(), world, backend)
        assert ranks_seen == world, "the backend connected %d of %d ranks" % (ranks_seen, world)
        from image_captioning_amd.parallel_model import reserve_cus_for_collectives
        persistent_cus = reserve_cus_for_collectives(world)      # before any encoder graph is captured
    else:
        persistent_cus = None
    S, V, T, R, B = args.image_size, args.vocab, args.tokens, args.rois, args.images_per_gpu

    if args.config == "joint":
        dt, losses, rois_per_step, inner = run_joint(args, dev, rank, world, barrier)
        dt = max_over_ranks(dt)
        ar = inner.grad_sync.exposed_ms() if (world > 1 and getattr(inner, "grad_sync", None) is not None) else None
        out = {
            "metric": "captions/sec (train step) on 1024px joint model (2000 proposals -> 200 RoI x 15tok) synth",
            "value": world * rois_per_step * args.steps / dt, "unit": "captions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.joint_dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"positive_rois": int(np.sum(inner.last_targets['npos'])), "workload": "BASELINE configs[4]: dense_img_cap joint model train step: frozen ResNet-101 + trainable FPN/RPN + "
                                   "ProposalLayer(2000) + DetectionTargetLayer(200 RoIs) + RoIAlign + trainable RoI head + Model-3 decoder "
                                   "+ 4 losses + Adam(amsgrad, clipnorm 0.5); %dx%d synth image, %d image(s)/GPU, V=%d, %d-token captions"
                                   % (S, S, args.joint_images_per_gpu, V, T), "images_per_gpu": args.joint_images_per_gpu,
                       "rois_per_image": rois_per_step // args.joint_images_per_gpu, "parallelism": "dp%d" % world,
                       "decoder_dtype": args.joint_dtype, "recurrent_dropout": args.joint_dropout, "conv_math": inner.conv_math_name, "losses": [float(v) for v in losses],
                       "rccl_ranks": ranks_seen, "dist_backend": backend, "persistent_cus": persistent_cus,
                       "allreduce_exposed_ms_per_step": None if ar is None else round(ar[0], 4),
                       "allreduce_host_wait_ms_per_step": None if ar is None else round(ar[1], 4),
                       # which schedule the timed steps ran: one GPU replays the step behind the encoder as ONE captured
                       # hipGraph with the RPN backward on a second branch; a data-parallel step issues the same launches eagerly, its
                       # collectives from Python as each layer group's backward has been enqueued (they cannot sit inside the capture)
                       "other_path_ms_per_step": None if getattr(inner, "other_path_ms_per_step", None) is None else
                       {("eager" if getattr(inner, "timed_path", "") == "graph" else "graph"): round(inner.other_path_ms_per_step, 4)},
                       "step_path_choice": getattr(inner, "step_path_choice", None),
                       "pipeline": ("backbone(i+1) || rest of step i (FPN, RPN, proposals .. AMSGrad), 2 encoder plans, 2 HIP streams + the RPN-backward branch"
                                    if getattr(inner, "pipelined", False) else "serial step"),
                       "serial_ms_per_step": None if getattr(inner, "serial_ms_per_step", None) is None else round(inner.serial_ms_per_step, 4),
                       "step_path": ("eager, data-parallel: RPN backward + its ranges' all-reduce on a second stream beside the proposals -> decoder chain; "
                                     "per-layer-group all-reduce issued from Python behind each group's backward"
                                     if world > 1 else ("captured hipGraph + RPN backward on a second branch" if getattr(inner, "timed_path", "graph") == "graph"
                                                        else "eager launches" + (" (the model's measured choice over the captured graph)" if getattr(inner, "timed_path", "") == "eager" else "")
                                                        + " + RPN backward on a second stream")),
                       "step_graph_fallback": inner.step_graph_fallback,
                       "grad_wire_dtype": getattr(getattr(inner, "grad_sync", None), "dtype", None) if world > 1 else None},
        }
        if rank == 0 and not args.no_roofline and args.joint_dtype == "bf16" and inner.conv_math_name == "bf16":
            inner.grad_sync = None
            out["roofline"] = joint_roofline(args, dev, inner)
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.backbone != "resnet101":                      # the alternative backbone is one GPU leg: no CPU port, no extra legs
        args.no_cpu_baseline = args.no_alt_math = args.no_other_configs = True
    e2e = E2E(args, dev, rank, world, B)
    dt, loss = e2e.timed(args.warmup, args.steps, barrier)
    dt = max_over_ranks(dt)
    ar = e2e.sync.exposed_ms() if e2e.sync is not None else None      # rank 0's view: time its decoder stream stood still for the exchange
    captions = world * B * R * args.steps
    final_loss = float(loss.item())

    out = {
        "metric": "captions/sec (train step) on 1024px x 32RoI x 15tok synth",
        "value": captions / dt, "unit": "captions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": e2e.dtype_label(), "data": "synthetic",
        "config": {"workload": ("BASELINE configs[2] (configs[3] at 8 GPUs): frozen ResNet-101+FPN fwd + PyramidROIAlign + "
                                "RoI head + v2-inject decoder fwd/bwd + AMSGrad, %dx%d synth images, %d RoI/img, %d-token captions, V=%d"
                                if args.backbone == "resnet101" else
                                "configs[2]'s label taken literally (alternative backbone, no counterpart in the reference's dense-captioning "
                                "paths): frozen VGG16 13-conv fwd + RoIAlign 7x7x512 on block5_conv3 + RoI head + v2-inject decoder fwd/bwd "
                                "+ AMSGrad, %dx%d synth images, %d RoI/img, %d-token captions, V=%d") % (S, S, R, T, V),
                   "backbone": args.backbone,
                   "images_per_gpu": B, "global_batch_images": B * world, "captions_per_step": B * R * world,
                   "parallelism": "dp%d" % world, "stage4_blocks": args.stage4_blocks, "final_loss": final_loss,
                   "pipeline": "encoder(i+1) || decoder(i), 2 HIP streams" if e2e.pipe is not None else "single stream",
                   "rccl_ranks": ranks_seen, "dist_backend": backend, "persistent_cus": persistent_cus,
                   "allreduce_exposed_ms_per_step": None if ar is None else round(ar[0], 4),
                   "allreduce_host_wait_ms_per_step": None if ar is None else round(ar[1], 4),
                   "grad_allreduce": ("per layer group, asynchronous, issued as each group's backward is enqueued" if world > 1 else None)},
    }

    if rank == 0 and not args.no_roofline:
        out["roofline"] = e2e.roofline(args)
    if rank == 0 and world == 1 and not args.no_alt_math and os.environ.get("DCAP_CONV_MATH", "f32") == "f32":
        # Same workload with the encoder's convolutions on the bf16 matrix pipe (operands split into bf16 pieces on the fly,
        # fp32 accumulate; csrc/igemm_bf16s.h), each mode timed by a child process of this one after the headline run.
        # Reported beside the headline, which stays on exact fp32 products.
        torch.cuda.synchronize()
        # Round 4: in both modes the 3x3 layers with frozen weights run the fp32 Winograd kernel like the headline; the split arithmetic
        # applies to the 1x1 / strided / stem layers.
        labels = {"bf16x3": "1x1 / strided / stem layers: 3-piece bf16 split of both operands, 6 MFMA products, fp32 accumulate (fp32-grade: same "
                            "test tolerances as f32); 3x3 layers: the headline's Winograd kernel",
                  "bf16x2": "1x1 / strided / stem layers: 2-piece bf16 split, 3 MFMA products, fp32 accumulate (2^-16 products; features within "
                            "1e-3 of the oracle); 3x3 layers: the headline's Winograd kernel"}
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-alt-math",
               "--no-cpu-baseline", "--no-other-configs", "--images-per-gpu", str(B), "--rois", str(R), "--tokens", str(T),
               "--vocab", str(V), "--image-size", str(S), "--stage4-blocks", str(args.stage4_blocks)] + (["--no-pipeline"] if args.no_pipeline else [])
        out["alt_math"] = {}
        for mode, label in labels.items():
            try:
                r = subprocess.run(cmd, env=dict(os.environ, DCAP_CONV_MATH=mode), capture_output=True, text=True, timeout=300)
                alt = json.loads(r.stdout.strip().splitlines()[-1])
                out["alt_math"][mode] = {"conv_math": label, "value": alt["value"], "unit": "captions/s", "ms_per_step": alt["ms_per_step"],
                                         "all_conv": alt.get("roofline", {}).get("all_conv")}
            except Exception as e:                             # the headline must not depend on the extra legs
                out["alt_math"][mode] = {"error": repr(e)[:200]}
    if rank == 0 and world == 1 and not args.no_other_configs:
        other = {}
        try:
            # configs[2] proper: ONE image per step, timed by a child process (a fresh context, like the alt_math legs)
            del e2e
            torch.cuda.empty_cache()
            cmd1 = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-alt-math",
                    "--no-cpu-baseline", "--no-other-configs", "--no-roofline", "--images-per-gpu", "1", "--rois", str(R), "--tokens", str(T),
                    "--vocab", str(V), "--image-size", str(S), "--stage4-blocks", str(args.stage4_blocks)] + (["--no-pipeline"] if args.no_pipeline else [])
            r1 = subprocess.run(cmd1, capture_output=True, text=True, timeout=300)
            one = json.loads(r1.stdout.strip().splitlines()[-1])
            other["configs2_one_image"] = {"workload": "BASELINE configs[2] as defined: 1 image x %d RoI per step, same model" % R,
                                           "value": one["value"], "unit": "captions/s", "ms_per_step": one["ms_per_step"], "steps": one["steps"]}
            # the same model at twice configs[3]'s per-GPU batch (4 images): what the fp32 kernels reach with fuller grids
            cmd4 = [a if a != "1" or cmd1[i - 1] != "--images-per-gpu" else "4" for i, a in enumerate(cmd1)]
            r4 = subprocess.run(cmd4, capture_output=True, text=True, timeout=300)
            four = json.loads(r4.stdout.strip().splitlines()[-1])
            other["four_images_per_gpu"] = {"workload": "the headline's model at 4 images x %d RoI per step and GPU (not a BASELINE config)" % R,
                                            "value": four["value"], "unit": "captions/s", "ms_per_step": four["ms_per_step"], "steps": four["steps"]}
            out["hbm_kernels"] = hbm_kernels_leg(dev, B, R, S)
            other["configs0_gpu"] = gpu_configs0(dev)
            other["configs1_gpu"] = gpu_configs1(dev)
            other["train_on_dataset"] = dataset_pipeline_leg(args, dev)
            # configs[2]'s label taken literally: the VGG16 13-conv backbone (child process; its roofline leg gives the conv TFLOP/s)
            cmdv = [sys.executable, os.path.abspath(__file__), "--backbone", "vgg16", "--steps", str(max(5, args.steps // 2)), "--warmup", "2",
                    "--images-per-gpu", str(B), "--rois", str(R), "--tokens", str(T), "--vocab", str(V), "--image-size", str(S)]
            rv = subprocess.run(cmdv, capture_output=True, text=True, timeout=300)
            vg = json.loads(rv.stdout.strip().splitlines()[-1])
            other["configs2_vgg16"] = {"workload": vg["config"]["workload"], "value": vg["value"], "unit": "captions/s",
                                       "ms_per_step": vg["ms_per_step"], "steps": vg["steps"], "images_per_gpu": B,
                                       "all_conv": vg.get("roofline", {}).get("all_conv")}
        except Exception as e:
            other["error_gpu"] = repr(e)[:300]
        try:
            # BASELINE configs[4]: the joint model's train step in bf16 (child process; carries its own roofline block against the
            # dense bf16 MFMA peak)
            cmdj = [sys.executable, os.path.abspath(__file__), "--config", "joint", "--steps", str(max(5, args.steps // 2)), "--warmup", "3",
                    "--image-size", str(S), "--tokens", str(T), "--stage4-blocks", str(args.stage4_blocks)]
            rj = subprocess.run(cmdj, capture_output=True, text=True, timeout=400)
            jt = json.loads(rj.stdout.strip().splitlines()[-1])
            other["configs4_joint"] = {"workload": jt["config"]["workload"], "value": jt["value"], "unit": "captions/s", "ms_per_step": jt["ms_per_step"],
                                       "steps": jt["steps"], "dtype": jt["dtype"], "positive_rois": jt["config"]["positive_rois"],
                                       "rois_per_image": jt["config"]["rois_per_image"], "recurrent_dropout": jt["config"]["recurrent_dropout"],
                                       "schedule": jt["config"].get("pipeline"), "serial_ms_per_step": jt["config"].get("serial_ms_per_step"),
                                       "roofline": jt.get("roofline")}
            # the same step with the reference's recurrent_dropout = 0.2 on both LSTMs (dense_img_cap/dense_model.py:769-770): device-side
            # Philox masks, one fused launch per LSTM timestep in both directions (round 4)
            rd = subprocess.run(cmdj + ["--joint-dropout", "0.2", "--no-roofline"], capture_output=True, text=True, timeout=400)
            jd = json.loads(rd.stdout.strip().splitlines()[-1])
            other["configs4_joint_reference_dropout"] = {"value": jd["value"], "unit": "captions/s", "ms_per_step": jd["ms_per_step"], "steps": jd["steps"],
                                                         "recurrent_dropout": jd["config"]["recurrent_dropout"],
                                                         "serial_ms_per_step": jd["config"].get("serial_ms_per_step")}
            # IMAGES_PER_GPU = 2 (the reference's batched graph, config.py:35): every latency-bound trunk layer sees twice the pixels
            r2 = subprocess.run(cmdj + ["--joint-images-per-gpu", "2", "--no-roofline"], capture_output=True, text=True, timeout=400)
            j2 = json.loads(r2.stdout.strip().splitlines()[-1])
            other["configs4_joint_2img"] = {"value": j2["value"], "unit": "captions/s", "captions_per_s_per_gpu": j2["value"] / max(1, j2["n_gpus"]),
                                            "ms_per_step": j2["ms_per_step"], "steps": j2["steps"], "images_per_gpu": j2["config"]["images_per_gpu"],
                                            "rois_per_image": j2["config"]["rois_per_image"], "positive_rois": j2["config"]["positive_rois"],
                                            "step_path": j2["config"]["step_path"], "schedule": j2["config"].get("pipeline"),
                                            "serial_ms_per_step": j2["config"].get("serial_ms_per_step")}
        except Exception as e:
            other["error_joint"] = repr(e)[:300]
        if not args.no_cpu_baseline:
            try:
                other.update(cpu_companions())
            except Exception as e:
                other["error_cpu"] = repr(e)[:300]
        out["other_configs"] = other
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
