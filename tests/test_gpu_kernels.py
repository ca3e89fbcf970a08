"""Kernel-level parity: every C-ABI entry point against the NumPy oracle on seeded inputs (-m gpu).
Tolerances: fp32 MFMA accumulation vs float64 oracle => 2e-5 of the output scale unless stated
(index outputs -- argmax, pyramid levels -- are bit-exact)."""
import os

import numpy as np
import pytest
import torch

from image_captioning_amd._lib import DcapError

from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from image_captioning_amd import ops as _ops, _lib
    _lib.load()
    return _ops


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


def close(got, want, tol=2e-5):
    got = (got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)).astype(np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max()) / scale
    assert err < tol, "max err %.3e (scaled) exceeds %.1e" % (err, tol)


@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (100, 72, 300), (8, 1000, 1024), (260, 132, 68), (960, 256, 2048), (512, 1024, 960)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_layouts(ops, M, N, K, ta, tb):
    rng = np.random.default_rng(M * 7 + N * 3 + K + ta * 2 + tb)
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((K, N))
    a = dev(A.T if ta else A)
    b = dev(B.T if tb else B)
    close(ops.gemm(a, b, a_trans=bool(ta), b_trans=bool(tb)), A @ B)


@pytest.mark.parametrize("split", [0, 1, 3, 8])
def test_gemm_epilogue_splitk_accumulate(ops, split):
    rng = np.random.default_rng(11 + split)
    M, N, K = 64, 1024, 12544 // 4
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((K, N)) / np.sqrt(K)
    sc, sh = rng.uniform(0.5, 1.5, N), rng.standard_normal(N)
    R = rng.standard_normal((M, N))
    C0 = rng.standard_normal((M, N))
    want = np.maximum((A @ B) * sc + sh + R, 0) + C0
    out = dev(C0)
    ops.gemm(dev(A), dev(B), out=out, scale=dev(sc), shift=dev(sh), residual=dev(R), relu=True, accumulate=True, split_k=split)
    close(out, want)


def test_gemm_gather_is_embedding_lookup(ops):
    rng = np.random.default_rng(5)
    V, E, N = 500, 300, 256
    table = rng.standard_normal((V, E))
    W = rng.standard_normal((E, N))
    ids = rng.integers(0, V, 77)
    got = ops.gemm(dev(table), dev(W), gather=dev(ids, torch.int32), shift=dev(np.ones(N)))
    close(got, table[ids] @ W + 1.0)


def test_gemm_strided_views(ops):
    rng = np.random.default_rng(6)
    big = rng.standard_normal((90, 1324))
    W = rng.standard_normal((1324, 64))
    a = dev(big)
    close(ops.gemm(a[:, 300:], dev(W[300:])), big[:, 300:] @ W[300:])          # the RoI-feature half of lstm1.kernel
    out = torch.zeros(90, 128, device="cuda")
    ops.gemm(a, dev(W), out=out[:, 64:])
    close(out[:, 64:], big @ W)
    assert float(out[:, :64].abs().max()) == 0.0


def test_gemm_rejects_bad_arguments(ops):
    from image_captioning_amd._lib import DcapError
    a = torch.zeros(8, 6, device="cuda")
    b = torch.zeros(6, 8, device="cuda")
    with pytest.raises(DcapError):
        ops.gemm(a, b)                     # lda = 6 is not a multiple of 4
    with pytest.raises(DcapError):
        ops.gemm(torch.zeros(8, 8), torch.zeros(8, 8))     # CPU tensors: no fallback


CONV_CASES = [
    # N,H,W,Cin,Cout,k,stride,padding,res_mode,relu
    (1, 16, 16, 64, 64, 1, 1, 'valid', 0, True),
    (2, 16, 24, 64, 256, 1, 1, 'valid', 1, True),
    (1, 32, 32, 256, 128, 1, 2, 'valid', 0, True),
    (2, 12, 20, 64, 64, 3, 1, 'same', 0, True),
    (1, 16, 16, 128, 128, 3, 1, 'same', 0, False),
    (1, 8, 8, 512, 512, 3, 1, 'same', 0, True),          # small M, long K: split-K path
    (1, 16, 16, 256, 256, 1, 1, 'valid', 2, False),      # FPN lateral + upsample-add
    (3, 7, 7, 256, 1024, 7, 1, 'valid', 0, True),        # RoI head conv as a conv
    (2, 64, 64, 64, 128, 3, 1, 'same', 1, True),         # 128-wide tiles
    (1, 9, 7, 128, 512, 1, 1, 'valid', 1, True),         # streaming pointwise kernel: ragged last strip, two weight slices
    (2, 16, 16, 256, 1024, 1, 1, 'valid', 1, True),      # ... Cin = 256, sixteen slices
    (1, 16, 16, 256, 64, 1, 1, 'valid', 0, True),        # ... one 64-channel slice
    (1, 8, 8, 128, 192, 1, 1, 'valid', 0, False),        # ... a partial last slice
]


@pytest.mark.parametrize("math", [0, 1, 3, 4], ids=["f32", "bf16x3", "bf16x2", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_matches_oracle(ops, case, math):
    """math 0: fp32 MFMA products; math 1: operands split into three bf16 pieces, six matrix-pipe products -- held to the
    SAME 2e-5 tolerance against the float64 oracle."""
    from image_captioning_amd.packing import pack_conv_kernel
    N, H, W, Cin, Cout, k, stride, padding, res_mode, relu = case
    rng = np.random.default_rng(sum(int(v) * (i + 1) for i, v in enumerate(case) if not isinstance(v, str)))
    x = rng.standard_normal((N, H, W, Cin))
    w = rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)
    sc, sh = rng.uniform(0.5, 1.5, Cout), rng.standard_normal(Cout)
    y = O.conv2d_nhwc(x, w, None, stride, padding) * sc + sh
    Ho, Wo = y.shape[1:3]
    res = None
    if res_mode == 1:
        res = rng.standard_normal(y.shape)
        y = y + res
    elif res_mode == 2:
        res = rng.standard_normal((N, Ho // 2, Wo // 2, Cout))
        y = y + O.upsample2x(res)
    if relu:
        y = np.maximum(y, 0)
    pt, pl = (O.same_pad(H, k, stride)[0], O.same_pad(W, k, stride)[0]) if padding == 'same' else (0, 0)
    wp = dev(pack_conv_kernel(w))
    got = ops.conv2d(dev(x), wp, k, k, stride, pt, pl, Ho, Wo, dev(sc), dev(sh),
                     None if res is None else dev(res), res_mode, relu, math={0: 0, 1: 1, 3: 2, 4: 3}[math])
    if math == 4:                                      # plain bf16 operands (configs[4]): exact against the oracle on bf16-rounded operands
        yb = O.conv2d_nhwc(O.to_bf16(x), O.to_bf16(w), None, stride, padding) * sc + sh
        if res_mode == 1:
            yb = yb + res
        elif res_mode == 2:
            yb = yb + O.upsample2x(res)
        close(got, np.maximum(yb, 0) if relu else yb, 3e-5)
        return
    close(got, y, 2e-5 if math < 3 else 1e-4)          # bf16x2: 16-bit-mantissa products (2^-16), TF32 would need 2e-3


WINO_CASES = [
    # N,H,W,Cin,Cout,relu,affine
    (1, 16, 16, 32, 32, True, True),          # one tile group per 8 x 16 pixels: 2 x 1 groups
    (2, 12, 20, 64, 64, True, True),          # ragged groups (6 x 10 tiles), two images, two K-chunks, two channel slices
    (1, 9, 7, 32, 96, False, True),           # odd sizes: half-empty edge tiles in both directions, three slices
    (1, 16, 16, 128, 128, False, False),      # no scale / shift (plain Conv2D, the FPN output layers)
    (2, 32, 32, 256, 64, True, True),         # eight K-chunks
    (1, 5, 40, 64, 32, True, True),           # one tile row of three, several group columns
    (2, 64, 64, 64, 256, True, True),         # 256 blocks of 64 tiles: the large-group kernel (wino64), four K-chunks of 16
    (1, 71, 119, 32, 288, False, True),       # wino64 with ragged 8 x 8 groups and half-empty edge tiles, two K-chunks, nine slices
    (1, 128, 128, 96, 32, True, False),       # 64 items of 64 tiles < 256 CUs: wino32 (128 items), three 32-channel pairs (odd), one slice
]


@pytest.mark.parametrize("products", ["f32", "b3"])
@pytest.mark.parametrize("case", WINO_CASES)
def test_conv2d_winograd_matches_oracle_and_direct(ops, case, products):
    """dc_conv_desc.w_wino: a 3x3 / stride 1 / 'same' layer in the Winograd F(2x2, 3x3) form (fp32 transforms, fp32 MFMA, 16 products
    per 2x2 tile instead of 36) against the float64 oracle at the direct kernel's own tolerance, and against the direct kernel.
    products = 'b3' (dc_conv_desc.w_wino_b3, round 5): the same form with the products on the BF16 matrix pipe in split arithmetic
    (U pre-split into three bf16 pieces, V split in registers, six MFMA products per fp32 product) -- held to the SAME tolerance."""
    from image_captioning_amd.packing import pack_conv_kernel
    N, H, W, Cin, Cout, relu, affine = case
    rng = np.random.default_rng(sum(int(v) * (i + 3) for i, v in enumerate(case)))
    x = rng.standard_normal((N, H, W, Cin))
    w = rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)
    sc, sh = (rng.uniform(0.5, 1.5, Cout), rng.standard_normal(Cout)) if affine else (None, None)
    y = O.conv2d_nhwc(x, w, None, 1, 'same')
    if affine:
        y = y * sc + sh
    if relu:
        y = np.maximum(y, 0)
    wp = dev(pack_conv_kernel(w))
    if products == "b3":
        u = ops.winograd_pack_b3(wp, Cin, Cout)
        assert u.shape == (48 * Cin * Cout,) and u.dtype == torch.int16
        kw = dict(w_wino_b3=u)
    else:
        u = ops.winograd_pack(wp, Cin, Cout)
        assert u.shape == (16 * Cin * Cout,)
        kw = dict(w_wino=u)
    args = (dev(x), wp, 3, 3, 1, 1, 1, H, W, None if sc is None else dev(sc), None if sh is None else dev(sh), None, 0, relu)
    out = torch.full((N, H, W, Cout), float("nan"), device="cuda")
    got = ops.conv2d(*args, out=out, **kw)
    th, tw = (H + 1) // 2, (W + 1) // 2
    big = N * ((th + 7) // 8) * ((tw + 7) // 8) * (Cout // 32) >= 256
    # fp32 products: 64-tile items where every CU gets one; split-bf16 products: 32-tile items everywhere (round 6, conv_winograd_tiles)
    assert ops.conv2d_kernel_name(*args, **kw) == ("wino32b_kernel" if products == "b3" else ("wino64_kernel" if big else "wino32_kernel"))
    close(got, y, 2e-5)
    direct = ops.conv2d(*args)
    assert not ops.conv2d_kernel_name(*args).startswith("wino")
    close(got, direct.cpu().numpy().astype(np.float64), 2e-5)


WINO_PERSISTENT_CASES = [
    # N,H,W,Cin,Cout -- the benchmark's own layer shapes at two images: every persistent wino64 block walks
    # SEVERAL work items, with an even number KP of 32-channel pairs (the buffer hand-over conv_wino.hip makes between items)
    (2, 256, 256, 64, 64),       # res2_2b:  1024 items = 4 per block, KP 2
    (2, 128, 128, 128, 128),     # res3_2b:   512 items = 2 per block, KP 4
    (2, 128, 128, 256, 256),     # fpn_p3:   1024 items = 4 per block, KP 8
    (2, 256, 256, 256, 256),     # fpn_p2:   4096 items = 16 per block, KP 8
    (2, 128, 128, 96, 128),      # odd KP = 3 with 2 items per block (every item starts in buffer 0 again)
    (3, 104, 88, 160, 96),       # ragged: 7 x 6 groups x 3 images x 3 slices = 378 items: blocks with 1 AND 2 items, KP 5, half-empty edge groups
]


def _wino_items_per_block(N, H, W, Cout, tiles=64):
    """Work items of a layer and items per persistent block: 64-tile items on 256 blocks (one per CU), 32-tile items (8 x 4 tiles) on 512."""
    th, tw = (H + 1) // 2, (W + 1) // 2
    items = N * ((th + (7 if tiles == 64 else 3)) // (8 if tiles == 64 else 4)) * ((tw + 7) // 8) * (Cout // 32)
    return items, items / (256.0 if tiles == 64 else 512.0)


@pytest.mark.parametrize("products", ["b3", "f32"])
@pytest.mark.parametrize("case", WINO_PERSISTENT_CASES)
def test_conv2d_winograd_persistent_multi_item_blocks_at_benchmark_shapes(ops, case, products):
    """The kernel behind the headline number on the code path the headline runs (products = 'b3': wino32b_kernel since round 6 -- the
    split-bf16 products of the default plan on 32-tile items, two persistent blocks per CU; 'f32': wino64_kernel): >= 2 work items per
    persistent block (next-item patch prefetch, M image in buffer 1 while buffer 0 refills, even / odd KP).  Whole tensor against the direct
    implicit-GEMM kernel (2e-5), and against the float64 oracle on windows that straddle image corners, image edges, the
    16-pixel item boundaries and the image-0 / image-1 seam, over ALL output channels (every slice of every sampled item)."""
    N, H, W, Cin, Cout = case
    items, per_block = _wino_items_per_block(N, H, W, Cout, 32 if products == "b3" else 64)
    assert items >= 256 and per_block > 1.0, "case does not reach the multi-item loop: %d items" % items
    g = torch.Generator(device="cuda").manual_seed(N * 1000003 + H * 1009 + Cin * 31 + Cout)
    x = torch.randn(N, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, 9 * Cin, device="cuda", generator=g) / (9 * Cin) ** 0.5            # packed [Cout][tap][Cin]
    sc = torch.rand(Cout, device="cuda", generator=g) + 0.5
    sh = torch.randn(Cout, device="cuda", generator=g)
    kw = dict(w_wino_b3=ops.winograd_pack_b3(w, Cin, Cout)) if products == "b3" else dict(w_wino=ops.winograd_pack(w, Cin, Cout))
    args = (x, w, 3, 3, 1, 1, 1, H, W, sc, sh, None, 0, True)
    assert ops.conv2d_kernel_name(*args, **kw) == ("wino32b_kernel" if products == "b3" else "wino64_kernel")
    out = torch.full((N, H, W, Cout), float("nan"), device="cuda")
    got = ops.conv2d(*args, out=out, **kw)
    assert bool(torch.isfinite(got).all())
    direct = ops.conv2d(*args)
    assert not ops.conv2d_kernel_name(*args).startswith("wino")
    scale = max(1.0, float(direct.abs().max()))
    assert float((got - direct).abs().max()) / scale < 2e-5
    # float64 oracle on windows
    xh = x.cpu().numpy().astype(np.float64)
    wk = w.cpu().numpy().astype(np.float64).reshape(Cout, 3, 3, Cin).transpose(1, 2, 3, 0)     # HWIO
    sch, shh = sc.cpu().numpy().astype(np.float64), sh.cpu().numpy().astype(np.float64)
    gh = got.cpu().numpy().astype(np.float64)
    S = 12
    ys = sorted({0, H - S, max(0, 16 - S // 2), max(0, (H // 32) * 16 - S // 2), max(0, H - 16 - S // 2)})
    xs = sorted({0, W - S, max(0, 16 - S // 2), max(0, (W // 32) * 16 - S // 2), max(0, W - 16 - S // 2)})
    worst = 0.0
    for n in sorted({0, N - 1}):
        for y0 in ys:
            for x0 in xs:
                y1, x1 = min(H, y0 + S), min(W, x0 + S)
                patch = np.zeros((1, y1 - y0 + 2, x1 - x0 + 2, Cin))
                sy0, sx0, sy1, sx1 = max(0, y0 - 1), max(0, x0 - 1), min(H, y1 + 1), min(W, x1 + 1)
                patch[0, sy0 - (y0 - 1):sy1 - (y0 - 1), sx0 - (x0 - 1):sx1 - (x0 - 1)] = xh[n, sy0:sy1, sx0:sx1]
                want = np.maximum(O.conv2d_nhwc(patch, wk, None, 1, 'valid')[0] * sch + shh, 0)
                worst = max(worst, float(np.abs(gh[n, y0:y1, x0:x1] - want).max()))
    assert worst / scale < 2e-5, "windows vs float64 oracle: %.3e" % (worst / scale)


@pytest.mark.parametrize("force", ["32", "64"])
def test_conv2d_winograd_forced_kernels_in_a_child_process(ops, force):
    """DCAP_WINO_TILES (read once per process) forces the 32- / 64-tile items whatever the layer's size and product form (the 64-tile
    split-bf16 kernel wino64b runs only when forced since round 6: it incl. its multi-item loop is covered here).  Each against the
    direct kernel."""
    import subprocess
    import sys
    code = (
        "import numpy as np, torch\n"
        "from image_captioning_amd import ops\n"
        "torch.manual_seed(0)\n"
        "for (N, H, W, Cin, Cout) in ((1, 12, 20, 64, 64), (2, 33, 17, 32, 96), (2, 128, 128, 128, 128)):\n"
        "    x = torch.randn(N, H, W, Cin, device='cuda'); w = torch.randn(Cout, 9 * Cin, device='cuda') / (9 * Cin) ** 0.5\n"
        "    sh = torch.randn(Cout, device='cuda')\n"
        "    b = ops.conv2d(x, w, 3, 3, 1, 1, 1, H, W, None, sh, None, 0, True)\n"
        "    for kw in (dict(w_wino=ops.winograd_pack(w, Cin, Cout)), dict(w_wino_b3=ops.winograd_pack_b3(w, Cin, Cout))):\n"
        "        a = ops.conv2d(x, w, 3, 3, 1, 1, 1, H, W, None, sh, None, 0, True, **kw)\n"
        "        name = ops.conv2d_kernel_name(x, w, 3, 3, 1, 1, 1, H, W, None, sh, None, 0, True, **kw)\n"
        "        err = float((a - b).abs().max() / b.abs().max())\n"
        "        assert err < 2e-5, (name, err)\n"
        "        print(name, err)\n")
    env = dict(os.environ, DCAP_WINO_TILES=force)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert ("wino64_kernel" if force == "64" else "wino32_kernel") in r.stdout and ("wino64b_kernel" if force == "64" else "wino32b_kernel") in r.stdout


@pytest.mark.parametrize("products", ["b3", "f32"])
@pytest.mark.parametrize("case", [(8192, 256, 1024, 256, True, True), (1000, 256, 1024, 256, True, False), (4096, 128, 512, 128, True, True),
                                  (77, 64, 512, 128, False, True), (33, 1024, 1024, 256, False, False), (200, 160, 512, 128, True, True),
                                  (16384, 64, 256, 64, True, True), (1003, 64, 256, 64, True, False), (40, 64, 256, 64, False, True)])
def test_pw_chain_two_pointwise_convolutions_in_one_launch(ops, case, products):
    """dc_pw_chain_f32 (round 5): a bottleneck's last convolution (1x1 + folded BN + shortcut + ReLU) and the next block's first (1x1 +
    folded BN + ReLU) in one launch -- both outputs against the float64 oracle at the convolution kernels' tolerance and against the two
    separate dc_conv2d_nhwc_f32 launches the plan used to make.  Cases: the stage-4 shape at two images (256 blocks = one per CU), a
    ragged pixel count (the last block has 8 rows), the stage-3 shape, tiny M, K1 = N1, a K1 whose 16-channel groups do not fill the
    weight ring (10 groups).  products = 'b3': both layers on the bf16 pipe in split arithmetic (dc_pw_chain_pack_b3) -- SAME tolerance.
    The 64 -> 256 -> 64 cases are the stage-2 seam: the streaming form (pw_chain_stream_kernel: layer 1's accumulators are layer 2's
    B operands), fp32 products only -- two strips per wave, a ragged last strip, fewer strips than waves."""
    M, K1, N1, N2, res, sc = case
    if N1 == 256 and products == "b3":
        with pytest.raises(Exception):                     # the bandwidth-bound seam takes fp32 products
            ops.pw_chain(dev(np.zeros((M, K1))), ops.pw_chain_pack_b3(dev(np.zeros((N1, K1)))), dev(np.zeros(N1)),
                         ops.pw_chain_pack_b3(dev(np.zeros((N2, N1)))), dev(np.zeros(N2)))
        return
    rng = np.random.default_rng(M + K1 + N1)
    x = rng.standard_normal((M, K1))
    w1 = rng.standard_normal((N1, K1)) / np.sqrt(K1)
    w2 = rng.standard_normal((N2, N1)) / np.sqrt(N1)
    s1, h1 = (rng.uniform(0.5, 1.5, N1) if sc else None), rng.standard_normal(N1)
    s2, h2 = (rng.uniform(0.5, 1.5, N2) if sc else None), rng.standard_normal(N2)
    r = rng.standard_normal((M, N1)) if res else None
    y = x @ w1.T * (1.0 if s1 is None else s1) + h1
    if r is not None:
        y = y + r
    y = np.maximum(y, 0)
    z = np.maximum(y @ w2.T * (1.0 if s2 is None else s2) + h2, 0)
    assert ops.pw_chain_supported(K1, N1, N2) and not ops.pw_chain_supported(K1, 256, 128) and not ops.pw_chain_supported(K1 + 16, N1, N2)
    xd, w1d, w2d = dev(x), dev(w1), dev(w2)
    pack = ops.pw_chain_pack_b3 if products == "b3" else ops.pw_chain_pack
    w1f, w2f = pack(w1d), pack(w2d)
    assert w1f.dtype == (torch.int16 if products == "b3" else torch.float32)
    yo = torch.full((M, N1), float("nan"), device="cuda")
    zo = torch.full((M, N2), float("nan"), device="cuda")
    gy, gz = ops.pw_chain(xd, w1f, dev(h1), w2f, dev(h2), scale1=None if s1 is None else dev(s1), scale2=None if s2 is None else dev(s2),
                          residual=None if r is None else dev(r), y=yo, z=zo)
    close(gy, y, 2e-5)
    close(gz, z, 2e-5)
    # the two launches it replaces (M pixels as one image row)
    x4 = xd.view(1, 1, M, K1)
    y2 = ops.conv2d(x4, w1d, 1, 1, 1, 0, 0, 1, M, None if s1 is None else dev(s1), dev(h1), None if r is None else dev(r).view(1, 1, M, N1), 1 if res else 0, True)
    z2 = ops.conv2d(y2, w2d, 1, 1, 1, 0, 0, 1, M, None if s2 is None else dev(s2), dev(h2), None, 0, True)
    close(gy, y2.view(M, N1).cpu().numpy().astype(np.float64), 2e-5)
    close(gz, z2.view(M, N2).cpu().numpy().astype(np.float64), 2e-5)
    with pytest.raises(Exception):
        ops.pw_chain(xd, w1f, dev(h1), pack(dev(rng.standard_normal((96, N1)))), dev(np.zeros(96)))     # N2 = 96: not a covered shape


def test_conv2d_winograd_falls_back_where_the_form_does_not_apply(ops):
    """w_wino on a layer the Winograd kernel does not cover (a residual, a stride, another arithmetic): the direct kernels run."""
    from image_captioning_amd.packing import pack_conv_kernel
    rng = np.random.default_rng(5)
    x, w = rng.standard_normal((1, 8, 8, 32)), rng.standard_normal((3, 3, 32, 32)) / 17.0
    wp = dev(pack_conv_kernel(w))
    u = ops.winograd_pack(wp, 32, 32)
    res = rng.standard_normal((1, 8, 8, 32))
    got = ops.conv2d(dev(x), wp, 3, 3, 1, 1, 1, 8, 8, None, None, dev(res), 1, False, w_wino=u)
    close(got, O.conv2d_nhwc(x, w, None, 1, 'same') + res, 2e-5)
    assert not ops.conv2d_kernel_name(dev(x), wp, 3, 3, 1, 1, 1, 8, 8, None, None, dev(res), 1, False, w_wino=u).startswith("wino")
    assert not ops.conv2d_kernel_name(dev(x), wp, 3, 3, 1, 1, 1, 8, 8, math=1, w_wino=u).startswith("wino")
    with pytest.raises(DcapError):
        ops.winograd_pack(wp, 32, 48)                      # not this kernel's size
    with pytest.raises(DcapError):
        ops.conv2d(dev(x), wp, 3, 3, 1, 1, 1, 8, 8, w_wino=u[:100])


def test_split_bf16x3_pieces(ops):
    """x = p0 + p1 + p2 to within 2^-25 |x|, every piece a bf16 (low 16 bits of its fp32 pattern zero)."""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(4097) * np.exp(rng.uniform(-20, 20, 4097))).astype(np.float32)
    x[:4] = [0.0, 1.0, -2.5, 2.0 ** -120]
    planes = ops.split_bf16x3(dev(x)).cpu().numpy().view(np.uint16).astype(np.uint32)
    pieces = (planes << 16).view(np.float32).astype(np.float64)
    err = np.abs(pieces.sum(0) - x.astype(np.float64))
    assert np.all(err <= np.abs(x.astype(np.float64)) * 2.0 ** -24 + 1e-44)
    assert np.all(np.abs(pieces[1]) <= np.abs(pieces[0]) * 2.0 ** -7 + 1e-44)


def test_conv2d_bf16x3_error_is_fp32_grade(ops):
    """The split-bf16 path against the exact-fp32 path on a long reduction (3x3x512): both err against float64 by a few
    1e-7 relative to the output scale; a two-piece split (1e-5) or a dropped product would fail this by orders of magnitude.
    Includes operands spanning 12 orders of magnitude and exact powers of two."""
    from image_captioning_amd.packing import pack_conv_kernel
    rng = np.random.default_rng(99)
    x = rng.standard_normal((1, 16, 16, 512)) * np.exp(rng.uniform(-14, 14, (1, 16, 16, 512)))
    x[0, 0, 0, :8] = [1.0, 2.0, 0.5, 1 + 2.0 ** -23, 3.0, 65536.0, 2.0 ** -20, 0.0]
    w = rng.standard_normal((3, 3, 512, 128)) / np.sqrt(9 * 512)
    want = O.conv2d_nhwc(x.astype(np.float32), w.astype(np.float32), None, 1, 'same')
    scale = np.abs(want).max()
    errs = []
    for math in (0, 1):
        got = ops.conv2d(dev(x), dev(pack_conv_kernel(w)), 3, 3, 1, 1, 1, 16, 16, math=math).cpu().numpy().astype(np.float64)
        errs.append(np.abs(got - want).max() / scale)
    assert errs[0] < 5e-6 and errs[1] < 5e-6 and errs[1] < 4 * errs[0] + 2e-7, errs


@pytest.mark.parametrize("math", [0, 1], ids=["f32", "bf16x3"])
@pytest.mark.parametrize("N,H,W", [(1, 32, 32), (2, 64, 96)])
def test_stem_mold_maxpool(ops, N, H, W, math):
    from image_captioning_amd.packing import pack_stem_kernel
    rng = np.random.default_rng(H)
    img = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    mean = [123.7, 116.8, 103.9]
    w = rng.standard_normal((7, 7, 3, 64)) / 12.0
    sc, sh = rng.uniform(0.5, 1.5, 64), rng.standard_normal(64)
    x = O.mold_image(img, mean)
    rgbx = ops.mold_image_rgbx(dev(img, torch.uint8), mean)
    close(rgbx[..., :3], x, 1e-6)
    assert float(rgbx[..., 3].abs().max()) == 0.0
    y = np.maximum(O.conv2d_nhwc(x, w, None, 2, (3, 3, 3, 3)) * sc + sh, 0)
    got = ops.conv2d(rgbx, dev(pack_stem_kernel(w)), 7, 7, 2, 3, 3, H // 2, W // 2, dev(sc), dev(sh), None, 0, True, math=math)
    close(got, y)
    close(ops.maxpool3x3s2_same(got), O.maxpool3x3s2_same(got.cpu().numpy().astype(np.float64)), 1e-7)


def test_roi_align_pyramid(ops):
    from image_captioning_amd import synth
    rng = np.random.default_rng(3)
    B, R, C, S = 2, 40, 256, 512
    maps = [rng.standard_normal((B, S // s, S // s, C)) for s in (4, 8, 16, 32)]
    rois = synth.rois(9, B, R, S, S, lo=16, hi=512)
    rois[0, 0] = [0, 0, S, S]
    rois[0, 1] = [100, 100, 100, 180]          # zero-area box
    rois[1, 0] = [S - 40, S - 30, S, S]        # touches the border
    boxes = O.normalize_boxes(rois, S, S)
    want = O.pyramid_roi_align(boxes, maps, (S, S, 3), 7)
    lv = torch.empty(B * R, dtype=torch.int32, device="cuda")
    got = ops.roi_align_pyramid([dev(m) for m in maps], dev(boxes), S * S, 7, levels_out=lv)
    np.testing.assert_array_equal(lv.cpu().numpy().reshape(B, R), O.roi_levels(boxes, (S, S, 3)))     # bit-exact routing
    assert set(lv.cpu().numpy().tolist()) == {2, 3, 4, 5}
    close(got, want, 1e-5)


def test_roi_align_boxes_clipped_to_the_image_border(ops):
    """Boxes that the ProposalLayer clipped to the image (y2 or x2 exactly 1): the last sample row/column lands ON the last pixel when
    every float operation is rounded on its own, as TF's kernel does -- y1*(H-1) + 6*((1-y1)*(H-1)/6) -- and for about one such box in
    twenty-five just OUTSIDE (extrapolation value 0) when the multiply is fused into the add.  Round 6 met exactly that on the box
    below once the build flags changed which multiplies the compiler fused; the file now forbids the fusion (csrc/roialign.hip)."""
    rng = np.random.default_rng(11)
    B, R, C, S = 1, 600, 256, 128
    maps = [rng.standard_normal((B, S // s, S // s, C)) + 3.0 for s in (4, 8, 16, 32)]       # far from 0: an extrapolated sample shows
    boxes = np.zeros((B, R, 4), np.float32)
    y1, x1 = rng.random(R) * 0.7, rng.random(R) * 0.7
    boxes[0, :, 0], boxes[0, :, 1] = y1, x1
    boxes[0, :, 2] = np.where(np.arange(R) % 3 != 1, 1.0, y1 + 0.05 + rng.random(R) * 0.25)
    boxes[0, :, 3] = np.where(np.arange(R) % 3 != 0, 1.0, x1 + 0.05 + rng.random(R) * 0.25)
    boxes[0, 0] = np.array([float.fromhex('0x1.c84c6ap-2'), float.fromhex('0x1.17b2bcp-3'), 1.0, float.fromhex('0x1.de575ep-2')], np.float32)
    want = O.pyramid_roi_align(boxes, maps, (S, S, 3), 7)
    got = ops.roi_align_pyramid([dev(m) for m in maps], dev(boxes), S * S, 7)
    close(got, want, 1e-5)
    assert float(np.abs(want[0, 0, 6]).min()) > 0        # the box of round 6: its last row is inside the map


@pytest.mark.parametrize("B,T,I,U,masked", [(3, 4, 8, 4, True), (64, 10, 300, 128, True), (8, 15, 64, 512, False), (5, 1, 2048, 256, False),
                                            (70, 6, 32, 512, True), (200, 4, 16, 256, True), (37, 3, 16, 16, True), (300, 3, 16, 512, False),
                                            (64, 15, 300, 1024, True),      # the headline's word-LSTM (64 captions x 15 tokens, 300 -> 1024)
                                            (960, 1, 2048, 256, False)])    # its inject-LSTM: one step over all 960 (caption, prefix) rows
def test_lstm_seq_forward_backward(ops, B, T, I, U, masked):
    rng = np.random.default_rng(B + T + U)
    x = rng.standard_normal((B, T, I))
    W = rng.standard_normal((I, 4 * U)) / np.sqrt(I)
    Ur = rng.standard_normal((U, 4 * U)) / np.sqrt(U)
    b = 0.1 * rng.standard_normal(4 * U)
    mask = None
    if masked:
        mask = rng.random((B, T)) > 0.3
        mask[0] = False                                   # a fully masked row
        mask[1] = True
    H, cache = O.lstm_forward(x, mask, W, Ur, b)
    dH = rng.standard_normal((B, T, U))
    dlast = rng.standard_normal((B, U))
    dx, dW, dU, db = O.lstm_backward(dH, cache, dh_last=dlast)
    # device: time-major rows t*B+b
    xt = dev(np.transpose(x, (1, 0, 2)).reshape(T * B, I))
    z = ops.gemm(xt, dev(W), shift=dev(b))
    mk = None if mask is None else dev(mask.T.reshape(-1), torch.uint8)
    Ud = dev(Ur)
    h_seq, c_seq = ops.lstm_seq_fwd(z, Ud, mk, B, T)
    close(h_seq.view(T, B, U).permute(1, 0, 2), H)
    dz, dUd = ops.lstm_seq_bwd(z, Ud, mk, h_seq, c_seq, B, T,
                               dh_seq=dev(np.transpose(dH, (1, 0, 2)).reshape(T * B, U)), dh_last=dev(dlast))
    close(dUd, dU, 5e-5)
    close(ops.gemm(xt, dz, a_trans=True), dW, 5e-5)                          # dkernel = x^T dz
    close(ops.colsum(dz), db, 5e-5)
    close(ops.gemm(dz, dev(W), b_trans=True).view(T, B, I).permute(1, 0, 2), dx, 5e-5)   # dx = dz kernel^T


@pytest.mark.parametrize("B,T,I,U", [(6, 5, 16, 32), (64, 10, 64, 512), (200, 4, 32, 128), (300, 3, 16, 512), (5, 3, 8, 24), (200, 15, 16, 512),
                                     (64, 3, 16, 1024), (8, 10, 16, 512)])
def test_lstm_recurrent_dropout_matches_oracle(ops, B, T, I, U):
    """Keras recurrent_dropout in the training phase: given the four per-gate masks, forward states and every gradient of the
    recurrence equal the oracle's (which finite differences pin, tests/test_oracle_kat.py).  ONE fused launch per timestep
    in both directions with masks too (forward, round 5: lstm_step_masked_kernel -- a wave per gate and K half, 16 x 16 tiles, the masks
    applied to the h fragments in registers; backward: the K = 4U split falls on gate boundaries, the masks apply where the partial tiles
    meet); the cases cover 16- and 32-row forward blocks (B = 300 at U = 512), ragged row tiles, U = 1024, BASELINE configs[0]'s own
    shape (8 x 10 x 512) and a U that is no multiple of 16 (the per-gate GEMM path)."""
    rng = np.random.default_rng(B + T + U)
    x = rng.standard_normal((B, T, I))
    W = rng.standard_normal((I, 4 * U)) / np.sqrt(I)
    Ur = rng.standard_normal((U, 4 * U)) / np.sqrt(U)
    b = 0.1 * rng.standard_normal(4 * U)
    mask = rng.random((B, T)) > 0.25
    mask[0] = False
    mask[1] = True
    rm = O.recurrent_dropout_masks(rng, B, U, 0.2)
    H, cache = O.lstm_forward(x, mask, W, Ur, b, rec_masks=rm)
    dH = rng.standard_normal((B, T, U))
    dx, dW, dU, db = O.lstm_backward(dH, cache)
    xt = dev(np.transpose(x, (1, 0, 2)).reshape(T * B, I))
    z = ops.gemm(xt, dev(W), shift=dev(b))
    mk = dev(mask.T.reshape(-1), torch.uint8)
    Ud, rmd = dev(Ur), dev(rm)
    h_seq, c_seq = ops.lstm_seq_fwd(z, Ud, mk, B, T, rec_masks=rmd)
    close(h_seq.view(T, B, U).permute(1, 0, 2), H)
    dz, dUd = ops.lstm_seq_bwd(z, Ud, mk, h_seq, c_seq, B, T, dh_seq=dev(np.transpose(dH, (1, 0, 2)).reshape(T * B, U)), rec_masks=rmd)
    close(dUd, dU, 5e-5)
    close(ops.gemm(xt, dz, a_trans=True), dW, 5e-5)
    close(ops.colsum(dz), db, 5e-5)
    ones = torch.ones_like(rmd)                                        # masks of ones == the plain (fused) recurrence
    z2 = ops.gemm(xt, dev(W), shift=dev(b))
    z3 = z2.clone()
    h_a, _ = ops.lstm_seq_fwd(z2, Ud, mk, B, T, rec_masks=ones)
    h_b, _ = ops.lstm_seq_fwd(z3, Ud, mk, B, T)
    close(h_a, h_b.cpu().numpy(), 1e-5)


@pytest.mark.parametrize("M,V", [(7, 1000), (64, 10000), (5, 1003)])
def test_softmax_ce(ops, M, V):
    rng = np.random.default_rng(V)
    z = 3.0 * rng.standard_normal((M, V))
    t = rng.integers(0, V, M)
    z[0, t[0]] = -60.0                                    # target probability below 1e-7: clipped row
    z[1, :] = -50.0
    z[1, t[1]] = 50.0                                     # target probability above 1-1e-7: clipped row
    p = O.softmax(z)
    ld = (V + 3) // 4 * 4
    zd = torch.zeros(M, ld, device="cuda")
    zd[:, :V] = dev(z)
    probs = torch.empty_like(zd)
    dl = torch.empty_like(zd)
    loss = torch.empty(M, device="cuda")
    ops.softmax_ce(zd[:, :V], dev(t, torch.int32), probs[:, :V], loss, dl[:, :V], grad_scale=1.0 / M)
    close(probs[:, :V], p, 1e-6)
    close(loss, O.categorical_crossentropy(t, p), 1e-5)
    close(dl[:, :V], O.softmax_ce_grad_logits(t, p, np.full(M, 1.0 / M)), 1e-6)
    assert float(dl[0, :V].abs().max()) == 0.0 and float(dl[1, :V].abs().max()) == 0.0
    close(ops.mean(loss), np.array([O.categorical_crossentropy(t, p).mean()]), 1e-5)


def test_softmax_ce_in_place_equals_out_of_place(ops):
    """dlogits may alias logits (dcap.h): the in-place call must give the bits of the out-of-place one, every time
    (a wave that finished its reductions early used to overwrite z[target] before a slower wave had read it)."""
    rng = np.random.default_rng(77)
    M, V = 960, 10000
    z = dev(3.0 * rng.standard_normal((M, V)))
    t = dev(rng.integers(0, V, M), torch.int32)
    rw = dev(rng.uniform(0.0, 1.0, M))
    for sparse in (False, True):
        ref_dl, ref_loss = torch.empty_like(z), torch.empty(M, device="cuda")
        ops.softmax_ce(z, t, None, ref_loss, ref_dl, grad_scale=1.0 / M, row_weights=rw, keras_sparse=sparse)
        for _ in range(20):
            zz, loss = z.clone(), torch.empty(M, device="cuda")
            ops.softmax_ce(zz, t, None, loss, zz, grad_scale=1.0 / M, row_weights=rw, keras_sparse=sparse)
            assert torch.equal(zz, ref_dl) and torch.equal(loss, ref_loss)


def test_sumsq_is_bit_reproducible(ops):
    x = dev(np.random.default_rng(3).standard_normal(10_000_019))
    first = ops.sumsq(x).clone()
    for _ in range(5):
        assert torch.equal(ops.sumsq(x), first)
    close(first, np.array([float((x.double() ** 2).sum())]), 1e-5)


def test_argmax_lowest_index_wins_ties(ops):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((33, 10000)).astype(np.float32)
    x[3, 77] = x[3, 9000] = 99.0
    x[4, :] = 1.0
    x[5, 9999] = 50.0
    got = ops.argmax_rows(dev(x)).cpu().numpy()
    np.testing.assert_array_equal(got, np.argmax(x, axis=1))
    assert got[3] == 77 and got[4] == 0 and got[5] == 9999


def test_colsum_sumsq(ops):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((960, 1000))
    close(ops.colsum(dev(x)), x.sum(0), 1e-5)
    acc = dev(np.ones(1000))
    ops.colsum(dev(x), out=acc, accumulate=True)
    close(acc, x.sum(0) + 1, 1e-5)
    close(ops.sumsq(dev(x)), np.array([(x ** 2).sum()]), 1e-5)


@pytest.mark.parametrize("clip", [None, 0.5])
def test_amsgrad_trajectory(ops, clip):
    rng = np.random.default_rng(8)
    n = 10007
    p0 = rng.standard_normal(n)
    p = dev(p0)
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    vh = torch.zeros(n, device="cuda")
    P, Mm, Vv, Vh = p0.copy(), 0.0, 0.0, 0.0
    for t in range(1, 6):
        g = rng.standard_normal(n) * (0.3 ** t)
        gg = g
        if clip is not None:
            (gg,), _ = O.clip_by_global_norm([g], clip)
        P, Mm, Vv, Vh = O.amsgrad_step(P, gg, Mm, Vv, Vh, t)
        gd = dev(g)
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        ops.amsgrad_step(p, gd, m, v, vh, lr_t, gnorm_sq=ops.sumsq(gd) if clip else None, clipnorm=clip or 0.0)
    close(p, P, 1e-6)
    close(vh, Vh, 1e-5)


def test_amsgrad_with_the_regulariser_inside_the_update(ops):
    """dc_reg_sumsq_f32 + dc_amsgrad_step_f32(reg=segments) (round 5: the joint model's L2 term, trainable mask and clip norm without
    rewriting the gradient bucket) against the float64 oracle -- g' = g mask + 2 coef w, clip by the global norm of g', AMSGrad -- and
    against the three-pass path it replaces (dc_l2_reg_f32, dc_sumsq_f32, dc_amsgrad_step_f32).  Segments end inside float4 vectors
    (the fused RPN head's bias: 6 + 12 + 2 elements) and the bucket length is no multiple of 4."""
    rng = np.random.default_rng(9)
    bounds = [0, 6, 18, 20, 1021, 1024, 5000, 5003, 9001, 10007]
    n = bounds[-1]
    coef = np.zeros(n, np.float32)
    mask = np.ones(n, np.float32)
    for s_, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
        coef[lo:hi] = 0.0 if s_ % 3 == 2 else 1e-4 / (hi - lo)
        mask[lo:hi] = 0.0 if s_ in (3, 6) else 1.0
    for use_mask in (True, False):
        mk = mask if use_mask else None
        segs = ops.RegSegmentTable(coef, mk, "cuda")
        assert segs.nseg <= len(bounds) - 1 and segs.n == n
        p0 = rng.standard_normal(n)
        pa, pb = dev(p0), dev(p0)
        st_a = [torch.zeros(n, device="cuda") for _ in range(3)]
        st_b = [torch.zeros(n, device="cuda") for _ in range(3)]
        P, Mm, Vv, Vh = p0.astype(np.float32).astype(np.float64), 0.0, 0.0, 0.0
        loss, gn = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
        for t in range(1, 5):
            g = rng.standard_normal(n) * (0.5 ** t)
            g32 = g.astype(np.float32).astype(np.float64)
            gr = g32 * (1.0 if mk is None else mk) + 2.0 * coef.astype(np.float64) * P
            want_loss = float((coef.astype(np.float64) * P * P).sum())
            (gc,), _ = O.clip_by_global_norm([gr], 0.5)
            lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
            # fused: the gradient bucket is read twice and never written
            ga = dev(g)
            keep = ga.clone()
            ops.reg_sumsq(pa, ga, segs, loss=loss, gnorm_sq=gn)
            assert abs(float(loss.item()) - want_loss) < 1e-5 * max(want_loss, 1e-12)
            assert abs(float(gn.item()) - float((gr * gr).sum())) < 1e-5 * float((gr * gr).sum())
            ops.amsgrad_step(pa, ga, *st_a, lr_t, gnorm_sq=gn, clipnorm=0.5, reg=segs)
            assert torch.equal(ga, keep)
            # the path it replaces
            gb = dev(g)
            ops.l2_reg(pb, dev(coef), gb, mask=None if mk is None else dev(mk))
            ops.amsgrad_step(pb, gb, *st_b, lr_t, gnorm_sq=ops.sumsq(gb), clipnorm=0.5)
            P, Mm, Vv, Vh = O.amsgrad_step(P, gc, Mm, Vv, Vh, t)
        close(pa, P, 1e-6)
        close(st_a[2], Vh, 1e-5)
        assert float((pa - pb).abs().max()) < 1e-6 * float(pb.abs().max())
    with pytest.raises(Exception):
        ops.amsgrad_step(pa[:100], ga[:100], *[t_[:100] for t_ in st_a], 1e-3, reg=segs)       # a table for another bucket


def test_subsample2(ops):
    x = np.random.default_rng(0).standard_normal((2, 8, 6, 8))
    close(ops.subsample2(dev(x)), O.subsample2(x), 1e-7)


@pytest.mark.parametrize("B,S", [(1, 256), (2, 128)])
def test_rpn_proposals_bit_exact_given_scores(ops, B, S):
    """ProposalLayer: the top-k order (ties included) and the NMS survivors are bit-exact against the float32
    oracle when both start from the same fp32 scores (the device's own softmax output); box coordinates agree to
    the last bit of exp()."""
    rng = np.random.default_rng(S)
    strides, scales, ratios = [4, 8, 16, 32, 64], (32, 64, 128, 256, 512), [0.5, 1, 2]
    shapes = [[S // s, S // s] for s in strides]
    heads = [rng.standard_normal((B, h, w, 18)).astype(np.float32) for h, w in shapes]
    for hd in heads:
        hd[..., 6:] *= 0.5
    heads[0][0, 0, 0, :6] = heads[0][0, 0, 1, :6]              # exact score ties between neighbouring anchors
    anchors = O.generate_pyramid_anchors(scales, ratios, shapes, strides, 1).astype(np.float32)
    count, pre = 100, 600
    props, (scores, order, keep) = ops.rpn_proposals([dev(h) for h in heads], dev(anchors), (S, S), count, 0.7,
                                                     pre_nms_limit=pre, debug=True)
    scores, order, keep, props = scores.cpu().numpy(), order.cpu().numpy(), keep.cpu().numpy(), props.cpu().numpy()
    cls = np.concatenate([h[..., :6].reshape(B, -1, 2) for h in heads], axis=1)
    box = np.concatenate([h[..., 6:].reshape(B, -1, 4) for h in heads], axis=1)
    assert np.abs(scores - O.softmax(cls)[:, :, 1]).max() < 1e-6
    for b in range(B):
        want, ix, kp = O.proposal_layer(scores[b], box[b], anchors, (S, S), count, 0.7, pre_nms_limit=pre)
        np.testing.assert_array_equal(order[b], ix)
        np.testing.assert_array_equal(keep[b][:len(kp)], kp)
        assert np.all(keep[b][len(kp):] == -1)
        # box values: the device's expf and numpy's float32 exp may differ in the last bit
        np.testing.assert_allclose(props[b], want, rtol=3e-7, atol=1e-7)
        assert np.all(props[b][len(kp):] == 0)                       # zero padding


@pytest.mark.parametrize("kind,pre", [("all_equal", 600), ("few_values", 1000), ("random", 9000), ("few_values", 12000), ("all", 10 ** 6)])
def test_rpn_topk_order_ties_and_sizes(ops, kind, pre):
    """The hand-written top-k (radix select + ordered compaction + bitonic sort) against a stable descending argsort of the
    device's own scores: massive ties (all scores equal; a handful of distinct values), k above the in-LDS sort's 8192, and
    k = every anchor."""
    S, B = 256, 2
    rng = np.random.default_rng(len(kind) + pre)
    strides, scales, ratios = [4, 8, 16, 32, 64], (32, 64, 128, 256, 512), [0.5, 1, 2]
    shapes = [[S // s, S // s] for s in strides]
    heads = []
    for h, w in shapes:
        hd = (0.1 * rng.standard_normal((B, h, w, 18))).astype(np.float32)
        if kind == "all_equal":
            hd[..., :6] = 0.0
        elif kind == "few_values":
            hd[..., :6] = rng.integers(0, 3, (B, h, w, 6)).astype(np.float32)
        else:
            hd[..., :6] = rng.standard_normal((B, h, w, 6)).astype(np.float32)
        heads.append(hd)
    anchors = O.generate_pyramid_anchors(scales, ratios, shapes, strides, 1).astype(np.float32)
    A = anchors.shape[0]
    k = min(pre, A)
    props, (scores, order, keep) = ops.rpn_proposals([dev(h) for h in heads], dev(anchors), (S, S), 50, 0.7, pre_nms_limit=pre, debug=True)
    scores, order = scores.cpu().numpy(), order.cpu().numpy()
    assert order.shape == (B, k)
    for b in range(B):
        want = np.argsort(-scores[b].astype(np.float64), kind="stable")[:k]
        np.testing.assert_array_equal(order[b], want)


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 64, 3, 1), (1, 32, 32, 128, 256, 3, 1), (2, 16, 16, 256, 64, 1, 1), (1, 32, 32, 64, 128, 1, 2)])
def test_conv2d_weight_and_data_gradients(ops, case):
    """Building blocks of the joint model's trainable convs: wgrad kernel and dgrad-as-forward-conv."""
    from image_captioning_amd.packing import pack_conv_kernel, pack_conv_kernel_dgrad
    N, H, W, Cin, Cout, k, stride = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((N, H, W, Cin))
    w = rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)
    padding = 'same' if k == 3 else 'valid'
    y = O.conv2d_nhwc(x, w, None, stride, padding)
    dy = rng.standard_normal(y.shape)
    dx, dw, db = O.conv2d_nhwc_backward(x, w, dy, stride, padding)
    pad = (k - 1) // 2
    got_dw = ops.conv2d_wgrad(dev(x), dev(dy), k, k, stride, pad, pad)
    close(got_dw, pack_conv_kernel(dw), 3e-5)
    close(ops.colsum(dev(dy.reshape(-1, Cout))), db, 3e-5)
    if stride == 1:
        got_dx = ops.conv2d(dev(dy), dev(pack_conv_kernel_dgrad(w)), k, k, 1, k - 1 - pad, k - 1 - pad, H, W)
        close(got_dx, dx, 3e-5)


def test_roi_align_backward_is_the_adjoint_of_forward(ops):
    """<forward(maps), g> == <maps, backward(g)> for random maps and g (bilinear gather / scatter adjointness),
    plus a direct check of the scatter against a NumPy loop on one level."""
    from image_captioning_amd import synth
    rng = np.random.default_rng(5)
    B, R, C, S = 1, 24, 64, 256
    maps = [rng.standard_normal((B, S // s, S // s, C)) for s in (4, 8, 16, 32)]
    rois = synth.rois(7, B, R, S, S, lo=16, hi=256)
    boxes = dev(O.normalize_boxes(rois, S, S))
    g = rng.standard_normal((B, R, 7, 7, C))
    fwd = ops.roi_align_pyramid([dev(m) for m in maps], boxes, S * S, 7).cpu().numpy().astype(np.float64)
    dm = [torch.zeros(m.shape, device="cuda") for m in maps]
    ops.roi_align_pyramid_bwd(dm, boxes, S * S, dev(g), 7)
    lhs = float((fwd * g).sum())
    rhs = sum(float((d.cpu().numpy().astype(np.float64) * m).sum()) for d, m in zip(dm, maps))
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
    # the scatter itself, element by element, against the oracle's NumPy loop (all four levels)
    want = O.pyramid_roi_align_backward(O.normalize_boxes(rois, S, S), [m.shape for m in maps], (S, S), g)
    for d, w in zip(dm, want):
        close(d, w, 2e-5)


def test_roi_align_backward_of_boxes_clipped_to_the_image_border(ops):
    """The backward's sample coordinates take the same decisions as the forward's (test_roi_align_boxes_clipped_to_the_image_border):
    300 boxes whose far edge is the image border -- the last sample row / column lands ON the last pixel, or just outside, exactly as the
    oracle's float32 arithmetic says -- scattered against the oracle's loop, and <forward, g> = <maps, backward>."""
    rng = np.random.default_rng(12)
    B, R, C, S = 1, 300, 64, 128
    maps = [rng.standard_normal((B, S // s, S // s, C)) for s in (4, 8, 16, 32)]
    boxes = np.zeros((B, R, 4), np.float32)
    y1, x1 = rng.random(R) * 0.7, rng.random(R) * 0.7
    boxes[0, :, 0], boxes[0, :, 1] = y1, x1
    boxes[0, :, 2] = np.where(np.arange(R) % 3 != 1, 1.0, y1 + 0.05 + rng.random(R) * 0.25)
    boxes[0, :, 3] = np.where(np.arange(R) % 3 != 0, 1.0, x1 + 0.05 + rng.random(R) * 0.25)
    boxes[0, 0] = np.array([float.fromhex('0x1.c84c6ap-2'), float.fromhex('0x1.17b2bcp-3'), 1.0, float.fromhex('0x1.de575ep-2')], np.float32)
    g = rng.standard_normal((B, R, 7, 7, C))
    fwd = ops.roi_align_pyramid([dev(m) for m in maps], dev(boxes), S * S, 7).cpu().numpy().astype(np.float64)
    dm = [torch.zeros(m.shape, device="cuda") for m in maps]
    ops.roi_align_pyramid_bwd(dm, dev(boxes), S * S, dev(g), 7)
    lhs = float((fwd * g).sum())
    rhs = sum(float((d.cpu().numpy().astype(np.float64) * m).sum()) for d, m in zip(dm, maps))
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
    want = O.pyramid_roi_align_backward(boxes, [m.shape for m in maps], (S, S), g)
    for d, w in zip(dm, want):
        close(d, w, 2e-5)


def test_roi_align_backward_is_deterministic_and_accumulates(ops):
    """The backward gathers per destination pixel in a fixed (RoI, bin) order -- no float atomics: two runs are bit-identical even
    with hundreds of overlapping RoIs (incl. boxes that leave the image, degenerate and duplicate boxes, two images), and it ADDS to
    what the maps already hold (the joint model puts the RPN branch's data gradients there first)."""
    from image_captioning_amd import synth
    rng = np.random.default_rng(11)
    B, R, C, S = 2, 200, 256, 256
    shapes = [(B, S // s, S // s, C) for s in (4, 8, 16, 32)]
    rois = synth.rois(3, B, R, S, S, lo=8, hi=256).astype(np.float64)
    rois[:, :8] = rois[:, 8:16]                                        # duplicates: several RoIs hit the same pixels
    rois[0, 16] = [-40, -30, 90, 120]                                  # sampling points outside the map (zero rows in the forward)
    rois[1, 17] = [200, 180, 300, 320]
    rois[0, 18] = [50, 60, 50, 60]                                     # zero area
    nb = O.normalize_boxes(rois, S, S)
    boxes = dev(nb)
    g = rng.standard_normal((B, R, 7, 7, C))
    gd = dev(g)
    base = [rng.standard_normal(sh) for sh in shapes]
    runs = []
    for _ in range(2):
        dm = [dev(b) for b in base]
        ops.roi_align_pyramid_bwd(dm, boxes, S * S, gd, 7)
        runs.append([d.cpu().numpy() for d in dm])
    for a, b in zip(*runs):
        np.testing.assert_array_equal(a, b)
    want = O.pyramid_roi_align_backward(nb, shapes, (S, S), g)
    for got, w, b in zip(runs[0], want, base):
        close(got, w + b, 3e-5)


def test_downsample2x_sum_is_upsample_adjoint(ops):
    rng = np.random.default_rng(6)
    fine = rng.standard_normal((2, 8, 12, 8))
    want = fine.reshape(2, 4, 2, 6, 2, 8).sum(axis=(2, 4))
    close(ops.downsample2x_sum(dev(fine)), want, 1e-6)
    acc = dev(np.ones_like(want))
    ops.downsample2x_sum(dev(fine), out=acc, accumulate=True)
    close(acc, want + 1, 1e-6)


# ---------------------------------------------------------------------------------------------
# joint-model (configs[4]) kernels
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("M,V", [(9, 24), (33, 1000), (6, 50001)])
def test_masked_keras_sparse_ce(ops, M, V):
    """imgcap_caption_loss_graph: K.sparse_categorical_crossentropy on a softmax row (clip + renormalise) times a
    per-row weight; rows with weight 0 contribute nothing."""
    rng = np.random.default_rng(V)
    z = 3.0 * rng.standard_normal((M, V))
    t = rng.integers(0, V, M)
    z[0, t[0]] = -60.0
    z[1, :] = -50.0
    z[1, t[1]] = 50.0
    w = rng.random(M)
    w[2] = 0.0
    p = O.softmax(z)
    want_loss, want_d = O.sparse_cce_keras_with_grad(t, p, w)
    ld = (V + 3) // 4 * 4
    zd = torch.zeros(M, ld, device="cuda")
    zd[:, :V] = dev(z)
    dl = torch.empty_like(zd)
    loss = torch.empty(M, device="cuda")
    ops.softmax_ce(zd[:, :V], dev(t, torch.int32), None, loss, dl[:, :V], grad_scale=1.0, row_weights=dev(w), keras_sparse=True)
    close(loss, want_loss, 2e-5)
    close(dl[:, :V], want_d, 2e-5)
    assert float(dl[2, :V].abs().max()) == 0.0


@pytest.mark.parametrize("S", [64, 256])
def test_rpn_losses_and_head_gradients(ops, S):
    rng = np.random.default_rng(S)
    A, stride = 3, 20
    shapes = [(S // s, S // s) for s in (4, 8, 16, 32, 64)]
    heads = [rng.standard_normal((1, h, w, stride)) for h, w in shapes]
    n_anchor = sum(h * w * A for h, w in shapes)
    match = np.zeros(n_anchor, np.int32)
    pick = rng.choice(n_anchor, 90, replace=False)
    match[pick] = np.where(rng.random(90) < 0.4, 1, -1)
    match[n_anchor - 1], match[0] = 1, -1                              # first and last anchor take part
    n_pos = int((match == 1).sum())
    target = 1.5 * rng.standard_normal((128, 4))
    logits = np.concatenate([h[0, :, :, :2 * A].reshape(-1, 2) for h in heads])
    bbox = np.concatenate([h[0, :, :, 2 * A:6 * A].reshape(-1, 4) for h in heads])
    l_cls, d_cls = O.rpn_class_loss(match, logits)
    l_box, d_box = O.rpn_bbox_loss(target, match, bbox)
    sizes = np.array([0] + [h * w * A for h, w in shapes]).cumsum()
    idx = np.nonzero(match)[0]
    lvl = np.searchsorted(sizes, idx, side="right") - 1
    dh = [torch.zeros(h.shape, device="cuda") for h in heads]
    losses = torch.empty(2, device="cuda")
    ops.rpn_loss_grad([dev(h) for h in heads], dh, dev(lvl, torch.int32), dev(idx - sizes[lvl], torch.int32),
                      dev(match[idx], torch.int32), dev(target), n_pos, losses)
    close(losses, np.array([l_cls, l_box]), 1e-5)
    for i, (h, w) in enumerate(shapes):
        n = h * w * A
        got = dh[i].cpu().numpy()[0]
        close(got[:, :, :2 * A].reshape(-1, 2), d_cls[sizes[i]:sizes[i] + n], 1e-5)
        close(got[:, :, 2 * A:6 * A].reshape(-1, 4), d_box[sizes[i]:sizes[i] + n], 1e-5)
        assert np.all(got[:, :, 6 * A:] == 0)


def test_dgrad_weight_pack_scatter_l2(ops):
    from image_captioning_amd.packing import pack_conv_kernel, pack_conv_kernel_dgrad
    rng = np.random.default_rng(11)
    for k, cin, cout in ((3, 64, 128), (1, 512, 20)):
        w = rng.standard_normal((k, k, cin, cout)).astype(np.float32)
        got = ops.conv_weight_dgrad_pack(dev(pack_conv_kernel(w)), k, k, cin).cpu().numpy()
        assert np.array_equal(got, pack_conv_kernel_dgrad(w))
    coarse, fine = rng.standard_normal((2, 3, 5, 8)), rng.standard_normal((2, 6, 10, 8))
    want = fine.copy()
    want[:, ::2, ::2, :] += coarse
    close(ops.scatter2_add(dev(coarse), dev(fine)), want, 1e-6)
    n = 5000
    wv, coef, g = rng.standard_normal(n), rng.random(n) * (rng.random(n) < 0.7), rng.standard_normal(n)
    gd, loss = dev(g), torch.empty(1, device="cuda")
    ops.l2_reg(dev(wv), dev(coef), gd, loss)
    close(gd, g + 2 * coef * wv, 1e-6)
    close(loss, np.array([(coef * wv * wv).sum()]), 1e-5)
    mask = (rng.random(n) < 0.5).astype(np.float64)           # the 0/1 trainable subset folded into the same pass
    gd2, loss2 = dev(g), torch.empty(1, device="cuda")
    ops.l2_reg(dev(wv), dev(coef), gd2, loss2, mask=dev(mask))
    close(gd2, g * mask + 2 * coef * wv, 1e-6)
    assert torch.equal(loss, loss2)                           # fixed-order sum: the same bits every time
    big = torch.randn(3_000_001, device="cuda")
    cb = torch.rand(3_000_001, device="cuda")
    l1, l2 = torch.empty(1, device="cuda"), torch.empty(1, device="cuda")
    ops.l2_reg(big, cb, None, l1)
    ops.l2_reg(big, cb, None, l2)
    assert torch.equal(l1, l2)
    assert abs(float(l1) - float((cb.double() * big.double() ** 2).sum())) < 1e-5 * float(l1)


@pytest.mark.parametrize("case", [(1, 2, 2, 64, 20, 1), (1, 4, 4, 256, 512, 3), (1, 16, 16, 512, 20, 1), (1, 8, 8, 2048, 256, 1)])
def test_wgrad_small_levels_and_accumulation(ops, case):
    """Pixel counts that are not a multiple of the K-tile (P5/P6 of small images), the padded 20-channel RPN head, and
    accumulation over pyramid levels."""
    from image_captioning_amd.packing import pack_conv_kernel
    N, H, W, Cin, Cout, k = case
    rng = np.random.default_rng(sum(case))
    x, dy = rng.standard_normal((N, H, W, Cin)), rng.standard_normal((N, H, W, Cout))
    w = np.zeros((k, k, Cin, Cout))
    _, dw, _ = O.conv2d_nhwc_backward(x, w, dy, 1, 'same' if k == 3 else 'valid')
    pad = (k - 1) // 2
    base = rng.standard_normal(pack_conv_kernel(dw).shape)
    out = dev(base)
    ops.conv2d_wgrad(dev(x), dev(dy), k, k, 1, pad, pad, out=out, accumulate=True)
    close(out, base + pack_conv_kernel(dw), 3e-5)
    ops.conv2d_wgrad(dev(x), dev(dy), k, k, 1, pad, pad, out=out)
    close(out, pack_conv_kernel(dw), 3e-5)
    if Cout == 20:                                                     # data gradient through the padded head: a K = 20 GEMM
        wk = rng.standard_normal((1, 1, Cin, Cout))                    # on the packed forward weights [Cout][Cin] as they are
        dx, _, _ = O.conv2d_nhwc_backward(x, wk, dy, 1, 'valid')
        close(ops.gemm(dev(dy.reshape(-1, Cout)), dev(pack_conv_kernel(wk))), dx.reshape(-1, Cin), 3e-5)


def test_conv_residual_in_place(ops):
    """dP += dgrad: the residual operand may alias the output."""
    from image_captioning_amd.packing import pack_conv_kernel
    rng = np.random.default_rng(12)
    x, w = rng.standard_normal((1, 16, 16, 512)), rng.standard_normal((3, 3, 512, 256)) / 60
    acc = rng.standard_normal((1, 16, 16, 256))
    want = acc + O.conv2d_nhwc(x, w, None, 1, 'same')
    out = dev(acc)
    ops.conv2d(dev(x), dev(pack_conv_kernel(w)), 3, 3, 1, 1, 1, 16, 16, residual=out, res_mode=1, out=out)
    close(out, want, 3e-5)


@pytest.mark.parametrize("M,N", [(5000, 20), (65536, 256), (3000, 1003), (7, 4), (300, 50000), (4097, 512)])
def test_colsum_shapes(ops, M, N):
    """Bias gradients: narrow RPN head, tall conv gradients (row chunks + atomics), odd widths, the vocabulary bias."""
    rng = np.random.default_rng(M + N)
    x = rng.standard_normal((M, N)).astype(np.float32)
    want = x.astype(np.float64).sum(0)
    close(ops.colsum(dev(x)), want, 2e-6 * np.sqrt(M))
    acc = dev(np.full(N, 2.0))
    ops.colsum(dev(x), out=acc, accumulate=True)
    close(acc, want + 2, 2e-6 * np.sqrt(M))


@pytest.mark.parametrize("ta,tb,M,N,K", [(1, 0, 1024, 520, 3000), (0, 1, 300, 1024, 5008), (0, 0, 200, 2048, 300), (1, 1, 64, 64, 200)])
def test_gemm_k_tail_split(ops, ta, tb, M, N, K):
    """K % 32 != 0: bulk on the fast loaders + one range-checked tail launch, with bias / residual / accumulate and split-K."""
    rng = np.random.default_rng(K)
    A, B = rng.standard_normal((M, K)), rng.standard_normal((K, N)) / np.sqrt(K)
    sh, R, C0 = rng.standard_normal(N), rng.standard_normal((M, N)), rng.standard_normal((M, N))
    a = dev(A.T if ta else A)
    b = dev(B.T if tb else B)
    for split in (0, 3):
        out = dev(C0)
        ops.gemm(a, b, out=out, a_trans=bool(ta), b_trans=bool(tb), shift=dev(sh), residual=dev(R), accumulate=True, split_k=split)
        close(out, A @ B + sh + R + C0)
    close(ops.gemm(a, b, a_trans=bool(ta), b_trans=bool(tb)), A @ B)


def test_gemm_k_tail_split_with_gather(ops):
    rng = np.random.default_rng(9)
    V, E, N, rows = 400, 300, 256, 200
    table, W = rng.standard_normal((V, E)), rng.standard_normal((E, N))
    ids = rng.integers(0, V, rows)
    close(ops.gemm(dev(table), dev(W), gather=dev(ids, torch.int32), shift=dev(np.ones(N))), table[ids] @ W + 1.0)   # K = 300
    dz = rng.standard_normal((rows, N))
    got = ops.gemm(dev(table), dev(dz), a_trans=True, gather=dev(ids, torch.int32))                                    # K = 200 rows
    close(got, table[ids].T @ dz, 5e-5)


def test_degenerate_sizes_are_rejected_not_launched(ops):
    """Empty operands (no RoIs, no rows, no columns) come back as DcapError from the C-ABI's validation -- never a zero-sized
    grid or an out-of-bounds access."""
    from image_captioning_amd._lib import DcapError
    maps = [torch.zeros(1, 64 // s, 64 // s, 8, device="cuda") for s in (4, 8, 16, 32)]
    with pytest.raises(DcapError):
        ops.roi_align_pyramid(maps, torch.zeros(1, 0, 4, device="cuda"), 64 * 64, 7)
    with pytest.raises(DcapError):
        ops.gemm(torch.zeros(0, 32, device="cuda"), torch.zeros(32, 64, device="cuda"))
    with pytest.raises(DcapError):
        ops.softmax_ce(torch.zeros(0, 8, device="cuda"), torch.zeros(0, dtype=torch.int32, device="cuda"), None, torch.zeros(0, device="cuda"), None)
    with pytest.raises(DcapError):
        ops.colsum(torch.zeros(0, 16, device="cuda"))
    with pytest.raises(DcapError):
        ops.conv2d(torch.zeros(1, 8, 8, 48, device="cuda"), torch.zeros(32, 48, device="cuda"), 1, 1, 1, 0, 0, 8, 8)       # Cin % 32 != 0
    with pytest.raises(DcapError):
        ops.lstm_seq_fwd(torch.zeros(0, 16, device="cuda"), torch.zeros(4, 16, device="cuda"), None, 0, 1)


def test_maxpool2x2s2_matches_numpy(ops):
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 6, 10, 8)).astype(np.float32)
    want = x.reshape(2, 3, 2, 5, 2, 8).max(axis=(2, 4))
    np.testing.assert_array_equal(ops.maxpool2x2s2(dev(x)).cpu().numpy(), want)
    from image_captioning_amd._lib import DcapError
    with pytest.raises(DcapError):
        ops.maxpool2x2s2(dev(x[:, :5]))                 # odd height


def test_dropout_mask_is_keras_inverted_dropout_of_ones(ops):
    """dc_dropout_mask_f32 = K.dropout(ones, rate): values 0 or 1/(1-rate), keep fraction 1-rate, a pure function of
    (element, seed, offset) -- the same stream twice is identical, another offset or seed is a different mask."""
    n, rate = 4 * 200 * 512, 0.2
    a = ops.dropout_mask(torch.empty(n, device="cuda"), rate, seed=77, offset=2).cpu().numpy()
    assert set(np.unique(a)) == {0.0, np.float32(1.25)}
    assert abs((a > 0).mean() - 0.8) < 5e-3 and abs(a.mean() - 1.0) < 6e-3
    again = ops.dropout_mask(torch.empty(n, device="cuda"), rate, seed=77, offset=2).cpu().numpy()
    np.testing.assert_array_equal(a, again)
    head = ops.dropout_mask(torch.empty(1000, device="cuda"), rate, seed=77, offset=2).cpu().numpy()
    np.testing.assert_array_equal(head, a[:1000])                       # independent of the launch geometry
    for seed, off in ((77, 3), (78, 2)):
        b = ops.dropout_mask(torch.empty(n, device="cuda"), rate, seed=seed, offset=off).cpu().numpy()
        assert 0.6 < ((a > 0) == (b > 0)).mean() < 0.76                  # independent masks agree on 0.8^2 + 0.2^2 = 0.68
    k = (a.reshape(4, 200, 512) > 0)
    assert abs(np.corrcoef(k[0].ravel(), k[1].ravel())[0, 1]) < 0.02    # the four gate masks are uncorrelated
    assert np.all(ops.dropout_mask(torch.empty(64, device="cuda"), 0.0, 1, 1).cpu().numpy() == 1.0)


def test_bn_bwd_recovers_the_normalised_activation_and_survives_dead_channels(ops):
    """dc_bn_bwd_f32: dacc = dz * scale, dzn = dz * n with n = (bn_out - beta) / gamma recovered from the stored BN output.  A dead
    channel (gamma == 0: pretrained ResNet BatchNorm layers have them) must give dzn = 0, not 0 / 0 = NaN: a NaN in
    dgamma would reach the AMSGrad state of the whole flat bucket through the clip norm."""
    rng = np.random.default_rng(21)
    rows, Cc = 37, 16
    n = rng.standard_normal((rows, Cc))
    gamma, beta = rng.uniform(0.5, 1.5, Cc), rng.standard_normal(Cc)
    gamma[3], gamma[10] = 0.0, 0.0
    short = rng.standard_normal((rows, Cc))
    a = n * gamma + beta + short                              # a block's last convolution: out - shortcut is the BN output
    dz = rng.standard_normal((rows, Cc)) * (rng.random((rows, Cc)) < 0.7)
    scale = gamma / np.sqrt(rng.uniform(0.5, 1.5, Cc) + 1e-3)
    dacc, dzn = torch.empty(rows, Cc, device="cuda"), torch.empty(rows, Cc, device="cuda")
    ops.bn_bwd(dev(dz), dev(a), dev(short), dev(gamma), dev(beta), dev(scale), dacc, dzn)
    assert bool(torch.isfinite(dzn).all()) and bool(torch.isfinite(dacc).all())
    want = dz * n
    want[:, [3, 10]] = 0.0
    close(dacc, dz * scale, 1e-6)
    close(dzn, want, 2e-5)


def _philox2x32(c0, c1, key):
    """NumPy restatement of the library's counter-based generator (csrc/dcap_internal.h philox2x32: Philox-2x32-10)."""
    c0 = np.asarray(c0, np.uint64) & 0xFFFFFFFF
    c1 = np.full_like(c0, int(c1) & 0xFFFFFFFF)
    key = int(key) & 0xFFFFFFFF
    for _ in range(10):
        p = (np.uint64(0xD256D193) * c0) & np.uint64(0xFFFFFFFFFFFFFFFF)
        hi, lo = p >> np.uint64(32), p & np.uint64(0xFFFFFFFF)
        c0 = (hi ^ np.uint64(key) ^ c1) & np.uint64(0xFFFFFFFF)
        c1 = lo
        key = (key + 0x9E3779B9) & 0xFFFFFFFF
    return c0.astype(np.uint32)


def _dt_case(seed, n_props, n_gt, pad_props, pad_gt, T=6, jitter=0.08):
    """Proposals scattered around GT boxes (some close: IoU above 0.5, some far), zero padding rows in both lists, one all-zero
    proposal in the middle of the list and a degenerate (zero-area) proposal."""
    rng = np.random.default_rng(seed)
    gt = np.zeros((n_gt + pad_gt, 4), np.float32)
    y1, x1 = rng.uniform(0, 0.6, n_gt), rng.uniform(0, 0.6, n_gt)
    gt[:n_gt] = np.stack([y1, x1, y1 + rng.uniform(0.1, 0.4, n_gt), x1 + rng.uniform(0.1, 0.4, n_gt)], 1)
    caps = np.zeros((n_gt + pad_gt, T), np.int32)
    caps[:n_gt] = rng.integers(1, 1000, (n_gt, T))
    props = np.zeros((n_props + pad_props, 4), np.float32)
    src = rng.integers(0, max(n_gt, 1), n_props)
    noise = rng.normal(0, jitter, (n_props, 4)) * (rng.random((n_props, 1)) < 0.6)
    base = gt[src] if n_gt else rng.uniform(0.1, 0.5, (n_props, 4)).astype(np.float32)
    props[:n_props] = np.clip(base + noise, 0, 1)
    far = rng.random(n_props) < 0.3
    fy, fx = rng.uniform(0, 0.9, n_props), rng.uniform(0, 0.9, n_props)
    props[:n_props][far] = np.stack([fy, fx, fy + 0.05, fx + 0.05], 1)[far]
    if n_props > 10:
        props[5] = 0                                         # a zero row that is NOT trailing padding: later indices shift when compacted
        props[7] = [0.3, 0.3, 0.3, 0.6]                      # zero area, non-zero row
    if pad_gt and n_gt > 2:
        gt[[1, n_gt]] = gt[[n_gt, 1]]                        # a zero GT row in the middle
        caps[[1, n_gt]] = caps[[n_gt, 1]]
    return props.astype(np.float32), gt, caps


@pytest.mark.parametrize("case", [(0, 2000, 40, 0, 60), (1, 1500, 7, 500, 3), (2, 300, 1, 0, 0), (3, 2000, 100, 48, 0), (4, 64, 0, 6, 4), (5, 3000, 12, 100, 0),
                                  (6, 90, 3, 10, 2),          # fewer positives than the quota: the negative count follows int32(f32(1/ratio) * n_pos) - n_pos
                                  (7, 4000, 500, 96, 12)])    # the entry point's limits: 4096 proposals, 512 GT boxes
@pytest.mark.parametrize("seed", [None, 1234])
def test_detection_targets_on_the_device_equal_the_oracle(ops, case, seed):
    """dc_detection_targets_f32 (DetectionTargetLayer, dense_img_cap/dense_model.py:450-572) against oracle.detection_targets: RoIs,
    captions and counts BIT-EXACT, the oracle's `shuffle` being the sort by the same Philox keys the kernel draws (or proposal order)."""
    cs, n_props, n_gt, pad_p, pad_g = case
    props, gt, caps = _dt_case(cs, n_props, n_gt, pad_p, pad_g)
    n_rois, ratio, offset = 200, 0.33, 77 + cs
    step = torch.tensor([5], dtype=torch.int32, device="cuda")
    rois, oc, counts = ops.detection_targets(dev(props), dev(gt), dev(caps, torch.int32), n_rois, ratio, seed=seed, offset=offset,
                                             offset_dev=step if seed is not None else None)
    if seed is None:
        shuffle = None
    else:
        keys = _philox2x32(np.arange(len(props)), offset + 5, seed)
        shuffle = lambda idx: idx[np.lexsort((idx, keys[idx]))]
    want_rois, want_caps, npos, nneg = O.detection_targets(props, caps, gt, n_rois, ratio, shuffle)
    assert counts.cpu().numpy().tolist() == [npos, nneg]
    assert np.array_equal(rois.cpu().numpy(), want_rois)
    assert np.array_equal(oc.cpu().numpy(), want_caps)
    if cs == 0:
        assert npos == 66 and nneg == 134                    # the benchmark's regime: both lists longer than their quota
    if cs == 6:
        assert 0 < npos < 66
    # the host implementation the joint model used until round 3 agrees as well
    from image_captioning_amd.dense_model import detection_targets as host_dt

    class Cfg:
        TRAIN_ROIS_PER_IMAGE, ROI_POSITIVE_RATIO = n_rois, ratio
    h_rois, h_caps, h_pos, h_neg = host_dt(props, caps, gt, Cfg, shuffle)
    assert (h_pos, h_neg) == (npos, nneg) and np.array_equal(h_rois, want_rois) and np.array_equal(h_caps, want_caps)


def test_detection_targets_rejects_what_it_does_not_cover(ops):
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device="cuda")
    with pytest.raises(DcapError):
        ops.detection_targets(z(5000, 4), z(4, 4), z(4, 6, dt=torch.int32), 200, 0.33)          # > 4096 proposals
    with pytest.raises(DcapError):
        ops.detection_targets(z(100, 4), z(600, 4), z(600, 6, dt=torch.int32), 200, 0.33)       # > 512 GT boxes
    with pytest.raises(DcapError):
        ops.detection_targets(z(100, 4), z(4, 4), z(5, 6, dt=torch.int32), 200, 0.33)           # captions / boxes mismatch
    with pytest.raises(DcapError):
        ops.detection_targets(torch.zeros(100, 4), z(4, 4), z(4, 6, dt=torch.int32), 200, 0.33)  # host tensor: no CPU path


def test_caption_tables_on_the_device(ops):
    """dc_caption_tables_i32: time-major ids / masks / shifted targets and the masked-mean row weights of imgcap_caption_loss_graph."""
    rng = np.random.default_rng(8)
    B, T = 200, 15
    caps = np.zeros((B, T), np.int32)
    for b in range(66):
        L = int(rng.integers(1, T + 1))
        caps[b, :L] = rng.integers(1, 50000, L)
    live = torch.zeros(1, dtype=torch.int32, device="cuda")
    ids, mask, tg, rw = ops.caption_tables(dev(caps, torch.int32), live_count=live)
    want_tg = np.concatenate([caps[:, 1:], np.zeros((B, 1), np.int32)], 1)
    count = int((want_tg > 0).sum())
    assert int(live.item()) == count
    assert np.array_equal(ids.cpu().numpy(), caps.T.reshape(-1))
    assert np.array_equal(mask.cpu().numpy(), (caps != 0).T.reshape(-1).astype(np.uint8))
    assert np.array_equal(tg.cpu().numpy(), want_tg.T.reshape(-1))
    assert np.array_equal(rw.cpu().numpy(), ((want_tg > 0).astype(np.float32) / np.float32(max(count, 1))).T.reshape(-1))
    # all padding: count 0 -> weights 0, no division by zero
    ids, mask, tg, rw = ops.caption_tables(torch.zeros(8, 4, dtype=torch.int32, device="cuda"))
    assert float(rw.abs().max()) == 0.0


def test_persistent_cu_budget_changes_the_grid_not_the_result(ops):
    """dc_set_persistent_cus: the persistent Winograd grids on fewer than 256 CUs (248 in a data-parallel run: RCCL's kernels get a
    CU per XCD; 8 = one block per XCD, every block walking 256 items here).  Results are bit-identical whatever the budget -- a work
    item's arithmetic does not depend on which block takes it; bad values are refused; 0 restores the default.  (The budget applies
    to launches of >= 8 rounds of items only: a short launch keeps the full grid, where 248 blocks for 256 items would mean two rounds.)"""
    from image_captioning_amd import _lib
    lib = _lib.load()
    N, H, W, Cin, Cout = 2, 256, 256, 64, 128                 # 2048 items of 64 tiles = 8 rounds on 256 CUs: the budget applies
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(N, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, 9 * Cin, device="cuda", generator=g) / (9 * Cin) ** 0.5
    u = ops.winograd_pack(w, Cin, Cout)
    args = (x, w, 3, 3, 1, 1, 1, H, W, None, None, None, 0, True)
    try:
        assert ops.set_persistent_cus(0) == 256
        ref = ops.conv2d(*args, w_wino=u).clone()
        for cus, want in ((248, 248), (8, 8), (100, 96)):      # (rounded down to a multiple of 8: one share per XCD)
            assert ops.set_persistent_cus(cus) == want
            got = ops.conv2d(*args, w_wino=u)
            assert torch.equal(got, ref), cus
        with pytest.raises(DcapError):
            ops.set_persistent_cus(4)
        with pytest.raises(DcapError):
            ops.set_persistent_cus(300)
    finally:
        assert ops.set_persistent_cus(0) == 256
    direct = ops.conv2d(*args)
    assert float((ref - direct).abs().max()) / float(direct.abs().max()) < 2e-5
