"""bf16 kernels (BASELINE configs[4]: bf16 storage, fp32 accumulate) against the NumPy oracle on operands rounded to the
same bf16 grid (-m gpu).  Products of bf16 values are exact in fp32, so the only difference to the float64 oracle is the
fp32 accumulation order: tolerance 3e-5 of the output scale (K up to a few thousand); a bf16 OUTPUT is within one bf16
ulp (2^-8 relative) of the rounded oracle value."""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from image_captioning_amd import ops as _ops, _lib
    _lib.load()
    return _ops


def dev(a, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


def close(got, want, tol=3e-5):
    got = got.detach().float().cpu().numpy().astype(np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    scale = max(1.0, float(np.abs(want).max()))
    diff = np.abs(got - want)
    err = float(diff.max()) / scale
    if not err < tol:                                          # where, and how many: a lone entry reads differently from a spread-out excess
        at = np.unravel_index(int(diff.argmax()), diff.shape)
        raise AssertionError("max err %.3e (scaled) exceeds %.1e at %s: got %.9g want %.9g; %d of %d entries above the tolerance"
                             % (err, tol, at, got[at], want[at], int((diff > tol * scale).sum()), diff.size))


def test_cast_is_round_to_nearest_even(ops):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(10007) * 10.0 ** rng.integers(-20, 20, 10007), [0.0, -0.0, 1.0, 1.00390625, 1.01171875, 3.3895314e38]])
    got = ops.to_bf16(dev(x)).float().cpu().numpy().astype(np.float64)
    np.testing.assert_array_equal(got, O.to_bf16(x))
    m = rng.standard_normal((5, 300))
    padded = ops.to_bf16(dev(m), pad_cols=304).float().cpu().numpy()
    np.testing.assert_array_equal(padded[:, :300], O.to_bf16(m))
    assert padded.shape == (5, 304) and not padded[:, 300:].any()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 1024, 1024), (8, 1000, 512), (264, 136, 72), (960, 256, 2048), (1024, 2048, 3000)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_bf16_layouts(ops, M, N, K, ta, tb):
    """NN / NT / TN / TT incl. ragged edges (M, N not multiples of the 128 tile), K tails (K % 64 != 0) and automatic split-K."""
    rng = np.random.default_rng(M * 7 + N * 3 + K + ta * 2 + tb)
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((K, N))
    a = ops.to_bf16(dev(A.T if ta else A))
    b = ops.to_bf16(dev(B.T if tb else B))
    close(ops.gemm_bf16(a, b, a_trans=bool(ta), b_trans=bool(tb)), O.to_bf16(A) @ O.to_bf16(B))


@pytest.mark.parametrize("M,N,K,split", [(3000, 4104, 200, 0), (1024, 1024, 8192, 16), (3072, 4096, 512, 0), (4096, 3072, 72, 0)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_bf16_256_tile_layouts(ops, M, N, K, split, ta, tb):
    """The 256 x 256 x 64 kernel (csrc/bgemm256_core.h): all four layouts, ragged edges in M and N, a K tail (200 = 3 x 64 + 8), a
    single short K-tile pair (72), exact tiling, split-K slabs (requested: the library's own cost model keeps that shape on the
    128-square tile); the library reports which tile ran."""
    rng = np.random.default_rng(M * 7 + N * 3 + K + ta * 2 + tb)
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((K, N))
    a = ops.to_bf16(dev(A.T if ta else A))
    b = ops.to_bf16(dev(B.T if tb else B))
    info = {}
    got = ops.gemm_bf16(a, b, a_trans=bool(ta), b_trans=bool(tb), split_k=split, info=info)
    assert info == {"tile": 256, "split_k": max(split, 1)}
    close(got, O.to_bf16(A) @ O.to_bf16(B))


def test_gemm_bf16_256_tile_epilogue_gather_and_bf16_output(ops):
    """Fused epilogue of the 256-square kernel straight from the accumulators: scale / shift / residual / ReLU / accumulate, the
    per-RoI (row-modulo) residual, the bf16 copy, a bf16-only output, and the row gather on A."""
    rng = np.random.default_rng(21)
    M, N, K = 3000, 4096, 512
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((K, N)) / np.sqrt(K)
    sc, sh = rng.uniform(0.5, 1.5, N), rng.standard_normal(N)
    R = rng.standard_normal((M, N))
    C0 = rng.standard_normal((M, N))
    ab, bb = ops.to_bf16(dev(A)), ops.to_bf16(dev(B))
    prod = O.to_bf16(A) @ O.to_bf16(B)
    want = np.maximum(prod * sc + sh + R, 0) + C0
    out = dev(C0)
    outb = torch.empty((M, N), dtype=BF, device="cuda")
    info = {}
    ops.gemm_bf16(ab, bb, out=out, out_bf16=outb, scale=dev(sc), shift=dev(sh), residual=dev(R), relu=True, accumulate=True, info=info)
    assert info["tile"] == 256
    close(out, want)
    assert np.abs(outb.float().cpu().numpy() - want).max() <= 2.0 ** -8 * np.abs(want).max() + 1e-6
    only_b = torch.empty((M, N), dtype=BF, device="cuda")
    ops.gemm_bf16(ab, bb, out_bf16=only_b, shift=dev(sh), info=info)
    assert info["tile"] == 256
    assert np.abs(only_b.float().cpu().numpy() - (prod + sh)).max() <= 2.0 ** -8 * np.abs(prod + sh).max() + 1e-6
    Bn = 200                                                           # per-RoI term broadcast over the 15 timesteps
    r = rng.standard_normal((Bn, N))
    close(ops.gemm_bf16(ab, bb, residual=dev(r), res_rows=Bn, info=info), prod + np.tile(r, (M // Bn, 1)))
    assert info["tile"] == 256
    ids = rng.integers(0, M, 3000)
    close(ops.gemm_bf16(ab, bb, gather=dev(ids, torch.int32), info=info), prod[ids])
    assert info["tile"] == 256
    # rows that are not 16-byte addressable go to the 128-square kernel (scalar epilogue)
    bt = ops.to_bf16(dev(B.T))
    close(ops.gemm_bf16(ab, bt[:4094], b_trans=True, info=info), prod[:, :4094])
    assert info["tile"] == 128


@pytest.mark.parametrize("split", [0, 1, 3, 7])
def test_gemm_bf16_epilogue_splitk_and_bf16_output(ops, split):
    rng = np.random.default_rng(11 + split)
    M, N, K = 200, 1024, 12544 // 4
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((K, N)) / np.sqrt(K)
    sc, sh = rng.uniform(0.5, 1.5, N), rng.standard_normal(N)
    R = rng.standard_normal((M, N))
    C0 = rng.standard_normal((M, N))
    want = np.maximum((O.to_bf16(A) @ O.to_bf16(B)) * sc + sh + R, 0) + C0
    out = dev(C0)
    outb = torch.empty((M, N), dtype=BF, device="cuda")
    ops.gemm_bf16(ops.to_bf16(dev(A)), ops.to_bf16(dev(B)), out=out, out_bf16=outb, scale=dev(sc), shift=dev(sh), residual=dev(R), relu=True,
                  accumulate=True, split_k=split)
    close(out, want)
    got_b = outb.float().cpu().numpy().astype(np.float64)
    assert np.abs(got_b - want).max() <= 2.0 ** -8 * np.abs(want).max() + 1e-6       # one bf16 ulp of the fp32 result
    only_b = torch.empty((M, N), dtype=BF, device="cuda")                            # bf16-only output (no fp32 C)
    ops.gemm_bf16(ops.to_bf16(dev(A)), ops.to_bf16(dev(B)), out_bf16=only_b, shift=dev(sh), split_k=split)
    wb = O.to_bf16(A) @ O.to_bf16(B) + sh
    assert np.abs(only_b.float().cpu().numpy() - wb).max() <= 2.0 ** -8 * np.abs(wb).max() + 1e-6


def test_gemm_bf16_gather_rows_and_k(ops):
    """a_gather: embedding lookup on the rows of A (forward) and on the K rows of A^T (the embedding-side weight gradient)."""
    rng = np.random.default_rng(5)
    V, E, Ep, N, U = 1000, 300, 304, 960, 512
    table = rng.standard_normal((V, E))
    ids = rng.integers(0, V, N)
    W = rng.standard_normal((E, U))
    tb = ops.to_bf16(dev(table), pad_cols=Ep)                         # K padded to a multiple of 8 with zeros
    Wp = np.zeros((Ep, U))
    Wp[:E] = W
    close(ops.gemm_bf16(tb, ops.to_bf16(dev(Wp)), gather=dev(ids, torch.int32)), O.to_bf16(table)[ids] @ O.to_bf16(W))
    dz = rng.standard_normal((N, U))
    got = ops.gemm_bf16(tb, ops.to_bf16(dev(dz)), a_trans=True, gather=dev(ids, torch.int32))
    want = np.zeros((Ep, U))
    want[:E] = O.to_bf16(table)[ids].T @ O.to_bf16(dz)
    close(got, want)


def test_gemm_bf16_per_roi_residual_and_strided_views(ops):
    rng = np.random.default_rng(6)
    Bn, T, K, N = 24, 5, 512, 256
    X = rng.standard_normal((T * Bn, K))
    W = rng.standard_normal((K + 64, N + 8))
    r = rng.standard_normal((Bn, N))
    wb = ops.to_bf16(dev(W))
    got = ops.gemm_bf16(ops.to_bf16(dev(X)), wb[64:, :N], residual=dev(r), res_rows=Bn)
    close(got, O.to_bf16(X) @ O.to_bf16(W)[64:, :N] + np.tile(r, (T, 1)))


def test_gemm_bf16_rejects_bad_arguments(ops):
    from image_captioning_amd._lib import DcapError
    a = torch.zeros((16, 60), dtype=BF, device="cuda")
    b = torch.zeros((60, 16), dtype=BF, device="cuda")
    with pytest.raises(DcapError):
        ops.gemm_bf16(a, b)                                           # K % 8 != 0
    with pytest.raises(DcapError):
        ops.gemm_bf16(torch.zeros((16, 64), device="cuda"), torch.zeros((64, 16), dtype=BF, device="cuda"))     # fp32 operand
    with pytest.raises(DcapError):
        ops.gemm_bf16(torch.zeros((16, 64), dtype=BF, device="cuda"), torch.zeros((64, 12), dtype=BF, device="cuda"))   # N % 8 (K-major B)


# ---------------------------------------------------------------------------------------------
# fused vocabulary projection + softmax + cross-entropy (dc_vocab_ce), fp32 and bf16 operands
# ---------------------------------------------------------------------------------------------

def _ce_case(rng, M, V, K, bf16):
    X = rng.standard_normal((M, K))
    W = rng.standard_normal((K, V)) * (2.0 / np.sqrt(K))
    b = rng.standard_normal(V)
    t = rng.integers(0, V, M)
    Xq, Wq = (O.to_bf16(X), O.to_bf16(W)) if bf16 else (X.astype(np.float32).astype(np.float64), W.astype(np.float32).astype(np.float64))
    z = Xq @ Wq + b
    return X, W, b, t, z


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("M,V,K", [(7, 1000, 256), (960, 10000, 256), (130, 1016, 2048), (64, 50000, 1024)])
def test_vocab_ce_categorical(ops, bf16, M, V, K):
    """Dense(V)+softmax+categorical CE fused: loss rows, d/dlogits and the bias gradient against the oracle applied to
    the logits X@W+b (operands rounded to bf16 first in the bf16 case); clipped rows have zero gradient."""
    rng = np.random.default_rng(V + K + M)
    X, W, b, t, z = _ce_case(rng, M, V, K, bf16)
    z[0] = X[0] @ W * 0 + b * 0                                        # row 0: uniform logits (loss = ln V)
    X[0] = 0
    b0 = b.copy()
    Xd, Wd = dev(X), dev(W)
    if bf16:
        Xd, Wd = ops.to_bf16(Xd), ops.to_bf16(Wd)
    z = (O.to_bf16(X) @ O.to_bf16(W) if bf16 else X.astype(np.float32).astype(np.float64) @ W.astype(np.float32).astype(np.float64)) + b0
    p = O.softmax(z)
    loss = torch.empty(M, device="cuda")
    dl = torch.full((M, V), 7.0, device="cuda")
    db = torch.empty(V, device="cuda")
    assert ops.vocab_ce_supported(Xd, Wd)
    ops.vocab_ce(Xd, Wd, dev(b0), dev(t, torch.int32), loss_rows=loss, dlogits=dl, dbias=db, grad_scale=1.0 / M)
    want_d = O.softmax_ce_grad_logits(t, p, np.full(M, 1.0 / M))
    close(loss, O.categorical_crossentropy(t, p), 2e-5)
    close(dl * M, want_d * M, 2e-5)
    close(db * M, want_d.sum(0) * M, 5e-5)
    if bf16:                                                           # bf16 gradient output, K-padded with zeros
        Vp = (V + 7) // 8 * 8
        dlb = torch.full((M, Vp), 7.0, dtype=BF, device="cuda")
        ops.vocab_ce(Xd, Wd, dev(b0), dev(t, torch.int32), dlogits=dlb, grad_scale=1.0, materialize_bf16=False)
        got = dlb.float().cpu().numpy().astype(np.float64)
        assert np.abs(got[:, :V] - want_d * M).max() <= 2.0 ** -8 * np.abs(want_d * M).max() + 1e-6 and not got[:, V:].any()


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("M,V,K", [(9, 24, 64), (33, 1000, 1024), (130, 50000, 1024)])
def test_vocab_ce_masked_keras_sparse(ops, bf16, M, V, K):
    """The joint model's imgcap_caption_loss_graph flavour: K.sparse_categorical_crossentropy (clip + renormalise) times a row
    weight, incl. a clipped-low and a clipped-high target and a zero-weight row."""
    rng = np.random.default_rng(V + K)
    X, W, b, t, _ = _ce_case(rng, M, V, K, bf16)
    b[t[0]] -= 80.0                                                    # (the bias is shared: every row sees it)
    X[1] = 0
    w = rng.random(M)
    w[2] = 0.0
    Xd, Wd = dev(X), dev(W)
    if bf16:
        Xd, Wd = ops.to_bf16(Xd), ops.to_bf16(Wd)
    z = (O.to_bf16(X) @ O.to_bf16(W) if bf16 else X.astype(np.float32).astype(np.float64) @ W.astype(np.float32).astype(np.float64)) + b
    p = O.softmax(z)
    want_loss, want_d = O.sparse_cce_keras_with_grad(t, p, w)
    Vp = (V + 3) // 4 * 4 if not bf16 else (V + 7) // 8 * 8
    if V % (8 if bf16 else 4):
        assert not ops.vocab_ce_supported(Xd, Wd)
        return
    loss = torch.empty(M, device="cuda")
    dl = torch.empty((M, Vp), device="cuda")
    db = torch.empty(V, device="cuda")
    ops.vocab_ce(Xd, Wd, dev(b), dev(t, torch.int32), loss_rows=loss, dlogits=dl, dbias=db, grad_scale=1.0, row_weights=dev(w), keras_sparse=True)
    close(loss, want_loss, 3e-5)
    close(dl[:, :V], want_d, 3e-5)
    close(db, want_d.sum(0), 1e-4)
    assert float(dl[2, :V].abs().max()) == 0.0
    only_loss = torch.empty(M, device="cuda")
    ops.vocab_ce(Xd, Wd, dev(b), dev(t, torch.int32), loss_rows=only_loss, row_weights=dev(w), keras_sparse=True)      # forward only
    assert torch.equal(only_loss, loss)


@pytest.mark.parametrize("bf16,M,V,K,calls", [(True, 130, 50000, 1024, 300), (False, 130, 10000, 256, 100), (True, 600, 50000, 1024, 60)])
def test_vocab_ce_is_bit_identical_from_call_to_call(ops, bf16, M, V, K, calls):
    """Identical inputs, identical bits, every call.  The (130, 50000, 1024, bf16) case is the one that did NOT hold this in round 6 while
    the library was built with the compiler's packed-f32 (v_pk_*_f32) vectorisation: about one call in fifteen -- and the first call of a
    process more often than not -- returned 16 wrong gradient entries (one row, one float4 component, lanes 48-63 of a wave) and the
    bias-gradient sums over them; nothing in the kernel is order-dependent (no atomics, fixed-order partial sums), so every call must
    equal the first (tools/determinism_vocab_ce.py prints the differing entries)."""
    rng = np.random.default_rng(V + K)
    X, W, b, t, _ = _ce_case(rng, M, V, K, bf16)
    b[t[0]] -= 80.0
    X[1] = 0
    w = rng.random(M)
    w[2] = 0.0
    Xd, Wd = dev(X), dev(W)
    if bf16:
        Xd, Wd = ops.to_bf16(Xd), ops.to_bf16(Wd)
    bd, td, wd = dev(b), dev(t, torch.int32), dev(w)
    first = None
    for call in range(calls):
        out = (torch.empty(M, device="cuda"), torch.empty((M, V), device="cuda"), torch.empty(V, device="cuda"))
        ops.vocab_ce(Xd, Wd, bd, td, loss_rows=out[0], dlogits=out[1], dbias=out[2], grad_scale=1.0, row_weights=wd, keras_sparse=True)
        if first is None:
            first = out
            continue
        for name, a, r in zip(("loss", "dlogits", "dbias"), out, first):
            assert torch.equal(a, r), "call %d: %s differs from the first call's in %d entries" % (call, name, int((a != r).sum()))


@pytest.mark.parametrize("bf16,M,V,K", [(False, 300, 1000, 64), (True, 300, 1000, 64), (True, 1500, 8200, 512)])
def test_vocab_ce_keras_sparse_clip_pass_runs_only_where_a_row_needs_it(ops, bf16, M, V, K):
    """The lazy CLIP pass (round 5): the STATS pass keeps every row's smallest logit, the row kernel marks the rows with a probability
    outside [1e-7, 1 - 1e-7], and only row tiles holding such a row run the second GEMM pass.  Three situations against the oracle:
    no row clipped at all (every tile skips the pass: S = UP = 1), one row tile with a clipped-low and a clipped-high row while the
    other tiles skip, and the same on the 256-row tile."""
    rng = np.random.default_rng(V + K + M)
    X, W, b, t, _ = _ce_case(rng, M, V, K, bf16)
    W *= 0.25                                                          # logits within a few units: every probability well inside the clip range
    b *= 0.5
    w = rng.random(M)
    for clipped_rows in ((), (3, 140)):
        Xc = X.copy()
        bc = b.copy()
        if clipped_rows:
            lo, hi = clipped_rows
            Xc[hi] = 400.0 * W[:, 7] / np.linalg.norm(W[:, 7])         # aligned with word 7: its probability exceeds 1 - 1e-7 (clipped high)
            Xc[lo] *= 24.0                                             # a wide row: some probabilities fall below 1e-7 (clipped low)
        Xd, Wd = dev(Xc), dev(W)
        if bf16:
            Xd, Wd = ops.to_bf16(Xd), ops.to_bf16(Wd)
        z = (O.to_bf16(Xc) @ O.to_bf16(W) if bf16 else Xc.astype(np.float32).astype(np.float64) @ W.astype(np.float32).astype(np.float64)) + bc
        p = O.softmax(z)
        outside = ((p < 1e-7) | (p > 1 - 1e-7)).any(axis=1)
        assert sorted(np.nonzero(outside)[0]) == sorted(clipped_rows)      # exactly the rows the case clips (both in the first row tile)
        if clipped_rows:
            assert p[clipped_rows[1]].max() > 1 - 1e-7 and p[clipped_rows[0]].max() < 1 - 1e-7 and p[clipped_rows[0]].min() < 1e-7
        want_loss, want_d = O.sparse_cce_keras_with_grad(t, p, w)
        Vp = (V + 7) // 8 * 8
        loss = torch.empty(M, device="cuda")
        dl = torch.empty((M, Vp), device="cuda")
        db = torch.empty(V, device="cuda")
        ops.vocab_ce(Xd, Wd, dev(bc), dev(t, torch.int32), loss_rows=loss, dlogits=dl, dbias=db, grad_scale=1.0, row_weights=dev(w), keras_sparse=True)
        close(loss, want_loss, 3e-5)
        close(dl[:, :V], want_d, 3e-5)
        close(db, want_d.sum(0), 1e-4)


@pytest.mark.parametrize("M,V,K,sparse", [(3000, 4104, 256, False), (2900, 5000, 200, False), (1500, 8200, 512, True), (3000, 50000, 1024, True)])
def test_vocab_ce_bf16_on_the_256_tile(ops, M, V, K, sparse):
    """Problems whose 256 x 256 grid covers the chip run the three passes on the large tile (csrc/bgemm256_core.h), reductions
    straight from the accumulators: ragged M and V, a K tail, both loss flavours, fp32 and bf16 gradient, the bias gradient.
    The last case is BASELINE configs[4]'s own shape (200 RoIs x 15 tokens, 50 000 words)."""
    rng = np.random.default_rng(V + K + M)
    X, W, b, t, _ = _ce_case(rng, M, V, K, True)
    if sparse:
        b[t[0]] -= 80.0
    X[1] = 0
    w = rng.random(M)
    w[2] = 0.0
    Xd, Wd = ops.to_bf16(dev(X)), ops.to_bf16(dev(W))
    p = O.softmax(O.to_bf16(X) @ O.to_bf16(W) + b)
    if sparse:
        want_loss, want_d = O.sparse_cce_keras_with_grad(t, p, w)
        kw = dict(grad_scale=1.0, row_weights=dev(w), keras_sparse=True)
        scale = 1.0
    else:
        want_loss, want_d = O.categorical_crossentropy(t, p), O.softmax_ce_grad_logits(t, p, np.full(M, 1.0 / M)) * M
        kw = dict(grad_scale=1.0 / M)
        scale = float(M)
    loss = torch.empty(M, device="cuda")
    dl = torch.full((M, V), 7.0, device="cuda")
    db = torch.empty(V, device="cuda")
    ops.vocab_ce(Xd, Wd, dev(b), dev(t, torch.int32), loss_rows=loss, dlogits=dl, dbias=db, **kw)
    close(loss, want_loss, 3e-5)
    close(dl * scale, want_d, 3e-5)
    close(db * scale, want_d.sum(0), 2e-4)
    Vp = V + 8
    dlb = torch.full((M, Vp), 7.0, dtype=BF, device="cuda")              # bf16 gradient output, K-padded with zeros
    ops.vocab_ce(Xd, Wd, dev(b), dev(t, torch.int32), dlogits=dlb, materialize_bf16=False, **kw)      # fp32 logits recomputed per pass
    got = dlb.float().cpu().numpy().astype(np.float64) * scale
    assert np.abs(got[:, :V] - want_d).max() <= 2.0 ** -8 * np.abs(want_d).max() + 1e-6 and not got[:, V:].any()


@pytest.mark.parametrize("M,V,K,sparse,clip", [(3000, 4104, 256, False, False), (1500, 8200, 512, True, False), (1500, 8200, 512, True, True),
                                               (3000, 50000, 1024, True, False), (2900, 5000, 200, True, True)])
def test_vocab_ce_bf16_materialised_logits(ops, M, V, K, sparse, clip):
    """dc_vocab_ce_desc.materialize_bf16 (round 6): ONE GEMM pass rounds the logits to bf16 and parks them in the gradient's buffer, the
    clip sums and the gradient are elementwise passes over it (in place).  Reference = the oracle applied to the bf16-ROUNDED logits:
    loss, gradient and bias gradient are those of the rounded logits, consistent with each other (every gradient row sums to zero).
    A logit within fp32 accumulation error of a bf16 rounding boundary may round the other way than in the float64 oracle: such an
    entry differs by one bf16 ulp of the logit, so tolerances are 2^-7 of the row's largest |logit| on the loss and 2^-7 on gradients
    (of the largest entry); against the recomputing flavour (fp32 logits) the same bounds hold with the rounding itself inside."""
    rng = np.random.default_rng(V + K + M + int(clip))
    X, W, b, t, _ = _ce_case(rng, M, V, K, True)
    if clip:
        b[t[0]] -= 80.0                                               # a word whose probability falls below 1e-7 in every row
        X[5] = 400.0 * W[:, 7] / np.linalg.norm(W[:, 7])              # a row whose word 7 exceeds 1 - 1e-7
    else:
        W *= 0.25                                                     # ordinary logits: no probability outside the clip range (lazy pass skipped)
        b *= 0.5
    X[1] = 0
    w = rng.random(M)
    w[2] = 0.0
    Xd, Wd = ops.to_bf16(dev(X)), ops.to_bf16(dev(W))
    z = O.to_bf16(X) @ O.to_bf16(W) + b
    zr = O.to_bf16(z)
    p = O.softmax(zr)
    if sparse:
        want_loss, want_d = O.sparse_cce_keras_with_grad(t, p, w)
        kw = dict(grad_scale=1.0, row_weights=dev(w), keras_sparse=True)
        scale = 1.0
    else:
        want_loss, want_d = O.categorical_crossentropy(t, p), O.softmax_ce_grad_logits(t, p, np.full(M, 1.0 / M)) * M
        kw = dict(grad_scale=1.0 / M)
        scale = float(M)
    Vp = V + 8
    loss = torch.empty(M, device="cuda")
    dlb = torch.full((M, Vp), 7.0, dtype=BF, device="cuda")
    db = torch.empty(V, device="cuda")
    ops.vocab_ce(Xd, Wd, dev(b), dev(t, torch.int32), loss_rows=loss, dlogits=dlb, dbias=db, materialize_bf16=True, **kw)
    got = dlb.float().cpu().numpy().astype(np.float64) * scale
    zmax = np.abs(z).max(axis=1)
    wrow = w if sparse else np.ones(M)
    lerr = np.abs(loss.cpu().numpy().astype(np.float64) - want_loss)
    assert np.all(lerr <= 2.0 ** -7 * wrow * np.maximum(1.0, zmax) + 3e-5), float(lerr.max())
    assert not got[:, V:].any()
    gmax = np.abs(want_d).max()
    assert np.abs(got[:, :V] - want_d).max() <= (2.0 ** -7 * max(1.0, float(zmax.max())) / 2 + 2.0 ** -8) * gmax + 1e-6
    # the median row is exact to output rounding: boundary flips are rare
    row_err = np.abs(got[:, :V] - want_d).max(axis=1)
    assert np.median(row_err) <= 2.0 ** -8 * gmax + 1e-6
    if not clip:                                                      # nothing clipped: a row's gradient sums to zero (to the bf16 rounding of its V outputs)
        assert np.abs(got[:, :V].sum(axis=1)).max() <= 2.0 ** -8 * np.abs(got[:, :V]).sum(axis=1).max() + 1e-6
    close(db * scale, got[:, :V].sum(0), 2e-3)                        # bias gradient = the column sums of the gradient it wrote (fp32 sums of pre-rounding values)
    # against the recomputing flavour: same call, fp32 logits
    loss2 = torch.empty(M, device="cuda")
    dl2 = torch.full((M, Vp), 7.0, dtype=BF, device="cuda")
    ops.vocab_ce(Xd, Wd, dev(b), dev(t, torch.int32), loss_rows=loss2, dlogits=dl2, materialize_bf16=False, **kw)
    d = (loss - loss2).abs().cpu().numpy()
    assert np.all(d <= 2.0 ** -7 * wrow * np.maximum(1.0, zmax) + 3e-5)
    assert float((dlb.float() - dl2.float()).abs().max()) * scale <= (2.0 ** -7 * max(1.0, float(zmax.max())) / 2 + 2.0 ** -7) * gmax + 1e-6


@pytest.mark.parametrize("case", [
    # N, H, W, Cin, Cout, k, stride, padding
    (1, 16, 16, 128, 64, 3, 1, 'same'),
    (2, 12, 20, 256, 256, 3, 1, 'same'),       # pixel count not a multiple of the 64-deep K-tile
    (1, 32, 32, 256, 512, 3, 1, 'same'),       # automatic split-K over the pixels
    (2, 8, 8, 1024, 256, 1, 1, 'valid'),       # an FPN lateral
    (1, 2, 2, 256, 256, 3, 1, 'same'),         # a P6-sized level: every tap partly outside
    (1, 128, 128, 256, 512, 3, 1, 'same'),     # the 256-square tile: rpn_conv_shared on a P3-sized level (one tap per column tile)
    (1, 96, 100, 128, 512, 3, 1, 'same'),      # the 256-square tile: two taps per column tile, ragged pixel count, partial last tile
    (2, 64, 64, 512, 256, 1, 1, 'valid'),      # a lateral (1x1) over two images: long K, small output
])
def test_conv2d_wgrad_bf16_matches_oracle(ops, case):
    """Weight gradient on the bf16 pipe (K-major dy^T x K-major im2col, both through transposing LDS reads) against the oracle's
    conv backward on operands rounded to bf16; accumulation over two pyramid levels like the shared RPN convolution."""
    N, H, W, Cin, Cout, k, stride, padding = case
    rng = np.random.default_rng(sum(int(v) * (i + 1) for i, v in enumerate(case) if not isinstance(v, str)))
    x = rng.standard_normal((N, H, W, Cin))
    w = rng.standard_normal((k, k, Cin, Cout))
    y = O.conv2d_nhwc(x, w, None, stride, padding)
    dy = rng.standard_normal(y.shape)
    _, dw, _ = O.conv2d_nhwc_backward(O.to_bf16(x), w, O.to_bf16(dy), stride, padding)           # dw [k,k,Cin,Cout]
    want = np.transpose(dw, (3, 0, 1, 2)).reshape(Cout, k * k * Cin)                             # packed [Cout][(ky,kx,ci)]
    pt, pl = (O.same_pad(H, k, stride)[0], O.same_pad(W, k, stride)[0]) if padding == 'same' else (0, 0)
    xb, dyb = ops.to_bf16(dev(x)), ops.to_bf16(dev(dy))
    assert ops.wgrad_bf16_supported(xb.shape, dyb.shape)
    info = {}
    got = ops.conv2d_wgrad_bf16(xb, dyb, k, k, stride, pt, pl, info=info)
    assert info["tile"] == (256 if H >= 96 else 128), info          # (the lateral's 256 x 512 output stays on the 128-square tile)
    close(got, want, 5e-5)
    ops.conv2d_wgrad_bf16(xb, dyb, k, k, stride, pt, pl, out=got, accumulate=True)
    close(got, 2 * want, 5e-5)


BCONV_CASES = [
    # N,H,W,Cin,Cout,k,stride,padding,res_mode,relu
    (1, 16, 16, 64, 64, 1, 1, 'valid', 0, True),
    (2, 16, 24, 64, 256, 1, 1, 'valid', 1, True),
    (1, 32, 32, 256, 128, 1, 2, 'valid', 0, True),          # strided 1x1 (stage-entry branches)
    (2, 12, 20, 64, 64, 3, 1, 'same', 0, True),             # ragged pixel count, borders
    (1, 16, 16, 128, 128, 3, 1, 'same', 0, False),
    (1, 8, 8, 512, 512, 3, 1, 'same', 0, True),             # small M, long K: split-K
    (1, 16, 16, 256, 256, 1, 1, 'valid', 2, False),         # FPN lateral + upsample-add
    (2, 9, 7, 128, 20, 1, 1, 'valid', 0, False),            # the padded RPN head: Cout = 20
    (1, 40, 40, 64, 192, 3, 1, 'same', 1, True),            # two column tiles, the second partial
]
BCONV_BIG = [   # the 256 x 256 tile (bconv256_kernel: grids of >= 192 tiles)
    (1, 128, 135, 64, 768, 3, 1, 'same', 1, True),          # 68 x 3 tiles, ragged pixel count, borders, residual
    (2, 96, 128, 128, 712, 1, 1, 'valid', 2, False),        # 96 x 3 tiles (the third 200 columns wide), upsample-add residual
    (1, 64, 64, 1024, 256, 3, 1, 'same', 0, True),          # 16 tiles, K = 9216: split-K on the large tile
]


BCONV_SMALL = [  # the 64 x 64 tile (bconv64_kernel: whole K loop per block): the one-image trunk layers of configs[4]
    (1, 64, 64, 256, 256, 3, 1, 'same', 0, True),           # res4*_2b at one image: 36 K-tiles through the four-stage ring
    (1, 64, 64, 1024, 256, 1, 1, 'valid', 1, True),         # res4*_2a
    (1, 32, 32, 512, 320, 3, 1, 'same', 0, False),          # 72 K-tiles, ragged columns (the cost model itself prefers split-K slabs here)
]


@pytest.mark.parametrize("tile", [0, 64, 128, 256])
@pytest.mark.parametrize("outs", ["f32", "bf16", "both"])
@pytest.mark.parametrize("case", BCONV_CASES + BCONV_BIG + BCONV_SMALL)
def test_conv2d_bf16_matches_oracle(ops, case, outs, tile):
    """dc_conv2d_bf16 (bf16 activations and weights in memory, LDS-DMA im2col) against the float64 oracle on the same
    bf16-rounded operands: only fp32 accumulation order separates them.  fp32 output, bf16 output (one more rounding) or both."""
    from image_captioning_amd.packing import pack_conv_kernel
    N, H, W, Cin, Cout, k, stride, padding, res_mode, relu = case
    rng = np.random.default_rng(sum(case[:7]))
    x = O.to_bf16(rng.standard_normal((N, H, W, Cin)))
    w = O.to_bf16(rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin))
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = (0.1 * rng.standard_normal(Cout)).astype(np.float32)
    y = O.conv2d_nhwc(x, w, None, stride, padding) * scale + shift
    Ho, Wo = y.shape[1:3]
    res = None
    if res_mode == 1:
        res = rng.standard_normal(y.shape).astype(np.float32)
        y = y + res
    elif res_mode == 2:
        res = rng.standard_normal((N, Ho // 2, Wo // 2, Cout)).astype(np.float32)
        y = y + res.repeat(2, axis=1).repeat(2, axis=2)
    if relu:
        y = np.maximum(y, 0)
    pad = (k - 1) // 2 if padding == 'same' else 0
    xb = ops.to_bf16(dev(x))
    wb = ops.to_bf16(dev(pack_conv_kernel(w.astype(np.float32))))
    info = {}
    got, gotb = ops.conv2d_bf16(xb, wb, k, k, stride, pad, pad, Ho, Wo, scale=dev(scale), shift=dev(shift),
                                residual=None if res is None else dev(res), res_mode=res_mode, relu=relu,
                                want_f32=outs != "bf16", want_bf16=outs != "f32", info=info, tile=tile,
                                split_k=16 if Cin == 1024 and k == 3 and tile != 64 else 0)   # (slabs on request: the cost model would not split that shape)
    if tile == 0:                                                  # the library's own choice
        assert info["tile"] == (256 if case in BCONV_BIG[:2] else 64 if case in BCONV_SMALL[:2] else info["tile"]), info
    elif tile == 256 and N * Ho * Wo * Cout * 4 < -(-N * Ho * Wo // 256) * -(-Cout // 256) * 65536:
        assert info["tile"] == 128                                 # a sliver of one 256-square tile is refused
    else:
        assert info["tile"] == tile, info
    if outs != "bf16":
        close(got, y, 3e-5)
    else:
        assert got is None
    if outs != "f32":
        want_b = O.to_bf16(got.cpu().numpy()) if outs == "both" else None
        gb = gotb.float().cpu().numpy()
        if want_b is not None:
            np.testing.assert_array_equal(gb, want_b)            # the bf16 copy is the rounded fp32 output, element for element
        assert np.abs(gb - y).max() <= 2.0 ** -8 * np.abs(y).max() + 1e-6
    else:
        assert gotb is None
