"""Thin torch-tensor wrappers over the C-ABI (include/dcap.h).  torch is plumbing here: device memory,
the current HIP stream and (elsewhere) torch.distributed; every arithmetic op is a dc_* kernel.
All wrappers enqueue on torch's current stream and never synchronise.
"""
import ctypes as C
import os

import numpy as np

import torch

from . import _lib
from ._lib import (PwChainDesc, ConvBf16Desc, AmsgradDesc, DetectionTargetsDesc, BnReluDesc, ConvDesc, ConvWgradBf16Desc, GemmBf16Desc, VocabCeDesc, ProposalDesc, RpnLossDesc, GemmDesc, LstmBwdDesc, LstmFwdDesc, RoiAlignDesc,
                   SoftmaxCeDesc, check)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, dtype=torch.float32, name="tensor"):
    if not t.is_cuda:
        raise _lib.DcapError("%s must live on the GPU (no CPU path exists)" % name)
    if t.dtype != dtype:
        raise _lib.DcapError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if t.dim() > 0 and t.stride(-1) != 1:
        raise _lib.DcapError("%s must be contiguous along its last dimension" % name)
    return t


class no_gc_during_capture(object):
    """Context manager for hipGraph captures: collect garbage BEFORE the capture and keep the cyclic collector off while it lasts.  A
    collection that happens to run inside a capture finalises whatever cyclic garbage is pending -- a dropped model's CUDAGraph, its
    page-locked buffers, its events -- and runtime calls made by those finalisers (hipGraphExecDestroy, hipHostFree, ...) are illegal
    on a capturing thread: round 4 saw the full GPU test suite die with SIGABRT inside `gc` in the middle of an encoder-plan capture."""

    def __enter__(self):
        import gc
        gc.collect()
        self._was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False


class _Workspace:
    """One grow-only scratch buffer per (device, stream) (kernels never allocate).  Per STREAM since round 4: the joint step runs its
    RPN backward on a side stream beside the proposals / decoder chain, and two streams must not scribble over one scratch buffer.
    A buffer that is first asked for DURING a hipGraph capture (a capture runs on a stream of its own, after eager steps have seen
    every size) starts at the largest size any stream of the device has asked for, so that the capture never has to grow (= free) a
    buffer whose address earlier nodes have baked; outside a capture a new stream's buffer is sized by what that stream asks for
    (round 5, ADVICE r4: side / pipeline streams no longer pin a copy of the device-wide high-water mark each).  release() drops the
    buffers of streams the caller has retired."""

    def __init__(self):
        self.buf = {}
        self.hi = {}

    def get(self, nbytes, device):
        if nbytes == 0:
            return None, 0
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        hi = max(self.hi.get(device, 0), nbytes)
        self.hi[device] = hi
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            want = hi if torch.cuda.is_current_stream_capturing() else nbytes + nbytes // 4
            b = torch.empty(max(want, 1 << 20), dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b, b.numel()

    def reserve(self, nbytes, device):
        self.get(nbytes, device)

    def current(self, device=None):
        """The scratch buffer of the current stream (or None): a captured graph keeps a reference to the one its launches point into."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        return self.buf.get((device, torch.cuda.current_stream(device).cuda_stream))

    def release(self, stream):
        """Drop the scratch buffer of a torch.cuda.Stream the caller is done with (no captured graph may still reference it)."""
        for key in [k for k in self.buf if k[1] == stream.cuda_stream]:
            del self.buf[key]


WORKSPACE = _Workspace()


def gemm(A, B, out=None, a_trans=False, b_trans=False, gather=None, scale=None, shift=None,
         residual=None, res_rows=0, relu=False, accumulate=False, split_k=0):
    """out[M,N] = epilogue(op(A) @ op(B)); see dc_gemm_f32."""
    lib = _lib.load()
    _chk(A, name="A"), _chk(B, name="B")
    if a_trans:
        K, M = (gather.numel() if gather is not None else A.shape[0]), A.shape[1]
    elif gather is not None:
        M, K = gather.numel(), A.shape[1]
    else:
        M, K = A.shape
    N = B.shape[0] if b_trans else B.shape[1]
    kb = B.shape[1] if b_trans else B.shape[0]
    if kb != K:
        raise _lib.DcapError("gemm: inner dimensions differ (%d vs %d)" % (K, kb))
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    _chk(out, name="out")
    if tuple(out.shape) != (M, N):
        raise _lib.DcapError("gemm: out has shape %s, expected %s" % (tuple(out.shape), (M, N)))
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.A, d.lda, d.a_trans = A.data_ptr(), A.stride(0), int(a_trans)
    d.a_gather = None if gather is None else _chk(gather, torch.int32, "gather").data_ptr()
    d.B, d.ldb, d.b_trans = B.data_ptr(), B.stride(0), int(b_trans)
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    d.scale = None if scale is None else _chk(scale, name="scale").data_ptr()
    d.shift = None if shift is None else _chk(shift, name="shift").data_ptr()
    if residual is not None:
        _chk(residual, name="residual")
        d.residual, d.ldr, d.res_rows = residual.data_ptr(), residual.stride(0), int(res_rows)
    d.relu, d.accumulate, d.split_k = int(relu), int(accumulate), int(split_k)
    ws, wsb = WORKSPACE.get(lib.dc_gemm_workspace_bytes(C.byref(d)), A.device)
    check(lib.dc_gemm_f32(C.byref(d), _ptr(ws), wsb, _stream()), "dc_gemm_f32")
    return out


BF16 = torch.bfloat16


def from_bf16(x, out):
    """bf16 -> fp32 (exact) into `out` (dc_cast_bf16_f32); both contiguous, one size."""
    lib = _lib.load()
    if not (x.is_contiguous() and out.is_contiguous()) or x.numel() != out.numel():
        raise _lib.DcapError("from_bf16: contiguous tensors of one size")
    check(lib.dc_cast_bf16_f32(_ptr(_chk(x, BF16, "x")), _ptr(_chk(out, name="out")), x.numel(), _stream()), "dc_cast_bf16_f32")
    return out


def to_bf16(x, out=None, pad_cols=None):
    """fp32 -> bf16 (round to nearest even) on the device.  pad_cols: for a 2-D x, the output's column count (>= x.shape[1],
    extra columns zero) -- pads a contraction dimension to the multiple of 8 dc_gemm_bf16 wants."""
    lib = _lib.load()
    _chk(x, name="x")
    if pad_cols is not None or (x.dim() == 2 and not x.is_contiguous()):
        rows, cols = x.shape
        co = cols if pad_cols is None else int(pad_cols)
        if out is None:
            out = torch.empty((rows, co), dtype=BF16, device=x.device)
        check(lib.dc_cast_f32_bf16_2d(_ptr(x), x.stride(0), _ptr(out), out.stride(0), rows, cols, co, _stream()), "dc_cast_f32_bf16_2d")
        return out
    if not x.is_contiguous():
        raise _lib.DcapError("to_bf16: x must be contiguous (or 2-D with unit column stride)")
    if out is None:
        out = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(lib.dc_cast_f32_bf16(_ptr(x), _ptr(out), x.numel(), _stream()), "dc_cast_f32_bf16")
    return out


def gemm_bf16(A, B, out=None, out_bf16=None, a_trans=False, b_trans=False, gather=None, scale=None, shift=None,
              residual=None, res_rows=0, relu=False, accumulate=False, split_k=0, info=None):
    """op(A) @ op(B) with bf16 operands, fp32 accumulate (dc_gemm_bf16).  out: fp32 [M,N] result (allocated when neither
    out nor out_bf16 is given); out_bf16: optional bf16 [M,N] copy of the result.  Returns out if present else out_bf16."""
    lib = _lib.load()
    _chk(A, BF16, "A"), _chk(B, BF16, "B")
    if a_trans:
        K, M = (gather.numel() if gather is not None else A.shape[0]), A.shape[1]
    elif gather is not None:
        M, K = gather.numel(), A.shape[1]
    else:
        M, K = A.shape
    N = B.shape[0] if b_trans else B.shape[1]
    kb = B.shape[1] if b_trans else B.shape[0]
    if kb != K:
        raise _lib.DcapError("gemm_bf16: inner dimensions differ (%d vs %d)" % (K, kb))
    if out is None and out_bf16 is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    d = GemmBf16Desc()
    d.M, d.N, d.K = M, N, K
    d.A, d.lda, d.a_trans = A.data_ptr(), A.stride(0), int(a_trans)
    if gather is not None:
        d.a_gather, d.a_gather_rows = _chk(gather, torch.int32, "gather").data_ptr(), A.shape[0]
    d.B, d.ldb, d.b_trans = B.data_ptr(), B.stride(0), int(b_trans)
    for t, name, dt in ((out, "out", torch.float32), (out_bf16, "out_bf16", BF16)):
        if t is not None and tuple(_chk(t, dt, name).shape) != (M, N):
            raise _lib.DcapError("gemm_bf16: %s has shape %s, expected %s" % (name, tuple(t.shape), (M, N)))
    if out is not None:
        d.C, d.ldc = out.data_ptr(), out.stride(0)
    if out_bf16 is not None:
        d.Cb, d.ldcb = out_bf16.data_ptr(), out_bf16.stride(0)
    d.scale = None if scale is None else _chk(scale, name="scale").data_ptr()
    d.shift = None if shift is None else _chk(shift, name="shift").data_ptr()
    if residual is not None:
        _chk(residual, name="residual")
        d.residual, d.ldr, d.res_rows = residual.data_ptr(), residual.stride(0), int(res_rows)
    d.relu, d.accumulate, d.split_k = int(relu), int(accumulate), int(split_k)
    if info is not None:                      # tests / benches: which kernel runs this problem
        sk = C.c_int(0)
        info["tile"] = int(lib.dc_gemm_bf16_tile(C.byref(d), C.byref(sk)))
        info["split_k"] = int(sk.value)
    ws, wsb = WORKSPACE.get(lib.dc_gemm_bf16_workspace_bytes(C.byref(d)), A.device)
    check(lib.dc_gemm_bf16(C.byref(d), _ptr(ws), wsb, _stream()), "dc_gemm_bf16")
    return out if out is not None else out_bf16


def conv2d(x, w_packed, kh, kw, stride, pad_t, pad_l, Ho, Wo, scale=None, shift=None, residual=None,
           res_mode=0, relu=False, out=None, split_k=0, math=0, w_wino=None, w_wino_b3=None, _name_only=False):
    """NHWC conv forward with fused epilogue; see dc_conv2d_nhwc_f32.  x [N,H,W,Cin] contiguous.  w_wino: winograd_pack(w_packed)
    of a frozen 3x3 / stride 1 / pad 1 kernel -> the layer runs in the Winograd F(2x2, 3x3) form (fp32, math=0 only)."""
    lib = _lib.load()
    _chk(x, name="x"), _chk(w_packed, name="w")
    N, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    if not x.is_contiguous() or not w_packed.is_contiguous():
        raise _lib.DcapError("conv2d: x and w must be contiguous")
    if out is None:
        out = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin = N, H, W, Cin
    d.Cout, d.kh, d.kw, d.stride, d.pad_t, d.pad_l, d.Ho, d.Wo = Cout, kh, kw, stride, pad_t, pad_l, Ho, Wo
    d.x, d.w, d.y = x.data_ptr(), w_packed.data_ptr(), _chk(out, name="out").data_ptr()
    d.scale = None if scale is None else scale.data_ptr()
    d.shift = None if shift is None else shift.data_ptr()
    d.residual = None if residual is None else _chk(residual, name="residual").data_ptr()
    d.res_mode, d.relu, d.split_k, d.math = int(res_mode), int(relu), int(split_k), int(math)
    if w_wino is not None:
        if not w_wino.is_contiguous() or w_wino.numel() != 16 * Cin * Cout:
            raise _lib.DcapError("conv2d: w_wino must be the contiguous winograd_pack() of this layer's kernel (16*Cin*Cout floats)")
        d.w_wino = _chk(w_wino, name="w_wino").data_ptr()
    if w_wino_b3 is not None:
        if not w_wino_b3.is_contiguous() or w_wino_b3.numel() != 48 * Cin * Cout:
            raise _lib.DcapError("conv2d: w_wino_b3 must be the contiguous winograd_pack_b3() of this layer's kernel (48*Cin*Cout bf16)")
        d.w_wino_b3 = _chk(w_wino_b3, torch.int16, "w_wino_b3").data_ptr()
    if _name_only:
        buf = C.create_string_buffer(128)
        check(lib.dc_conv2d_kernel_name(C.byref(d), buf, 128), "dc_conv2d_kernel_name")
        return buf.value.decode()
    ws, wsb = WORKSPACE.get(lib.dc_conv2d_workspace_bytes(C.byref(d)), x.device)
    check(lib.dc_conv2d_nhwc_f32(C.byref(d), _ptr(ws), wsb, _stream()), "dc_conv2d_nhwc_f32")
    return out


def conv2d_kernel_name(*args, **kw):
    """The kernel template instantiation conv2d(*args, **kw) would launch (dc_conv2d_kernel_name); nothing is launched."""
    return conv2d(*args, _name_only=True, **kw)


def pw_chain_supported(k1, n1, n2):
    return bool(_lib.load().dc_pw_chain_supported(int(k1), int(n1), int(n2)))


def pw_chain_pack(w_packed, out=None):
    """A packed 1x1 kernel [Cout, Cin] in the fragment order dc_pw_chain_f32 reads (same size)."""
    lib = _lib.load()
    _chk(w_packed, name="w")
    if w_packed.dim() != 2 or not w_packed.is_contiguous():
        raise _lib.DcapError("pw_chain_pack: w must be the contiguous packed 1x1 kernel [Cout, Cin]")
    if out is None:
        out = torch.empty_like(w_packed)
    check(lib.dc_pw_chain_pack_f32(_ptr(w_packed), _ptr(out), w_packed.shape[0], w_packed.shape[1], _stream()), "dc_pw_chain_pack_f32")
    return out


def pw_chain_pack_b3(w_packed, out=None):
    """A packed 1x1 kernel [Cout, Cin] as three bf16 pieces per element in the fragment order of the split-bf16 chain kernel:
    int16 [Cout, 3 * Cin] (dc_pw_chain_pack_b3)."""
    lib = _lib.load()
    _chk(w_packed, name="w")
    if w_packed.dim() != 2 or not w_packed.is_contiguous():
        raise _lib.DcapError("pw_chain_pack_b3: w must be the contiguous packed 1x1 kernel [Cout, Cin]")
    if out is None:
        out = torch.empty((w_packed.shape[0], 3 * w_packed.shape[1]), dtype=torch.int16, device=w_packed.device)
    check(lib.dc_pw_chain_pack_b3(_ptr(w_packed), _ptr(out), w_packed.shape[0], w_packed.shape[1], _stream()), "dc_pw_chain_pack_b3")
    return out


def pw_chain(x, w1f, shift1, w2f, shift2, scale1=None, scale2=None, residual=None, relu1=True, relu2=True, y=None, z=None):
    """y = act1((x W1^T) scale1 + shift1 [+ residual]); z = act2((y W2^T) scale2 + shift2) in one launch (dc_pw_chain_f32).
    x [M, K1] (any leading dims), w1f / w2f: pw_chain_pack() of the packed kernels [N1, K1] / [N2, N1] (fp32 MFMA products), or both
    pw_chain_pack_b3() (int16 [N, 3 K]: the products on the bf16 pipe in split arithmetic).  Returns (y, z)."""
    lib = _lib.load()
    K1 = x.shape[-1]
    M = x.numel() // K1
    N1, N2 = w1f.shape[0], w2f.shape[0]
    b3 = w1f.dtype == torch.int16
    if b3 != (w2f.dtype == torch.int16):
        raise _lib.DcapError("pw_chain: both kernels in the same form (pw_chain_pack or pw_chain_pack_b3)")
    if w1f.shape[1] != (3 if b3 else 1) * K1 or w2f.shape[1] != (3 if b3 else 1) * N1:
        raise _lib.DcapError("pw_chain: kernel shapes do not chain: x [.., %d], w1 %s, w2 %s" % (K1, tuple(w1f.shape), tuple(w2f.shape)))
    if y is None:
        y = torch.empty(tuple(x.shape[:-1]) + (N1,), dtype=torch.float32, device=x.device)
    if z is None:
        z = torch.empty(tuple(x.shape[:-1]) + (N2,), dtype=torch.float32, device=x.device)
    for t in (x, y, z) + (() if residual is None else (residual,)):
        if not t.is_contiguous():
            raise _lib.DcapError("pw_chain: tensors must be contiguous")
    d = PwChainDesc()
    d.M, d.K1, d.N1, d.N2 = M, K1, N1, N2
    d.x, d.shift1, d.y = _chk(x, name="x").data_ptr(), _chk(shift1, name="shift1").data_ptr(), _chk(y, name="y").data_ptr()
    d.shift2, d.z = _chk(shift2, name="shift2").data_ptr(), _chk(z, name="z").data_ptr()
    if b3:
        d.w1_b3, d.w2_b3 = _chk(w1f, torch.int16, "w1").data_ptr(), _chk(w2f, torch.int16, "w2").data_ptr()
    else:
        d.w1, d.w2 = _chk(w1f, name="w1").data_ptr(), _chk(w2f, name="w2").data_ptr()
    d.scale1 = None if scale1 is None else _chk(scale1, name="scale1").data_ptr()
    d.scale2 = None if scale2 is None else _chk(scale2, name="scale2").data_ptr()
    d.residual = None if residual is None else _chk(residual, name="residual").data_ptr()
    d.relu1, d.relu2 = int(relu1), int(relu2)
    check(lib.dc_pw_chain_f32(C.byref(d), _stream()), "dc_pw_chain_f32")
    return y, z


def winograd_pack_b3(w_packed, cin, cout, out=None):
    """U = G g G^T of a packed 3x3 kernel [Cout, 9*Cin], every element as three bf16 pieces, in the fragment order of the split-bf16
    Winograd kernel: int16 [48*Cin*Cout] (dc_conv2d_winograd_pack_b3)."""
    lib = _lib.load()
    _chk(w_packed, name="w")
    if not w_packed.is_contiguous() or w_packed.numel() != 9 * cin * cout:
        raise _lib.DcapError("winograd_pack_b3: w must be the contiguous packed 3x3 kernel [Cout, 9*Cin]")
    nbytes = lib.dc_conv2d_winograd_b3_weight_bytes(cin, cout)
    if nbytes == 0:
        raise _lib.DcapError("winograd_pack_b3: Cin and Cout must be multiples of 32")
    if out is None:
        out = torch.empty((nbytes // 2,), dtype=torch.int16, device=w_packed.device)
    check(lib.dc_conv2d_winograd_pack_b3(_ptr(w_packed), _ptr(out), cin, cout, _stream()), "dc_conv2d_winograd_pack_b3")
    return out


def winograd_pack(w_packed, cin, cout, out=None):
    """U = G g G^T of a packed 3x3 kernel [Cout, 9*Cin] in the fragment order the Winograd kernel reads: fp32 [16*Cin*Cout]."""
    lib = _lib.load()
    _chk(w_packed, name="w")
    if not w_packed.is_contiguous() or w_packed.numel() != 9 * cin * cout:
        raise _lib.DcapError("winograd_pack: w must be the contiguous packed 3x3 kernel [Cout, 9*Cin]")
    nbytes = lib.dc_conv2d_winograd_weight_bytes(cin, cout)
    if nbytes == 0:
        raise _lib.DcapError("winograd_pack: Cin and Cout must be multiples of 32")
    if out is None:
        out = torch.empty((nbytes // 4,), dtype=torch.float32, device=w_packed.device)
    check(lib.dc_conv2d_winograd_pack_f32(_ptr(w_packed), _ptr(out), cin, cout, _stream()), "dc_conv2d_winograd_pack_f32")
    return out


def split_bf16x3(x, out=None):
    """fp32 tensor -> int16 tensor [3, *x.shape] of bf16 bit patterns with x = p0 + p1 + p2 (pre-split conv weights)."""
    lib = _lib.load()
    _chk(x, name="x")
    if not x.is_contiguous():
        raise _lib.DcapError("split_bf16x3: x must be contiguous")
    if out is None:
        out = torch.empty((3,) + tuple(x.shape), dtype=torch.int16, device=x.device)
    check(lib.dc_split_bf16x3_f32(_ptr(x), _ptr(out), x.numel(), _stream()), "dc_split_bf16x3_f32")
    return out


def conv2d_wgrad(x, dy, kh, kw, stride, pad_t, pad_l, out=None, split_k=0, accumulate=False):
    """dw packed [Cout][kh*kw*Cin] = sum_pixels dy (x) im2col(x); x [N,H,W,Cin], dy [N,Ho,Wo,Cout] contiguous."""
    lib = _lib.load()
    _chk(x, name="x"), _chk(dy, name="dy")
    N, H, W, Cin = x.shape
    _, Ho, Wo, Cout = dy.shape
    if not x.is_contiguous() or not dy.is_contiguous():
        raise _lib.DcapError("conv2d_wgrad: x and dy must be contiguous")
    if out is None:
        out = torch.empty((Cout, kh * kw * Cin), dtype=torch.float32, device=x.device)
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin = N, H, W, Cin
    d.Cout, d.kh, d.kw, d.stride, d.pad_t, d.pad_l, d.Ho, d.Wo = Cout, kh, kw, stride, pad_t, pad_l, Ho, Wo
    d.x, d.y, d.w, d.split_k = x.data_ptr(), dy.data_ptr(), _chk(out, name="dw").data_ptr(), int(split_k)
    d.accumulate = int(accumulate)
    ws, wsb = WORKSPACE.get(lib.dc_conv2d_wgrad_workspace_bytes(C.byref(d)), x.device)
    check(lib.dc_conv2d_wgrad_f32(C.byref(d), _ptr(ws), wsb, _stream()), "dc_conv2d_wgrad_f32")
    return out


def conv2d_wgrad_bf16(x, dy, kh, kw, stride, pad_t, pad_l, out=None, split_k=0, accumulate=False, info=None):
    """conv2d_wgrad on the bf16 pipe: x [N,H,W,Cin] and dy [N,Ho,Wo,Cout] contiguous bf16; dw fp32 packed [Cout][kh*kw*Cin]."""
    lib = _lib.load()
    _chk(x, BF16, "x"), _chk(dy, BF16, "dy")
    N, H, W, Cin = x.shape
    _, Ho, Wo, Cout = dy.shape
    if not x.is_contiguous() or not dy.is_contiguous():
        raise _lib.DcapError("conv2d_wgrad_bf16: x and dy must be contiguous")
    if out is None:
        out = torch.empty((Cout, kh * kw * Cin), dtype=torch.float32, device=x.device)
    d = ConvWgradBf16Desc()
    d.N, d.H, d.W, d.Cin = N, H, W, Cin
    d.Cout, d.kh, d.kw, d.stride, d.pad_t, d.pad_l, d.Ho, d.Wo = Cout, kh, kw, stride, pad_t, pad_l, Ho, Wo
    d.x, d.dy, d.dw = x.data_ptr(), dy.data_ptr(), _chk(out, name="dw").data_ptr()
    d.accumulate, d.split_k = int(accumulate), int(split_k)
    if info is not None:
        sk = C.c_int(0)
        info["tile"] = int(lib.dc_conv2d_wgrad_bf16_tile(C.byref(d), C.byref(sk)))
        info["split_k"] = int(sk.value)
    ws, wsb = WORKSPACE.get(lib.dc_conv2d_wgrad_bf16_workspace_bytes(C.byref(d)), x.device)
    check(lib.dc_conv2d_wgrad_bf16(C.byref(d), _ptr(ws), wsb, _stream()), "dc_conv2d_wgrad_bf16")
    return out


def conv2d_bf16(x, w, kh, kw, stride, pad_t, pad_l, Ho, Wo, scale=None, shift=None, residual=None, res_mode=0, relu=False,
                out=None, out_bf16=None, want_f32=True, want_bf16=False, split_k=0, info=None, tile=0):
    """conv2d with bf16 storage (dc_conv2d_bf16): x [N,H,W,Cin] bf16 (Cin % 64 == 0), w packed [Cout, kh*kw*Cin] bf16; the fp32
    epilogue operands as in conv2d.  Returns (fp32 output or None, bf16 output or None)."""
    lib = _lib.load()
    _chk(x, BF16, "x"), _chk(w, BF16, "w")
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    if not x.is_contiguous() or not w.is_contiguous() or w.shape[1] != kh * kw * Cin:
        raise _lib.DcapError("conv2d_bf16: x must be contiguous NHWC, w contiguous [Cout, kh*kw*Cin]")
    if out is None and want_f32:
        out = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    if out_bf16 is None and want_bf16:
        out_bf16 = torch.empty((N, Ho, Wo, Cout), dtype=BF16, device=x.device)
    d = ConvBf16Desc()
    d.N, d.H, d.W, d.Cin = N, H, W, Cin
    d.Cout, d.kh, d.kw, d.stride, d.pad_t, d.pad_l, d.Ho, d.Wo = Cout, kh, kw, stride, pad_t, pad_l, Ho, Wo
    d.x, d.w = x.data_ptr(), w.data_ptr()
    d.y = None if out is None else _chk(out, name="out").data_ptr()
    d.y_bf16 = None if out_bf16 is None else _chk(out_bf16, BF16, "out_bf16").data_ptr()
    d.scale = None if scale is None else _chk(scale, name="scale").data_ptr()
    d.shift = None if shift is None else _chk(shift, name="shift").data_ptr()
    d.residual = None if residual is None else _chk(residual, name="residual").data_ptr()
    d.res_mode, d.relu, d.split_k, d.tile = int(res_mode), int(relu), int(split_k), int(tile)
    if info is not None:                      # tests / benches: which kernel runs this layer
        sk = C.c_int(0)
        info["tile"] = int(lib.dc_conv2d_bf16_tile(C.byref(d), C.byref(sk)))
        info["split_k"] = int(sk.value)
    ws, wsb = WORKSPACE.get(lib.dc_conv2d_bf16_workspace_bytes(C.byref(d)), x.device)
    check(lib.dc_conv2d_bf16(C.byref(d), _ptr(ws), wsb, _stream()), "dc_conv2d_bf16")
    return out, out_bf16


def conv_bf16_supported(cin):
    return cin % 64 == 0


def wgrad_bf16_supported(x_shape, dy_shape):
    return x_shape[-1] % 128 == 0 and dy_shape[-1] % 8 == 0


def downsample2x_sum(fine, out=None, accumulate=False, out_bf16=None):
    """out (+)= the 2x2 block sums of fine [N,H,W,C] (the adjoint of UpSampling2D(2)); out_bf16: the bf16 copy of out, same pass."""
    lib = _lib.load()
    _chk(fine, name="fine")
    N, H, W, Cc = fine.shape
    if out is None:
        out = torch.empty((N, H // 2, W // 2, Cc), dtype=torch.float32, device=fine.device)
    check(lib.dc_downsample2x_sum_dual_f32(_ptr(fine), _ptr(out), None if out_bf16 is None else _ptr(_chk(out_bf16, BF16, "out_bf16")), N, H // 2, W // 2, Cc,
                                           int(accumulate), _stream()), "dc_downsample2x_sum_dual_f32")
    return out


def roi_align_pyramid_bwd(dmaps, boxes, image_area, dout, pool=7):
    """dmaps: four zero-initialised gradient maps [B,H,W,C] (accumulated into); dout [B,R,pool,pool,C]."""
    lib = _lib.load()
    B, R, _ = boxes.shape
    d = RoiAlignDesc()
    d.B, d.R, d.C, d.pool = B, R, dmaps[0].shape[-1], pool
    for i, m in enumerate(dmaps):
        if not _chk(m, name="dmap").is_contiguous():
            raise _lib.DcapError("roi_align_bwd: gradient maps must be contiguous")
        d.maps[i] = m.data_ptr()
        d.Hs[i], d.Ws[i] = m.shape[1], m.shape[2]
    d.boxes, d.image_area = _chk(boxes, name="boxes").data_ptr(), float(image_area)
    d.out = _chk(dout, name="dout").data_ptr()
    check(lib.dc_roi_align_pyramid_bwd_f32(C.byref(d), _stream()), "dc_roi_align_pyramid_bwd_f32")
    return dmaps


def maxpool3x3s2_same(x, out=None):
    lib = _lib.load()
    _chk(x, name="x")
    N, H, W, Cc = x.shape
    if out is None:
        out = torch.empty((N, (H + 1) // 2, (W + 1) // 2, Cc), dtype=torch.float32, device=x.device)
    check(lib.dc_maxpool3x3s2_same_f32(_ptr(x), _ptr(out), N, H, W, Cc, _stream()), "dc_maxpool3x3s2_same_f32")
    return out


def maxpool2x2s2(x, out=None):
    """MaxPooling2D((2, 2), strides 2) on NHWC float32 (even H, W)."""
    lib = _lib.load()
    _chk(x, name="x")
    N, H, W, Cc = x.shape
    if out is None:
        out = torch.empty((N, H // 2, W // 2, Cc), dtype=torch.float32, device=x.device)
    check(lib.dc_maxpool2x2s2_f32(_ptr(x), _ptr(out), N, H, W, Cc, _stream()), "dc_maxpool2x2s2_f32")
    return out


def mold_image_rgbx(img_u8, mean_pixel, out=None):
    lib = _lib.load()
    _chk(img_u8, torch.uint8, "images")
    N, H, W, c = img_u8.shape
    if c != 3 or not img_u8.is_contiguous():
        raise _lib.DcapError("mold_image: images must be contiguous [N,H,W,3] uint8")
    if out is None:
        out = torch.empty((N, H, W, 4), dtype=torch.float32, device=img_u8.device)
    check(lib.dc_mold_image_rgbx_f32(_ptr(img_u8), _ptr(out), N, H, W, float(mean_pixel[0]), float(mean_pixel[1]),
                                     float(mean_pixel[2]), _stream()), "dc_mold_image_rgbx_f32")
    return out


def mold_image_padded(img_u8, mean_pixel, out):
    """mold_image into pixels zero-padded to out.shape[-1] channels (a multiple of 4)."""
    lib = _lib.load()
    _chk(img_u8, torch.uint8, "images"), _chk(out, name="out")
    N, H, W, c = img_u8.shape
    if c != 3 or not img_u8.is_contiguous() or not out.is_contiguous() or tuple(out.shape[:3]) != (N, H, W):
        raise _lib.DcapError("mold_image_padded: images contiguous [N,H,W,3] uint8, out contiguous [N,H,W,C]")
    check(lib.dc_mold_image_padded_f32(_ptr(img_u8), _ptr(out), N, H, W, out.shape[3], float(mean_pixel[0]), float(mean_pixel[1]),
                                       float(mean_pixel[2]), _stream()), "dc_mold_image_padded_f32")
    return out


def roi_align_pyramid(maps, boxes, image_area, pool=7, out=None, levels_out=None):
    """maps: [P2,P3,P4,P5] each [B,H,W,C]; boxes [B,R,4] normalised float32 -> [B,R,pool,pool,C]."""
    lib = _lib.load()
    B, R, _ = boxes.shape
    Cc = maps[0].shape[-1]
    _chk(boxes, name="boxes")
    if not boxes.is_contiguous():
        raise _lib.DcapError("roi_align: boxes must be contiguous")
    if out is None:
        out = torch.empty((B, R, pool, pool, Cc), dtype=torch.float32, device=boxes.device)
    d = RoiAlignDesc()
    d.B, d.R, d.C, d.pool = B, R, Cc, pool
    for i, m in enumerate(maps):
        _chk(m, name="map")
        if not m.is_contiguous() or m.shape[0] != B or m.shape[-1] != Cc:
            raise _lib.DcapError("roi_align: feature maps must be contiguous [B,H,W,C]")
        d.maps[i] = m.data_ptr()
        d.Hs[i], d.Ws[i] = m.shape[1], m.shape[2]
    d.boxes, d.image_area, d.out = boxes.data_ptr(), float(image_area), out.data_ptr()
    d.levels_out = None if levels_out is None else _chk(levels_out, torch.int32, "levels").data_ptr()
    check(lib.dc_roi_align_pyramid_f32(C.byref(d), _stream()), "dc_roi_align_pyramid_f32")
    return out


def subsample2(x, out=None):
    lib = _lib.load()
    _chk(x, name="x")
    N, H, W, Cc = x.shape
    if out is None:
        out = torch.empty((N, (H + 1) // 2, (W + 1) // 2, Cc), dtype=torch.float32, device=x.device)
    check(lib.dc_subsample2_f32(_ptr(x), _ptr(out), N, H, W, Cc, _stream()), "dc_subsample2_f32")
    return out


def rpn_proposals(heads, anchors, image_hw, proposal_count, nms_threshold, std_dev=(0.1, 0.1, 0.2, 0.2), pre_nms_limit=6000,
                  anchors_per_loc=3, out=None, debug=False, head_stride=0):
    """heads: per-level fused RPN head outputs [B,H,W,A*6]; anchors [A_total,4] float32 device tensor.
    Returns proposals [B,count,4] (normalised, zero padded) and, with debug, (scores, order, keep)."""
    lib = _lib.load()
    B = heads[0].shape[0]
    d = ProposalDesc()
    d.B, d.levels, d.anchors_per_loc = B, len(heads), anchors_per_loc
    for i, h in enumerate(heads):
        if not _chk(h, name="head").is_contiguous() or h.shape[-1] != (head_stride or anchors_per_loc * 6):
            raise _lib.DcapError("rpn_proposals: heads must be contiguous [B,H,W,A*6] (or [B,H,W,head_stride])")
        d.heads[i] = h.data_ptr()
        d.Hs[i], d.Ws[i] = h.shape[1], h.shape[2]
    d.head_stride = int(head_stride)
    d.anchors, d.A_total = _chk(anchors, name="anchors").data_ptr(), anchors.shape[0]
    for i in range(4):
        d.std_dev[i] = float(std_dev[i])
    d.image_h, d.image_w = float(image_hw[0]), float(image_hw[1])
    d.pre_nms_limit, d.proposal_count, d.nms_threshold = int(pre_nms_limit), int(proposal_count), float(nms_threshold)
    dev = heads[0].device
    if out is None:
        out = torch.empty((B, proposal_count, 4), dtype=torch.float32, device=dev)
    d.proposals = out.data_ptr()
    extra = None
    if debug:
        k = min(pre_nms_limit, anchors.shape[0])
        extra = (torch.empty((B, anchors.shape[0]), dtype=torch.float32, device=dev),
                 torch.empty((B, k), dtype=torch.int32, device=dev),
                 torch.empty((B, proposal_count), dtype=torch.int32, device=dev))
        d.scores_out, d.order_out, d.keep_out = (t.data_ptr() for t in extra)
    ws, wsb = WORKSPACE.get(lib.dc_proposals_workspace_bytes(C.byref(d)), dev)
    check(lib.dc_proposals_f32(C.byref(d), _ptr(ws), wsb, _stream()), "dc_proposals_f32")
    return (out, extra) if debug else out


def _rec_masks(rec_masks, B, U):
    if rec_masks is None:
        return None
    if tuple(_chk(rec_masks, name="rec_masks").shape) != (4, B, U) or not rec_masks.is_contiguous():
        raise _lib.DcapError("rec_masks must be a contiguous float32 [4,B,U] tensor")
    return rec_masks.data_ptr()


def lstm_seq_fwd(z, U_rec, mask, B, T, h_seq=None, c_seq=None, rec_masks=None):
    """z [T*B,4U] (x-projection + bias, overwritten with the full pre-activation) -> h_seq, c_seq [T*B,U].
    rec_masks [4,B,U]: Keras recurrent_dropout masks (training phase); None = no dropout."""
    lib = _lib.load()
    U = U_rec.shape[0]
    _chk(z, name="z"), _chk(U_rec, name="U_rec")
    if not z.is_contiguous() or not U_rec.is_contiguous() or tuple(z.shape) != (T * B, 4 * U):
        raise _lib.DcapError("lstm_seq_fwd: z must be contiguous [T*B,4U], U_rec contiguous [U,4U]")
    if h_seq is None:
        h_seq = torch.empty((T * B, U), dtype=torch.float32, device=z.device)
    if c_seq is None:
        c_seq = torch.empty((T * B, U), dtype=torch.float32, device=z.device)
    d = LstmFwdDesc()
    d.B, d.T, d.U = B, T, U
    d.z, d.U_rec = z.data_ptr(), U_rec.data_ptr()
    d.mask = None if mask is None else _chk(mask, torch.uint8, "mask").data_ptr()
    d.h_seq, d.c_seq = h_seq.data_ptr(), c_seq.data_ptr()
    d.rec_masks = _rec_masks(rec_masks, B, U)
    ws, wsb = WORKSPACE.get(lib.dc_lstm_seq_workspace_bytes(B, T, U), z.device)
    check(lib.dc_lstm_seq_fwd_f32(C.byref(d), _ptr(ws), wsb, _stream()), "dc_lstm_seq_fwd_f32")
    return h_seq, c_seq


def lstm_seq_bwd(z, U_rec, mask, h_seq, c_seq, B, T, dh_seq=None, dh_last=None, dz=None, dU=None, accumulate_dU=False, rec_masks=None):
    lib = _lib.load()
    U = U_rec.shape[0]
    if dz is None:
        dz = torch.empty((T * B, 4 * U), dtype=torch.float32, device=z.device)
    if dU is None:
        dU = torch.empty((U, 4 * U), dtype=torch.float32, device=z.device)
    elif dU is False:                                  # the caller forms the recurrent kernel's gradient itself (returns (dz, None))
        dU = None
    d = LstmBwdDesc()
    d.B, d.T, d.U = B, T, U
    d.z, d.U_rec = _chk(z, name="z").data_ptr(), _chk(U_rec, name="U_rec").data_ptr()
    d.mask = None if mask is None else _chk(mask, torch.uint8, "mask").data_ptr()
    d.h_seq, d.c_seq = h_seq.data_ptr(), c_seq.data_ptr()
    for name, t in (("dh_seq", dh_seq), ("dh_last", dh_last)):
        if t is not None and not _chk(t, name=name).is_contiguous():
            raise _lib.DcapError("lstm_seq_bwd: %s must be contiguous" % name)
    d.dh_seq = None if dh_seq is None else dh_seq.data_ptr()
    d.dh_last = None if dh_last is None else dh_last.data_ptr()
    d.dz, d.dU_rec, d.accumulate_dU = dz.data_ptr(), (None if dU is None else _chk(dU, name="dU").data_ptr()), int(accumulate_dU)
    d.rec_masks = _rec_masks(rec_masks, B, U)
    ws, wsb = WORKSPACE.get(lib.dc_lstm_seq_workspace_bytes(B, T, U), z.device)
    check(lib.dc_lstm_seq_bwd_f32(C.byref(d), _ptr(ws), wsb, _stream()), "dc_lstm_seq_bwd_f32")
    return dz, dU


def softmax_ce(logits, targets=None, probs=None, loss_rows=None, dlogits=None, grad_scale=1.0, row_weights=None,
               keras_sparse=False):
    lib = _lib.load()
    _chk(logits, name="logits")
    d = SoftmaxCeDesc()
    d.M, d.V, d.ld = logits.shape[0], logits.shape[1], logits.stride(0)
    d.logits = logits.data_ptr()
    d.targets = None if targets is None else _chk(targets, torch.int32, "targets").data_ptr()
    for name, t in (("probs", probs), ("dlogits", dlogits)):
        if t is not None and (_chk(t, name=name).stride(0) != logits.stride(0)):
            raise _lib.DcapError("softmax_ce: %s must share the logits' row stride" % name)
    d.probs = None if probs is None else probs.data_ptr()
    d.loss_rows = None if loss_rows is None else _chk(loss_rows, name="loss_rows").data_ptr()
    d.dlogits = None if dlogits is None else dlogits.data_ptr()
    d.grad_scale = float(grad_scale)
    d.row_weights = None if row_weights is None else _chk(row_weights, name="row_weights").data_ptr()
    d.keras_sparse = int(keras_sparse)
    check(lib.dc_softmax_ce_f32(C.byref(d), _stream()), "dc_softmax_ce_f32")


def vocab_ce_supported(X, W):
    """Shapes the fused vocabulary softmax / cross-entropy takes (otherwise: gemm + softmax_ce)."""
    K, V = W.shape
    if X.dtype == BF16:
        return W.dtype == BF16 and K % 8 == 0 and V % 8 == 0 and X.stride(0) % 8 == 0 and W.stride(0) % 8 == 0
    return K % 32 == 0 and V % 4 == 0 and V >= 4 and X.stride(0) % 4 == 0 and W.stride(0) % 4 == 0


def vocab_ce(X, W, bias, targets, loss_rows=None, dlogits=None, dbias=None, grad_scale=1.0, row_weights=None, keras_sparse=False,
             materialize_bf16=None):
    """Fused Dense(V) + softmax + Keras cross-entropy (dc_vocab_ce): X [M,K] and W [K,V] both float32 or both bf16; the
    [M,V] logits are never materialised in fp32.  dlogits: float32 or bf16 [M, >=V] receives d(loss)/d(logits); dbias [V] its
    column sums.  materialize_bf16 (default: DCAP_VOCAB_MATERIALIZE, on): bf16 operands with a bf16 dlogits buffer -- the logits are
    rounded to bf16 and parked in that buffer by the first GEMM pass, the gradient is an in-place elementwise pass (dcap.h)."""
    lib = _lib.load()
    bf = X.dtype == BF16
    _chk(X, BF16 if bf else torch.float32, "X"), _chk(W, BF16 if bf else torch.float32, "W")
    M, K = X.shape
    if W.shape[0] != K:
        raise _lib.DcapError("vocab_ce: inner dimensions differ (%d vs %d)" % (K, W.shape[0]))
    d = VocabCeDesc()
    d.M, d.V, d.K, d.bf16 = M, W.shape[1], K, int(bf)
    d.X, d.ldx, d.W, d.ldw = X.data_ptr(), X.stride(0), W.data_ptr(), W.stride(0)
    d.bias = None if bias is None else _chk(bias, name="bias").data_ptr()
    d.targets = _chk(targets, torch.int32, "targets").data_ptr()
    d.row_weights = None if row_weights is None else _chk(row_weights, name="row_weights").data_ptr()
    d.grad_scale, d.keras_sparse = float(grad_scale), int(keras_sparse)
    d.loss_rows = None if loss_rows is None else _chk(loss_rows, name="loss_rows").data_ptr()
    if dlogits is not None:
        if dlogits.dtype not in (torch.float32, BF16) or not dlogits.is_cuda or dlogits.stride(1) != 1 or dlogits.shape[0] != M:
            raise _lib.DcapError("vocab_ce: dlogits must be a float32 or bf16 device matrix with M rows")
        d.dlogits, d.lddl, d.dl_bf16 = dlogits.data_ptr(), dlogits.stride(0), int(dlogits.dtype == BF16)
    d.dbias = None if dbias is None else _chk(dbias, name="dbias").data_ptr()
    if materialize_bf16 is None:
        materialize_bf16 = os.environ.get("DCAP_VOCAB_MATERIALIZE", "1") != "0"
    d.materialize_bf16 = int(bool(materialize_bf16) and bf and dlogits is not None and dlogits.dtype == BF16)
    ws, wsb = WORKSPACE.get(lib.dc_vocab_ce_workspace_bytes(C.byref(d)), X.device)
    check(lib.dc_vocab_ce(C.byref(d), _ptr(ws), wsb, _stream()), "dc_vocab_ce")


def argmax_rows(x, out=None):
    lib = _lib.load()
    _chk(x, name="x")
    if out is None:
        out = torch.empty((x.shape[0],), dtype=torch.int32, device=x.device)
    check(lib.dc_argmax_rows_f32(_ptr(x), x.shape[0], x.shape[1], x.stride(0), _ptr(out), _stream()), "dc_argmax_rows_f32")
    return out


def gather_rows(src, idx, out, width=None):
    """out[n, :width] = src[idx[n], :width] (zeros where idx[n] < 0); out may be a column slice."""
    lib = _lib.load()
    _chk(src, name="src"), _chk(out, name="out"), _chk(idx, torch.int32, "idx")
    width = out.shape[1] if width is None else width
    check(lib.dc_gather_rows_f32(_ptr(src), src.stride(0), _ptr(idx), _ptr(out), out.stride(0), idx.numel(), width, _stream()),
          "dc_gather_rows_f32")
    return out


def _bn_desc(acc, bias, gamma, beta, mean, var, eps):
    d = BnReluDesc()
    d.M, d.N, d.ld = acc.shape[0], acc.shape[1], acc.stride(0)
    d.acc = _chk(acc, name="acc").data_ptr()
    d.bias, d.gamma, d.beta, d.mean, d.var = (_chk(t, name="bn param").data_ptr() for t in (bias, gamma, beta, mean, var))
    d.eps = float(eps)
    return d


def bn_relu_fwd(acc, bias, gamma, beta, mean, var, out, eps=1e-3):
    """out = relu(gamma*(acc + bias - mean)/sqrt(var+eps) + beta); acc/out [M,N] sharing a row stride."""
    lib = _lib.load()
    d = _bn_desc(acc, bias, gamma, beta, mean, var, eps)
    if _chk(out, name="out").stride(0) != acc.stride(0):
        raise _lib.DcapError("bn_relu_fwd: out must share acc's row stride")
    d.y = out.data_ptr()
    check(lib.dc_bn_relu_fwd_f32(C.byref(d), _stream()), "dc_bn_relu_fwd_f32")
    return out


def bn_relu_bwd(acc, bias, gamma, beta, mean, var, dy, dacc, dgamma, dbeta, dbias, eps=1e-3):
    lib = _lib.load()
    d = _bn_desc(acc, bias, gamma, beta, mean, var, eps)
    for t in (dy, dacc):
        if _chk(t, name="dy/dacc").stride(0) != acc.stride(0):
            raise _lib.DcapError("bn_relu_bwd: dy and dacc must share acc's row stride")
    d.dy, d.dacc = dy.data_ptr(), dacc.data_ptr()
    d.dgamma, d.dbeta, d.dbias = (_chk(t, name="bn grad").data_ptr() for t in (dgamma, dbeta, dbias))
    check(lib.dc_bn_relu_bwd_f32(C.byref(d), _stream()), "dc_bn_relu_bwd_f32")
    return dacc


def conv_weight_dgrad_pack(w_packed, kh, kw, cin, out=None):
    """[Cout][kh*kw*Cin] -> [Cin][kh*kw*Cout] with the taps rotated: the weights that make conv2d() the data gradient."""
    lib = _lib.load()
    cout = w_packed.shape[0]
    if out is None:
        out = torch.empty((cin, kh * kw * cout), dtype=torch.float32, device=w_packed.device)
    check(lib.dc_conv_weight_dgrad_pack_f32(_ptr(_chk(w_packed, name="w")), _ptr(out), cout, kh, kw, cin, _stream()),
          "dc_conv_weight_dgrad_pack_f32")
    return out


def rpn_loss_grad(heads, dheads, sel_level, sel_index, sel_match, target_deltas, n_pos, losses, image=0, anchors_per_loc=3, counts_dev=None,
                  batched=False):
    """RPN class + bbox losses of image `image` and their gradients scattered into the (pre-zeroed) dheads.  counts_dev: int32 device
    tensor {n_sel, n_pos} overriding the host counts (sel_* / target_deltas then have fixed capacities: graph-capturable).
    batched=True: the selection indexes the WHOLE [B, h, w, C] head tensors (image b's anchor i of a level = b * h * w * A + i, images
    in order, target rows packed in the same order): both means then run over the batch's union of selected anchors, as the
    reference's batched loss graphs do (dense_img_cap/dense_model.py:877-933)."""
    lib = _lib.load()
    d = RpnLossDesc()
    d.levels, d.anchors_per_loc, d.head_stride = len(heads), anchors_per_loc, heads[0].shape[-1]
    for i, (h, g) in enumerate(zip(heads, dheads)):
        per = h.shape[1] * h.shape[2] * h.shape[3]
        d.heads[i] = h.data_ptr() + 4 * per * image
        d.dheads[i] = g.data_ptr() + 4 * per * image
        d.Hs[i], d.Ws[i] = (h.shape[0] * h.shape[1] if batched else h.shape[1]), h.shape[2]      # (a batch is a taller map to the kernel's bound check)
    d.n_sel, d.n_pos = sel_level.numel(), int(n_pos)
    if d.n_sel:                                         # no selected anchor (all neutral): the kernel only zeroes the losses
        d.sel_level, d.sel_index, d.sel_match = (_chk(t, torch.int32, "sel").data_ptr() for t in (sel_level, sel_index, sel_match))
    d.target_deltas = _chk(target_deltas, name="target_deltas").data_ptr()
    d.losses = _chk(losses, name="losses").data_ptr()
    if counts_dev is not None:
        d.counts_dev = _chk(counts_dev, torch.int32, "counts_dev").data_ptr()
        d.n_pos = min(int(n_pos), target_deltas.shape[0])
    check(lib.dc_rpn_loss_grad_f32(C.byref(d), _stream()), "dc_rpn_loss_grad_f32")


def detection_targets(proposals, gt_boxes, gt_captions, n_rois, positive_ratio, seed=None, offset=0, offset_dev=None, out=None):
    """DetectionTargetLayer for one image on the device (dc_detection_targets_f32).  proposals [N,4] / gt_boxes [G,4] float32
    normalised, gt_captions [G,T] int32.  seed None: proposal order (no shuffle).  Returns (rois [n_rois,4], captions [n_rois,T] int32,
    counts int32 [2] = {n_pos, n_neg}) -- device tensors, nothing is read back."""
    lib = _lib.load()
    _chk(proposals, name="proposals"), _chk(gt_boxes, name="gt_boxes"), _chk(gt_captions, torch.int32, "gt_captions")
    N, G, T = proposals.shape[0], gt_boxes.shape[0], gt_captions.shape[1]
    if gt_captions.shape[0] != G or not (proposals.is_contiguous() and gt_boxes.is_contiguous() and gt_captions.is_contiguous()):
        raise _lib.DcapError("detection_targets: gt_captions must have one row per GT box; operands contiguous")
    if out is None:
        out = (torch.empty((n_rois, 4), dtype=torch.float32, device=proposals.device),
               torch.empty((n_rois, T), dtype=torch.int32, device=proposals.device),
               torch.empty((2,), dtype=torch.int32, device=proposals.device))
    rois, caps, counts = out
    d = DetectionTargetsDesc()
    d.n_proposals, d.n_gt, d.n_rois, d.T = N, G, int(n_rois), T
    d.proposals, d.gt_boxes, d.gt_captions = proposals.data_ptr(), gt_boxes.data_ptr(), gt_captions.data_ptr()
    d.max_positive = int(n_rois * positive_ratio)
    import numpy as _np
    d.inv_ratio = float(_np.float32(1.0 / positive_ratio))
    d.shuffle, d.seed, d.offset = int(seed is not None), (int(seed or 0) & 0xFFFFFFFF), int(offset) & 0xFFFFFFFF
    d.offset_dev = None if offset_dev is None else _chk(offset_dev, torch.int32, "offset_dev").data_ptr()
    d.rois, d.captions, d.counts = _chk(rois, name="rois").data_ptr(), _chk(caps, torch.int32, "captions").data_ptr(), _chk(counts, torch.int32, "counts").data_ptr()
    check(lib.dc_detection_targets_f32(C.byref(d), _stream()), "dc_detection_targets_f32")
    return rois, caps, counts


def caption_tables(captions, out=None, live_count=None):
    """captions int32 [B,T] (device) -> (ids_tm int32 [T*B], mask uint8 [T*B], targets_tm int32 [T*B], row_weights f32 [T*B])."""
    lib = _lib.load()
    _chk(captions, torch.int32, "captions")
    B, T = captions.shape
    dev = captions.device
    if out is None:
        out = (torch.empty(T * B, dtype=torch.int32, device=dev), torch.empty(T * B, dtype=torch.uint8, device=dev),
               torch.empty(T * B, dtype=torch.int32, device=dev), torch.empty(T * B, dtype=torch.float32, device=dev))
    ids, mask, tg, rw = out
    check(lib.dc_caption_tables_i32(_ptr(captions), B, T, _ptr(ids), _ptr(mask), _ptr(tg), _ptr(rw),
                                    None if live_count is None else _ptr(live_count), _stream()), "dc_caption_tables_i32")
    return ids, mask, tg, rw


def scatter2_add(coarse, fine):
    lib = _lib.load()
    N, Hc, Wc, Cc = coarse.shape
    check(lib.dc_scatter2_add_f32(_ptr(_chk(coarse, name="coarse")), _ptr(_chk(fine, name="fine")), N, Hc, Wc, Cc, _stream()), "dc_scatter2_add_f32")
    return fine


def l2_reg(w, coef, grad=None, loss=None, mask=None):
    """grad = grad * mask + 2 coef w (mask None: all ones); loss[0] = sum coef w^2 in a fixed order."""
    lib = _lib.load()
    ws, wsb = WORKSPACE.get(lib.dc_l2_reg_workspace_bytes(w.numel()), w.device) if loss is not None else (None, 0)
    check(lib.dc_l2_reg_f32(_ptr(_chk(w, name="w")), _ptr(_chk(coef, name="coef")), None if mask is None else _ptr(_chk(mask, name="mask")),
                            None if grad is None else _ptr(grad), w.numel(), None if loss is None else _ptr(loss),
                            None if ws is None else _ptr(ws), wsb, _stream()), "dc_l2_reg_f32")
    return loss


def set_persistent_cus(n):
    """CUs the persistent Winograd grids may occupy (0 = default); see dc_set_persistent_cus in include/dcap.h."""
    lib = _lib.load()
    check(lib.dc_set_persistent_cus(int(n)), "dc_set_persistent_cus")
    return int(lib.dc_get_persistent_cus())


def zero_fill(t):
    """t[...] = 0 by the library's zero-fill kernel (contiguous tensor of 4-byte-multiple size): no torch fill launch on the product path."""
    lib = _lib.load()
    if not t.is_contiguous() or not t.is_cuda:
        raise _lib.DcapError("zero_fill: a contiguous GPU tensor")
    check(lib.dc_zero_fill(_ptr(t), t.numel() * t.element_size(), _stream()), "dc_zero_fill")
    return t


def axpy(a, x, y):
    lib = _lib.load()
    check(lib.dc_axpy_f32(float(a), _ptr(_chk(x, name="x")), _ptr(_chk(y, name="y")), x.numel(), _stream()), "dc_axpy_f32")
    return y


def relu_bwd(dy, y, out, out_bf16=None):
    """out = dy where y > 0 else 0 (all [M,N] with one row stride).  out_bf16 (contiguous operands only): the bf16 copy of the result,
    written in the same pass (dc_relu_bwd_dual_f32)."""
    lib = _lib.load()
    _chk(dy, name="dy"), _chk(y, name="y"), _chk(out, name="out")
    if out_bf16 is not None:
        if not (dy.is_contiguous() and y.is_contiguous() and out.is_contiguous() and out_bf16.is_contiguous()) or out_bf16.numel() != out.numel():
            raise _lib.DcapError("relu_bwd: the bf16 copy needs contiguous operands of one size")
        check(lib.dc_relu_bwd_dual_f32(_ptr(dy), _ptr(y), _ptr(out), _ptr(_chk(out_bf16, BF16, "out_bf16")), out.numel(), _stream()), "dc_relu_bwd_dual_f32")
        return out
    if not (dy.stride(0) == y.stride(0) == out.stride(0)):
        raise _lib.DcapError("relu_bwd: tensors must share a row stride")
    check(lib.dc_relu_bwd_f32(_ptr(dy), _ptr(y), _ptr(out), y.shape[0], y.shape[1], y.stride(0), _stream()), "dc_relu_bwd_f32")
    return out


def fold_time(x, T, B, out):
    """out[b] = sum_t x[t*B + b] (time-major rows)."""
    lib = _lib.load()
    _chk(x, name="x"), _chk(out, name="out")
    check(lib.dc_fold_time_f32(_ptr(x), T, B, x.shape[1], x.stride(0), _ptr(out), out.stride(0), _stream()), "dc_fold_time_f32")
    return out


def colsum(x, out=None, accumulate=False):
    lib = _lib.load()
    _chk(x, name="x")
    if out is None:
        out = torch.empty((x.shape[1],), dtype=torch.float32, device=x.device)
    ws, wsb = WORKSPACE.get(lib.dc_colsum_workspace_bytes(x.shape[0], x.shape[1], x.stride(0)), x.device)
    check(lib.dc_colsum_f32(_ptr(x), x.shape[0], x.shape[1], x.stride(0), _ptr(out), int(accumulate), _ptr(ws), wsb, _stream()), "dc_colsum_f32")
    return out


def sumsq(x, out=None, accumulate=False):
    lib = _lib.load()
    if out is None:
        out = torch.empty((1,), dtype=torch.float32, device=x.device)
    ws, wsb = WORKSPACE.get(lib.dc_sumsq_workspace_bytes(x.numel()), x.device)
    check(lib.dc_sumsq_f32(_ptr(_chk(x, name="x")), x.numel(), _ptr(out), int(accumulate), _ptr(ws), wsb, _stream()), "dc_sumsq_f32")
    return out


def mean(x, out=None):
    lib = _lib.load()
    if out is None:
        out = torch.empty((1,), dtype=torch.float32, device=x.device)
    check(lib.dc_mean_f32(_ptr(_chk(x, name="x")), x.numel(), _ptr(out), _stream()), "dc_mean_f32")
    return out


def dropout_mask(out, rate, seed, offset, offset_dev=None):
    """K.dropout(ones, rate): out filled with 1/(1-rate) (kept) or 0, element i a pure function of (i, seed, offset [+ *offset_dev])."""
    lib = _lib.load()
    check(lib.dc_dropout_mask_f32(_ptr(_chk(out, name="out")), out.numel(), float(rate), int(seed) & 0xFFFFFFFF, int(offset) & 0xFFFFFFFF,
                                  None if offset_dev is None else _ptr(_chk(offset_dev, torch.int32, "offset_dev")), _stream()),
          "dc_dropout_mask_f32")
    return out


def bn_fold(gamma, beta, bias, mean, var, scale, shift, eps=1e-3):
    """Fused-epilogue operands of a convolution followed by a frozen-statistics BatchNorm (dc_bn_fold_f32)."""
    lib = _lib.load()
    for t in (gamma, beta, bias, mean, var, scale, shift):
        _chk(t, name="bn_fold operand")
    check(lib.dc_bn_fold_f32(_ptr(gamma), _ptr(beta), _ptr(bias), _ptr(mean), _ptr(var), float(eps), _ptr(scale), _ptr(shift), gamma.numel(), _stream()),
          "dc_bn_fold_f32")


def bn_bwd(dz, a, b, gamma, beta, scale, dacc, dzn):
    """dacc = dz * scale, dzn = dz * (a - b - beta) / gamma over [..., C] tensors (dc_bn_bwd_f32); b may be None."""
    lib = _lib.load()
    for t in (dz, a, gamma, beta, scale, dacc, dzn):
        if not _chk(t, name="bn_bwd operand").is_contiguous():
            raise _lib.DcapError("bn_bwd: operands must be contiguous")
    Cc = dz.shape[-1]
    check(lib.dc_bn_bwd_f32(_ptr(dz), _ptr(a), _ptr(b), _ptr(gamma), _ptr(beta), _ptr(scale), _ptr(dacc), _ptr(dzn), dz.numel() // Cc, Cc, _stream()),
          "dc_bn_bwd_f32")
    return dacc, dzn


def mul(a, b, out):
    lib = _lib.load()
    check(lib.dc_mul_f32(_ptr(_chk(a, name="a")), _ptr(_chk(b, name="b")), _ptr(_chk(out, name="out")), out.numel(), _stream()), "dc_mul_f32")
    return out


def maxpool3x3s2_same_bwd(x, y, dy, out=None):
    lib = _lib.load()
    _chk(x, name="x"), _chk(y, name="y"), _chk(dy, name="dy")
    N, H, W, Cc = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib.dc_maxpool3x3s2_same_bwd_f32(_ptr(x), _ptr(y), _ptr(dy), _ptr(out), N, H, W, Cc, _stream()), "dc_maxpool3x3s2_same_bwd_f32")
    return out


class RegSegmentTable(object):
    """dc_reg_segments on the device: the run-length form of the per-element regulariser coefficient / trainable-mask vectors of a flat
    parameter bucket (coef, mask: host float32 arrays of the bucket's length; mask None = everything trains).  The kernels keep the
    table in LDS: at most MAX_SEGMENTS runs (csrc/loss.hip: kMaxRegSegs); callers with more take the unfused passes."""
    MAX_SEGMENTS = 1024

    def __init__(self, coef, mask, device):
        coef = np.ascontiguousarray(coef, np.float32)
        mask = np.ones_like(coef) if mask is None else np.ascontiguousarray(mask, np.float32)
        n = coef.shape[0]
        cut = np.flatnonzero((coef[1:] != coef[:-1]) | (mask[1:] != mask[:-1])) + 1
        start = np.concatenate([[0], cut, [n]]).astype(np.int32)
        self.n, self.nseg = n, len(start) - 1
        self.start = torch.tensor(start, device=device)
        self.coef = torch.tensor(coef[start[:-1]], device=device)
        self.mask = torch.tensor(mask[start[:-1]], device=device)
        self.c = _lib.RegSegments(self.start.data_ptr(), self.coef.data_ptr(), self.mask.data_ptr(), self.nseg)


def reg_sumsq(w, g, segs, loss=None, gnorm_sq=None):
    """One read-only pass: loss[0] = sum coef w^2, gnorm_sq[0] = sum (g mask + 2 coef w)^2 (dc_reg_sumsq_f32), fixed summation order."""
    lib = _lib.load()
    if segs.n != w.numel() or g.numel() != w.numel():
        raise _lib.DcapError("reg_sumsq: the segment table covers %d elements, the bucket has %d" % (segs.n, w.numel()))
    ws, wsb = WORKSPACE.get(lib.dc_reg_sumsq_workspace_bytes(w.numel()), w.device)
    check(lib.dc_reg_sumsq_f32(_ptr(_chk(w, name="w")), _ptr(_chk(g, name="g")), C.byref(segs.c), w.numel(), None if loss is None else _ptr(loss),
                               None if gnorm_sq is None else _ptr(gnorm_sq), _ptr(ws), wsb, _stream()), "dc_reg_sumsq_f32")
    return loss, gnorm_sq


def amsgrad_step(p, g, m, v, vhat, lr_t, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0, gnorm_sq=None, clipnorm=0.0, p_bf16=None,
                 lr_t_dev=None, reg=None):
    lib = _lib.load()
    d = AmsgradDesc()
    d.n = p.numel()
    d.p, d.g, d.m, d.v, d.vhat = (_chk(t, name="amsgrad buffer").data_ptr() for t in (p, g, m, v, vhat))
    d.lr_t, d.beta1, d.beta2, d.eps = float(lr_t), float(beta1), float(beta2), float(eps)
    d.grad_scale = float(grad_scale)
    d.gnorm_sq = None if gnorm_sq is None else gnorm_sq.data_ptr()
    d.clipnorm = float(clipnorm or 0.0)
    if p_bf16 is not None:
        d.p_bf16, d.n_bf16 = _chk(p_bf16, BF16, "p_bf16").data_ptr(), p_bf16.numel()
    if lr_t_dev is not None:
        d.lr_t_dev = _chk(lr_t_dev, name="lr_t_dev").data_ptr()
    if reg is not None:                                    # regulariser + trainable mask applied inside the update (RegSegmentTable)
        if reg.n != p.numel():
            raise _lib.DcapError("amsgrad_step: the segment table covers %d elements, the bucket has %d" % (reg.n, p.numel()))
        d.reg = C.pointer(reg.c)
    check(lib.dc_amsgrad_step_f32(C.byref(d), _stream()), "dc_amsgrad_step_f32")
