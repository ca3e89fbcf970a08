"""Encoder plan: ResNet-101 + FPN + PyramidROIAlign on the GPU, built once per (batch, H, W).

What the reference does with a Keras graph + predict() (feature_generation/dense_model.py:143-173
resnet_graph, :1404-1427 FPN, :317-418 PyramidROIAlign; GT-RoI variant
dense_img_cap_separate_models/modified_dense_model.py:1410-1433, :1522-1527) is here a flat list of
pre-built kernel descriptors over pre-allocated, stage-wise recycled activation buffers:
one dc_conv2d_nhwc_f32 launch per convolution with BN / bias / residual / ReLU / upsample-add fused
into its epilogue, replayed from a hipGraph after the first call.  The RPN branch, whose output the
GT-RoI path never reads (207.6 GF of dead work per image in the reference), is not evaluated.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, ops
from ._lib import ConvBf16Desc, ConvDesc, PwChainDesc, check
from .layers import ConvSpec, resnet_fpn_convs
from .packing import fold_bn, pack_conv_kernel, pack_stem_kernel


DEFAULT_CONV_MATH = "f32"


def conv_math_mode(math=None):
    import os
    name = math if math is not None else os.environ.get("DCAP_CONV_MATH", DEFAULT_CONV_MATH)
    try:
        return {"f32": _lib.MATH_F32, "bf16x3": _lib.MATH_BF16X3, "bf16x2": _lib.MATH_BF16X2, "bf16": _lib.MATH_BF16}[name]
    except KeyError:
        raise ValueError("conv math must be 'f32', 'bf16x3', 'bf16x2' or 'bf16', got %r" % (name,))


def conv_math_name(mode):
    return {_lib.MATH_F32: "f32", _lib.MATH_BF16X3: "bf16x3", _lib.MATH_BF16X2: "bf16x2", _lib.MATH_BF16: "bf16"}.get(mode, str(mode))


def fuse_rpn_head(weights, channels=None):
    """rpn_class_raw (2A channels) ++ rpn_bbox_pred (4A) as ONE 1x1 convolution, zero-padded to `channels` outputs."""
    k = np.concatenate([weights["rpn_class_raw/kernel"], weights["rpn_bbox_pred/kernel"]], axis=3).astype(np.float32)
    b = np.concatenate([weights["rpn_class_raw/bias"], weights["rpn_bbox_pred/bias"]]).astype(np.float32)
    pad = (channels or k.shape[3]) - k.shape[3]
    if pad < 0:
        raise ValueError("rpn head needs %d channels, got %d" % (k.shape[3], channels))
    if pad:
        k = np.concatenate([k, np.zeros(k.shape[:3] + (pad,), np.float32)], axis=3)
        b = np.concatenate([b, np.zeros(pad, np.float32)])
    return k, b


class EncoderPlan:
    feat_channels = 256              # channels of a RoI feature (the FPN depth)
    fast_bf16 = False                # bf16 STORAGE between the convolutions (math = 'bf16' only; see __init__)

    def __init__(self, weights, batch, height, width, device, stage4_blocks=22,
                 mean_pixel=(123.7, 116.8, 103.9), use_graph=True, rpn=None, external=None, math=None, external_bn=None, train_stages=(),
                 winograd=None, wino_products=None, pw_chain=None, layer_math=None):
        """rpn: None (GT-RoI variant: the RPN is never evaluated) or a dict with the config values the
        proposal path needs: scales, ratios, strides, anchor_stride, bbox_std, nms_threshold, proposal_count and
        optionally head_channels (the fused class+bbox head padded to a multiple of 4 channels for the wgrad kernel).
        external: {conv name: (packed kernel, scale or None, shift)} device tensors owned by the caller (the joint
        model's trainable FPN/RPN weights live in its flat parameter bucket; the plan reads them in place).
        external_bn: {conv name: dict(gamma, beta, bias, mean, var)} device tensors for TRAINABLE ResNet layers (train(layers=
        "3+" ...), dense_img_cap/dense_model.py:1829-1845): the plan owns scale / shift vectors for them and refreshes both
        from the trained parameters at the start of every forward (dc_bn_fold_f32); their packed kernels come through `external`
        with scale = shift = None.  train_stages: the ResNet stages (2..5; 1 = the stem) whose activations must survive the
        forward for the backward pass: they get buffers of their own instead of the recycled ones (self.saved)."""
        if height % 64 or width % 64:
            raise ValueError("Image size must be dividable by 2 at least 6 times (got %dx%d)" % (height, width))
        self.lib = _lib.load()
        # conv arithmetic: 'f32' (fp32 MFMA products) or 'bf16x3' (three-piece bf16 split, six matrix-pipe products, fp32
        # accumulate); DCAP_CONV_MATH overrides the default for experiments
        self.math = conv_math_mode(math)
        # math = 'bf16' (BASELINE configs[4]): activations travel between the convolutions as bf16 tensors and every convolution with
        # Cin % 64 == 0 runs on dc_conv2d_bf16 (LDS-DMA im2col on the bf16 GEMM core); fp32 copies are written only where something
        # reads fp32 (residual adds, RoIAlign, the joint model's backward).
        import os
        self.fast_bf16 = self.math == _lib.MATH_BF16
        # math = 'f32': the 3x3 / stride 1 layers with FROZEN weights (ResNet 2b branches, FPN output convolutions) run in the Winograd
        # F(2x2, 3x3) form -- fp32 transforms, fp32 MFMA, 16 products per 2x2 output tile instead of 36 (csrc/conv_wino.hip); their
        # kernels are transformed once here.  winograd=False / DCAP_WINOGRAD=0: the direct implicit GEMM everywhere.
        self.winograd = (os.environ.get("DCAP_WINOGRAD", "1") != "0") if winograd is None else bool(winograd)
        # ... and their 16 products per tile run on the BF16 matrix pipe in split arithmetic (round 5: U pre-split into three bf16 pieces
        # once, V split in registers, six bf16 MFMA products per fp32 product, fp32 accumulation -- DC_MATH_BF16X3's fp32-grade arithmetic,
        # held to the fp32 kernel's tolerances at kernel level and at full depth): wino_products = 'b3' (default) | 'f32' (fp32 MFMA
        # products; DCAP_WINO_PRODUCTS overrides the default)
        self.wino_products = wino_products or os.environ.get("DCAP_WINO_PRODUCTS", "b3")
        if self.wino_products not in ("b3", "f32"):
            raise ValueError("wino_products must be 'b3' or 'f32'")
        # math = 'f32', frozen stages 3 and 4: a bottleneck's last 1x1 convolution (+ shortcut + ReLU) and the next block's first run as ONE
        # launch that keeps the 4x-wide intermediate rows in LDS (csrc/conv_chain.hip, round 5).  pw_chain=False / DCAP_PW_CHAIN=0: two launches.
        self.pw_chain = (os.environ.get("DCAP_PW_CHAIN", "1") != "0") if pw_chain is None else bool(pw_chain)
        # per-layer choice between the fp32 pipe and split-bf16 arithmetic for the direct pointwise layers (_layer_math)
        self.layer_math = (os.environ.get("DCAP_LAYER_MATH", "1") != "0") if layer_math is None else bool(layer_math)
        self._wchain = {}
        self._wwino = {}
        self._twin = {}
        self._wb = {}
        self.B, self.H, self.W = batch, height, width
        self.device = torch.device(device)
        self.mean_pixel = [float(v) for v in mean_pixel]
        self.stage4_blocks = stage4_blocks
        self.use_graph = bool(use_graph)
        self._graph = None
        self._part_graphs, self._part_warm = {}, set()
        self._warm = False
        self._specs = {s.name: s for s in resnet_fpn_convs(stage4_blocks)}
        self.rpn = rpn
        self._external = dict(external or {})
        self._external_bn = dict(external_bn or {})
        self.train_stages = tuple(sorted(train_stages))
        self.saved = {}
        for name, bnp in self._external_bn.items():         # plan-owned epilogue vectors of the trainable BatchNorm layers
            n = bnp["gamma"].numel()
            sc, sh = torch.empty(n, dtype=torch.float32, device=self.device), torch.empty(n, dtype=torch.float32, device=self.device)
            kern = self._external[name][0]
            self._external[name] = (kern, sc, sh)
        if rpn is not None:
            a = len(rpn["ratios"])
            self.head_channels = hc = int(rpn.get("head_channels", 6 * a))
            self._specs["rpn_conv_shared"] = ConvSpec("rpn_conv_shared", None, 3, 256, 512, 1, "same")
            self._specs["rpn_head"] = ConvSpec("rpn_head", None, 1, 512, hc, 1, "valid")         # class_raw ++ bbox_pred (++ pad)
            if "rpn_head" not in self._external:
                weights = dict(weights)
                weights["rpn_head/kernel"], weights["rpn_head/bias"] = fuse_rpn_head(weights, hc)
        self._w = {}
        self._upload(weights)
        self._build()

    # ------------------------------------------------------------------ weights
    def _upload(self, W):
        dev = self.device
        for s in self._specs.values():
            if s.name in self._external:
                self._w[s.name] = self._external[s.name]
                continue
            k = np.asarray(W[s.name + "/kernel"], np.float32)
            if tuple(k.shape) != (s.k, s.k, s.cin, s.cout):
                raise ValueError("%s/kernel has shape %s, expected %s" % (s.name, k.shape, (s.k, s.k, s.cin, s.cout)))
            packed = pack_stem_kernel(k) if s.name == "conv1" else pack_conv_kernel(k)
            bias = np.asarray(W[s.name + "/bias"], np.float32)
            if s.bn:
                scale, shift = fold_bn(W[s.bn + "/gamma"], W[s.bn + "/beta"], W[s.bn + "/moving_mean"],
                                       W[s.bn + "/moving_variance"], bias)
                sc = torch.tensor(scale, device=dev)
            else:
                sc, shift = None, bias
            self._w[s.name] = (torch.tensor(packed, device=dev), sc, torch.tensor(shift, device=dev))

    # ------------------------------------------------------------------ plan
    def _buf(self, h, w, c):
        t = torch.empty((self.B, h, w, c), dtype=torch.float32, device=self.device)
        self._bufs.append(t)          # descriptors hold raw pointers: the plan must own every buffer
        return t

    def _bf(self, t):
        """The bf16 twin of a plan buffer (allocated on first use; the plan owns it)."""
        k = t.data_ptr()
        if k not in self._twin:
            self._twin[k] = torch.empty(tuple(t.shape), dtype=torch.bfloat16, device=self.device)
        return self._twin[k]

    def bf16_of(self, t):
        """bf16 twin of plan buffer `t` if the last forward() wrote one, else None (the joint model's backward reuses them)."""
        return self._twin.get(t.data_ptr()) if self.fast_bf16 else None

    def _conv_bf16(self, name, x, y, residual, res_mode, relu, f32, bf16):
        s = self._specs[name]
        wp, sc, sh = self._w[name]
        N, H, W, Cin = x.shape
        _, Ho, Wo, Cout = y.shape
        if x.data_ptr() not in self._twin:
            raise RuntimeError("%s: its input has no bf16 twin (the producing op must be built with bf16=True)" % name)
        if name not in self._wb:
            if name in self._external:                 # trainable weights change every step: re-cast inside the forward, once
                self._wb[name] = torch.empty(tuple(wp.shape), dtype=torch.bfloat16, device=self.device)
                self._ops.append(("cast", wp, self._wb[name]))
            else:                                      # frozen weights: rounded once
                self._wb[name] = ops.to_bf16(wp)
        pad = (s.k - 1) // 2 if s.padding == "same" else (3 if s.padding == "pad3" else 0)
        d = ConvBf16Desc()
        d.N, d.H, d.W, d.Cin = N, H, W, Cin
        d.Cout, d.kh, d.kw, d.stride, d.pad_t, d.pad_l, d.Ho, d.Wo = Cout, s.k, s.k, s.stride, pad, pad, Ho, Wo
        d.x, d.w = self._twin[x.data_ptr()].data_ptr(), self._wb[name].data_ptr()
        d.y = y.data_ptr() if f32 else None
        d.y_bf16 = self._bf(y).data_ptr() if bf16 else None
        d.scale = None if sc is None else sc.data_ptr()
        d.shift = sh.data_ptr()
        d.residual = None if residual is None else residual.data_ptr()
        d.res_mode, d.relu, d.split_k = res_mode, int(relu), 0
        self._ws_bytes = max(self._ws_bytes, self.lib.dc_conv2d_bf16_workspace_bytes(C.byref(d)))
        self._ops.append(("bconv", d, name))
        self.flops += 2.0 * N * Ho * Wo * Cout * s.k * s.k * s.cin

    def _conv(self, name, x, y, residual=None, res_mode=0, relu=True, f32=True, bf16=False):
        """f32 / bf16: which copies of the output its consumers read (only the bf16-storage mode acts on them: there a layer
        whose consumers are all convolutions writes no fp32 copy at all)."""
        s = self._specs[name]
        if self.fast_bf16 and s.cin % 64 == 0 and s.stride in (1, 2) and (s.padding != "same" or s.stride == 1):
            return self._conv_bf16(name, x, y, residual, res_mode, relu, f32, bf16)
        wp, sc, sh = self._w[name]
        N, H, W, Cin = x.shape
        _, Ho, Wo, Cout = y.shape
        d = ConvDesc()
        d.N, d.H, d.W, d.Cin = N, H, W, Cin
        if s.padding == "same":
            pad = (s.k - 1) // 2 if s.stride == 1 else None
            if pad is None:
                raise ValueError("strided SAME conv is not on this path")
        elif s.padding == "pad3":
            pad = 3
        else:
            pad = 0
        d.Cout, d.kh, d.kw, d.stride, d.pad_t, d.pad_l, d.Ho, d.Wo = Cout, s.k, s.k, s.stride, pad, pad, Ho, Wo
        d.x, d.w, d.y = x.data_ptr(), wp.data_ptr(), y.data_ptr()
        d.scale = None if sc is None else sc.data_ptr()
        d.shift = sh.data_ptr()
        d.residual = None if residual is None else residual.data_ptr()
        d.res_mode, d.relu, d.split_k, d.math = res_mode, int(relu), 0, self._layer_math(s, name)
        use_wino = (self.winograd and self.math in (_lib.MATH_F32, _lib.MATH_BF16X3, _lib.MATH_BF16X2) and s.k == 3 and s.stride == 1 and
                    s.padding == "same" and residual is None and name not in self._external and Cin % 32 == 0 and Cout % 32 == 0 and
                    wp.shape[1] == 9 * Cin)
        if use_wino:
            # (Cin = the channels of the tensor the layer READS: VGG16's block1_conv1 reads RGB zero-padded to 32 channels)
            # The split-bf16 modes (round 4): their 3x3 layers with frozen weights ALSO run the fp32 Winograd kernel -- exact fp32 products,
            # 2.25x fewer of them, faster than six (three) bf16 products per direct-form product --, the 1x1 / strided / residual layers
            # keep the split arithmetic on the bf16 matrix pipe.
            if name not in self._wwino:
                self._wwino[name] = (ops.winograd_pack_b3 if self.wino_products == "b3" else ops.winograd_pack)(wp, Cin, Cout)
            if self.wino_products == "b3":
                d.w_wino_b3 = self._wwino[name].data_ptr()
            else:
                d.w_wino = self._wwino[name].data_ptr()
            d.math = _lib.MATH_F32
        self._ws_bytes = max(self._ws_bytes, self.lib.dc_conv2d_workspace_bytes(C.byref(d)))
        self._ops.append(("conv", d, name))
        if self.fast_bf16 and bf16:                    # a layer the bf16 kernel does not take (the stem): cast its output
            self._ops.append(("cast", y, self._bf(y)))
        self.flops += 2.0 * N * Ho * Wo * Cout * s.k * s.k * s.cin

    def _layer_math(self, s, name):
        """Arithmetic of one direct-kernel layer.  In the fp32-grade default plan (conv_math='f32') the pointwise layers that measured
        faster in split-bf16 arithmetic (DC_MATH_BF16X3: three bf16 pieces per operand, six matrix-pipe products, fp32 accumulation --
        the same fp32-grade result, the same test tolerances) run there: the stem, the strided projection shortcuts (res3a/4a/5a_branch1) and the
        stride-1 layers with 128 <= Cin <= 512 and Cout >= 256 (FPN laterals C2 / C3, the un-chained 2c layers).  tools/conv_bench.py
        --filter 1x1 [--math 1], two images: res3a_1 85 -> 61 us, res4a_1 77 -> 55, fpn_c2p2 163 -> 142, fpn_c3p3 74 -> 58, res5_2c 41 -> 38;
        Cin >= 1024 and the strided 2a layers are faster on the fp32 pipe and stay there.  layer_math=False / DCAP_LAYER_MATH=0: one
        arithmetic for all direct layers."""
        if self.math != _lib.MATH_F32 or not self.layer_math or name in self._external:
            return self.math
        if s.k == 7 and s.cin == 3:                        # the stem: 185 -> 122 us at two images (output within 6e-7 of the fp32 pipe's)
            return _lib.MATH_BF16X3
        if s.k != 1:
            return self.math
        if s.stride == 2 and s.cout >= 512 and s.cout > s.cin:   # the projection shortcuts (res5a_branch2a, 1024 -> 512 strided, is faster on the fp32 pipe)
            return _lib.MATH_BF16X3
        if s.stride == 1 and 128 <= s.cin <= 512 and s.cout >= 256:
            return _lib.MATH_BF16X3
        return self.math

    def _chain_ok(self, name_c, name_a, mid, cout, pixels):
        sc, sa = self._specs[name_c], self._specs[name_a]
        # the tile form (cout >= 512) is one 32-pixel block per CU: below ~3/4 of the chip's CUs the two separate launches are faster
        # (one image per GPU, stage 4: 128 blocks -- 7 800 captions/s chained against 7 920 unchained); the streaming form has no such limit
        if cout >= 512 and pixels < 32 * 192:
            return False
        return (self.pw_chain and self.math == _lib.MATH_F32 and not self.fast_bf16 and sc.k == 1 and sa.k == 1 and sc.stride == 1 and sa.stride == 1
                and name_c not in self._external and name_a not in self._external and ops.pw_chain_supported(mid, cout, mid))

    def _chain(self, name_c, x, y, residual, name_a, z):
        """One launch for conv `name_c` (1x1, x -> y, + residual, ReLU) followed by conv `name_a` (1x1, y -> z, ReLU): dc_pw_chain_f32."""
        (w1, sc1, sh1), (w2, sc2, sh2) = self._w[name_c], self._w[name_a]
        b3 = self.wino_products == "b3" and x.shape[3] >= 128  # the split-bf16 products, like the Winograd layers (the stage-2 seam is bandwidth-bound: fp32 products)
        for n, w in ((name_c, w1), (name_a, w2)):
            if n not in self._wchain:
                self._wchain[n] = (ops.pw_chain_pack_b3 if b3 else ops.pw_chain_pack)(w)
        N, H, W, K1 = x.shape
        d = PwChainDesc()
        d.M, d.K1, d.N1, d.N2 = N * H * W, K1, y.shape[3], z.shape[3]
        d.x, d.shift1, d.y = x.data_ptr(), sh1.data_ptr(), y.data_ptr()
        if b3:
            d.w1_b3, d.w2_b3 = self._wchain[name_c].data_ptr(), self._wchain[name_a].data_ptr()
        else:
            d.w1, d.w2 = self._wchain[name_c].data_ptr(), self._wchain[name_a].data_ptr()
        d.scale1 = None if sc1 is None else sc1.data_ptr()
        d.residual = None if residual is None else residual.data_ptr()
        d.shift2, d.z = sh2.data_ptr(), z.data_ptr()
        d.scale2 = None if sc2 is None else sc2.data_ptr()
        d.relu1 = d.relu2 = 1
        self._ops.append(("chain", d, name_c + "+" + name_a))
        self.flops += 2.0 * d.M * (d.K1 * d.N1 + d.N1 * d.N2)

    def _build(self):
        B, H, W = self.B, self.H, self.W
        self._ops, self._ws_bytes, self.flops, self._bufs = [], 0, 0.0, []
        self._twin, self._wb = {}, {}
        self.images = torch.empty((B, H, W, 3), dtype=torch.uint8, device=self.device)
        rgbx = torch.empty((B, H, W, 4), dtype=torch.float32, device=self.device)
        for name, bnp in self._external_bn.items():
            _, sc, sh = self._external[name]
            self._ops.append(("bnfold", bnp, sc, sh))
        self._ops.append(("mold", self.images, rgbx))
        c1 = self._buf(H // 2, W // 2, 64)
        self._conv("conv1", rgbx, c1)
        x = self._buf(H // 4, W // 4, 64)
        self._ops.append(("pool", c1, x))
        if self.fast_bf16:
            self._ops.append(("cast", x, self._bf(x)))
        self.saved[1] = dict(rgbx=rgbx, c1=c1, pooled=x)

        def stage(s, blocks, mid, cout, stride, x):
            h, w = x.shape[1] // stride, x.shape[2] // stride
            keep = s in self.train_stages              # a trainable stage: every activation survives for the backward (fp32 copies too)
            if not keep:
                m1, m2, sc = self._buf(h, w, mid), self._buf(h, w, mid), self._buf(h, w, cout)
                pp = [self._buf(h, w, cout), self._buf(h, w, cout)]
            final = self._buf(h, w, cout)
            self.saved[s] = []
            pending = None                                  # the previous block's deferred 2c: it runs chained with this block's 2a
            for i, blk in enumerate(blocks):
                cn = "res%d%s_branch" % (s, blk)
                if keep:
                    m1, m2 = self._buf(h, w, mid), self._buf(h, w, mid)
                    sc = self._buf(h, w, cout) if i == 0 else None
                    out = final if i == len(blocks) - 1 else self._buf(h, w, cout)
                    self.saved[s].append(dict(name=cn, x=x, m1=m1, m2=m2, sc=sc, out=out, stride=stride if i == 0 else 1))
                else:
                    out = final if i == len(blocks) - 1 else pp[i & 1]
                if pending is not None:                                            # previous 2c (-> x) and this 2a (x -> m1) in one launch
                    self._chain(pending[0], pending[1], x, pending[2], cn + "2a", m1)
                    pending = None
                else:
                    self._conv(cn + "2a", x, m1, f32=keep, bf16=True)            # read by the next convolution only (and by a backward pass)
                self._conv(cn + "2b", m1, m2, f32=keep, bf16=True)
                res = x
                if i == 0:
                    self._conv(cn + "1", x, sc, relu=False)                        # read as a residual only
                    res = sc
                nxt = "res%d%s_branch2a" % (s, blocks[i + 1]) if i + 1 < len(blocks) else None
                if nxt is not None and not keep and self._chain_ok(cn + "2c", nxt, mid, cout, x.shape[0] * m2.shape[1] * m2.shape[2]):
                    # m2 is rewritten only by the next block's 2b, which runs behind the chained launch: it may read it
                    pending = (cn + "2c", m2, res)
                else:
                    self._conv(cn + "2c", m2, out, residual=res, res_mode=1, bf16=True)   # fp32 for the next residual add, bf16 for the next conv
                x = out
            return x

        C2 = stage(2, "abc", 64, 256, 1, x)
        C3 = stage(3, "abcd", 128, 512, 2, C2)
        C4 = stage(4, ["a"] + [chr(98 + i) for i in range(self.stage4_blocks)], 256, 1024, 2, C3)
        C5 = stage(5, "abc", 512, 2048, 2, C4)
        self.C = (C2, C3, C4, C5)
        self._n_trunk = len(self._ops)                     # ops [0, _n_trunk): image -> C2..C5; the rest: FPN (+ RPN)  (forward_trunk / forward_top)
        self.pre = None
        t5, t4 = self._buf(H // 32, W // 32, 256), self._buf(H // 16, W // 16, 256)
        t3, t2 = self._buf(H // 8, W // 8, 256), self._buf(H // 4, W // 4, 256)
        self._conv("fpn_c5p5", C5, t5, relu=False, bf16=True)
        self._conv("fpn_c4p4", C4, t4, residual=t5, res_mode=2, relu=False, bf16=True)
        self._conv("fpn_c3p3", C3, t3, residual=t4, res_mode=2, relu=False, bf16=True)
        self._conv("fpn_c2p2", C2, t2, residual=t3, res_mode=2, relu=False, bf16=True)
        P2, P3 = self._buf(H // 4, W // 4, 256), self._buf(H // 8, W // 8, 256)
        P4, P5 = self._buf(H // 16, W // 16, 256), self._buf(H // 32, W // 32, 256)
        rp = self.rpn is not None                       # the RPN's shared convolution reads the pyramid maps
        self._conv("fpn_p2", t2, P2, relu=False, bf16=rp)
        self._conv("fpn_p3", t3, P3, relu=False, bf16=rp)
        self._conv("fpn_p4", t4, P4, relu=False, bf16=rp)
        self._conv("fpn_p5", t5, P5, relu=False, bf16=rp)
        self.P = (P2, P3, P4, P5)
        self.pre = (t2, t3, t4, t5)                        # top-down sums: the inputs of fpn_p2..p5 (kept for the joint backward)
        if self.rpn is not None:
            from .utils import generate_pyramid_anchors
            P6 = self._buf(H // 64, W // 64, 256)
            self._ops.append(("sub2", P5, P6))
            if self.fast_bf16:
                self._ops.append(("cast", P6, self._bf(P6)))
            self.P6 = P6
            self.rpn_heads, self.rpn_shared = [], []
            for p in (P2, P3, P4, P5, P6):
                sh = self._buf(p.shape[1], p.shape[2], 512)
                hd = self._buf(p.shape[1], p.shape[2], self.head_channels)
                self._conv("rpn_conv_shared", p, sh, bf16=True)
                self._conv("rpn_head", sh, hd, relu=False)
                self.rpn_heads.append(hd)
                self.rpn_shared.append(sh)
            shapes = [[-(-H // st), -(-W // st)] for st in self.rpn["strides"]]
            anchors = generate_pyramid_anchors(self.rpn["scales"], self.rpn["ratios"], shapes, self.rpn["strides"],
                                               self.rpn.get("anchor_stride", 1)).astype(np.float32)
            self.anchors = torch.tensor(anchors, device=self.device)
        self._bufs.append(rgbx)
        # plan-owned split-K workspace: its address is baked into the captured hipGraph
        self._ws = torch.empty(max(self._ws_bytes, 16), dtype=torch.uint8, device=self.device)

    # ------------------------------------------------------------------ run
    def _run_ops(self, lo=0, hi=None):
        """Enqueue ops [lo, hi) of the plan on the current stream."""
        lib = self.lib
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        wsp, wsb = C.c_void_p(self._ws.data_ptr()), self._ws.numel()
        for op in self._ops[lo:hi]:
            kind = op[0]
            if kind == "conv":
                rc = lib.dc_conv2d_nhwc_f32(C.byref(op[1]), wsp, wsb, stream)
                if rc:
                    check(rc, "dc_conv2d_nhwc_f32(%s)" % op[2])
            elif kind == "chain":
                rc = lib.dc_pw_chain_f32(C.byref(op[1]), stream)
                if rc:
                    check(rc, "dc_pw_chain_f32(%s)" % op[2])
            elif kind == "bconv":
                rc = lib.dc_conv2d_bf16(C.byref(op[1]), wsp, wsb, stream)
                if rc:
                    check(rc, "dc_conv2d_bf16(%s)" % op[2])
            elif kind == "cast":
                ops.to_bf16(op[1], out=op[2])
            elif kind == "bnfold":
                b = op[1]
                ops.bn_fold(b["gamma"], b["beta"], b["bias"], b["mean"], b["var"], op[2], op[3])
            elif kind == "mold":
                ops.mold_image_rgbx(op[1], self.mean_pixel, out=op[2])
            elif kind == "sub2":
                ops.subsample2(op[1], out=op[2])
            else:
                ops.maxpool3x3s2_same(op[1], out=op[2])

    def conv_table(self):
        """[(layer, flops, bm, bn, split_k, kernel)] for every conv of the plan, in launch order; kernel = rocprof's spelling of the
        template instantiation the library launches for the layer (dc_conv2d_kernel_name)."""
        rows = []
        for op in self._ops:
            if op[0] == "chain":
                d, buf = op[1], C.create_string_buffer(64)
                check(self.lib.dc_pw_chain_kernel_name(C.byref(d), buf, 64), "dc_pw_chain_kernel_name")
                rows.append((op[2], 2.0 * d.M * (d.K1 * d.N1 + d.N1 * d.N2), 32, d.N1, 1, buf.value.decode()))
                continue
            if op[0] != "conv":
                continue
            d, bm, bn, sk = op[1], C.c_int(), C.c_int(), C.c_int()
            check(self.lib.dc_conv2d_tile_config(C.byref(d), C.byref(bm), C.byref(bn), C.byref(sk)), "dc_conv2d_tile_config")
            buf = C.create_string_buffer(128)
            check(self.lib.dc_conv2d_kernel_name(C.byref(d), buf, 128), "dc_conv2d_kernel_name")
            s = self._specs[op[2]]
            rows.append((op[2], 2.0 * d.N * d.Ho * d.Wo * d.Cout * s.k * s.k * s.cin, bm.value, bn.value, sk.value, buf.value.decode()))
        return rows

    def conv_algorithmic_bytes(self):
        """{layer: compulsory HBM bytes of one launch} = input + output + weights (the 16-position Winograd image where that is what
        the kernel reads) + the residual / upsample-add operand: what `roofline.algorithmic_bytes` in bench.py sums (fp32 plans)."""
        out = {}
        for op in self._ops:
            if op[0] == "chain":                              # input + intermediate (written once) + shortcut + output + both kernels
                d = op[1]
                out[op[2]] = 4.0 * (d.M * (d.K1 + (2 if d.residual else 1) * d.N1 + d.N2) + d.K1 * d.N1 + d.N1 * d.N2)
                continue
            if op[0] != "conv":
                continue
            d, s = op[1], self._specs[op[2]]
            wbytes = ((16 * (6 if self.wino_products == "b3" else 4)) if op[2] in self._wwino else s.k * s.k * 4) * d.Cin * s.cout
            b = 4.0 * d.N * d.H * d.W * d.Cin + 4.0 * d.N * d.Ho * d.Wo * d.Cout + wbytes
            if d.res_mode == 1:
                b += 4.0 * d.N * d.Ho * d.Wo * d.Cout
            elif d.res_mode == 2:
                b += 1.0 * d.N * d.Ho * d.Wo * d.Cout
            out[op[2]] = b
        return out

    def time_convs(self, reps=3, beside=None):
        """Eager replay with a HIP event pair around every conv launch on the launch stream; returns
        [(layer, mean milliseconds)] (the roofline leg of bench.py).  beside: optional callable that enqueues the work
        that shares the GPU with the encoder in the training pipeline (one decoder train step on the decoder's own
        stream); it is called right before every timed replay, so the convolutions are timed IN the pipeline's
        conditions -- what a rocprofv3 kernel trace of the whole bench sees -- instead of alone on the chip."""
        lib = self.lib
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        wsp, wsb = C.c_void_p(self._ws.data_ptr()), self._ws.numel()
        convs = [op for op in self._ops if op[0] in ("conv", "chain")]
        acc = [0.0] * len(convs)
        for _ in range(reps):
            self._run_ops()
            torch.cuda.synchronize()
            if beside is not None:
                beside()
            evs = []
            for op in convs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = lib.dc_pw_chain_f32(C.byref(op[1]), stream) if op[0] == "chain" else lib.dc_conv2d_nhwc_f32(C.byref(op[1]), wsp, wsb, stream)
                e1.record()
                if rc:
                    check(rc, "dc_conv2d_nhwc_f32(%s)" % op[2])
                evs.append((e0, e1))
            torch.cuda.synchronize()
            for i, (e0, e1) in enumerate(evs):
                acc[i] += e0.elapsed_time(e1)
        return [(op[2], a / reps) for op, a in zip(convs, acc)]

    def time_bconvs(self, reps=3):
        """bf16-storage mode (configs[4]): eager replay with a HIP event pair around every dc_conv2d_bf16 launch on the launch
        stream; returns [(layer, flops, mean milliseconds, block tile (256 | 128), split-K slices)] in launch order -- the
        roofline leg of `bench.py --config joint` (a split-K layer's time includes its slab reduction)."""
        lib = self.lib
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        wsp, wsb = C.c_void_p(self._ws.data_ptr()), self._ws.numel()
        convs = [op for op in self._ops if op[0] == "bconv"]
        acc = [0.0] * len(convs)
        for _ in range(reps):
            self._run_ops()
            torch.cuda.synchronize()
            evs = []
            for op in convs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = lib.dc_conv2d_bf16(C.byref(op[1]), wsp, wsb, stream)
                e1.record()
                if rc:
                    check(rc, "dc_conv2d_bf16(%s)" % op[2])
                evs.append((e0, e1))
            torch.cuda.synchronize()
            for i, (e0, e1) in enumerate(evs):
                acc[i] += e0.elapsed_time(e1)
        rows = []
        for op, a in zip(convs, acc):
            d, sk = op[1], C.c_int(0)
            tile = int(lib.dc_conv2d_bf16_tile(C.byref(d), C.byref(sk)))
            rows.append((op[2], 2.0 * d.N * d.Ho * d.Wo * d.Cout * d.kh * d.kw * d.Cin, a / reps, tile, int(sk.value)))
        return rows

    def forward(self, images_u8=None):
        """images_u8: [B,H,W,3] uint8 torch tensor (any device) or None to reuse self.images.
        Returns (P2, P3, P4, P5), plan-owned buffers valid until the next forward()."""
        if images_u8 is not None:
            # host arrays arrive as pageable memory: the runtime moves one 1024x1024 image in ~64 KB pieces (48 copy kernels, 0.7 ms
            # on the compute queue).  A pinned staging buffer + one DMA was tried and measured SLOWER inside the joint step (35 vs
            # 12.9 ms: the DMA and the CPU-side refill of the buffer stall each other on this box); callers that care hand over a
            # device-resident uint8 tensor (bench.py does, the data generator can prefetch one).
            # (pageable memory: the copy must be complete when this call returns -- on this runtime a non_blocking copy from unpinned
            # memory really is asynchronous, and the caller's array may be freed or refilled right away: round 3 found the training
            # pipeline reading half-overwritten images that way.  Device tensors and pinned buffers stay asynchronous.)
            self.images.copy_(images_u8, non_blocking=bool(images_u8.is_cuda or images_u8.is_pinned()))
        if not self.use_graph:
            self._run_ops()
        elif self._graph is not None:
            self._graph.replay()
        elif not self._warm:
            self._run_ops()                 # first call: eager (sets kernel attributes, sizes the workspace)
            self._warm = True
        else:
            g = torch.cuda.CUDAGraph()
            # thread-local capture: the RCCL watchdog thread of a multi-GPU run polls events while this thread captures;
            # under the default (global) mode that would invalidate the capture
            with ops.no_gc_during_capture(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._run_ops()
            self._graph = g
            g.replay()
        return self.P

    # The pass in two halves, for a caller that runs the frozen backbone of the NEXT batch beside the rest of this batch's step
    # (pipeline.JointTrainPipeline): forward_trunk() = image -> C2..C5 (reads nothing a train step changes when no ResNet stage is
    # trainable), forward_top() = the FPN and the RPN on the C maps of the last forward_trunk().  Each half is its own hipGraph;
    # forward_trunk(); forward_top() enqueues exactly the launches of forward().
    def _run_part(self, part, lo, hi):
        if not self.use_graph:
            self._run_ops(lo, hi)
            return
        g = self._part_graphs.get(part)
        if g is not None:
            g.replay()
        elif part not in self._part_warm:
            self._run_ops(lo, hi)                           # first call: eager (kernel attributes, workspace sizes)
            self._part_warm.add(part)
        else:
            g = torch.cuda.CUDAGraph()
            with ops.no_gc_during_capture(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._run_ops(lo, hi)
            self._part_graphs[part] = g
            g.replay()

    def forward_trunk(self, images_u8=None):
        if self._external_bn:
            raise RuntimeError("forward_trunk: this plan folds trainable BatchNorm layers inside the backbone pass; use forward()")
        if images_u8 is not None:
            self.images.copy_(images_u8, non_blocking=bool(images_u8.is_cuda or images_u8.is_pinned()))
        self._run_part("trunk", 0, self._n_trunk)
        return self.C

    def forward_top(self):
        self._run_part("top", self._n_trunk, None)
        return self.P

    def proposals(self, debug=False):
        """ProposalLayer on the RPN heads of the last forward(): normalised boxes [B,count,4], zero padded."""
        if self.rpn is None:
            raise RuntimeError("this plan was built without the RPN")
        r = self.rpn
        return ops.rpn_proposals(self.rpn_heads, self.anchors, (self.H, self.W), r["proposal_count"], r["nms_threshold"],
                                 r.get("bbox_std", (0.1, 0.1, 0.2, 0.2)), anchors_per_loc=len(r["ratios"]), debug=debug,
                                 head_stride=self.head_channels)

    def normalize_boxes(self, rois_px):
        """rois / [h,w,h,w] in float32, as modified_dense_model.py:1522-1527 (the molded image's size,
        no window/scale correction -- the reference's own quirk).  Returns a device tensor [B,R,4]."""
        hw = np.array([self.H, self.W, self.H, self.W], np.float32)
        if isinstance(rois_px, torch.Tensor):
            rois_px = rois_px.detach().cpu().numpy()
        return torch.tensor(np.asarray(rois_px, np.float32) / hw, device=self.device).contiguous()

    def roi_features(self, rois_px=None, out=None, boxes_norm=None):
        """rois_px [B,R,4] (y1,x1,y2,x2) pixels of the molded image (or boxes_norm from
        normalize_boxes, device-resident) -> [B,R,7,7,256]."""
        boxes = boxes_norm if boxes_norm is not None else self.normalize_boxes(rois_px)
        return ops.roi_align_pyramid(list(self.P), boxes, float(self.H * self.W), 7, out=out)


class Vgg16Plan(EncoderPlan):
    """Alternative backbone for BASELINE configs[2]'s label ("VGG16 backbone + RoIAlign + inject-LSTM"): the 13 convolutions of
    keras.applications VGG16 (`image captioning/vgg16.py:9-18` loads that model) at the benchmark's image size, RoIAlign
    (crop_and_resize, 7x7) on block5_conv3 (stride 16, 512 channels).  No dense-captioning path of the reference runs VGG16
    (SURVEY.md section 1), so this plan has no parity target of its own: every kernel in it is the one the ResNet-FPN plan is
    validated with (conv 3x3 + bias + ReLU, RoIAlign) plus a 2x2 max pool checked against NumPy; 641.43 GFLOP per 1024x1024 image.
    The first convolution reads the RGBX image zero-padded to 32 channels (the implicit-GEMM loader wants Cin % 32 == 0)."""
    feat_channels = 512

    def __init__(self, weights, batch, height, width, device, mean_pixel=(123.7, 116.8, 103.9), use_graph=True, math=None, winograd=None,
                 wino_products=None):
        from .layers import vgg16_convs
        import os
        if height % 16 or width % 16:
            raise ValueError("Image size must be dividable by 16 (got %dx%d)" % (height, width))
        self.lib = _lib.load()
        self.math = conv_math_mode(math)
        self.winograd = (os.environ.get("DCAP_WINOGRAD", "1") != "0") if winograd is None else bool(winograd)      # all 13 layers are 3x3 / stride 1
        self.wino_products = wino_products or os.environ.get("DCAP_WINO_PRODUCTS", "b3")
        self._wwino = {}
        self.layer_math = False                         # (13 3x3 layers: nothing to choose)
        self.B, self.H, self.W = batch, height, width
        self.device = torch.device(device)
        self.mean_pixel = [float(v) for v in mean_pixel]
        self.use_graph = use_graph
        self._graph = None
        self._warm = False
        self._specs = {s.name: s for s in vgg16_convs()}
        self.rpn = None
        self._external = {}
        self._w = {}
        self._upload(weights)
        self._build()

    def _upload(self, W):
        dev = self.device
        for s in self._specs.values():
            k = np.asarray(W[s.name + "/kernel"], np.float32)
            if tuple(k.shape) != (s.k, s.k, s.cin, s.cout):
                raise ValueError("%s/kernel has shape %s, expected %s" % (s.name, k.shape, (s.k, s.k, s.cin, s.cout)))
            if s.cin == 3:                                  # R, G, B, then 29 zero input channels
                k = np.concatenate([k, np.zeros((s.k, s.k, 29, s.cout), np.float32)], axis=2)
            self._w[s.name] = (torch.tensor(pack_conv_kernel(k), device=dev), None,
                               torch.tensor(np.asarray(W[s.name + "/bias"], np.float32), device=dev))

    def _build(self):
        B, H, W = self.B, self.H, self.W
        self._ops, self._ws_bytes, self.flops, self._bufs = [], 0, 0.0, []
        self.images = torch.empty((B, H, W, 3), dtype=torch.uint8, device=self.device)
        x32 = torch.empty((B, H, W, 32), dtype=torch.float32, device=self.device)
        self._bufs += [x32]
        self._ops.append(("mold32", self.images, x32))       # mean-subtracted RGB + 29 zero channels, one kernel (dc_mold_image_padded_f32)
        x, h, w = x32, H, W
        for b, (n, cout) in enumerate(((2, 64), (2, 128), (3, 256), (3, 512), (3, 512)), 1):
            for i in range(1, n + 1):
                y = self._buf(h, w, cout)
                self._conv("block%d_conv%d" % (b, i), x, y)
                x = y
            if b < 5:
                h, w = h // 2, w // 2
                y = self._buf(h, w, cout)
                self._ops.append(("pool2", x, y))
                x = y
        self.C = (x,)
        self.P = (x, x, x, x)            # every RoI reads the one stride-16 map, whatever level its size routes it to
        self.pre = None
        self._ws = torch.empty(max(self._ws_bytes, 16), dtype=torch.uint8, device=self.device)

    def _run_ops(self):
        lib = self.lib
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        wsp, wsb = C.c_void_p(self._ws.data_ptr()), self._ws.numel()
        for op in self._ops:
            kind = op[0]
            if kind == "conv":
                rc = lib.dc_conv2d_nhwc_f32(C.byref(op[1]), wsp, wsb, stream)
                if rc:
                    check(rc, "dc_conv2d_nhwc_f32(%s)" % op[2])
            elif kind == "mold32":
                ops.mold_image_padded(op[1], self.mean_pixel, out=op[2])
            else:
                ops.maxpool2x2s2(op[1], out=op[2])

    def roi_features(self, rois_px=None, out=None, boxes_norm=None):
        """-> [B,R,7,7,512] crops of block5_conv3."""
        boxes = boxes_norm if boxes_norm is not None else self.normalize_boxes(rois_px)
        return ops.roi_align_pyramid(list(self.P), boxes, float(self.H * self.W), 7, out=out)
