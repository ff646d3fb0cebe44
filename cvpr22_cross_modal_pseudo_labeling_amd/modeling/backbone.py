"""ResNet-50-C4 trunk and the per-RoI res5 head, with FrozenBatchNorm folded into the convolutions.

Structure, strides (stride on the 1x1: MODEL.RESNETS.STRIDE_IN_1X1) and parameter / buffer names follow
maskrcnn_benchmark/modeling/backbone/resnet.py:81-152 (``ResNet``), :155-204 (``ResNetHead``),
:239-344 (``Bottleneck``), :347-366 (``BaseStem``) and backbone/backbone.py:12-20, so reference
checkpoints (``backbone.body.*``, ``roi_heads.box.feature_extractor.head.layer4.*``) load by name.

MI355X-first difference: the reference runs ``x * scale + shift`` as a separate memory-bound
pass after every convolution (layers/batch_norm.py:19-31).  Here each conv+FrozenBN pair is one
MIOpen call with the affine folded into the weights (``w * scale``) and bias (``shift``); the fold
is a weight-sized op, differentiable w.r.t. the conv weight, and cached for frozen modules.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..layers import Conv2d, DFConv2d, FrozenBatchNorm2d, bias_relu_, split_conv_same, split_linear
from ..layers.pair_bottleneck import bottleneck_pair, is_placeholder, pair_weight


class ConvBN(nn.Module):
    """``conv`` (bias-free) followed by a FrozenBatchNorm2d ``bn``, evaluated as one convolution."""

    def __init__(self, conv, bn):
        super().__init__()
        self.conv = conv
        self.bn = bn
        self._cache = None

    def folded(self):
        w = self.conv.weight
        frozen = not w.requires_grad
        key = (id(w), w._version, w.device) + self.bn.fold_key()  # the FrozenBN buffers are part of the folded weight
        if frozen and self._cache is not None and self._cache[0] == key:
            return self._cache[1], self._cache[2]
        scale, shift = self.bn.fold()
        fw = w * scale.reshape(-1, 1, 1, 1)
        if frozen:
            self._cache = (key, fw.detach(), shift.detach())
        return fw, shift

    def forward(self, x):
        w, b = self.folded()
        c = self.conv
        return F.conv2d(x, w, b, c.stride, c.padding, c.dilation, c.groups)


class Bottleneck(nn.Module):
    def __init__(self, in_channels, bottleneck_channels, out_channels, num_groups=1, stride_in_1x1=True,
                 stride=1, dilation=1, dcn_config=None):
        super().__init__()
        dcn_config = dcn_config or {}
        self.downsample = None
        if in_channels != out_channels:
            down_stride = stride if dilation == 1 else 1
            self.downsample = nn.Sequential(
                Conv2d(in_channels, out_channels, kernel_size=1, stride=down_stride, bias=False),
                FrozenBatchNorm2d(out_channels))
            nn.init.kaiming_uniform_(self.downsample[0].weight, a=1)
        if dilation > 1:
            stride = 1
        stride_1x1, stride_3x3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=stride_1x1, bias=False)
        self.bn1 = FrozenBatchNorm2d(bottleneck_channels)
        # resnet.py:286-300: a stage listed in MODEL.RESNETS.STAGE_WITH_DCN builds its 3x3 as a deformable convolution
        # (offsets -- and masks, WITH_MODULATED_DCN -- predicted by an ordinary 3x3; layers/misc.py::DFConv2d, parameter
        # names ``conv2.offset.*`` / ``conv2.conv.*`` as in the reference)
        self.with_dcn = bool(dcn_config.get("stage_with_dcn", False))
        if self.with_dcn:
            self.conv2 = DFConv2d(bottleneck_channels, bottleneck_channels,
                                  with_modulated_dcn=dcn_config.get("with_modulated_dcn", False), kernel_size=3,
                                  stride=stride_3x3, groups=num_groups, dilation=dilation,
                                  deformable_groups=dcn_config.get("deformable_groups", 1), bias=False)
        else:
            self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=stride_3x3,
                                padding=dilation, bias=False, groups=num_groups, dilation=dilation)
        self.bn2 = FrozenBatchNorm2d(bottleneck_channels)
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False)
        self.bn3 = FrozenBatchNorm2d(out_channels)
        # (the initialisation ORDER is kept as it was before the deformable variant existed: the seeded tiny models of
        # tests/ draw the same weights)
        for l in (self.conv1,) + (() if self.with_dcn else (self.conv2,)) + (self.conv3,):
            nn.init.kaiming_uniform_(l.weight, a=1)
        # fused views over the same parameters (no extra state_dict entries)
        self._f1 = [ConvBN(self.conv1, self.bn1)]
        self._f2 = [ConvBN(self.conv2, self.bn2)] if not self.with_dcn else None
        self._f3 = [ConvBN(self.conv3, self.bn3)]
        self._fd = [ConvBN(self.downsample[0], self.downsample[1])] if self.downsample is not None else None
        self.conv3x3_nchw = None  # None = by autograd mode (see forward_nhwc)
        # pair-layout split GEMM with fused epilogues and implicit 3x3 (csrc/split_gemm.hip): the NHWC route.  The other
        # attribute combinations (K-concatenated split through a library GEMM, fp32 GEMM, per-layer 3x3) are the earlier
        # forms of the same block, kept as cross-checks: tests/test_heads_gpu.py sets them explicitly
        self.split_gemm = self.split_conv = self.pair_gemm = True
        # a block that hands its result on in pair layout (want_pair) does not also write it as fp32: the next block
        # takes its identity shortcut from the pair form (hi + lo).  False = fp32 + pair (cross-check in the tests)
        self.pair_only_chain = True
        self._pair_cache = None

    def forward(self, x):
        out = F.relu_(self._f1[0](x))
        if self.with_dcn:  # the deformable 3x3 (csrc/split_gemm.hip DEFORM mode / deform_conv*.hip), then its FrozenBN affine
            out = F.relu_(self.bn2(self.conv2(out)))
        else:
            out = F.relu_(self._f2[0](out))
        out = self._f3[0](out)
        identity = self._fd[0](x) if self._fd is not None else x
        out += identity
        return F.relu_(out)

    def nhwc_supported(self):
        c1, c2, c3 = self.conv1, self.conv2, self.conv3
        return (not self.with_dcn and c1.kernel_size == (1, 1) and c3.kernel_size == (1, 1) and c1.groups == 1 and c3.groups == 1
                and c1.padding == (0, 0) and c3.padding == (0, 0) and c3.stride == (1, 1))

    def pair_supported(self):
        c1, c2, c3 = self.conv1, self.conv2, self.conv3
        d = self.downsample[0] if self.downsample is not None else None
        return (self.nhwc_supported() and c2.stride == (1, 1) and c2.dilation == (1, 1) and c2.groups == 1
                and c2.kernel_size[0] % 2 == 1 and c2.kernel_size[1] % 2 == 1
                and c2.padding == (c2.kernel_size[0] // 2, c2.kernel_size[1] // 2)
                and all(ch % 32 == 0 for ch in (c1.in_channels, c1.out_channels, c2.out_channels, c3.out_channels))
                and (d is None or (d.kernel_size == (1, 1) and d.padding == (0, 0) and d.groups == 1
                                   and d.stride == c1.stride))
                and (d is not None or (c1.stride == (1, 1) and c1.in_channels == c3.out_channels)))

    def prep_plan_convs(self):
        """[(weight, scale)] of conv1, conv2, conv3 (, downsample) for ``prepare_weights_ahead`` when this block trains on the
        pair GEMM of a device; the scales are the very tensors ``_pair_node`` passes on (``FrozenBatchNorm2d.fold``)."""
        if not (self.pair_gemm and self.pair_supported()):
            return None
        ws = [self.conv1.weight, self.conv2.weight, self.conv3.weight] + ([self.downsample[0].weight] if self.downsample is not None else [])
        if not (any(w.requires_grad for w in ws) and all(w.is_cuda and w.is_contiguous() for w in ws)):
            return None
        bns = [self.bn1, self.bn2, self.bn3] + ([self.downsample[1]] if self.downsample is not None else [])
        return [(w, bn.fold()[0]) for w, bn in zip(ws, bns)]

    def takes_pair_only_input(self):
        """This block can consume an input that exists in pair layout only (conv1 operand and shortcut from the pair form):
        what a producer must check before it drops the fp32 copy of its output (``pair_only``)."""
        return bool(self.pair_gemm and self.pair_supported())

    def _forward_pair(self, x, prestrided, xp, want_pair, pool=False, select=None, pair_only=False, pool_only_ok=False):
        """The block as ONE autograd node on the pair-layout split GEMM (layers/pair_bottleneck.py): bias, shortcut,
        ReLU and the next layer's operand split live in the GEMM epilogues, the 3x3 is an implicit GEMM."""
        r, h, w, c = x.shape
        sy, sx = self.conv1.stride
        if not prestrided and (sy, sx) != (1, 1):
            if xp is not None:
                xp = xp.view(r, h, w, 2 * c)[:, ::sy, ::sx, :].contiguous().view(-1, 2 * c)
            # the fp32 rows are only read by an identity shortcut or by the input gradient
            keep = self._fd is None or (torch.is_grad_enabled() and x.requires_grad) or xp is None
            x = x[:, ::sy, ::sx, :]
            hs, ws = x.shape[1], x.shape[2]
            x2d = (x.reshape(-1, c) if is_placeholder(x) else x.contiguous().view(-1, c)) if keep else None
        else:
            hs, ws = h, w
            x2d = x.reshape(-1, c)
        return self._pair_node(x2d, xp, r, hs, ws, want_pair, pool, select, pair_only, pool_only_ok)

    def forward_pair_rows(self, xp, r, hs, ws, want_pair=False, pool=False, pair_only=False):
        """The block on rows that exist only in pair layout ([r*hs*ws, 2*Cin] bf16, e.g. written by the pooler): needs
        the projection shortcut (no fp32 rows for an identity shortcut) and no gradient w.r.t. the input."""
        assert self._fd is not None
        return self._pair_node(None, xp, r, hs, ws, want_pair, pool, None, pair_only)

    def _pair_node(self, x2d, xp, r, hs, ws, want_pair, pool, select=None, pair_only=False, pool_only_ok=False):
        from .. import _C
        # RAW weights + folded FrozenBN (scale, shift) pairs: the fold itself happens inside the node's weight-prep kernel
        w1, w2, w3 = self.conv1.weight, self.conv2.weight, self.conv3.weight
        (s1, b1), (s2, b2), (s3, b3) = self.bn1.fold(), self.bn2.fold(), self.bn3.fold()
        wd = sd = bd = None
        if self.downsample is not None:
            wd = self.downsample[0].weight
            sd, bd = self.downsample[1].fold()
        ws_all = [t for t in (w1, w2, w3, wd) if t is not None]
        bns = [self.bn1, self.bn2, self.bn3] + ([self.downsample[1]] if self.downsample is not None else [])
        wpairs = None
        if not any(t.requires_grad for t in ws_all):
            # frozen block: the pair forms of the folded weights (and the summed shift) are computed once
            key = tuple((id(t), t._version, t.device) for t in ws_all) + tuple(k for bn in bns for k in bn.fold_key())
            if self._pair_cache is None or self._pair_cache[0] != key:
                wp = {"w1": _C.weight_prep_pair(w1, s1)[0], "w2": _C.weight_prep_pair(w2, s2)[0],
                      "w3": _C.weight_prep_pair(w3, s3)[0],
                      "wd": _C.weight_prep_pair(wd, sd)[0] if wd is not None else None}
                if wd is not None:
                    wp["w3d"] = torch.cat([wp["w3"], wp["wd"]], 1)  # conv3 + projection shortcut as one product
                self._pair_cache = (key, wp, (b3 if bd is None else b3 + bd).contiguous())
            wpairs, b3s = self._pair_cache[1], self._pair_cache[2]
        else:
            b3s = b3 if bd is None else b3 + bd
            plan = self.__dict__.get("_prep_plan")  # trainable block: operands prepared behind the optimizer step, if still fresh
            if plan is not None:
                wpairs = plan.lookup(id(self), (s1, s2, s3) if sd is None else (s1, s2, s3, sd))
        res = bottleneck_pair(x2d, xp, (hs, ws), w1, b1, w2, b2, w3, b3s, wd, want_pair, wpairs, pool,
                              scales=(s1, s2, s3, sd),
                              want_f32=not (want_pair and pair_only and self.pair_only_chain), select=select,
                              pool_only_ok=pool_only_ok)
        out = res[0].view(r, hs, ws, res[0].shape[-1])
        if pool:
            out._ovis_pooled = res[2]  # [R, C] mean over the map, an output of the same autograd node (see pooled())
        sel = getattr(res[0], "_ovis_selected", None)
        if sel is not None:
            out._ovis_selected = sel
        return (out, res[1]) if want_pair else out

    def forward_nhwc(self, x, prestrided=False, xp=None, want_pair=False, pool=False, select=None, pair_only=False,
                     pool_only_ok=False):
        """Same block on an NHWC tensor ``x`` [R, H, W, C] (contiguous); ``prestrided``: x already holds only the
        positions conv1 / the shortcut read (the pooler applied their common stride).  The 1x1 convolutions -- 53 % of the
        res5 FLOPs -- become ONE row-major GEMM over all R*H*W positions each ([R*H*W, Cin] x [Cin, Cout], bias
        = the folded FrozenBN shift) instead of R batched [Cout, Cin] x [Cin, 49] products behind layout
        transposes; a stride-2 1x1 (STRIDE_IN_1X1) first drops the rows it never reads.  By default the GEMMs run as
        bf16 hi/lo split products on the bf16 matrix pipe (~4e-6 relative error, ``split_gemm = False`` selects
        the fp32 GEMM); the 3x3 goes through MIOpen.  Values equal ``forward`` up to that error.
        ``xp``: the pair-layout form of x when the producer already wrote it; ``want_pair``: also return the pair
        form of the result (or None) for the next block -- both only used by the pair-layout route.  ``pair_only``: the
        consumer of the result ``takes_pair_only_input()``, so the fp32 copy may be dropped (the caller looks ahead:
        ``chain_nhwc``); a block that is NOT on the pair route never receives such an input."""
        if self.pair_gemm and x.is_cuda and self.pair_supported():
            return self._forward_pair(x, prestrided, xp, want_pair, pool, select, pair_only, pool_only_ok)
        if is_placeholder(x):
            raise RuntimeError("Bottleneck.forward_nhwc: the input exists in pair layout only (its fp32 handle is a "
                               "placeholder) but this block is not on the pair-GEMM route; the producer must keep the "
                               "fp32 copy (pair_only=False)")
        r, h, w, c = x.shape
        sy, sx = self.conv1.stride
        if prestrided:
            assert self._fd is None or self.downsample[0].stride == (sy, sx)
            xs = x
        else:
            xs = x[:, ::sy, ::sx, :].contiguous() if (sy, sx) != (1, 1) else x
        hs, ws = xs.shape[1], xs.shape[2]
        x2d = xs.view(-1, c)
        # raw products (no bias): bf16 hi/lo split GEMMs on the bf16 matrix pipe (layers/cross_modal.py::split_linear)
        # or fp32 GEMMs; every bias / shortcut add / ReLU below is ONE fused in-place pass (bias_relu_)
        if self.split_gemm:
            def products(a, *ws):
                return split_linear(a, *[t for w_ in ws for t in (w_, None)])
        else:
            def products(a, *ws):
                return tuple(torch.mm(a, w_.t()) for w_ in ws)
        w1, b1 = self._f1[0].folded()
        w1 = w1.view(w1.shape[0], -1)
        idn, bd = None, None
        if self._fd is not None:
            wd, bd = self._fd[0].folded()
            wd = wd.view(wd.shape[0], -1)
            dy, dx = self.downsample[0].stride
            if prestrided or (dy, dx) == (sy, sx):  # conv1 and the projection shortcut read the same rows
                out, idn = products(x2d, w1, wd)
            else:
                (out,) = products(x2d, w1)
                (idn,) = products(x[:, ::dy, ::dx, :].contiguous().view(-1, c), wd)
        else:
            (out,) = products(x2d, w1)
        out = bias_relu_(out, b1)
        w2, b2 = self._f2[0].folded()
        c2 = self.conv2
        if (self.split_conv and c2.stride == (1, 1) and c2.dilation == (1, 1) and c2.groups == 1
                and c2.kernel_size[0] % 2 == 1 and c2.kernel_size[1] % 2 == 1
                and c2.padding == (c2.kernel_size[0] // 2, c2.kernel_size[1] // 2)):
            # the 3x3 as a bf16 hi/lo split GEMM over its im2col rows (layers/cross_modal.py::split_conv_same)
            out = split_conv_same(out.view(r, hs, ws, out.shape[-1]), w2)  # [r*hs*ws, Cout]
            ho, wo = hs, ws
        else:
            nchw = self.conv3x3_nchw
            if nchw is None:
                # MIOpen's NCHW fp32 Winograd is its fastest 3x3 forward for these shapes (the NHWC pick at R = 2000
                # is a 50 TFLOP/s grouped-conv kernel), worth two 0.1 ms layout copies around it; under autograd
                # MIOpen runs NHWC implicit-GEMM kernels for all three directions: the tensor stays channels_last
                nchw = not (torch.is_grad_enabled() and (out.requires_grad or w2.requires_grad))
            if nchw:
                out = F.conv2d(out.view(r, hs, ws, out.shape[-1]).permute(0, 3, 1, 2).contiguous(), w2, None, c2.stride,
                               c2.padding, c2.dilation, c2.groups)
                out = out.permute(0, 2, 3, 1).contiguous()
            else:
                out = F.conv2d(out.view(r, hs, ws, out.shape[-1]).permute(0, 3, 1, 2),
                               w2.contiguous(memory_format=torch.channels_last), None, c2.stride, c2.padding,
                               c2.dilation, c2.groups)
                out = out.permute(0, 2, 3, 1)
                if not out.is_contiguous():
                    out = out.contiguous()
            ho, wo = out.shape[1], out.shape[2]
            out = out.view(-1, out.shape[3])
        out = bias_relu_(out, b2)
        w3, b3 = self._f3[0].folded()
        (out,) = products(out, w3.view(w3.shape[0], -1))
        out = bias_relu_(out, b3 if bd is None else b3 + bd, idn if idn is not None else x.view(-1, c))
        out = out.view(r, ho, wo, out.shape[-1])
        return (out, None) if want_pair else out


def chain_nhwc(blocks, y, yp=None, first=None, last=None, then=None):
    """Run consecutive bottlenecks on an NHWC activation.  Every block but the last hands its result on in pair layout;
    it drops the fp32 copy (``pair_only``) only when the NEXT block can take a pair-only input -- otherwise (e.g.
    STRIDE_IN_1X1 False: the stage's first block has a strided 3x3 and runs the per-layer route) both forms are
    written.  ``first`` / ``last``: extra keyword arguments of the first / last block's ``forward_nhwc``.  ``then``: the
    block a LATER call continues the chain with -- the last block then hands on ``(y, yp)`` exactly as it would inside
    one chain (the frozen prefix of the trunk, run ahead on a side stream: ``ResNetC4.forward_prefix``)."""
    n = len(blocks)
    for i, b in enumerate(blocks):
        kw = dict(first or {}) if i == 0 else {}
        nxt = blocks[i + 1] if i + 1 < n else then
        if nxt is not None:
            y, yp = b.forward_nhwc(y, xp=yp, want_pair=True, pair_only=nxt.takes_pair_only_input(), **kw)
        else:
            kw.update(last or {})
            y = b.forward_nhwc(y, xp=yp, **kw)
    return (y, yp) if then is not None else y


class Stem(nn.Module):
    def __init__(self, out_channels=64):
        super().__init__()
        self.conv1 = Conv2d(3, out_channels, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(out_channels)
        nn.init.kaiming_uniform_(self.conv1.weight, a=1)
        self._f = [ConvBN(self.conv1, self.bn1)]

    def forward(self, x):
        x = F.relu_(self._f[0](x))
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)

    def forward_gemm(self, x):
        """Frozen stem on the GPU: patches -> pair rows (csrc/split_gemm.hip::im2col_nchw_pair_kernel), the 7x7 as one
        split GEMM with the FrozenBN shift and the ReLU in its epilogue, max-pool on the NHWC result.  Returns the
        channels_last view [N, 64, H/4, W/4] (MIOpen's fp32 kernel for this 3-channel convolution runs at 6 TFLOP/s)."""
        from .. import _C
        w, b = self._f[0].folded()
        c = self.conv1
        key = (id(w), w._version) + self.bn1.fold_key()
        if getattr(self, "_wp", None) is None or self._wp[0] != key:
            k = w.shape[1] * w.shape[2] * w.shape[3]
            kp = -(-k // 32) * 32
            wm = w.new_zeros((w.shape[0], kp))
            wm[:, :k] = w.permute(0, 2, 3, 1).reshape(w.shape[0], k)   # k = (ky*KW + kx)*C + c
            self._wp = (key, pair_weight(wm))
        rows, (ho, wo) = _C.im2col_nchw_pair(x.contiguous(), c.kernel_size[0], c.kernel_size[1], c.stride[0], c.padding[0])
        y, _ = _C.split_gemm_pair(rows, self._wp[1], b, None, True)
        y = y.view(x.shape[0], ho, wo, -1).permute(0, 3, 1, 2)          # NCHW view of NHWC memory (channels_last)
        return F.max_pool2d(y, kernel_size=3, stride=2, padding=1)

    def gemm_supported(self, x):
        c = self.conv1
        return (x.is_cuda and not c.weight.requires_grad and not x.requires_grad and c.groups == 1 and c.dilation == (1, 1)
                and c.stride[0] == c.stride[1] and c.padding[0] == c.padding[1] and c.out_channels % 4 == 0)


def _make_stage(in_channels, bottleneck_channels, out_channels, block_count, num_groups, stride_in_1x1,
                first_stride, dilation=1, dcn_config=None):
    blocks, stride = [], first_stride
    for _ in range(block_count):
        blocks.append(Bottleneck(in_channels, bottleneck_channels, out_channels, num_groups, stride_in_1x1,
                                 stride, dilation, dcn_config))
        stride = 1
        in_channels = out_channels
    return nn.Sequential(*blocks)


_C4_STAGES = ((1, 3), (2, 4), (3, 6))  # (index, block_count) of R-50-C4 (resnet.py:27-31)


class ResNetC4(nn.Module):
    """stem + layer1..layer3, output stride 16, 1024 channels."""

    def __init__(self, cfg):
        super().__init__()
        r = cfg.MODEL.RESNETS
        if cfg.MODEL.BACKBONE.CONV_BODY != "R-50-C4":
            raise NotImplementedError("only R-50-C4 (the body every shipped config uses) is built")
        self.stem = Stem(r.STEM_OUT_CHANNELS)
        in_channels = r.STEM_OUT_CHANNELS
        width = r.NUM_GROUPS * r.WIDTH_PER_GROUP
        self.stages = []
        for index, count in _C4_STAGES:
            factor = 2 ** (index - 1)
            out_channels = r.RES2_OUT_CHANNELS * factor
            stage = _make_stage(in_channels, width * factor, out_channels, count, r.NUM_GROUPS,
                                r.STRIDE_IN_1X1, first_stride=int(index > 1) + 1,
                                dcn_config={"stage_with_dcn": r.STAGE_WITH_DCN[index - 1],          # resnet.py:110-124
                                            "with_modulated_dcn": r.WITH_MODULATED_DCN,
                                            "deformable_groups": r.DEFORMABLE_GROUPS})
            name = f"layer{index}"
            self.add_module(name, stage)
            self.stages.append(name)
            in_channels = out_channels
        self.out_channels = in_channels
        self.nhwc = self.train_nhwc = True  # False = per-layer NCHW convolutions (cross-check in the tests)
        self._freeze(cfg.MODEL.BACKBONE.FREEZE_CONV_BODY_AT)

    def _freeze(self, freeze_at):
        for i in range(max(freeze_at, 0)):
            m = self.stem if i == 0 else getattr(self, f"layer{i}")
            for p in m.parameters():
                p.requires_grad = False

    def _chain_ok(self, x, blocks):
        frozen = not any(p.requires_grad for p in self.parameters())
        plain = [b for b in blocks if not b.with_dcn]
        return (x.is_cuda and self.nhwc and all(b.nhwc_supported() for b in plain)
                and (frozen or (self.train_nhwc and all(b.pair_gemm and b.pair_supported() for b in plain))))

    def forward_prefix(self, x):
        """Stem and the FROZEN leading blocks (FREEZE_CONV_BODY_AT) on the NHWC chain: ``(y, yp, n_blocks)`` for
        ``forward(prefix=...)``, or None when there is nothing to run ahead (no frozen block before a trainable one, a
        deformable trunk, the per-layer NCHW route, a host tensor).  Nothing in it depends on a trainable parameter, so a
        training loop may run it for the NEXT batch beside this batch's backward (engine/trainer.py::PipelinedTrainer);
        the values are those of the un-split chain -- the last frozen block hands on what the next block takes."""
        blocks = [b for name in self.stages for b in getattr(self, name)]
        k = 0
        while k < len(blocks) and not any(p.requires_grad for p in blocks[k].parameters()):
            k += 1
        if (k == 0 or k == len(blocks) or any(b.with_dcn for b in blocks) or not self._chain_ok(x, blocks)
                or not (self.nhwc and self.stem.gemm_supported(x))):
            return None
        with torch.no_grad():
            x = self.stem.forward_gemm(x)
            y, yp = chain_nhwc(blocks[:k], x.permute(0, 2, 3, 1).contiguous(), then=blocks[k])
        return y, yp, k

    def forward(self, x, prefix=None):
        blocks = [b for name in self.stages for b in getattr(self, name)]
        if prefix is not None:  # the rest of the chain on the result of ``forward_prefix`` (same values as the whole chain)
            y, yp, k = prefix
            return [chain_nhwc(blocks[k:], y, yp).permute(0, 3, 1, 2)]
        x = self.stem.forward_gemm(x) if (self.nhwc and self.stem.gemm_supported(x)) else self.stem(x)
        plain = [b for b in blocks if not b.with_dcn]
        if self._chain_ok(x, blocks):
            if len(plain) != len(blocks):
                return [self._forward_mixed(x, blocks)]
            # layer1-3 in NHWC with the split-GEMM bottlenecks of the res5 head (1x1 = row-major GEMM, 3x3 = implicit
            # GEMM).  Frozen trunk (student-teacher configuration): always.
            # Trainable stages (teacher training): through the pair-layout autograd nodes as well -- 39.4 vs 42.2 ms per
            # step against MIOpen's NCHW kernels (both with a warm MIOpen kernel cache; ``train_nhwc = False``
            # selects MIOpen).
            y = chain_nhwc(blocks, x.permute(0, 2, 3, 1).contiguous())
            # the C4 map stays in NHWC memory (an NCHW-shaped view of it): the poolers read channels-last maps in place
            # (csrc/roi_align.hip::roi_align_fwd_nhwc_in_strided_kernel) and the RPN head wants NHWC rows anyway
            return [y.permute(0, 3, 1, 2)]
        x = x.contiguous()  # the GEMM stem hands over channels_last memory; MIOpen's NCHW kernels are the faster ones here
        for name in self.stages:
            x = getattr(self, name)(x)
        return [x]

    @staticmethod
    def _forward_mixed(x, blocks):
        """A trunk with deformable stages (STAGE_WITH_DCN): runs of ordinary bottlenecks stay on the NHWC pair-GEMM chain,
        a deformable block runs its own NCHW forward (``_C`` deformable ops take the reference's NCHW tensors) between two
        layout copies.  Returns the NCHW-shaped view of NHWC memory the poolers / RPN head read."""
        y = x.permute(0, 2, 3, 1).contiguous()
        run = []
        for b in blocks:
            if not b.with_dcn:
                run.append(b)
                continue
            if run:
                y = chain_nhwc(run, y)
                run = []
            y = b(y.permute(0, 3, 1, 2).contiguous()).permute(0, 2, 3, 1).contiguous()
        if run:
            y = chain_nhwc(run, y)
        return y.permute(0, 3, 1, 2)


class Backbone(nn.Sequential):
    """``backbone.body`` wrapper so parameter names match backbone/backbone.py:12-20."""

    def __init__(self, cfg):
        body = ResNetC4(cfg)
        super().__init__()
        self.add_module("body", body)
        self.out_channels = body.out_channels


class ResNetHead(nn.Module):
    """res5 (layer4) applied to every pooled RoI: [R,1024,14,14] -> [R,2048,7,7]."""

    def __init__(self, cfg):
        super().__init__()
        r = cfg.MODEL.RESNETS
        factor = 2 ** 3
        out_channels = r.RES2_OUT_CHANNELS * factor
        self.layer4 = _make_stage(out_channels // 2, r.NUM_GROUPS * r.WIDTH_PER_GROUP * factor, out_channels, 3,
                                  r.NUM_GROUPS, r.STRIDE_IN_1X1, first_stride=2, dilation=r.RES5_DILATION)
        self.out_channels = out_channels
        self.nhwc = self.fuse_pooler = True  # False = per-layer NCHW convolutions / full 14x14 pooling (cross-checks)

    def forward(self, x):
        """x [R, C, 14, 14] -> [R, 2048, 7, 7].  On the GPU the stage runs in NHWC with GEMM 1x1s
        (``Bottleneck.forward_nhwc``) and returns the channels_last view of the result; ``nhwc = False``
        keeps the plain per-layer convolution path (also taken for grouped / exotic configurations)."""
        if x.is_cuda and self.nhwc and all(b.nhwc_supported() for b in self.layer4):
            # the first block's stride-2 slice makes this the only NCHW -> NHWC copy
            return chain_nhwc(list(self.layer4), x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return self.layer4(x)

    def pooler_stride(self):
        """Stride the pooler may apply itself (``ROIAlign.forward_strided_nhwc``): the common stride of the first
        block's conv1 and projection shortcut when the NHWC path is on, else 0."""
        b0 = self.layer4[0]
        s = b0.conv1.stride
        ok = (self.nhwc and self.fuse_pooler and all(b.nhwc_supported() for b in self.layer4) and s[0] == s[1] and s[0] > 1
              and (b0.downsample is None or b0.downsample[0].stride == s))
        return s[0] if ok else 0

    def pooled_pair_ok(self):
        """The pooler may hand its bins over in pair layout only (no fp32 rows): the first block runs on the pair GEMM
        route and has a projection shortcut."""
        b0 = self.layer4[0]
        return bool(self.pooler_stride()) and b0.pair_gemm and b0.pair_supported() and b0._fd is not None

    def forward_pooled_nhwc(self, y, yp=None, shape=None, select=None, pooled_only=False):
        """y [R, 7, 7, C]: the pooled bins conv1 reads, NHWC (from ``forward_strided_nhwc``) -> [R, 2048, 7, 7] view;
        or y None and yp the same bins in pair layout with shape = (R, 7, 7) (from ``roi_align_forward_strided_pair``).
        ``pooled_only``: the caller reads nothing but the pooled [R, 2048] rows (``_ovis_pooled``) of a no-grad pass -- the
        last block may then skip writing its [R*49, 2048] result (the returned view is a NaN placeholder)."""
        blocks = list(self.layer4)
        if y is None:  # bins in pair layout only: the first block runs on them directly
            y, yp = blocks[0].forward_pair_rows(yp, shape[0], shape[1], shape[2], want_pair=True,
                                                pair_only=blocks[1].takes_pair_only_input())
            y = chain_nhwc(blocks[1:], y, yp, last={"pool": True, "select": select, "pool_only_ok": pooled_only})
        else:
            y = chain_nhwc(blocks, y, yp, first={"prestrided": True},
                           last={"pool": True, "select": select, "pool_only_ok": pooled_only})
        out = y.permute(0, 3, 1, 2)
        for attr in ("_ovis_pooled", "_ovis_selected"):  # outputs of the last block's autograd node, carried on its result
            v = getattr(y, attr, None)
            if v is not None:
                setattr(out, attr, v)
        return out


def _trainable_pair_blocks(model):
    """[(module, [(weight, scale) ...])] of the modules that offer ``prep_plan_convs()``: trainable bottlenecks on the pair
    GEMM (three or four convolutions) and lone trainable convolutions on it (the RPN head's 3x3)."""
    out = []
    for m in model.modules():
        offer = getattr(m, "prep_plan_convs", None)
        convs = offer() if offer is not None else None
        if convs:
            out.append((m, convs))
    return out


def prepare_weights_ahead(model):
    """After an optimizer step: the pair-layout operands of every trainable bottleneck of ``model`` in ONE launch
    (layers/pair_bottleneck.py::WeightPrepPlan) -- the next forward finds them instead of preparing 3-4 weights per block in
    front of its GEMMs.  The plan (buffers + device tables) is built at the first call and rebuilt when a weight's storage or
    a FrozenBN fold moved; a block whose weights were touched afterwards by anything else prepares them itself.  No-op on
    hosts / without such blocks.  Returns the number of blocks served."""
    from ..layers.pair_bottleneck import WeightPrepPlan
    plan = model.__dict__.get("_ovis_prep_plan")
    if isinstance(plan, torch.device):  # nothing to prepare when last looked: look again only when the model has moved
        first = next(model.parameters(), None)
        if first is None or first.device == plan:
            return 0
        plan = None
    if plan is None or not plan.describes():
        blocks = _trainable_pair_blocks(model)  # (module walk + fold(): not per step)
        if not blocks:
            first = next(model.parameters(), None)
            model.__dict__["_ovis_prep_plan"] = first.device if first is not None else torch.device("cpu")
            return 0
        plan = WeightPrepPlan([(id(m), convs) for m, convs in blocks])
        model.__dict__["_ovis_prep_plan"] = plan
        for m, _ in blocks:
            m.__dict__["_prep_plan"] = plan
    plan.run()
    return len(plan.entries)
