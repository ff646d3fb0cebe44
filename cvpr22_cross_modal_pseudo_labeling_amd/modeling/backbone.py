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

from ..layers import Conv2d, FrozenBatchNorm2d


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
        if (frozen and self._cache is not None and self._cache[0] is w and self._cache[1] == w._version
                and self._cache[2].device == w.device):
            return self._cache[2], self._cache[3]
        scale, shift = self.bn.fold()
        fw = w * scale.reshape(-1, 1, 1, 1)
        if frozen:
            self._cache = (w, w._version, fw.detach(), shift.detach())
        return fw, shift

    def forward(self, x):
        w, b = self.folded()
        c = self.conv
        return F.conv2d(x, w, b, c.stride, c.padding, c.dilation, c.groups)


class Bottleneck(nn.Module):
    def __init__(self, in_channels, bottleneck_channels, out_channels, num_groups=1, stride_in_1x1=True,
                 stride=1, dilation=1):
        super().__init__()
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
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=stride_3x3,
                            padding=dilation, bias=False, groups=num_groups, dilation=dilation)
        self.bn2 = FrozenBatchNorm2d(bottleneck_channels)
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False)
        self.bn3 = FrozenBatchNorm2d(out_channels)
        for l in (self.conv1, self.conv2, self.conv3):
            nn.init.kaiming_uniform_(l.weight, a=1)
        # fused views over the same parameters (no extra state_dict entries)
        self._f1 = [ConvBN(self.conv1, self.bn1)]
        self._f2 = [ConvBN(self.conv2, self.bn2)]
        self._f3 = [ConvBN(self.conv3, self.bn3)]
        self._fd = [ConvBN(self.downsample[0], self.downsample[1])] if self.downsample is not None else None

    def forward(self, x):
        out = F.relu_(self._f1[0](x))
        out = F.relu_(self._f2[0](out))
        out = self._f3[0](out)
        identity = self._fd[0](x) if self._fd is not None else x
        out += identity
        return F.relu_(out)


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


def _make_stage(in_channels, bottleneck_channels, out_channels, block_count, num_groups, stride_in_1x1,
                first_stride, dilation=1):
    blocks, stride = [], first_stride
    for _ in range(block_count):
        blocks.append(Bottleneck(in_channels, bottleneck_channels, out_channels, num_groups, stride_in_1x1,
                                 stride, dilation))
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
        if any(r.STAGE_WITH_DCN):
            raise NotImplementedError("STAGE_WITH_DCN is not wired into the trunk in this build")
        self.stem = Stem(r.STEM_OUT_CHANNELS)
        in_channels = r.STEM_OUT_CHANNELS
        width = r.NUM_GROUPS * r.WIDTH_PER_GROUP
        self.stages = []
        for index, count in _C4_STAGES:
            factor = 2 ** (index - 1)
            out_channels = r.RES2_OUT_CHANNELS * factor
            stage = _make_stage(in_channels, width * factor, out_channels, count, r.NUM_GROUPS,
                                r.STRIDE_IN_1X1, first_stride=int(index > 1) + 1)
            name = f"layer{index}"
            self.add_module(name, stage)
            self.stages.append(name)
            in_channels = out_channels
        self.out_channels = in_channels
        self._freeze(cfg.MODEL.BACKBONE.FREEZE_CONV_BODY_AT)

    def _freeze(self, freeze_at):
        for i in range(max(freeze_at, 0)):
            m = self.stem if i == 0 else getattr(self, f"layer{i}")
            for p in m.parameters():
                p.requires_grad = False

    def forward(self, x):
        x = self.stem(x)
        for name in self.stages:
            x = getattr(self, name)(x)
        return [x]


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

    def forward(self, x):
        return self.layer4(x)
