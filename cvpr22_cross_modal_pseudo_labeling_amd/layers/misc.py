"""Empty-batch-safe wrappers -- maskrcnn_benchmark/layers/misc.py:18-111.

Current PyTorch handles zero-element batches natively for these modules (the reference's
``_NewEmptyTensorOp`` shape arithmetic dates from torch 1.x), so the wrappers keep the
reference's names and only special-case what still needs it."""
import torch
from torch import nn


class Conv2d(nn.Conv2d):
    pass


class ConvTranspose2d(nn.ConvTranspose2d):
    pass


class BatchNorm2d(nn.BatchNorm2d):
    def forward(self, x):
        if x.numel() > 0:
            return super().forward(x)
        return x.new_empty(x.shape)


def interpolate(input, size=None, scale_factor=None, mode="nearest", align_corners=None):
    return torch.nn.functional.interpolate(input, size, scale_factor, mode, align_corners)


class DFConv2d(nn.Module):
    """Deformable convolutional layer: an ordinary conv predicts the offsets (+ sigmoid masks for v2) that drive
    ``DeformConv`` / ``ModulatedDeformConv`` (maskrcnn_benchmark/layers/misc.py:114-203)."""

    def __init__(self, in_channels, out_channels, with_modulated_dcn=True, kernel_size=3, stride=1, groups=1,
                 dilation=1, deformable_groups=1, bias=False):
        super().__init__()
        from .dcn import DeformConv, ModulatedDeformConv

        if isinstance(kernel_size, (list, tuple)):
            assert isinstance(stride, (list, tuple)) and isinstance(dilation, (list, tuple))
            assert len(kernel_size) == 2 and len(stride) == 2 and len(dilation) == 2
            padding = (dilation[0] * (kernel_size[0] - 1) // 2, dilation[1] * (kernel_size[1] - 1) // 2)
            base = kernel_size[0] * kernel_size[1]
        else:
            padding = dilation * (kernel_size - 1) // 2
            base = kernel_size * kernel_size
        self.offset_base_channels = base
        self.deformable_groups = deformable_groups
        offset_channels = base * (3 if with_modulated_dcn else 2)
        conv_block = ModulatedDeformConv if with_modulated_dcn else DeformConv
        self.offset = Conv2d(in_channels, deformable_groups * offset_channels, kernel_size=kernel_size, stride=stride,
                             padding=padding, groups=1, dilation=dilation)
        nn.init.kaiming_uniform_(self.offset.weight, a=1)
        nn.init.constant_(self.offset.bias, 0.0)
        self.conv = conv_block(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                               dilation=dilation, groups=groups, deformable_groups=deformable_groups, bias=bias)
        self.with_modulated_dcn = with_modulated_dcn
        self.kernel_size, self.stride, self.padding, self.dilation = kernel_size, stride, padding, dilation

    def forward(self, x):
        if x.numel() == 0:
            k, s, p, d = (v if isinstance(v, (list, tuple)) else (v, v)
                          for v in (self.kernel_size, self.stride, self.padding, self.dilation))
            hw = [(i + 2 * pp - (dd * (kk - 1) + 1)) // ss + 1 for i, pp, dd, kk, ss in zip(x.shape[-2:], p, d, k, s)]
            return x.new_empty([x.shape[0], self.conv.weight.shape[0]] + hw)
        if not self.with_modulated_dcn:
            return self.conv(x, self.offset(x))
        offset_mask = self.offset(x)
        # the reference hard-codes 18 / 9 (3x3 kernels, ONE deformable group; with more groups its slices no longer match
        # the channels DeformConv expects): here the split follows the groups -- identical for deformable_groups = 1
        n = self.offset_base_channels * self.deformable_groups
        return self.conv(x, offset_mask[:, : 2 * n], offset_mask[:, -n:].sigmoid())
