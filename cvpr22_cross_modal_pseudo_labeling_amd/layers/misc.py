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
