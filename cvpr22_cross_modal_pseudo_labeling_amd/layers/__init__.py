"""Operator API of ``maskrcnn_benchmark.layers`` (maskrcnn_benchmark/layers/__init__.py:23-46)."""
from .batch_norm import FrozenBatchNorm2d
from .cross_modal import bias_relu_, linear_mfma, smooth_l1_picked, split_conv_same, split_linear, stochastic_mask_bce, text_logits, weighted_cross_entropy
from .dcn import DeformConv, ModulatedDeformConv, ModulatedDeformConvPack, deform_conv, modulated_deform_conv
from .dcn import DeformRoIPooling, DeformRoIPoolingPack, ModulatedDeformRoIPoolingPack, deform_roi_pooling
from .misc import BatchNorm2d, Conv2d, ConvTranspose2d, DFConv2d, interpolate
from .nms import nms, nms_padded
from .roi_align import ROIAlign, roi_align
from .roi_pool import ROIPool, roi_pool
from .sigmoid_focal_loss import SigmoidFocalLoss
from .smooth_l1_loss import smooth_l1_loss

__all__ = [
    "nms",
    "nms_padded",
    "roi_align",
    "ROIAlign",
    "roi_pool",
    "ROIPool",
    "smooth_l1_loss",
    "Conv2d",
    "ConvTranspose2d",
    "interpolate",
    "BatchNorm2d",
    "FrozenBatchNorm2d",
    "SigmoidFocalLoss",
    "DFConv2d",
    "deform_conv",
    "modulated_deform_conv",
    "DeformConv",
    "ModulatedDeformConv",
    "ModulatedDeformConvPack",
    "deform_roi_pooling",
    "DeformRoIPooling",
    "DeformRoIPoolingPack",
    "ModulatedDeformRoIPoolingPack",
    "linear_mfma",
    "split_linear",
    "split_conv_same",
    "bias_relu_",
    "text_logits",
    "smooth_l1_picked",
    "weighted_cross_entropy",
    "stochastic_mask_bce",
]
