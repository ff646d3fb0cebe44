"""Deformable convolution layers (maskrcnn_benchmark/layers/dcn/*)."""
from .deform_conv_func import deform_conv, modulated_deform_conv  # noqa: F401
from .deform_conv_module import DeformConv, ModulatedDeformConv, ModulatedDeformConvPack  # noqa: F401
