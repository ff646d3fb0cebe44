"""Deformable convolution / RoI pooling layers (maskrcnn_benchmark/layers/dcn/*)."""
from .deform_conv_func import deform_conv, modulated_deform_conv  # noqa: F401
from .deform_conv_module import DeformConv, ModulatedDeformConv, ModulatedDeformConvPack  # noqa: F401
from .deform_pool_func import deform_roi_pooling  # noqa: F401
from .deform_pool_module import DeformRoIPooling, DeformRoIPoolingPack, ModulatedDeformRoIPoolingPack  # noqa: F401
