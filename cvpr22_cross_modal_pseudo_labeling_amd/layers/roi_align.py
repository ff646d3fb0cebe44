"""``ROIAlign`` / ``roi_align`` -- same API as maskrcnn_benchmark/layers/roi_align.py:12-69."""
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .. import _C
from ._fp32 import float_function


class _ROIAlign(Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio, matrix_core=False):
        ctx.save_for_backward(roi)
        ctx.output_size = _pair(output_size)
        ctx.spatial_scale = spatial_scale
        ctx.sampling_ratio = sampling_ratio
        ctx.input_shape = input.size()
        fwd = _C.roi_align_forward_mfma if matrix_core else _C.roi_align_forward
        return fwd(input, roi, spatial_scale, ctx.output_size[0], ctx.output_size[1], sampling_ratio)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        bs, ch, h, w = ctx.input_shape
        grad_input = _C.roi_align_backward(grad_output, rois, ctx.spatial_scale, ctx.output_size[0],
                                           ctx.output_size[1], bs, ch, h, w, ctx.sampling_ratio)
        return grad_input, None, None, None, None, None


roi_align = _ROIAlign.apply


class _ROIAlignStridedNHWC(Function):
    """RoIAlign fused with the stride of the layer that consumes it: only bins (s*i, s*j), NHWC output
    [R, ceil(PH/s), ceil(PW/s), C] (``_C.roi_align_forward_strided_nhwc``).  Backward: the gradient of the
    skipped bins is zero; the plane-owner kernel reads the computed bins' tiles directly (tables built for the strided
    bins, ``_C.roi_align_backward_strided``) -- for maps it does not cover the tile is scattered into a zero
    [R, C, PH, PW] one and handed to ``roi_align_backward``."""

    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio, bin_stride):
        ctx.save_for_backward(roi)
        ctx.output_size = _pair(output_size)
        ctx.spatial_scale = spatial_scale
        ctx.sampling_ratio = sampling_ratio
        ctx.bin_stride = bin_stride
        ctx.input_shape = input.size()
        return _C.roi_align_forward_strided_nhwc(input, roi, spatial_scale, ctx.output_size[0], ctx.output_size[1],
                                                 sampling_ratio, bin_stride)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        bs, ch, h, w = ctx.input_shape
        ph, pw = ctx.output_size
        s = ctx.bin_stride
        # the NHWC gradient as it is (tiles up to 8 x 8): the library re-lays it into pre-split per-(RoI, channel) tiles
        grad_input = _C.roi_align_backward_strided_nhwc(grad_output.contiguous(), rois, ctx.spatial_scale, ph, pw, bs, ch, h, w,
                                                        ctx.sampling_ratio, s) if grad_output.is_cuda else None
        if grad_input is not None:
            return grad_input, None, None, None, None, None
        # [R, oh, ow, C] -> [R, C, oh, ow] tiles of the computed bins only: a quarter of the bytes of the full tile
        grad_input = _C.roi_align_backward_strided(grad_output.permute(0, 3, 1, 2).contiguous(), rois, ctx.spatial_scale, ph, pw,
                                                   bs, ch, h, w, ctx.sampling_ratio, s)
        if grad_input is not None:
            return grad_input, None, None, None, None, None
        full = grad_output.new_zeros((grad_output.shape[0], ch, ph, pw))
        full[:, :, ::s, ::s] = grad_output.permute(0, 3, 1, 2)
        grad_input = _C.roi_align_backward(full, rois, ctx.spatial_scale, ph, pw, bs, ch, h, w, ctx.sampling_ratio)
        return grad_input, None, None, None, None, None


def roi_align_strided_nhwc(input, rois, output_size, spatial_scale, sampling_ratio, bin_stride):
    return _ROIAlignStridedNHWC.apply(input, rois, output_size, spatial_scale, sampling_ratio, bin_stride)


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio, matrix_core=False):
        """``matrix_core`` (extension, default off = the reference's bit-exact forward): pool on the bf16 matrix
        pipe (``_C.roi_align_forward_mfma``); the backward is the matrix-core plane-owner kernel either way."""
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio
        self.matrix_core = matrix_core

    @float_function
    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.matrix_core)

    @float_function
    def forward_strided_nhwc(self, input, rois, bin_stride):
        """Extension for a consumer that reads every ``bin_stride``-th bin in NHWC (the res5 head)."""
        return roi_align_strided_nhwc(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio,
                                      bin_stride)

    def __repr__(self):
        return (f"{self.__class__.__name__}(output_size={self.output_size}, "
                f"spatial_scale={self.spatial_scale}, sampling_ratio={self.sampling_ratio})")
