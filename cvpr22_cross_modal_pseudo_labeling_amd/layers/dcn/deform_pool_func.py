"""Autograd function of deformable position-sensitive RoI pooling (maskrcnn_benchmark/layers/dcn/deform_pool_func.py:8-95)
on the HIP kernels of csrc/deform_pool.hip."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ... import _C


class DeformRoIPoolingFunction(Function):
    @staticmethod
    def forward(ctx, data, rois, offset, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                sample_per_part=4, trans_std=.0):
        if not 0.0 <= trans_std <= 1.0:
            raise AssertionError("trans_std must lie in [0, 1]")
        if not data.is_cuda:
            raise NotImplementedError("deform_roi_pooling has no CPU implementation (as upstream)")
        part_size = out_size if part_size is None else part_size
        ctx.cfg = (bool(no_trans), float(spatial_scale), int(out_channels), int(group_size), int(out_size), int(part_size),
                   int(sample_per_part), float(trans_std))
        n = rois.shape[0]
        output = data.new_empty(n, out_channels, out_size, out_size)
        count = data.new_empty(n, out_channels, out_size, out_size)
        _C.deform_psroi_pooling_forward(data, rois, offset, output, count, *ctx.cfg)
        if data.requires_grad or rois.requires_grad or offset.requires_grad:
            ctx.save_for_backward(data, rois, offset)
        ctx.count = count
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError("deform_roi_pooling has no CPU implementation (as upstream)")
        data, rois, offset = ctx.saved_tensors
        grad_input = torch.zeros_like(data)
        grad_offset = torch.zeros_like(offset)
        _C.deform_psroi_pooling_backward(grad_output.contiguous(), data, rois, offset, ctx.count, grad_input, grad_offset,
                                         *ctx.cfg)
        return (grad_input, None, grad_offset) + (None,) * 8


deform_roi_pooling = DeformRoIPoolingFunction.apply
