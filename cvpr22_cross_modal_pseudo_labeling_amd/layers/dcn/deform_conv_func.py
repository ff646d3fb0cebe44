"""``deform_conv`` / ``modulated_deform_conv`` autograd functions -- same call signatures as
maskrcnn_benchmark/layers/dcn/deform_conv_func.py:9-147 (v1) and :150-259 (v2)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from ... import _C


def _out_size(input, weight, padding, dilation, stride):
    size = [input.size(0), weight.size(0)]
    for d in range(2):
        kernel = dilation[d] * (weight.size(d + 2) - 1) + 1
        size.append((input.size(d + 2) + 2 * padding[d] - kernel) // stride[d] + 1)
    if not all(s > 0 for s in size):
        raise ValueError("convolution input is too small (output would be {})".format("x".join(map(str, size))))
    return tuple(size)


class DeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError(f"Expected 4D tensor as input, got {input.dim()}D tensor instead.")
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.groups, ctx.deformable_groups, ctx.im2col_step = groups, deformable_groups, im2col_step
        ctx.save_for_backward(input, offset, weight)
        output = input.new_empty(_out_size(input, weight, ctx.padding, ctx.dilation, ctx.stride))
        ctx.bufs_ = [input.new_empty(0), input.new_empty(0)]
        step = min(im2col_step, input.shape[0])
        assert input.shape[0] % step == 0, "im2col step must divide batchsize"
        _C.deform_conv_forward(input, weight, offset, output, ctx.bufs_[0], ctx.bufs_[1], weight.size(3), weight.size(2),
                               ctx.stride[1], ctx.stride[0], ctx.padding[1], ctx.padding[0], ctx.dilation[1],
                               ctx.dilation[0], groups, deformable_groups, step)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, weight = ctx.saved_tensors
        grad_input = grad_offset = grad_weight = None
        grad_output = grad_output.contiguous()
        step = min(ctx.im2col_step, input.shape[0])
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            grad_input, grad_offset = torch.zeros_like(input), torch.zeros_like(offset)
            _C.deform_conv_backward_input(input, offset, grad_output, grad_input, grad_offset, weight, ctx.bufs_[0],
                                          weight.size(3), weight.size(2), ctx.stride[1], ctx.stride[0], ctx.padding[1],
                                          ctx.padding[0], ctx.dilation[1], ctx.dilation[0], ctx.groups,
                                          ctx.deformable_groups, step)
        if ctx.needs_input_grad[2]:
            grad_weight = torch.zeros_like(weight)
            _C.deform_conv_backward_parameters(input, offset, grad_output, grad_weight, ctx.bufs_[0], ctx.bufs_[1],
                                               weight.size(3), weight.size(2), ctx.stride[1], ctx.stride[0],
                                               ctx.padding[1], ctx.padding[0], ctx.dilation[1], ctx.dilation[0],
                                               ctx.groups, ctx.deformable_groups, 1, step)
        return grad_input, grad_offset, grad_weight, None, None, None, None, None, None


class ModulatedDeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1):
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.groups, ctx.deformable_groups = groups, deformable_groups
        ctx.with_bias = bias is not None
        if not ctx.with_bias:
            bias = input.new_empty(1)
        ctx.save_for_backward(input, offset, mask, weight, bias)
        output = input.new_empty(_out_size(input, weight, ctx.padding, ctx.dilation, ctx.stride))
        ctx._bufs = [input.new_empty(0), input.new_empty(0)]
        _C.modulated_deform_conv_forward(input, weight, bias, ctx._bufs[0], offset, mask, output, ctx._bufs[1],
                                         weight.shape[2], weight.shape[3], ctx.stride[0], ctx.stride[1], ctx.padding[0],
                                         ctx.padding[1], ctx.dilation[0], ctx.dilation[1], groups, deformable_groups,
                                         ctx.with_bias)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask, weight, bias = ctx.saved_tensors
        grad_input, grad_offset, grad_mask = torch.zeros_like(input), torch.zeros_like(offset), torch.zeros_like(mask)
        grad_weight, grad_bias = torch.zeros_like(weight), torch.zeros_like(bias)
        _C.modulated_deform_conv_backward(input, weight, bias, ctx._bufs[0], offset, mask, ctx._bufs[1], grad_input,
                                          grad_weight, grad_bias, grad_offset, grad_mask, grad_output.contiguous(),
                                          weight.shape[2], weight.shape[3], ctx.stride[0], ctx.stride[1], ctx.padding[0],
                                          ctx.padding[1], ctx.dilation[0], ctx.dilation[1], ctx.groups,
                                          ctx.deformable_groups, ctx.with_bias)
        if not ctx.with_bias:
            grad_bias = None
        return grad_input, grad_offset, grad_mask, grad_weight, grad_bias, None, None, None, None, None


deform_conv = DeformConvFunction.apply
modulated_deform_conv = ModulatedDeformConvFunction.apply
