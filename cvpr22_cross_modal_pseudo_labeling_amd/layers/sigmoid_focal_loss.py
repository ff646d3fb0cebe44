"""``SigmoidFocalLoss`` -- maskrcnn_benchmark/layers/sigmoid_focal_loss.py:9-74.

Like the reference, the module switches on the logits' device: device tensors run the HIP kernels, host tensors the
pure-torch formula (``sigmoid_focal_loss_cpu``, :40-50 -- the reference has no native host kernel for this op)."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C


class _SigmoidFocalLoss(Function):
    @staticmethod
    def forward(ctx, logits, targets, gamma, alpha):
        ctx.save_for_backward(logits, targets)
        ctx.num_classes = logits.shape[1]
        ctx.gamma = gamma
        ctx.alpha = alpha
        return _C.sigmoid_focalloss_forward(logits, targets, ctx.num_classes, gamma, alpha)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_loss):
        logits, targets = ctx.saved_tensors
        d_logits = _C.sigmoid_focalloss_backward(logits, targets, d_loss.contiguous(), ctx.num_classes,
                                                 ctx.gamma, ctx.alpha)
        return d_logits, None, None, None, None


sigmoid_focal_loss_cuda = _SigmoidFocalLoss.apply


def sigmoid_focal_loss_cpu(logits, targets, gamma, alpha):
    """Host tensors: -alpha (1 - p)^gamma log p for the target class, -(1 - alpha) p^gamma log(1 - p) for the other classes of
    a labelled row (targets >= 0; class c of the logits is label c + 1), elementwise [M, C]."""
    classes = torch.arange(1, logits.shape[1] + 1, dtype=targets.dtype, device=targets.device).unsqueeze(0)
    t = targets.unsqueeze(1)
    p = torch.sigmoid(logits)
    positive = (1 - p) ** gamma * torch.log(p)
    negative = p ** gamma * torch.log(1 - p)
    return -(t == classes).float() * positive * alpha - ((t != classes) * (t >= 0)).float() * negative * (1 - alpha)


class SigmoidFocalLoss(nn.Module):
    def __init__(self, gamma, alpha):
        super().__init__()
        self.gamma = gamma
        self.alpha = alpha

    def forward(self, logits, targets):
        loss_func = sigmoid_focal_loss_cuda if logits.is_cuda else sigmoid_focal_loss_cpu
        return loss_func(logits, targets, self.gamma, self.alpha).sum()

    def __repr__(self):
        return f"{self.__class__.__name__}(gamma={self.gamma}, alpha={self.alpha})"
