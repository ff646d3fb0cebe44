"""``smooth_l1_loss`` with the extra ``beta`` -- maskrcnn_benchmark/layers/smooth_l1_loss.py:6-16."""
import torch


def smooth_l1_loss(input, target, beta=1.0 / 9, size_average=True):
    n = torch.abs(input - target)
    loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    return loss.mean() if size_average else loss.sum()
