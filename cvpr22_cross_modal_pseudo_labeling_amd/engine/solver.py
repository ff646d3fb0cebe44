"""Optimizer / LR schedule of the training step.

maskrcnn_benchmark/solver/build.py:8-37 (per-parameter SGD groups: bias lr x BIAS_LR_FACTOR and
WEIGHT_DECAY_BIAS, ``uncertain_pred`` lr x UNCERTAINTY_LR_FACTOR) and
solver/lr_scheduler.py:10-52 (``WarmupMultiStepLR``).
"""
from bisect import bisect_right

import torch


def make_optimizer(cfg, model):
    params = []
    for key, value in model.named_parameters():
        if not value.requires_grad:
            continue
        lr = cfg.SOLVER.BASE_LR
        weight_decay = cfg.SOLVER.WEIGHT_DECAY
        if "bias" in key:
            lr = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR
            weight_decay = cfg.SOLVER.WEIGHT_DECAY_BIAS
        if "uncertain_pred" in key:
            lr = lr * cfg.SOLVER.UNCERTAINTY_LR_FACTOR
        params.append({"params": [value], "lr": lr, "weight_decay": weight_decay})
    return torch.optim.SGD(params, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM)


class WarmupMultiStepLR(torch.optim.lr_scheduler._LRScheduler):
    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=1.0 / 3, warmup_iters=500,
                 warmup_method="linear", last_epoch=-1):
        if list(milestones) != sorted(milestones):
            raise ValueError(f"Milestones should be increasing integers, got {milestones}")
        if warmup_method not in ("constant", "linear"):
            raise ValueError(f"Only 'constant' or 'linear' warmup_method accepted, got {warmup_method}")
        self.milestones, self.gamma = list(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        warmup_factor = 1
        if self.last_epoch < self.warmup_iters:
            if self.warmup_method == "constant":
                warmup_factor = self.warmup_factor
            else:
                alpha = float(self.last_epoch) / self.warmup_iters
                warmup_factor = self.warmup_factor * (1 - alpha) + alpha
        return [base_lr * warmup_factor * self.gamma ** bisect_right(self.milestones, self.last_epoch)
                for base_lr in self.base_lrs]


def make_lr_scheduler(cfg, optimizer):
    s = cfg.SOLVER
    return WarmupMultiStepLR(optimizer, s.STEPS, s.GAMMA, warmup_factor=s.WARMUP_FACTOR,
                             warmup_iters=s.WARMUP_ITERS, warmup_method=s.WARMUP_METHOD)
