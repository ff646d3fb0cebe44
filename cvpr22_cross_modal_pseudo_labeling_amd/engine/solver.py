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
    return GroupFusedSGD(params, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM)


class GroupFusedSGD(torch.optim.SGD):
    """``torch.optim.SGD`` with the reference's one-group-per-parameter layout (same ``param_groups`` / ``state``, so
    optimizer checkpoints interchange) whose ``step`` runs the update of ALL groups as six multi-tensor ops with
    per-tensor scalars, instead of four launches per parameter:  d = g + wd_i p;  buf = momentum_i buf + d (buf = d on
    the first step);  p -= lr_i buf.  Falls back to the stock step for closures, dampening, nesterov, maximize or sparse
    gradients."""

    @torch.no_grad()
    def step(self, closure=None):
        groups = self.param_groups
        if closure is not None or any(g["dampening"] != 0 or g["nesterov"] or g.get("maximize", False) for g in groups):
            return super().step(closure)
        params, grads, wds, lrs, moms = [], [], [], [], []
        for g in groups:
            for p in g["params"]:
                if p.grad is None or not p.requires_grad:  # frozen during the run: no decay / momentum update either
                    continue
                if p.grad.is_sparse:
                    return super().step(closure)
                params.append(p)
                grads.append(p.grad)
                wds.append(float(g["weight_decay"]))
                lrs.append(float(g["lr"]))
                moms.append(float(g["momentum"]))
        if not params:
            return None
        d = torch._foreach_add(grads, torch._foreach_mul(params, wds)) if any(wds) else grads
        if any(moms):
            fresh = {i for i, p in enumerate(params) if self.state[p].get("momentum_buffer") is None}
            for i in fresh:
                self.state[params[i]]["momentum_buffer"] = torch.clone(d[i]).detach()
            old = [i for i in range(len(params)) if i not in fresh]
            if old:
                bufs = [self.state[params[i]]["momentum_buffer"] for i in old]
                torch._foreach_mul_(bufs, [moms[i] for i in old])
                torch._foreach_add_(bufs, [d[i] for i in old])
            upd = [self.state[p]["momentum_buffer"] if m != 0 else di for p, m, di in zip(params, moms, d)]
        else:
            upd = d
        torch._foreach_add_(params, torch._foreach_mul(upd, [-lr for lr in lrs]))
        return None


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
