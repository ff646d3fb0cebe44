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
    optimizer checkpoints interchange) whose ``step`` updates ALL groups together:  d = g + wd_i p;  buf = momentum_i buf + d
    (buf = d on the first step);  p -= lr_i buf.
    Device fp32 parameters: ONE native launch per distinct (lr, wd, momentum) -- weights / biases / ``uncertain_pred``: two
    or three per step -- of ``csrc/optim.hip`` (p, g, buf read, buf, p written once; the table of tensor pointers is built
    once and rebuilt only when a pointer changes; gradients are stable views of the reducer's flat buckets).  Otherwise
    (host tensors, other dtypes) six multi-tensor ops with per-tensor scalars -- the same operation sequence, so the two
    paths give the same bits.  Falls back to the stock step for closures, dampening, nesterov, maximize or sparse
    gradients."""

    native = True  # False: always the multi-tensor form (tests compare the two)

    @torch.no_grad()
    def step(self, closure=None):
        from ..layers.pair_bottleneck import note_weights_written
        note_weights_written()  # the native launch writes parameters through raw pointers: no version counter moves
        groups = self.param_groups
        if closure is None and self.native and self._step_cached(groups):
            return None
        if closure is not None or any(g["dampening"] != 0 or g["nesterov"] or g.get("maximize", False) for g in groups):
            return super().step(closure)
        params, grads, wds, lrs, moms = [], [], [], [], []
        for g in groups:
            for p in g["params"]:
                if p.grad is None:  # (torch.optim.SGD's rule; a parameter frozen during the run keeps a zero gradient -- comm.py)
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
        if self.native and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and q.is_contiguous() and q.dtype == torch.float32
               for p, q in zip(params, grads)):
            return self._step_native(params, grads, wds, lrs, moms)
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

    def _step_native(self, params, grads, wds, lrs, moms):
        import numpy as np

        from .. import _C

        bufs = []
        for p, m in zip(params, moms):
            if m == 0:
                bufs.append(None)
                continue
            st = self.state[p]
            if st.get("momentum_buffer") is None:
                st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)  # 0 * m + d = d: the first step
            bufs.append(st["momentum_buffer"])
        key = tuple((p.data_ptr(), q.data_ptr(), 0 if b is None else b.data_ptr(), lr, wd, m)
                    for p, q, b, lr, wd, m in zip(params, grads, bufs, lrs, wds, moms))
        cache = self.__dict__.setdefault("_native_tables", {})
        sig = tuple(k[:3] for k in key)
        part = tuple((lr, wd, m) for _, _, _, lr, wd, m in key)
        hit = cache.get("tables")
        if hit is None or hit[0] != sig or hit[1] != self._partition_of(part):
            chunk = _C.sgd_chunk_elements()
            by = {}
            for i, t in enumerate(part):
                by.setdefault(t, []).append(i)
            tables = []
            dev = params[0].device
            for t, idx in by.items():
                items = np.zeros((len(idx), 4), dtype=np.int64)
                blocks = []
                for j, i in enumerate(idx):
                    items[j] = (key[i][0], key[i][1], key[i][2], params[i].numel())
                    blocks.extend((j, c) for c in range((params[i].numel() + chunk - 1) // chunk))
                tables.append((idx[0], torch.from_numpy(items.view(np.uint8).reshape(-1)).to(dev),
                               torch.tensor(blocks, dtype=torch.int32, device=dev).reshape(-1, 2)))
            hit = (sig, self._partition_of(part), tables)
            cache["tables"] = hit
        use_wd = any(wds)
        for first, items, blocks in hit[2]:
            _C.sgd_momentum_multi(items, blocks, lrs[first], wds[first], moms[first], use_wd)
        # what the next steps check instead of rebuilding all of the above: every parameter's gradient pointer (None = no
        # gradient) and, per launch, the group whose scalars it reads and the groups that must agree with it
        index = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g["params"]}
        classes = {}
        for i, p in enumerate(params):
            classes.setdefault((lrs[i], wds[i], moms[i]), []).append(index[id(p)])
        launches = [(classes[(lrs[first], wds[first], moms[first])], items, blocks, moms[first] != 0) for first, items, blocks in hit[2]]
        # ... and every pointer the device tables hold: the parameter's own storage and its momentum buffer (an
        # ``optimizer.load_state_dict``, a ``state.clear()`` or a ``p.data`` swap replaces them behind the cache's back)
        used = {id(p): b for p, b in zip(params, bufs)}
        every = []
        for g in self.param_groups:
            for p in g["params"]:
                live = p.grad is not None
                b = used.get(id(p)) if live else None
                every.append((p, p.data_ptr() if live else 0, p.grad.data_ptr() if live else 0, b, 0 if b is None else b.data_ptr()))
        cache["fast"] = (len(self.param_groups), every, launches)
        return None

    def _step_cached(self, groups):
        """The step of a run in steady state -- same parameters, same gradient buffers, same partition into (lr, wd, momentum)
        classes as the last full step -- without the per-parameter list building of the general path (~0.5 ms of host time
        per step for the teacher's 110 parameters, spent at the step boundary where the GPU has nothing queued).  Returns
        False when anything differs; the general path then runs and refreshes the cache."""
        cache = self.__dict__.get("_native_tables")
        fast = cache.get("fast") if cache else None
        if fast is None or fast[0] != len(groups):
            return False
        _, every, launches = fast
        state = self.state
        for p, pptr, gptr, buf, bptr in every:
            g = p.grad
            if g is None:
                if gptr != 0:
                    return False
                continue
            if g.data_ptr() != gptr or p.data_ptr() != pptr:
                return False
            if buf is not None and (state[p].get("momentum_buffer") is not buf or buf.data_ptr() != bptr):
                return False
        scalars = []
        for members, _, _, has_buf in launches:
            g0 = groups[members[0]]
            lr, wd, mom = g0["lr"], g0["weight_decay"], g0["momentum"]
            if (mom != 0) != has_buf:
                return False  # momentum switched on / off: the tables hold no (or stale) buffer pointers
            for gi in members:
                g = groups[gi]
                if g["lr"] != lr or g["weight_decay"] != wd or g["momentum"] != mom or g["dampening"] != 0 or g["nesterov"] \
                        or g.get("maximize", False):
                    return False
            scalars.append((float(lr), float(wd), float(mom)))
        if len(set(scalars)) != len(scalars):
            return False  # two classes have met (e.g. a rate of zero): let the general path re-partition
        from .. import _C
        use_wd = any(wd != 0 for _, wd, _ in scalars)  # from the CURRENT scalars: a decay switched on later must apply
        for (lr, wd, mom), (_, items, blocks, _) in zip(scalars, launches):
            _C.sgd_momentum_multi(items, blocks, lr, wd, mom, use_wd)
        return True

    def _drop_native_tables(self):
        self.__dict__.pop("_native_tables", None)

    def load_state_dict(self, state_dict):
        """A resumed run gets NEW momentum buffers (and possibly new scalars): the cached device tables point at the old ones."""
        self._drop_native_tables()
        return super().load_state_dict(state_dict)

    def __setstate__(self, state):
        super().__setstate__(state)
        self._drop_native_tables()

    def add_param_group(self, param_group):
        self._drop_native_tables()
        return super().add_param_group(param_group)

    @staticmethod
    def _partition_of(part):
        """Which tensors share their scalars (the VALUES change with the schedule, the partition does not)."""
        seen = {}
        return tuple(seen.setdefault(t, len(seen)) for t in part)


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

    def step(self, epoch=None):
        """The base class's step without its bookkeeping (closed-form detection, per-group tensor checks): this runs once per
        iteration at the step boundary, over one group per parameter."""
        if epoch is not None:
            return super().step(epoch)
        self._step_count += 1
        self.last_epoch += 1
        values = self.get_lr()
        for group, lr in zip(self.optimizer.param_groups, values):
            group["lr"] = lr
        self._last_lr = values

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
