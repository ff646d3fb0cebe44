"""Training loop of the hot path: engine/trainer.py:54-267 (``do_train``) and the model / optimizer /
DDP set-up of tools/train_net.py:43-126, rebuilt around ``BucketedGradReducer``.

Only the step itself is here (forward -> sum of losses -> backward with overlapped all-reduce -> SGD ->
LR schedule) plus rank-0 logging at LOG_PERIOD; checkpoint / evaluation cadence is outside the
hot-path scope.
"""
import logging
import time

import torch

from . import comm


def train_step(model, optimizer, reducer, images, targets, scheduler=None):
    """One optimisation step; returns the (un-reduced) loss dict of this rank."""
    reducer.zero_grad()
    loss_dict = model(images, targets)
    losses = sum(loss for loss in loss_dict.values())
    losses.backward()
    reducer.finish()
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return loss_dict


def do_train(cfg, model, data_iter, optimizer, scheduler, max_iter, start_iter=0, log_period=None, logger=None):
    logger = logger or logging.getLogger("ovis.trainer")
    log_period = log_period or cfg.SOLVER.LOG_PERIOD
    model.train()
    reducer = comm.BucketedGradReducer(model)
    start = time.time()
    last = start
    history = []
    for iteration in range(start_iter, max_iter):
        images, targets = next(data_iter)
        if any(len(t) < 1 for t in targets):  # trainer.py:96-98
            logger.error("iteration %d skipped: an image has no targets", iteration + 1)
            continue
        loss_dict = train_step(model, optimizer, reducer, images, targets, scheduler)
        if (iteration + 1) % log_period == 0 or iteration + 1 == max_iter:
            reduced = comm.reduce_loss_dict(loss_dict)  # the only host sync of the loop
            if comm.get_rank() == 0:
                vals = {k: float(v) for k, v in reduced.items()}
                now = time.time()
                history.append((iteration + 1, vals))
                logger.info("iter %d  loss %.4f  %s  lr %.6f  %.3f s/it", iteration + 1, sum(vals.values()),
                            "  ".join(f"{k} {v:.4f}" for k, v in vals.items()), optimizer.param_groups[0]["lr"],
                            (now - last) / log_period)
                last = now
    reducer.remove()
    total = time.time() - start
    logger.info("Total training time: %.1f s (%.4f s / it)", total, total / max(max_iter - start_iter, 1))
    return history
