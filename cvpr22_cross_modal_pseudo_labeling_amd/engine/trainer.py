"""Training loop of the hot path: engine/trainer.py:54-267 (``do_train``) and the model / optimizer /
DDP set-up of tools/train_net.py:43-126, rebuilt around ``BucketedGradReducer``.

Only the step itself is here (forward -> sum of losses -> backward with overlapped all-reduce -> SGD ->
LR schedule) plus rank-0 logging at LOG_PERIOD and the reference's checkpoint cadence (``model_<iter>.pth`` every
CHECKPOINT_PERIOD, ``model_final.pth`` at the end: trainer.py:172-173, 252-253); periodic evaluation is outside the
hot-path scope.
"""
import logging
import time

import torch

from . import comm
from .. import _C


def total_loss(loss_dict):
    """Sum of the loss terms (trainer.py:96 ``sum(loss for loss in loss_dict.values())``) as one stack + one reduction
    instead of a chain of scalar adds."""
    terms = [v if torch.is_tensor(v) else torch.as_tensor(float(v)) for v in loss_dict.values()]
    if len(terms) < 3 or not all(t.is_cuda for t in terms):
        return sum(terms)
    return torch.stack([t.reshape(()) for t in terms]).sum()


class StepPolicy:
    """SOLVER.GRADIENT_ACCUMULATION_STEPS and SOLVER.CLIP_GRAD_NORM_AT of the reference loop (engine/trainer.py:117,
    135-141): the summed loss is divided by the accumulation count; the optimizer (and the LR schedule) steps on the
    accumulated gradients, clipped to a total L2 norm first when the threshold is positive, at every iteration whose NUMBER is
    a multiple of the count -- so a run resumed at an odd iteration, or a window with a skipped batch (trainer.py:96-98: the
    batch is dropped before the check), steps where the reference's does.  Callers without iteration numbers (``iteration``
    None) step every k-th call."""

    def __init__(self, accumulation_steps=1, clip_grad_norm_at=-1.0):
        self.accumulation_steps = max(int(accumulation_steps), 1)
        self.clip_grad_norm_at = float(clip_grad_norm_at)
        self.prepare_weights_ahead = True  # False = every block prepares its own weights in its forward (A/B switch)
        self.micro = 0     # micro-steps since the last optimizer step
        self.fresh = True  # the gradient buffers hold nothing of an open window: the next micro-step starts from zero

    @classmethod
    def from_cfg(cls, cfg):
        return cls(cfg.SOLVER.GRADIENT_ACCUMULATION_STEPS, cfg.SOLVER.CLIP_GRAD_NORM_AT)

    def begin(self, reducer):
        reducer.zero_grad(set_to_zero=self.fresh)
        self.fresh = False

    def scale(self, losses):
        return losses / float(self.accumulation_steps) if self.accumulation_steps > 1 else losses

    def end(self, reducer, optimizer, scheduler, iteration=None, model=None):
        """After backward(): reduce, and -- when the window closes -- clip, step, advance the schedule.  ``iteration``: the
        1-based number of this iteration in the run (None: count calls).  ``model``: its trainable bottlenecks' GEMM operands
        are prepared right behind the optimizer step, in one launch (modeling/backbone.py::prepare_weights_ahead)."""
        reducer.finish()
        self.micro += 1
        due = self.micro >= self.accumulation_steps if iteration is None else iteration % self.accumulation_steps == 0
        if not due:
            return False
        self.micro = 0
        self.fresh = True
        if self.clip_grad_norm_at > 0:
            reducer.clip_grad_norm_(self.clip_grad_norm_at)
        optimizer.step()
        if model is not None and self.prepare_weights_ahead:
            from ..modeling.backbone import prepare_weights_ahead
            prepare_weights_ahead(model)
        if scheduler is not None:
            scheduler.step()
        return True


def train_step(model, optimizer, reducer, images, targets, scheduler=None, policy=None, iteration=None):
    """One iteration of the loop (an optimisation step unless ``policy`` accumulates); returns the (un-reduced) loss dict
    of this rank."""
    policy = policy or _default_policy(reducer)
    loss_dict = model(images, targets)
    # (the reference zeroes the gradients right behind ``optimizer.step()``, engine/trainer.py:139-141; any point in front of
    # the backward is equivalent.  Here: between forward and backward, so that the 0.4 ms of host time of re-arming the reducer
    # -- 110 parameters' slots checked -- fall where the device has the forward queued, not at the step boundary; the step
    # itself measured equal either way: tools/experiments/boundary_probe.py, ab_bench.py begin_first)
    policy.begin(reducer)
    losses = policy.scale(total_loss(loss_dict))
    losses.backward()
    policy.end(reducer, optimizer, scheduler, iteration, model)
    return loss_dict


def _default_policy(reducer):
    """One policy object per reducer when the caller passes none (plain step: no accumulation, no clipping)."""
    pol = getattr(reducer, "_step_policy", None)
    if pol is None:
        pol = reducer._step_policy = StepPolicy()
    return pol


def _record_stream(obj, stream):
    """Tell the caching allocator that every tensor reachable from ``obj`` is (also) used on ``stream``."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)
    elif hasattr(obj, "bbox") and hasattr(obj, "extra_fields"):  # BoxList
        _record_stream(obj.bbox, stream)
        _record_stream(obj.extra_fields, stream)
        _record_stream(getattr(obj, "pos_index", None), stream)  # sampled lists: where their positives sit
    elif hasattr(obj, "polygon_start"):  # PolygonMasks
        for t in (obj.coords, obj.polygon_start, obj.instance_start):
            _record_stream(t, stream)
    elif hasattr(obj, "probs") and hasattr(obj, "boxes"):  # PastedMasks
        _record_stream(obj.probs, stream)
        _record_stream(obj.boxes, stream)
    elif hasattr(obj, "tensors") and hasattr(obj, "image_sizes"):  # ImageList: the padded batch to_image_list allocates
        _record_stream(obj.tensors, stream)


_SIDE_STREAMS = {}


def side_stream(priority):
    """ONE side stream per (device, priority) and process: streams share a few hardware queues, so a process must not grow a
    new one per trainer object (data/prefetch.py::copy_stream has the measurement)."""
    key = (torch.cuda.current_device(), priority)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(priority=priority)
    return _SIDE_STREAMS[key]


def branch_stream():
    """The stream a step's independent branches run on beside the main one (the teacher's RPN branch and its backward,
    the trunk's weight gradients): the SAME side stream the look-ahead half of ``PipelinedTrainer`` uses -- that half is done
    long before a step reaches its RPN, and with the copy stream and RCCL's own a process then stays at the four hardware
    queues HIP multiplexes streams onto (a fifth stream shares the compute stream's queue: the 31-vs-23 ms lesson of round 5).
    A/B against a stream of its own: equal (profiles/r6_ab_one_side_stream.txt)."""
    return side_stream(-1)


class PipelinedTrainer:
    """Student-teacher step, software-pipelined across iterations on two HIP streams.

    Trunk, RPN and teacher heads are frozen in the student-teacher configuration, so features, proposals and pseudo
    labels of batch i+1 (``model.forward_frozen``) do not depend on the optimizer step of batch i.  That half is
    host-bound (hundreds of small kernels between host syncs: NMS counts, ``nonzero`` ...) while the student half
    is GPU-bound (res5 GEMMs, backward), and run back to back each leaves the other resource idle (14 % of the
    wall time in the un-pipelined step).  Here the frozen half of the NEXT batch is enqueued on a side stream
    right after the backward of the current batch has been enqueued on the main stream: its host syncs only wait
    for the side stream, and its small kernels fill the gaps of the main stream instead of waiting behind them.
    Results are those of the plain step (same modules, same weights); only the order in which independent work
    reaches the GPU changes.  Teacher training (``GeneralizedRCNN``) has a smaller frozen half -- stem + the leading trunk
    stages below FREEZE_CONV_BODY_AT (``ResNetC4.forward_prefix``) -- which runs ahead the same way (24.6 -> 23.1 ms per
    step: its ~1.1 ms of GEMMs fill the under-filled launches of the backward); models without ``forward_frozen`` /
    ``forward_student`` fall back to ``train_step``."""

    def __init__(self, model, optimizer, reducer, scheduler=None, threaded=True, side_priority=-1, policy=None):
        self.model, self.optimizer, self.reducer, self.scheduler = model, optimizer, reducer, scheduler
        self.policy = policy or StepPolicy()
        self.enabled = hasattr(model, "forward_frozen") and torch.cuda.is_available()
        # The look-ahead half must not read anything the optimizer writes: with MODEL.LANGUAGE_BACKBONE.FT_EMB the BERT
        # table is trained AND read by the frozen half's noun embeddings (st_generalized_rcnn.py:242) -- the side stream
        # would read it while optimizer.step() writes it.  Run such models un-pipelined.
        bert = getattr(model, "bert", None)
        if self.enabled and bert is not None and any(p.requires_grad for p in bert.parameters()):
            self.enabled = False
        # The side stream gets the HIGH queue priority: the frozen half is a chain of small kernels between host reads
        # (RPN counts, sampler counts); behind the main stream's full-machine GEMMs each of those round trips waited for
        # a GEMM to drain, and a late frozen half stalls the next student half.  Measured: 33.9 -> 33.4 ms per step.
        self.side = side_stream(side_priority) if self.enabled else None
        self.pending = None  # (key, frozen outputs, event recorded on the side stream)
        # threaded: the frozen half of the next batch is ISSUED by a worker thread concurrently with the student half
        # (both halves are host-bound between their own host syncs, which release the GIL), not after it
        self.threaded = threaded
        self.worker = None
        self.worker_error = None

    def _frozen_on_side(self, images, targets, after_event, device):
        try:
            torch.cuda.set_device(device)
            with torch.cuda.stream(self.side):
                if after_event is not None:
                    self.side.wait_event(after_event)
                with _C.co_scheduled():  # these launches run beside the student half: K cut for least total work
                    frozen = self.model.forward_frozen(images, targets)
                done = torch.cuda.Event()
                done.record(self.side)
            self.pending = ((id(images), id(targets)), frozen, done)
        except BaseException as e:  # re-raised on the training thread
            self.worker_error = e

    def _launch_frozen(self, images, targets, after_event=None, threaded=False):
        device = torch.cuda.current_device()
        if threaded:
            import threading
            self.worker = threading.Thread(target=self._frozen_on_side, args=(images, targets, after_event, device))
            self.worker.start()
        else:
            self._frozen_on_side(images, targets, after_event, device)
            self._check_worker()

    def _check_worker(self):
        if self.worker is not None:
            self.worker.join()
            self.worker = None
        if self.worker_error is not None:
            e, self.worker_error = self.worker_error, None
            raise e

    def step(self, images, targets, next_batch=None, iteration=None):
        """One optimisation step on (images, targets); ``next_batch`` = the (images, targets) of the following call
        (already resident on the device), whose frozen half is started before this call returns; ``iteration`` = the 1-based
        number of this iteration (``StepPolicy.end``)."""
        if not self.enabled:
            return train_step(self.model, self.optimizer, self.reducer, images, targets, self.scheduler, self.policy, iteration)
        main = torch.cuda.current_stream()
        inputs_ready = torch.cuda.Event()
        inputs_ready.record(main)  # uploads of this and the next batch precede this point on the main stream
        self._check_worker()
        if self.pending is None or self.pending[0] != (id(images), id(targets)):
            self._launch_frozen(images, targets, inputs_ready)  # cold start / unexpected batch: no overlap
        _, frozen, done = self.pending
        self.pending = None
        main.wait_event(done)
        _record_stream(frozen, main)
        if next_batch is not None and self.threaded:
            self._launch_frozen(next_batch[0], next_batch[1], inputs_ready, threaded=True)
        loss_dict = self.model.forward_student(frozen, targets)
        self.policy.begin(self.reducer)  # behind the forward's launches: see train_step
        losses = self.policy.scale(total_loss(loss_dict))
        losses.backward()
        self.policy.end(self.reducer, self.optimizer, self.scheduler, iteration, self.model)
        if next_batch is not None and not self.threaded:
            # the GPU now has the whole backward queued: overlap the next frozen half with it
            self._launch_frozen(next_batch[0], next_batch[1], inputs_ready)
        return loss_dict

    def drain(self, discard=False):
        """Wait for the look-ahead frozen half.  Its outputs depend on frozen weights only, so they stay valid across a
        checkpoint save and are kept for the next step unless ``discard``."""
        self._check_worker()
        if self.pending is not None:
            self.pending[2].synchronize()
            if discard:
                self.pending = None


def do_train(cfg, model, data_iter, optimizer, scheduler, max_iter, start_iter=0, log_period=None, logger=None,
             checkpointer=None, checkpoint_period=None):
    logger = logger or logging.getLogger("ovis.trainer")
    log_period = log_period or cfg.SOLVER.LOG_PERIOD
    model.train()
    reducer = comm.BucketedGradReducer(model)
    # (plain train_step for models without a frozen half; gradient accumulation / clipping as the config says)
    pipe = PipelinedTrainer(model, optimizer, reducer, scheduler, policy=StepPolicy.from_cfg(cfg))
    start = time.time()
    last = start
    history = []

    def fetch():
        try:
            return next(data_iter)
        except StopIteration:
            return None

    batch = fetch()
    completed = start_iter
    for iteration in range(start_iter, max_iter):
        if batch is None:
            break
        images, targets = batch
        nxt = fetch() if iteration + 1 < max_iter else None  # one batch of look-ahead feeds the side-stream half
        if any(len(t) < 1 for t in targets):  # trainer.py:96-98
            logger.error("iteration %d skipped: an image has no targets", iteration + 1)
            batch = nxt
            continue
        ok_next = nxt is not None and not any(len(t) < 1 for t in nxt[1])
        loss_dict = pipe.step(images, targets, nxt if ok_next else None, iteration=iteration + 1)
        completed = iteration + 1
        batch = nxt
        if (iteration + 1) % log_period == 0 or iteration + 1 == max_iter:
            reduced = comm.reduce_loss_dict(loss_dict)  # the only host sync of the loop
            if comm.get_rank() == 0:
                vals = {k: float(v) for k, v in reduced.items()}
                now = time.time()
                history.append((iteration + 1, vals))
                # (engine/trainer.py:156-170 of the reference: iteration, meters, lr, max mem in MB)
                mem = torch.cuda.max_memory_allocated() / 1024.0 / 1024.0 if torch.cuda.is_available() else 0.0
                logger.info("iter %d  loss %.4f  %s  lr %.6f  %.3f s/it  max mem: %.0f", iteration + 1, sum(vals.values()),
                            "  ".join(f"{k} {v:.4f}" for k, v in vals.items()), optimizer.param_groups[0]["lr"],
                            (now - last) / log_period, mem)
                last = now
        if checkpointer is not None and checkpoint_period and (iteration + 1) % checkpoint_period == 0:
            pipe.drain()  # the look-ahead half holds no state, but the weights must be quiescent while they are read
            checkpointer.save("model_{:07d}".format(iteration + 1), iteration=iteration + 1)
    pipe.drain(discard=True)
    reducer.remove()
    if checkpointer is not None and completed > start_iter:
        if completed < max_iter:
            logger.warning("the data stream ended after %d of %d iterations", completed, max_iter)
        checkpointer.save("model_final", iteration=completed)
    total = time.time() - start
    logger.info("Total training time: %.1f s (%.4f s / it)", total, total / max(max_iter - start_iter, 1))
    return history
