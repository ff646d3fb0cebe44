"""Data-parallel gradient exchange for one process per GPU (RCCL over xGMI; gloo on CPU in tests).

The reference wraps the model in ``DistributedDataParallel(find_unused_parameters=True)``
(tools/train_net.py:66-71) and reduces the loss dict to rank 0 every iteration
(engine/trainer.py:19-41).  Here:

* the trainable set is static: every trainable parameter's ``.grad`` is a view into one of a few flat
  fp32 bucket buffers (reverse parameter order ~ backward order), so there is no per-step graph walk
  and no gradient copy-in/copy-out;
* a bucket's all-reduce is issued from an autograd post-accumulate hook the moment its last gradient is
  written, i.e. it overlaps with the rest of backward; ``finish()`` waits (the average is taken
  inside the collective on RCCL);
* bucket size defaults to 32 MiB: xGMI is point-to-point (7 links/GPU), ring collectives are per-link
  bound, so a few large messages beat many small ones (student 75 MB -> 3 buckets, teacher 140 MB -> 5).
"""
import torch
import torch.distributed as dist


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


def broadcast_parameters(model, src=0):
    """Initial parameter + buffer broadcast (the DDP constructor's job in the reference).  The tensors themselves are
    written (not ``t.data``, which carries its own version counter), so every cache keyed on ``_version`` -- folded
    FrozenBN weights, pair-layout weight forms -- sees the new values."""
    if get_world_size() == 1:
        return
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t, src)


def reduce_loss_dict(loss_dict):
    """All ranks' losses averaged onto rank 0 in ONE small collective (trainer.py:19-41)."""
    world = get_world_size()
    if world < 2:
        return {k: v.detach() for k, v in loss_dict.items()}
    with torch.no_grad():
        names = sorted(loss_dict.keys())
        stacked = torch.stack([loss_dict[k].detach().reshape(()) for k in names], dim=0)
        dist.reduce(stacked, dst=0)
        if dist.get_rank() == 0:
            stacked /= world
        return dict(zip(names, stacked))


class BucketedGradReducer:
    """``never_used``: parameters that are trainable but take part in no loss (the reference's ``lambda_exemplar``,
    st_generalized_rcnn.py:50): their hooks never fire, so they are left out of the per-bucket hook count (their slots
    of the flat buffer stay zero) -- otherwise their bucket, and every bucket behind it, would only be reduced in
    ``finish()``, after the backward.  Parameters whose ``requires_grad`` is turned off during the run (the student's
    ``uncertain_pred`` at MODEL.UNCERTAINTY_TRAIN_ITER) are dropped from the count by the next ``zero_grad()`` and keep a
    ZERO gradient: the optimizer still applies weight decay and momentum to them, as the reference's loop does under its
    pinned torch 1.7.1, whose ``zero_grad()`` zeroes gradients in place (pinned by tests/golden/step_student_freeze.npz)."""

    def __init__(self, model, bucket_bytes=32 << 20, never_used=None):
        self.world = get_world_size()
        # RCCL averages inside the collective (ReduceOp.AVG): no pass over the buckets after the wait.  gloo (CPU tests)
        # has no AVG: there the sum is divided in finish().
        self._avg_in_collective = self.world > 1 and dist.get_backend() == "nccl"
        params = [p for p in model.parameters() if p.requires_grad]
        params.reverse()
        if never_used is None and hasattr(model, "never_used_parameters"):
            never_used = model.never_used_parameters()
        self._never_used = {id(p) for p in (never_used or ())}
        self.hook_launches = 0  # buckets whose all-reduce was issued from a backward hook in the current step
        # measure_wait: record, per step, how long finish() stalls on the collectives -- the EXPOSED part of the exchange
        # (what backward did not hide).  Device buckets: an event pair on the current stream around the waits (RCCL's
        # ``wait()`` only makes the stream wait, the host runs on); host buckets (gloo): wall time of the waits.
        self.measure_wait = False
        self._wait_records = []  # (start_event, end_event) or seconds
        self.buckets = []
        cur, cur_bytes = [], 0
        for p in params:
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        self.flat, self.pending, self.handles, self.launched = [], [], [], []
        self._hooks = []
        self._next = 0  # first bucket whose all-reduce has not been issued yet
        self._hook_streams = []  # per bucket: the streams its gradients were accumulated on this step (device runs, world > 1)
        for bi, bucket in enumerate(self.buckets):
            flat = torch.zeros(sum(p.numel() for p in bucket), dtype=bucket[0].dtype, device=bucket[0].device)
            off = 0
            for p in bucket:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
            self.flat.append(flat)
            self._hook_streams.append(set())
            self.pending.append(self._expected(bucket))
            self.handles.append(None)
            self.launched.append(False)

    def _expected(self, bucket):
        return sum(1 for p in bucket if p.requires_grad and id(p) not in self._never_used)

    def _make_hook(self, bi):
        def hook(param):
            if id(param) in self._never_used:
                if self.launched[bi]:
                    raise RuntimeError("BucketedGradReducer: a parameter declared never_used received a gradient after "
                                       "its bucket was reduced")
                return
            self.pending[bi] -= 1
            if self.world > 1 and param.is_cuda:
                # autograd accumulates a gradient on the stream its parameter was first used on in the forward (the teacher
                # step's RPN head runs on the side stream): remember it, the bucket's collective must be ordered behind it
                self._hook_streams[bi].add(torch.cuda.current_stream())
            before = self._next
            self._launch_ready()
            self.hook_launches += self._next - before
        return hook

    def _launch_ready(self):
        """Collectives are issued strictly in bucket order on every rank: a bucket whose gradients are complete waits
        for the buckets before it (a rank on which some parameter got no gradient this step -- an image without
        positives, say -- would otherwise issue its all-reduces in a different order than its peers)."""
        while self._next < len(self.buckets) and self.pending[self._next] <= 0:
            self._launch(self._next)
            self._next += 1

    def _launch(self, bi):
        self.launched[bi] = True
        if self.world > 1 and self._hook_streams[bi]:
            # The collective is ordered behind the stream it is issued from.  Gradients of this bucket that were accumulated
            # on ANOTHER stream (see the hook) are only ordered against that one: make the issuing stream wait for each.
            cur = torch.cuda.current_stream()
            for st in self._hook_streams[bi]:
                if st != cur:
                    cur.wait_stream(st)
        if self.world > 1:
            op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
            self.handles[bi] = dist.all_reduce(self.flat[bi], op=op, async_op=True)

    def zero_grad(self, set_to_zero=True):
        """Replaces optimizer.zero_grad(): grads stay views of the flat buffers.  ``set_to_zero=False`` starts the next
        micro-step of a gradient accumulation (SOLVER.GRADIENT_ACCUMULATION_STEPS, engine/trainer.py:117,135-141): the hook
        bookkeeping is reset, the buffers keep what the earlier micro-steps left -- values every rank holds identically after
        their all-reduce, so reducing the buffer again after this micro-step's backward adds exactly its averaged gradient."""
        self._next = 0
        self.hook_launches = 0
        for bi, (flat, bucket) in enumerate(zip(self.flat, self.buckets)):
            if set_to_zero:
                flat.zero_()
            self.pending[bi] = self._expected(bucket)
            self.handles[bi] = None
            self.launched[bi] = False
            self._hook_streams[bi].clear()
            off, base, esz = 0, flat.data_ptr(), flat.element_size()
            for p in bucket:  # pointer comparison only: no tensor op per parameter on the per-step path
                # (a parameter frozen during the run keeps its view: the slot stays zero on every rank, and the optimizer goes
                # on applying weight decay and momentum to it -- what the reference's pinned torch 1.7.1 / apex
                # ``zero_grad()``, which zeroes in place, does to ``uncertain_pred`` after MODEL.UNCERTAINTY_TRAIN_ITER;
                # tests/golden/step_student_freeze.npz)
                if p.grad is None or p.grad.data_ptr() != base + off * esz:
                    p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()

    def finish(self):
        """Call after backward(): flush buckets whose hooks did not all fire (parameters that got no
        gradient this step contribute zeros), wait for the collectives and average."""
        for bi in range(self._next, len(self.buckets)):  # in order, after everything the hooks issued
            self._launch(bi)
        self._next = len(self.buckets)
        if self.world > 1:
            measure = self.measure_wait
            on_device = measure and self.flat and self.flat[0].is_cuda
            if on_device:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            elif measure:
                import time
                t0 = time.perf_counter()
            for h in self.handles:
                if h is not None:
                    h.wait()
            if on_device:
                e1.record()
                self._wait_records.append((e0, e1))
            elif measure:
                self._wait_records.append(time.perf_counter() - t0)
            if not self._avg_in_collective:
                for flat in self.flat:
                    flat.div_(self.world)

    def clip_grad_norm_(self, max_norm):
        """``torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)`` (engine/trainer.py:136-138) on the flat buckets:
        total L2 norm over all gradients, every gradient scaled by min(1, max_norm / (norm + 1e-6)) -- no host read."""
        if not self.flat:
            return None
        total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(f.float()) for f in self.flat]))
        coef = (float(max_norm) / (total + 1e-6)).clamp(max=1.0)
        for f in self.flat:
            f.mul_(coef.to(f.dtype))
        return total

    def exposed_wait_ms(self, clear=True):
        """Per-step stall of finish() on the collectives, in ms (list; needs ``measure_wait``; device events must have
        completed: call after a synchronize)."""
        out = [r[0].elapsed_time(r[1]) if isinstance(r, tuple) else 1e3 * r for r in self._wait_records]
        if clear:
            self._wait_records = []
        return out

    def remove(self):
        for h in self._hooks:
            h.remove()
