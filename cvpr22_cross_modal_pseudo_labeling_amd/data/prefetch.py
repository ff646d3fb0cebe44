"""Host -> device staging of the input stream, off the training thread.

The reference feeds the step from a ``torch.utils.data.DataLoader`` and moves every batch with a blocking
``images.to(device)`` inside the loop (maskrcnn_benchmark/engine/trainer.py:100-104).  On this design the training thread
must keep two HIP streams fed (engine/trainer.py::PipelinedTrainer), and a pageable host-to-device copy issued from it
waits behind everything already queued on its stream.  ``DevicePrefetcher`` therefore runs the source iterator on its own
thread, stages every batch through pinned memory on a dedicated copy stream a few batches ahead, and hands the consumer
device tensors plus an event its current stream waits on -- the copies overlap the previous steps' compute.
"""
import queue
import threading

import torch


def _map(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(o, fn) for o in obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    if hasattr(obj, "bbox") and hasattr(obj, "extra_fields"):  # BoxList
        out = type(obj)(fn(obj.bbox), obj.size)
        for k, v in obj.extra_fields.items():
            out.add_field(k, _map(v, fn))
        return out
    return obj


_COPY_STREAMS = {}


def copy_stream(device):
    """ONE copy stream per device and process.  HIP multiplexes streams onto a handful of hardware queues (four by default):
    a process that keeps creating streams -- a second prefetcher after the first was closed -- ends up with its new copy stream
    on the hardware queue of the compute stream, and every staged batch then queues behind the step's kernels (measured: the
    teacher step at 31 instead of 22.5 ms when it ran as the SECOND workload of a bench process)."""
    key = (device.type, device.index)
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(device)
    return _COPY_STREAMS[key]


class DevicePrefetcher:
    """Iterates ``source`` (host batches: tensors / BoxLists / nested lists, tuples, dicts of them) ``depth`` batches
    ahead on a worker thread and yields the same structures on ``device``.  On a CPU device it is a plain look-ahead
    queue.  Exceptions of the source are re-raised in the consumer."""

    _END = object()

    def __init__(self, source, device, depth=2):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        if self.cuda and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.stream = copy_stream(self.device) if self.cuda else None
        self.q = queue.Queue(maxsize=max(int(depth), 1))
        # pinned staging buffers: ``depth`` batches wait in the queue, one is with the consumer, one is being staged
        self._slots = [{"buffer": None, "event": None} for _ in range(max(int(depth), 1) + 2)]
        self._next_slot = 0
        self.stop = False
        self.thread = threading.Thread(target=self._run, args=(iter(source),), daemon=True)
        self.thread.start()

    def _stage(self, batch):
        """Host batch -> device tensors + the event of their copies.  Pageable tensors go through one of ``depth + 2``
        REUSED pinned staging buffers (one buffer per batch, every tensor at a 64-byte aligned offset): ``t.pin_memory()``
        per tensor allocated and first-touched fresh pinned pages for every batch -- 34 ms per 2-image batch (41 MB in ~12
        tensors), which made ``tools/train_net.py`` at 2 images per GPU input-bound at 58 ms per iteration against a 33 ms
        step (tools/experiments/data_path_rate.py).  Already-pinned tensors are copied from where they are."""
        if not self.cuda:
            return _map(batch, lambda t: t.to(self.device)), None
        torch.cuda.set_device(self.device)
        tensors = []
        _map(batch, lambda t: (tensors.append(t), t)[1] if (not t.is_cuda and not t.is_pinned() and t.numel()) else t)
        need = sum((t.numel() * t.element_size() + 63) // 64 * 64 for t in tensors)
        slot = None
        if need:
            slot = self._slots[self._next_slot % len(self._slots)]
            self._next_slot += 1
            if slot["event"] is not None:
                slot["event"].synchronize()          # the copies of this buffer's previous batch have left it
            if slot["buffer"] is None or slot["buffer"].numel() < need:
                slot["buffer"] = torch.empty(int(need * 1.25), dtype=torch.uint8).pin_memory()
        off = 0
        staged = {}
        for t in tensors:
            nbytes = t.numel() * t.element_size()
            view = slot["buffer"][off:off + nbytes].view(t.dtype).view(t.shape)
            view.copy_(t)
            staged[id(t)] = view
            off += (nbytes + 63) // 64 * 64
        with torch.cuda.stream(self.stream):
            moved = _map(batch, lambda t: t if t.is_cuda else staged.get(id(t), t).to(self.device, non_blocking=True))
            ev = torch.cuda.Event()
            ev.record(self.stream)
        if slot is not None:
            slot["event"] = ev
        return moved, ev

    def _put(self, item):
        """Blocking put that gives up when the consumer has closed the prefetcher (a full queue nobody reads)."""
        while not self.stop:
            try:
                self.q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _run(self, it):
        try:
            for batch in it:
                if self.stop or not self._put(self._stage(batch)):
                    break
            else:
                self._put((self._END, None))
        except BaseException as e:  # handed to the consumer
            self._put((e, None))
        finally:
            del it  # lets a DataLoader iterator shut its worker processes down

    def __iter__(self):
        return self

    def __next__(self):
        item, ev = self.q.get()
        if item is self._END:
            self.q.put((self._END, None))
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            _map(item, lambda t: (t.record_stream(cur), t)[1] if t.is_cuda else t)  # allocated on the copy stream
        return item

    def close(self, timeout=10.0):
        """Stop the staging thread and wait for it: no copy is issued on the copy stream after this returns (the caller
        may tear the process group / the device context down)."""
        self.stop = True
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        if self.thread.is_alive() and threading.current_thread() is not self.thread:
            self.thread.join(timeout)
        if self.stream is not None:
            self.stream.synchronize()
