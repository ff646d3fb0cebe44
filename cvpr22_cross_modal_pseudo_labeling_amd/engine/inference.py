"""Evaluation-side data path: run the detector over a stream of batches and collect every rank's detections on rank 0.

Counterpart of maskrcnn_benchmark/engine/inference.py:25-47 (``compute_on_dataset``), :82-101
(``_accumulate_predictions_from_multiple_gpus``) and :104-170 (``inference``).  The reference ships each rank's
``{image_id: BoxList}`` dict as a pickle through a byte-tensor ``all_gather`` (utils/comm.py:65-103); here the
detections are packed into a handful of flat tensors per rank and exchanged with padded tensor all-gathers over the
process group's own backend (RCCL on the GPUs, gloo on CPU): the payload is never pickled and stays on the device for
the NCCL backend (only the field names / trailing shapes -- configuration, the same on every rank -- travel as a small
object, so that a rank without images knows the layout).  Dataset-specific scoring (COCO / LVIS ``evaluate``) needs the
annotation files and ``pycocotools``, neither of which exists in this build: ``inference`` returns / saves the ordered
prediction list the reference hands to ``evaluate`` (``predictions.pth``, inference.py:163-164).
"""
import logging
import os
import time

import torch
import torch.distributed as dist

from . import comm
from ..modeling.structures import BoxList


@torch.no_grad()
def compute_on_dataset(model, batches, device, timer=None):
    """batches: iterable of (images, targets, image_ids) -> {image_id: BoxList on CPU} (inference.py:25-47)."""
    model.eval()
    results = {}
    cpu = torch.device("cpu")
    for images, targets, image_ids in batches:
        t0 = time.perf_counter()
        output = model(images.to(device), targets)
        if timer is not None:
            if torch.device(device).type == "cuda":
                torch.cuda.synchronize()
            timer.append(time.perf_counter() - t0)
        for img_id, det in zip(image_ids, output):
            results[int(img_id)] = det.to(cpu)
    return results


def _pack(predictions, device):
    """{id: BoxList} -> (header int64 [n, 4] = id, count, width, height; {field: tensor [sum count, ...]})."""
    ids = sorted(predictions)
    header = torch.tensor([[i, len(predictions[i]), predictions[i].size[0], predictions[i].size[1]] for i in ids],
                          dtype=torch.int64).reshape(-1, 4)
    fields = {"bbox": torch.cat([predictions[i].bbox for i in ids], 0) if ids else torch.zeros(0, 4)}
    names = sorted(predictions[ids[0]].fields()) if ids else []
    for name in names:
        fields[name] = torch.cat([predictions[i].get_field(name) for i in ids], 0)
    return header.to(device), {k: v.to(device) for k, v in fields.items()}


def _unpack(header, fields):
    out, start = {}, 0
    for img_id, count, w, h in header.tolist():
        det = BoxList(fields["bbox"][start:start + count], (w, h))
        for name, v in fields.items():
            if name != "bbox":
                det.add_field(name, v[start:start + count])
        out[img_id] = det
        start += count
    return out


def _all_gather_rows(t, device):
    """All-gather of tensors that differ in their first dimension only: sizes first, then one padded exchange."""
    world = comm.get_world_size()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    padded = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=device)
    padded[: t.shape[0]] = t
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    return [p[:s].cpu() for p, s in zip(parts, sizes)]


def gather_predictions(predictions, device=None):
    """Every rank's {image_id: BoxList} -> the merged dict on rank 0, None elsewhere (inference.py:82-91)."""
    if comm.get_world_size() == 1:
        return dict(predictions)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    header, fields = _pack(predictions, device)
    # field names / trailing shapes / dtypes come from the model configuration and are the same on every rank, but a
    # rank may hold no image at all: agree on the layout through rank-independent metadata first
    meta = [None] * comm.get_world_size()
    dist.all_gather_object(meta, {k: (tuple(v.shape[1:]), str(v.dtype)) for k, v in fields.items()})
    layout = max(meta, key=len)
    for name, (shape, dtype) in layout.items():
        if name not in fields:
            fields[name] = torch.zeros((0,) + shape, dtype=getattr(torch, dtype.split(".")[1]), device=device)
    headers = _all_gather_rows(header, device)
    gathered = {name: _all_gather_rows(fields[name], device) for name in sorted(layout)}
    if comm.get_rank() != 0:
        return None
    merged = {}
    for r, h in enumerate(headers):
        merged.update(_unpack(h, {name: parts[r] for name, parts in gathered.items()}))
    return merged


def accumulate_predictions(predictions, logger=None):
    """Merged dict -> list ordered by image id, with the reference's warning for gaps (inference.py:92-101)."""
    merged = gather_predictions(predictions)
    if merged is None:
        return None
    image_ids = sorted(merged)
    if image_ids and len(image_ids) != image_ids[-1] + 1:
        (logger or logging.getLogger("ovis.inference")).warning(
            "Number of images that were gathered from multiple processes is not a contiguous set. "
            "Some images might be missing from the evaluation")
    return [merged[i] for i in image_ids]


def inference(model, batches, dataset_name="synthetic", device="cuda", output_folder=None, class_embeddings=None,
              logger=None):
    """Detections of every image of ``batches`` (this rank's shard), gathered on rank 0 in image-id order and saved
    as ``predictions.pth`` (inference.py:104-170, without the dataset-specific ``evaluate`` call)."""
    logger = logger or logging.getLogger("ovis.inference")
    device = torch.device(device)
    if class_embeddings is not None:  # zero-shot heads map predicted embeddings to classes (inference.py:124-131)
        model.set_class_embeddings(class_embeddings.to(device))
    times = []
    start = time.perf_counter()
    predictions = compute_on_dataset(model, batches, device, times)
    comm.synchronize()
    total = time.perf_counter() - start
    world = comm.get_world_size()
    n_local = len(predictions)
    logger.info("%s: %d images on this rank in %.2f s (%.4f s / img per device, on %d devices); model time %.4f s / img",
                dataset_name, n_local, total, total * world / max(n_local * world, 1), world,
                sum(times) / max(n_local, 1))
    ordered = accumulate_predictions(predictions, logger)
    if comm.get_rank() != 0:
        return None
    if output_folder:
        os.makedirs(output_folder, exist_ok=True)
        torch.save(ordered, os.path.join(output_folder, "predictions.pth"))
    return ordered
