"""Operator surface of ``maskrcnn_benchmark._C`` (maskrcnn_benchmark/csrc/vision.cpp:9-25) on MI355X.

Same names, argument order and return conventions as the reference's pybind module; every op
is a thin tensor<->pointer shim over the C ABI in ``include/ovis_hip.h``.  Device tensors
only: CPU tensors raise ``RuntimeError`` (the reference itself raises "Not implemented on the
CPU" for most of these, csrc/ROIAlign.h:44, csrc/SigmoidFocalLoss.h:23,40).
"""
import torch

from . import _lib

_L = _lib.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a HIP device tensor: this package has no CPU implementation")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


# ---- RoIAlign (csrc/ROIAlign.h:11-46) ---------------------------------------------------------
def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    input, rois = _dev(input, "input"), _dev(rois, "rois")
    if rois.dim() != 2 or rois.size(1) != 5 or input.dim() != 4:
        raise RuntimeError("roi_align_forward: expected input [N,C,H,W] and rois [R,5]")
    n, c, h, w = input.shape
    r = rois.size(0)
    out = torch.empty((r, c, pooled_height, pooled_width), dtype=input.dtype, device=input.device)
    if out.numel() == 0:
        return out
    with torch.cuda.device(input.device):
        rc = _L.ovis_roi_align_forward_f32(input.data_ptr(), rois.data_ptr(), out.data_ptr(), r, n, c, h, w,
                                           pooled_height, pooled_width, spatial_scale, sampling_ratio, _stream())
    _lib.check(rc, "roi_align_forward")
    return out


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height,
                       width, sampling_ratio):
    grad, rois = _dev(grad, "grad"), _dev(rois, "rois")
    r = rois.size(0)
    gin = torch.empty((batch_size, channels, height, width), dtype=grad.dtype, device=grad.device)
    if gin.numel() == 0:
        return gin
    with torch.cuda.device(grad.device):
        rc = _L.ovis_roi_align_backward_f32(grad.data_ptr(), rois.data_ptr(), gin.data_ptr(), r, batch_size,
                                            channels, height, width, pooled_height, pooled_width,
                                            spatial_scale, sampling_ratio, _stream())
    _lib.check(rc, "roi_align_backward")
    return gin


# ---- NMS (csrc/nms.h:10-28) -------------------------------------------------------------------
def nms_padded(dets, scores, threshold, ge_mode=False):
    """Sync-free form: returns (keep[K] int64 -- first n entries valid, ascending; n as a
    1-element int32 device tensor).  Extension of the reference API for device pipelines."""
    dets, scores = _dev(dets, "dets"), _dev(scores, "scores")
    k = dets.size(0)
    keep = torch.empty((k,), dtype=torch.int64, device=dets.device)
    num = torch.zeros((1,), dtype=torch.int32, device=dets.device)
    if k == 0:
        return keep, num
    if dets.dim() != 2 or dets.size(1) != 4 or scores.numel() != k:
        raise RuntimeError("nms: expected dets [K,4] and scores [K]")
    with torch.cuda.device(dets.device):
        nbytes = _L.ovis_nms_workspace_bytes(k)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dets.device)
        rc = _L.ovis_nms_f32(dets.data_ptr(), scores.data_ptr(), k, threshold, int(bool(ge_mode)), ws.data_ptr(),
                             nbytes, keep.data_ptr(), num.data_ptr(), _stream())
    _lib.check(rc, "nms")
    return keep, num


def nms(dets, scores, threshold):
    if dets.is_cuda and dets.numel() == 0:
        # the reference returns a CPU tensor for the empty case (csrc/nms.h:17-18)
        return torch.empty((0,), dtype=torch.int64, device="cpu")
    keep, num = nms_padded(dets, scores, threshold)
    return keep[: int(num.item())]


# ---- sigmoid focal loss (csrc/SigmoidFocalLoss.h:10-41) ----------------------------------------
def sigmoid_focalloss_forward(logits, targets, num_classes, gamma, alpha):
    logits, targets = _dev(logits, "logits"), _dev(targets, "targets", torch.int32)
    if logits.dim() != 2:
        raise RuntimeError("logits should be NxClass")
    losses = torch.empty_like(logits)
    if losses.numel() == 0:
        return losses
    with torch.cuda.device(logits.device):
        rc = _L.ovis_sigmoid_focal_loss_forward_f32(logits.data_ptr(), targets.data_ptr(), losses.data_ptr(),
                                                    logits.size(0), logits.size(1), gamma, alpha, _stream())
    _lib.check(rc, "sigmoid_focalloss_forward")
    return losses


def sigmoid_focalloss_backward(logits, targets, d_losses, num_classes, gamma, alpha):
    logits, targets = _dev(logits, "logits"), _dev(targets, "targets", torch.int32)
    d_losses = _dev(d_losses, "d_losses")
    if logits.dim() != 2 or logits.size(1) != num_classes:
        raise RuntimeError("logits.size(1) should be num_classes")
    d_logits = torch.zeros_like(logits)
    if d_logits.numel() == 0:
        return d_logits
    with torch.cuda.device(logits.device):
        rc = _L.ovis_sigmoid_focal_loss_backward_f32(logits.data_ptr(), targets.data_ptr(), d_losses.data_ptr(),
                                                     d_logits.data_ptr(), logits.size(0), num_classes, gamma,
                                                     alpha, _stream())
    _lib.check(rc, "sigmoid_focalloss_backward")
    return d_logits
