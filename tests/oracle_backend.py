"""TEST-ONLY: lets the host-side model code run on CPU tensors by routing the native-op entry points
of ``cvpr22_cross_modal_pseudo_labeling_amd._C`` to the CPU oracle.  The product package never does this
(its ops raise on CPU tensors); tests use it to (a) exercise the Python plumbing without a GPU and
(b) produce the CPU side of GPU-vs-oracle comparisons of whole-model steps."""
import contextlib

import torch

import oracle
from cvpr22_cross_modal_pseudo_labeling_amd import _C


def _nms_padded(dets, scores, thr, ge_mode=False):
    keep = oracle.nms(dets, scores, thr, ge_mode)
    out = torch.zeros(dets.shape[0], dtype=torch.int64)
    out[: keep.numel()] = keep
    return out, torch.tensor([keep.numel()], dtype=torch.int32)


@contextlib.contextmanager
def oracle_ops():
    saved = {k: getattr(_C, k) for k in ("roi_align_forward", "roi_align_backward", "nms", "nms_padded",
                                         "sigmoid_focalloss_forward", "sigmoid_focalloss_backward")}
    _C.roi_align_forward = lambda x, r, s, ph, pw, sr: oracle.roi_align_forward(x, r, s, ph, pw, sr)
    _C.roi_align_backward = lambda g, r, s, ph, pw, n, c, h, w, sr: oracle.roi_align_backward(g, r, s, ph, pw, n, c, h, w, sr)
    _C.nms = lambda d, s, t: oracle.nms(d, s, t)
    _C.nms_padded = _nms_padded
    _C.sigmoid_focalloss_forward = lambda l, t, nc, g, a: oracle.sigmoid_focal_loss_forward(l, t, g, a)
    _C.sigmoid_focalloss_backward = lambda l, t, d, nc, g, a: oracle.sigmoid_focal_loss_backward(l, t, d, g, a)
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(_C, k, v)
