"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  It wraps

* ``libovis_oracle.so`` -- the plain-C restatement (``ovis_oracle.c``; kind "port"), and
* ``_ref/ref_C.so``     -- the reference's own CPU kernels compiled by ``build_ref.py`` from
  ``/root/reference`` (kind "reference"; RoIAlign forward and NMS only -- everything else in
  that module raises "Not implemented on the CPU", csrc/ROIAlign.h:44).

All functions take / return CPU ``torch`` tensors.
"""
import ctypes
import importlib.util
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def build(with_ref=True):
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    if with_ref and os.path.isdir("/root/reference") and not os.path.exists(os.path.join(_HERE, "_ref", "ref_C.so")):
        subprocess.check_call(["python3", os.path.join(_HERE, "build_ref.py")])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libovis_oracle.so")
        if not os.path.exists(path):
            build(with_ref=False)
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_nms_f32.restype = ctypes.c_int
    return _LIB


def ref_module():
    """The reference's pybind module (or None when oracle/_ref was not built)."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "ref_C.so")
        if not os.path.exists(path):
            return None
        spec = importlib.util.spec_from_file_location("ref_C", path)
        _REF = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_REF)
    return _REF


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _f(t, dtype=torch.float32):
    return t.detach().to("cpu", dtype).contiguous()


def roi_align_forward(inp, rois, scale, ph, pw, sampling_ratio, dtype=torch.float32):
    inp, rois = _f(inp, dtype), _f(rois, dtype)
    n, c, h, w = inp.shape
    r = rois.shape[0]
    out = torch.empty(r, c, ph, pw, dtype=dtype)
    if dtype == torch.float32:
        lib().oracle_roi_align_forward_f32(_p(inp), _p(rois), _p(out), r, n, c, h, w, ph, pw, ctypes.c_float(scale), sampling_ratio)
    else:
        lib().oracle_roi_align_forward_f64(_p(inp), _p(rois), _p(out), r, n, c, h, w, ph, pw, ctypes.c_double(scale), sampling_ratio)
    return out


def roi_align_backward(grad, rois, scale, ph, pw, n, c, h, w, sampling_ratio, dtype=torch.float32):
    grad, rois = _f(grad, dtype), _f(rois, dtype)
    r = rois.shape[0]
    gin = torch.empty(n, c, h, w, dtype=dtype)
    if dtype == torch.float32:
        lib().oracle_roi_align_backward_f32(_p(grad), _p(rois), _p(gin), r, n, c, h, w, ph, pw, ctypes.c_float(scale), sampling_ratio)
    else:
        lib().oracle_roi_align_backward_f64(_p(grad), _p(rois), _p(gin), r, n, c, h, w, ph, pw, ctypes.c_double(scale), sampling_ratio)
    return gin


def nms(boxes, scores, thr, ge_mode=False):
    boxes, scores = _f(boxes), _f(scores)
    k = boxes.shape[0]
    keep = torch.empty(max(k, 1), dtype=torch.int64)
    n = lib().oracle_nms_f32(_p(boxes), _p(scores), k, ctypes.c_float(thr), int(bool(ge_mode)), _p(keep))
    return keep[:n].clone()


def sigmoid_focal_loss_forward(logits, targets, gamma, alpha):
    logits = _f(logits)
    targets = targets.detach().to("cpu", torch.int32).contiguous()
    out = torch.empty_like(logits)
    lib().oracle_sigmoid_focal_loss_forward_f32(_p(logits), _p(targets), _p(out), logits.shape[0], logits.shape[1], ctypes.c_float(gamma), ctypes.c_float(alpha))
    return out


def sigmoid_focal_loss_backward(logits, targets, d_losses, gamma, alpha):
    logits, d_losses = _f(logits), _f(d_losses)
    targets = targets.detach().to("cpu", torch.int32).contiguous()
    out = torch.empty_like(logits)
    lib().oracle_sigmoid_focal_loss_backward_f32(_p(logits), _p(targets), _p(d_losses), _p(out), logits.shape[0], logits.shape[1], ctypes.c_float(gamma), ctypes.c_float(alpha))
    return out


def roi_pool_forward(inp, rois, scale, ph, pw):
    """(output, argmax int32) -- csrc/cuda/ROIPool_cuda.cu:17-77."""
    inp, rois = _f(inp), _f(rois)
    n, c, h, w = inp.shape
    r = rois.shape[0]
    out = torch.empty((r, c, ph, pw), dtype=torch.float32)
    arg = torch.empty((r, c, ph, pw), dtype=torch.int32)
    lib().oracle_roi_pool_forward_f32(_p(inp), _p(rois), _p(out), _p(arg), r, c, h, w, ph, pw, ctypes.c_float(scale))
    return out, arg


def roi_pool_backward(grad, argmax, rois, n, c, h, w):
    """csrc/cuda/ROIPool_cuda.cu:80-108."""
    grad, rois = _f(grad), _f(rois)
    argmax = argmax.detach().to("cpu", torch.int32).contiguous()
    r, _, ph, pw = grad.shape
    gin = torch.empty((n, c, h, w), dtype=torch.float32)
    lib().oracle_roi_pool_backward_f32(_p(grad), _p(argmax), _p(rois), _p(gin), r, n, c, h, w, ph, pw)
    return gin
