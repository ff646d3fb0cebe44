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


def polygon_to_mask(polygons, height, width):
    """Union of the polygons' masks at height x width (pycocotools frPyObjects -> merge -> decode, restated in
    ovis_oracle.c::oracle_polygon_to_mask).  polygons: iterable of flat (x0, y0, x1, y1, ...) float sequences."""
    import numpy as np

    mask = np.zeros((height, width), dtype=np.uint8)
    for p in polygons:
        xy = np.ascontiguousarray(np.asarray(p, dtype=np.float64))  # _mask.pyx: np.array(p, dtype=np.double)
        lib().oracle_polygon_to_mask(ctypes.c_void_p(xy.ctypes.data), int(xy.size // 2), int(height), int(width),
                                     ctypes.c_void_p(mask.ctypes.data))
    return torch.from_numpy(mask)


def project_polygons_on_boxes(instances, gt_index, boxes, image_size, resolution):
    """mask_head/loss.py:11-42 for polygon targets: per box, PolygonInstance.crop (segmentation_mask.py:270-296) ->
    resize((M, M)) (:298-324) -> convert_to_binarymask (:326-334) -> float32 [P, M, M].  instances: per ground truth a list
    of flat polygons; image_size = (width, height).  The crop / resize arithmetic is float32 tensor (op) python float, as
    in the reference."""
    import numpy as np

    W, H = image_size
    M = resolution
    out = torch.zeros((len(gt_index), M, M), dtype=torch.float32)
    for i, (g, box) in enumerate(zip(gt_index.tolist(), boxes.tolist())):
        xmin, ymin, xmax, ymax = map(float, box)
        xmin = min(max(xmin, 0), W - 1)
        ymin = min(max(ymin, 0), H - 1)
        xmax = min(max(xmax, 0), W)
        ymax = min(max(ymax, 0), H)
        xmax = max(xmax, xmin + 1)
        ymax = max(ymax, ymin + 1)
        w, h = xmax - xmin, ymax - ymin
        polys = []
        for p in instances[g]:
            p = torch.as_tensor(p, dtype=torch.float32)
            if len(p) < 6:
                continue
            p = p.clone()
            p[0::2] = p[0::2] - xmin
            p[1::2] = p[1::2] - ymin
            rw, rh = float(M) / float(w), float(M) / float(h)
            if rw == rh:
                p = p * rw
            else:
                p[0::2] *= rw
                p[1::2] *= rh
            polys.append(p.numpy())
        out[i] = polygon_to_mask(polys, M, M).float()
    return out


def text_embed(table, input_ids, special_tokens_mask):
    """st_generalized_rcnn.py:202-209 (extract_emb) on token ids: masked mean of the table rows, then F.normalize -- in
    torch on the CPU, expression by expression."""
    import torch.nn.functional as F

    table = _f(table)
    emb = table[input_ids.long().cpu()]
    mask = (1 - special_tokens_mask.cpu()).to(torch.float32)
    e = (emb * mask[:, :, None]).sum(1) / mask.sum(1)[:, None]
    return F.normalize(e, dim=-1)
