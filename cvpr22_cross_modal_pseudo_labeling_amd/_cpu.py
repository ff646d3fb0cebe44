"""HOST-tensor side of ``_C``: the reference's CPU-only configuration (``MODEL.DEVICE cpu``, BASELINE.json configs[0]).

The reference's native module dispatches on the tensor's device (csrc/ROIAlign.h:11-25, csrc/nms.h:10-28): host tensors go to
``csrc/cpu/ROIAlign_cpu.cpp`` / ``csrc/cpu/nms_cpu.cpp``, and the heads / losses are plain torch ops on whatever device the
tensors live on.  This module is that side: RoIAlign forward, its transpose and NMS in ``libovis_cpu.so`` (C++,
``csrc/cpu/ovis_cpu.cpp``, C ABI ``include/ovis_cpu.h``), the head / loss entry points as the torch-op formulas of the
reference's modules.  It is reached ONLY with host tensors (``_C.py`` branches on ``is_cuda``); device tensors are never
routed here and host tensors never to the device, and a missing library raises."""
import ctypes
import os

import torch
import torch.nn.functional as F

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libovis_cpu.so")

_i, _f, _vp = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
SIGNATURES = {
    "ovis_cpu_roi_align_forward_f32": (_i, [_vp, _vp, _vp] + [_i] * 7 + [_f, _i, _i]),
    "ovis_cpu_roi_align_backward_f32": (_i, [_vp, _vp, _vp] + [_i] * 7 + [_f, _i, _i]),
    "ovis_cpu_nms_f32": (_i, [_vp, _vp, _i, _f, _vp]),
    "ovis_cpu_project_polygon_masks_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i]),
    "ovis_cpu_polygons_to_masks_u8": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i]),
    "ovis_cpu_version": (ctypes.c_char_p, []),
}
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it first (make -C cvpr22_cross_modal_pseudo_labeling_amd/csrc); "
                              "host tensors have no other implementation")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _host(t, name, dtype=torch.float32):
    if t.is_cuda:
        raise RuntimeError(f"{name}: host and device tensors mixed in one call")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def _check(rc, what):
    if rc < 0:
        raise RuntimeError(f"{what}: bad argument (a RoI's batch index outside the batch, or a null / negative size)")


# ---- csrc/cpu/ROIAlign_cpu.cpp:114-219; the transpose has no host form in the reference (csrc/ROIAlign.h:44) ----------------
def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    input, rois = _host(input, "input"), _host(rois, "rois")
    if rois.dim() != 2 or rois.size(1) != 5 or input.dim() != 4:
        raise RuntimeError("roi_align_forward: expected input [N,C,H,W] and rois [R,5]")
    n, c, h, w = input.shape
    r = rois.size(0)
    out = torch.empty((r, c, pooled_height, pooled_width), dtype=torch.float32)
    if out.numel():
        _check(load().ovis_cpu_roi_align_forward_f32(input.data_ptr(), rois.data_ptr(), out.data_ptr(), r, n, c, h, w,
                                                      pooled_height, pooled_width, spatial_scale, sampling_ratio, 0),
               "roi_align_forward")
    return out


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height, width,
                       sampling_ratio):
    grad, rois = _host(grad, "grad"), _host(rois, "rois")
    gin = torch.empty((batch_size, channels, height, width), dtype=torch.float32)
    if gin.numel():
        _check(load().ovis_cpu_roi_align_backward_f32(grad.data_ptr(), rois.data_ptr(), gin.data_ptr(), rois.size(0), batch_size,
                                                       channels, height, width, pooled_height, pooled_width, spatial_scale,
                                                       sampling_ratio, 0), "roi_align_backward")
    return gin


# ---- csrc/cpu/nms_cpu.cpp:6-65 (a box goes when IoU >= threshold -- the host kernel's comparison, whatever ge_mode says) ------
def nms_padded(dets, scores, threshold, ge_mode=True):
    dets, scores = _host(dets, "dets"), _host(scores, "scores")
    k = dets.size(0)
    keep = torch.zeros((k,), dtype=torch.int64)
    if k == 0:
        return keep, torch.zeros((1,), dtype=torch.int32)
    if dets.dim() != 2 or dets.size(1) != 4 or scores.numel() != k:
        raise RuntimeError("nms: expected dets [K,4] and scores [K]")
    n = load().ovis_cpu_nms_f32(dets.data_ptr(), scores.data_ptr(), k, threshold, keep.data_ptr())
    _check(n, "nms")
    return keep, torch.tensor([n], dtype=torch.int32)


def nms(dets, scores, threshold):
    keep, num = nms_padded(dets, scores, threshold)
    return keep[: int(num)]


# ---- heads / losses: the reference's own torch-op formulas ------------------------------------------------------------
def gemm_nt(a, b, bias=None):
    """a [M, K] @ b [N, K]^T (+ bias): nn.Linear of roi_box_predictors.py:62-81 / roi_mask_predictors.py:41-65."""
    y = a @ b.t()
    return y if bias is None else y + bias


def region_noun_align(region_emb, noun_emb):
    """st_generalized_rcnn.py:236-246: per noun the best region -> (raw maximum, its sigmoid, argmax)."""
    raw, idx = torch.max(region_emb @ noun_emb.t(), dim=0)
    return raw, torch.sigmoid(raw), idx


def weighted_ce_fwd_bwd(logits, labels, bg_weight, need_grad=True):
    """box_head/loss.py:160-176: cross entropy with the background class weighted, summed over the rows and divided by
    their number -> (loss, d loss / d logits)."""
    p = labels.numel()
    logp = F.log_softmax(logits, dim=1)
    w = torch.ones(logits.shape[1], dtype=logits.dtype)
    w[0] = bg_weight
    wy = w[labels]
    loss = -(wy * logp.gather(1, labels.view(-1, 1)).squeeze(1)).sum() / max(p, 1)
    if not need_grad:
        return loss, None
    g = logp.exp()
    g.scatter_add_(1, labels.view(-1, 1), torch.full((p, 1), -1.0, dtype=logits.dtype))
    return loss, g * (wy / max(p, 1)).view(-1, 1)


def mask_bce_stochastic_fwd_bwd(mu, sigma, eps, pos_index, targets, channel, need_grad=True):
    """mask_head/loss.py:107-148 with the stochastic logits of roi_mask_predictors.py:52-63 (z = mu + eps * sigma):
    mean binary cross entropy of the positives' selected channel -> (loss, d mu, d sigma)."""
    z = mu if sigma is None else mu + eps * sigma
    sel = z[pos_index, channel]                                   # [Pp, M, M]
    npos = pos_index.numel()
    t = targets.reshape(sel.shape)
    loss = F.binary_cross_entropy_with_logits(sel, t, reduction="mean") if npos else sel.sum() * 0
    if not need_grad:
        return loss, None, None
    dsel = (torch.sigmoid(sel) - t) / max(sel.numel(), 1)
    dmu = torch.zeros_like(mu)
    dmu.index_put_((pos_index, channel if torch.is_tensor(channel) else torch.full_like(pos_index, channel)), dsel, accumulate=True)
    dsigma = None
    if sigma is not None:
        dsigma = torch.zeros_like(sigma)
        dsigma.index_put_((pos_index, torch.zeros_like(pos_index)), dsel * eps[pos_index, channel], accumulate=True)
    return loss, dmu, dsigma


# ---- mask_head/loss.py:11-42 for polygon targets (segmentation_mask.py:270-334 + pycocotools' rasteriser) ----------------------
def project_polygon_masks(coords, polygon_start, instance_start, gt_index, boxes, image_size, resolution):
    boxes = _host(boxes, "boxes")
    coords = _host(coords, "coords") if coords.numel() else coords
    polygon_start, instance_start = _host(polygon_start, "polygon_start", torch.int32), _host(instance_start, "instance_start", torch.int32)
    gt_index = _host(gt_index, "gt_index", torch.int64)
    p = boxes.shape[0]
    out = torch.empty((p, resolution, resolution), dtype=torch.float32)
    if p:
        _check(load().ovis_cpu_project_polygon_masks_f32(coords.data_ptr() if coords.numel() else 0, polygon_start.data_ptr(),
                                                        instance_start.data_ptr(), gt_index.data_ptr(), boxes.data_ptr(), p,
                                                        int(image_size[0]), int(image_size[1]), int(resolution), out.data_ptr(), 0),
               "project_polygon_masks")
    return out


def polygons_to_masks(coords, polygon_start, instance_start, image_size):
    """uint8 [G, height, width] whole-image masks of the G polygon instances (segmentation_mask.py:326-334)."""
    coords = _host(coords, "coords") if coords.numel() else coords
    polygon_start, instance_start = _host(polygon_start, "polygon_start", torch.int32), _host(instance_start, "instance_start", torch.int32)
    g = instance_start.numel() - 1
    w, h = int(image_size[0]), int(image_size[1])
    out = torch.empty((g, h, w), dtype=torch.uint8)
    if g:
        _check(load().ovis_cpu_polygons_to_masks_u8(coords.data_ptr() if coords.numel() else 0, polygon_start.data_ptr(),
                                                   instance_start.data_ptr(), g, w, h, out.data_ptr(), 0), "polygons_to_masks")
    return out


# ---- st_generalized_rcnn.py:202-209 (extract_emb) on host tensors: the reference's own tensor-op formula -----------------------
def text_embed(table, input_ids, special_tokens_mask):
    keep = (1 - special_tokens_mask).to(torch.float32)
    emb = (table[input_ids] * keep[:, :, None]).sum(1) / keep.sum(1)[:, None]
    return F.normalize(emb, dim=-1)
