"""Operator surface of ``maskrcnn_benchmark._C`` (maskrcnn_benchmark/csrc/vision.cpp:9-25) on MI355X.

Same names, argument order and return conventions as the reference's pybind module; every op
is a thin tensor<->pointer shim over the C ABI in ``include/ovis_hip.h``.  Like the reference's module
(csrc/ROIAlign.h:11-25, csrc/nms.h:10-28) the entry points the CPU-only configuration needs dispatch on the
tensor's device: HOST tensors go to ``_cpu.py`` (RoIAlign, NMS: ``libovis_cpu.so``; heads / losses: the
reference's torch-op formulas), DEVICE tensors to the HIP kernels -- never across.  Every other op raises on
host tensors (the reference raises "Not implemented on the CPU", csrc/ROIAlign.h:44,
csrc/SigmoidFocalLoss.h:23,40).
"""
import contextlib
import functools
import ctypes
import threading

import torch

from . import _cpu, _lib

_L = _lib.load()


def _stream():
    # the raw handle of the current stream of the current device (the `with _on(...)` around every launch makes that
    # the operands' device); ~10x cheaper than building a torch.cuda.Stream object per launch
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


class _Same:
    """No-op context: the operands already live on the current device (every launch of a one-GPU-per-process run)."""

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_SAME = _Same()


def _on(device):
    """``torch.cuda.device(device)`` only when it would change the current device."""
    return _SAME if device.index == torch.cuda.current_device() else torch.cuda.device(device)


def _dev(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a HIP device tensor: this op has no host implementation (as in the reference)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


# ---- RoIAlign (csrc/ROIAlign.h:11-46) ---------------------------------------------------------
def _roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio, exact):
    if not input.is_cuda:
        return _cpu.roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio)
    input, rois = _dev(input, "input"), _dev(rois, "rois")
    if rois.dim() != 2 or rois.size(1) != 5 or input.dim() != 4:
        raise RuntimeError("roi_align_forward: expected input [N,C,H,W] and rois [R,5]")
    n, c, h, w = input.shape
    r = rois.size(0)
    out = torch.empty((r, c, pooled_height, pooled_width), dtype=input.dtype, device=input.device)
    if out.numel() == 0:
        return out
    with _on(input.device):
        if exact:
            rc = _L.ovis_roi_align_forward_f32(input.data_ptr(), rois.data_ptr(), out.data_ptr(), r, n, c, h, w,
                                               pooled_height, pooled_width, spatial_scale, sampling_ratio, _stream())
        else:
            nbytes = _L.ovis_roi_align_forward_workspace_bytes(r, h, w)
            ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=input.device)
            rc = _L.ovis_roi_align_forward_ws_f32(input.data_ptr(), rois.data_ptr(), out.data_ptr(), r, n, c, h, w,
                                                  pooled_height, pooled_width, spatial_scale, sampling_ratio,
                                                  ws.data_ptr(), nbytes, _stream())
    _lib.check(rc, "roi_align_forward")
    return out


def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    """The reference's `_C.roi_align_forward`: bit-identical to the reference CPU kernel
    (cpu/ROIAlign_cpu.cpp:114-219) on finite inputs -- same IEEE operation sequence, FP contraction off."""
    return _roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio, True)


def roi_align_forward_mfma(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    """Matrix-core forward (extension): separable weights as bf16 hi/lo MFMA operands, same values up to ~2^-15 of
    sum |w x|.  Round-1 timing: on par with the exact kernel (0.33-0.36 vs 0.31-0.32 ms on uniform RoIs, 0.44-0.45 vs
    0.46-0.52 ms on RPN-like ones at [2,1024,50,84], R=1024) -- both sit on the 0.8 MB/RoI output write.  Falls back to the exact kernel for maps below
    16 x 16 or pooled sizes above 16 x 16."""
    return _roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio, False)


def _channels_last_map(input, channel_multiple):
    """An [N, C, H, W] HIP tensor whose MEMORY is NHWC-contiguous (the NCHW view of the trunk's NHWC result) with a channel
    count the NHWC-input pooler takes: no layout copy is needed."""
    return (input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 and input.shape[1] % channel_multiple == 0
            and not input.is_contiguous() and input.permute(0, 2, 3, 1).is_contiguous() and input.data_ptr() % 16 == 0)


def _strided_from_nhwc(input, rois, out, spatial_scale, pooled_height, pooled_width, sampling_ratio, bin_stride, pair):
    n, c, h, w = input.shape
    with _on(input.device):
        rc = _L.ovis_roi_align_forward_strided_from_nhwc_f32(input.data_ptr(), rois.data_ptr(), out.data_ptr(), rois.size(0), n, c,
                                                             h, w, pooled_height, pooled_width, bin_stride, spatial_scale,
                                                             sampling_ratio, int(pair), _stream())
    _lib.check(rc, "roi_align_forward_strided_from_nhwc")


def roi_align_forward_strided_nhwc(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio, bin_stride):
    """Extension: bins (bin_stride*i, bin_stride*j) only, as [R, ceil(PH/s), ceil(PW/s), C] (NHWC); bit-identical to
    ``roi_align_forward(...)[:, :, ::s, ::s].permute(0, 2, 3, 1)``.  A channels-last ``input`` is pooled in place (no
    window staging: ``ovis_roi_align_forward_strided_from_nhwc_f32``), an NCHW one through the window-staging kernel."""
    if _channels_last_map(input, 4):
        rois = _dev(rois, "rois")
        oh, ow = -(-pooled_height // bin_stride), -(-pooled_width // bin_stride)
        out = torch.empty((rois.size(0), oh, ow, input.shape[1]), dtype=input.dtype, device=input.device)
        if out.numel():
            _strided_from_nhwc(input, rois, out, spatial_scale, pooled_height, pooled_width, sampling_ratio, bin_stride, False)
        return out
    input, rois = _dev(input, "input"), _dev(rois, "rois")
    if rois.dim() != 2 or rois.size(1) != 5 or input.dim() != 4:
        raise RuntimeError("roi_align_forward_strided_nhwc: expected input [N,C,H,W] and rois [R,5]")
    n, c, h, w = input.shape
    r = rois.size(0)
    oh, ow = -(-pooled_height // bin_stride), -(-pooled_width // bin_stride)
    out = torch.empty((r, oh, ow, c), dtype=input.dtype, device=input.device)
    if out.numel() == 0:
        return out
    with _on(input.device):
        rc = _L.ovis_roi_align_forward_strided_nhwc_f32(input.data_ptr(), rois.data_ptr(), out.data_ptr(), r, n, c, h,
                                                        w, pooled_height, pooled_width, bin_stride, spatial_scale,
                                                        sampling_ratio, _stream())
    _lib.check(rc, "roi_align_forward_strided_nhwc")
    return out


def roi_align_forward_strided_pair(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio, bin_stride):
    """``roi_align_forward_strided_nhwc`` written in pair layout: [R*oh*ow, 2*C] bf16 (exactly ``split_pair`` of the fp32
    bins), the operand of the res5 head's first split GEMM.  Returns (pair rows, (oh, ow)).  Channels-last inputs are
    pooled in place (see ``roi_align_forward_strided_nhwc``)."""
    if _channels_last_map(input, 32):
        rois = _dev(rois, "rois")
        oh, ow = -(-pooled_height // bin_stride), -(-pooled_width // bin_stride)
        out = torch.empty((rois.size(0) * oh * ow, 2 * input.shape[1]), dtype=torch.bfloat16, device=input.device)
        if out.numel():
            _strided_from_nhwc(input, rois, out, spatial_scale, pooled_height, pooled_width, sampling_ratio, bin_stride, True)
        return out, (oh, ow)
    input, rois = _dev(input, "input"), _dev(rois, "rois")
    if rois.dim() != 2 or rois.size(1) != 5 or input.dim() != 4:
        raise RuntimeError("roi_align_forward_strided_pair: expected input [N,C,H,W] and rois [R,5]")
    n, c, h, w = input.shape
    r = rois.size(0)
    oh, ow = -(-pooled_height // bin_stride), -(-pooled_width // bin_stride)
    out = torch.empty((r * oh * ow, 2 * c), dtype=torch.bfloat16, device=input.device)
    if out.numel():
        with _on(input.device):
            rc = _L.ovis_roi_align_forward_strided_pair_f32(input.data_ptr(), rois.data_ptr(), out.data_ptr(), r, n, c, h, w,
                                                            pooled_height, pooled_width, bin_stride, spatial_scale,
                                                            sampling_ratio, _stream())
        _lib.check(rc, "roi_align_forward_strided_pair")
    return out, (oh, ow)


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height,
                       width, sampling_ratio):
    if not grad.is_cuda:
        return _cpu.roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height,
                                       width, sampling_ratio)
    grad, rois = _dev(grad, "grad"), _dev(rois, "rois")
    r = rois.size(0)
    gin = torch.empty((batch_size, channels, height, width), dtype=grad.dtype, device=grad.device)
    if gin.numel() == 0:
        return gin
    with _on(grad.device):
        # plane-owner MFMA kernel when the H x W plane fits LDS (needs a per-call table workspace); the
        # library falls back to the window-gather + atomic kernel for larger maps
        nbytes = _L.ovis_roi_align_backward_workspace_bytes(r, batch_size, height, width)
        ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=grad.device)
        rc = _L.ovis_roi_align_backward_ws_f32(grad.data_ptr(), rois.data_ptr(), gin.data_ptr(), r, batch_size,
                                               channels, height, width, pooled_height, pooled_width,
                                               spatial_scale, sampling_ratio, ws.data_ptr(), nbytes, _stream())
    _lib.check(rc, "roi_align_backward")
    return gin


def roi_align_backward_strided(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height,
                               width, sampling_ratio, bin_stride):
    """Backward of ``roi_align_forward_strided_nhwc``: ``grad`` [R, C, ceil(PH/s), ceil(PW/s)] (the gradient of the bins
    the strided pooler produced) -> grad_input [N, C, H, W].  Returns None when the plane-owner kernel does not cover the
    shape (the caller scatters into a full tile and uses ``roi_align_backward``)."""
    grad, rois = _dev(grad, "grad"), _dev(rois, "rois")
    r = rois.size(0)
    gin = torch.empty((batch_size, channels, height, width), dtype=grad.dtype, device=grad.device)
    if gin.numel() == 0:
        return gin
    with _on(grad.device):
        nbytes = _L.ovis_roi_align_backward_workspace_bytes(r, batch_size, height, width)
        ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=grad.device)
        rc = _L.ovis_roi_align_backward_strided_ws_f32(grad.data_ptr(), rois.data_ptr(), gin.data_ptr(), r, batch_size,
                                                       channels, height, width, pooled_height, pooled_width, bin_stride,
                                                       spatial_scale, sampling_ratio, ws.data_ptr(), nbytes, _stream())
    if rc == -3:  # OVIS_ERANGE
        return None
    _lib.check(rc, "roi_align_backward_strided")
    return gin


def roi_align_backward_strided_nhwc(grad_nhwc, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                                    height, width, sampling_ratio, bin_stride):
    """Backward of ``roi_align_forward_strided_nhwc`` from the NHWC gradient itself: ``grad_nhwc`` [R, ceil(PH/s), ceil(PW/s), C]
    contiguous (as the res5 head's data-gradient GEMM leaves it) -> grad_input [N, C, H, W].  The library re-lays it into
    pre-split bf16 hi | lo tiles (no permute copy here, no split arithmetic in the plane-owner kernel).  Returns None when the
    shape is not covered (tiles above 8 x 8, planes that do not fit): the caller uses ``roi_align_backward_strided``."""
    grad_nhwc, rois = _dev(grad_nhwc, "grad"), _dev(rois, "rois")
    r, th, tw, c = grad_nhwc.shape
    if c != channels or r != rois.size(0):
        raise RuntimeError(f"roi_align_backward_strided_nhwc: gradient {tuple(grad_nhwc.shape)} for {rois.size(0)} RoIs x {channels} channels")
    if th > 8 or tw > 8:
        return None
    gin = torch.empty((batch_size, channels, height, width), dtype=grad_nhwc.dtype, device=grad_nhwc.device)
    if gin.numel() == 0:
        return gin
    with _on(grad_nhwc.device):
        nbytes = _L.ovis_roi_align_backward_strided_nhwc_workspace_bytes(r, batch_size, channels, height, width)
        ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=grad_nhwc.device)
        rc = _L.ovis_roi_align_backward_strided_nhwc_ws_f32(grad_nhwc.data_ptr(), rois.data_ptr(), gin.data_ptr(), r, batch_size,
                                                            channels, height, width, pooled_height, pooled_width, bin_stride,
                                                            spatial_scale, sampling_ratio, ws.data_ptr(), nbytes, _stream())
    if rc == -3:  # OVIS_ERANGE
        return None
    _lib.check(rc, "roi_align_backward_strided_nhwc")
    return gin


# ---- NMS (csrc/nms.h:10-28) -------------------------------------------------------------------
def nms_padded(dets, scores, threshold, ge_mode=False):
    """Sync-free form: returns (keep[K] int64 -- first n entries valid, ascending; n as a
    1-element int32 device tensor).  Extension of the reference API for device pipelines."""
    if not dets.is_cuda:
        return _cpu.nms_padded(dets, scores, threshold)
    dets, scores = _dev(dets, "dets"), _dev(scores, "scores")
    k = dets.size(0)
    keep = torch.empty((k,), dtype=torch.int64, device=dets.device)
    num = torch.zeros((1,), dtype=torch.int32, device=dets.device)
    if k == 0:
        return keep, num
    if dets.dim() != 2 or dets.size(1) != 4 or scores.numel() != k:
        raise RuntimeError("nms: expected dets [K,4] and scores [K]")
    with _on(dets.device):
        nbytes = _L.ovis_nms_workspace_bytes(k)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dets.device)
        rc = _L.ovis_nms_f32(dets.data_ptr(), scores.data_ptr(), k, threshold, int(bool(ge_mode)), ws.data_ptr(),
                             nbytes, keep.data_ptr(), num.data_ptr(), _stream())
    _lib.check(rc, "nms")
    return keep, num


def nms_grouped_padded(dets, scores, groups, threshold, ge_mode=False):
    """NMS inside every group of boxes in one launch (groups [K] integer labels): (keep[K] int64 -- first n entries
    valid, ascending indices into the K candidates; n as a 1-element int32 device tensor).  Equals ``nms`` run on
    every group separately (the per-class loop of box_head/inference.py:137-150)."""
    dets, scores = _dev(dets, "dets"), _dev(scores, "scores")
    groups = _dev(groups.to(torch.int32), "groups", torch.int32)
    k = dets.size(0)
    keep = torch.empty((k,), dtype=torch.int64, device=dets.device)
    num = torch.zeros((1,), dtype=torch.int32, device=dets.device)
    if k == 0:
        return keep, num
    if dets.dim() != 2 or dets.size(1) != 4 or scores.numel() != k or groups.numel() != k:
        raise RuntimeError("nms_grouped: expected dets [K,4], scores [K] and groups [K]")
    with _on(dets.device):
        nbytes = _L.ovis_nms_workspace_bytes(k)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dets.device)
        rc = _L.ovis_nms_grouped_f32(dets.data_ptr(), scores.data_ptr(), groups.data_ptr(), k, threshold,
                                     int(bool(ge_mode)), ws.data_ptr(), nbytes, keep.data_ptr(), num.data_ptr(), _stream())
    _lib.check(rc, "nms_grouped")
    return keep, num


def nms_grouped(dets, scores, groups, threshold):
    keep, num = nms_grouped_padded(dets, scores, groups, threshold)
    return keep[: int(num.item())]


def nms(dets, scores, threshold):
    if dets.is_cuda and dets.numel() == 0:
        # the reference returns a CPU tensor for the empty case (csrc/nms.h:17-18)
        return torch.empty((0,), dtype=torch.int64, device="cpu")
    keep, num = nms_padded(dets, scores, threshold)
    return keep[: int(num.item())]


def nms_presorted_batched(boxes, drop, threshold, below=0, ge_mode=False):
    """NMS of every image's score-sorted candidates in one mask + one reduce launch (``ovis_nms_presorted_batched_f32``):
    boxes [N, K, 4] in descending score order, drop [N, K] int32 (negative = removed in front of the NMS) or None ->
    (keep [N, K] int64: per image the survivors' indices ascending, zeros behind them; counts [N, 2] int32 = number of
    survivors, number of survivors with index < ``below``)."""
    boxes = _dev(boxes, "boxes")
    n, k = boxes.shape[0], boxes.shape[1]
    keep = torch.empty((n, k), dtype=torch.int64, device=boxes.device)
    counts = torch.zeros((n, 2), dtype=torch.int32, device=boxes.device)
    if n == 0 or k == 0:
        return keep, counts
    if boxes.dim() != 3 or boxes.size(2) != 4:
        raise RuntimeError("nms_presorted_batched: expected boxes [N,K,4]")
    if drop is not None:
        drop = _dev(drop, "drop", torch.int32)
        if drop.shape != (n, k):
            raise RuntimeError("nms_presorted_batched: drop must be [N,K] int32")
    with _on(boxes.device):
        nbytes = _nms_presorted_ws_bytes(n, k)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=boxes.device)
        rc = _L.ovis_nms_presorted_batched_f32(boxes.data_ptr(), 0 if drop is None else drop.data_ptr(), n, k, threshold,
                                               int(bool(ge_mode)), int(below), ws.data_ptr(), nbytes, keep.data_ptr(),
                                               counts.data_ptr(), _stream())
    _lib.check(rc, "nms_presorted_batched")
    return keep, counts


def topk_sorted(scores, k):
    """``scores.topk(k, dim=1, sorted=True)`` of a [N, A] score matrix in three launches (``ovis_topk_sorted_f32``: one device
    radix sort of the batch; rpn/inference.py:95) -> (values [N, k] descending, indices [N, k] int64).  Equal scores come in
    ascending index order."""
    if not scores.is_cuda or scores.dtype != torch.float32 or scores.dim() != 2:
        raise RuntimeError("topk_sorted: float32 HIP tensor [N, A] expected (the product path has no CPU fallback)")
    n, a = scores.shape
    if not 0 <= k <= a:
        raise RuntimeError(f"topk_sorted: k = {k} out of range for rows of {a}")
    if a > 1 and scores.stride(1) != 1:
        scores = scores.contiguous()
    vals = torch.empty((n, k), dtype=torch.float32, device=scores.device)
    idx = torch.empty((n, k), dtype=torch.int64, device=scores.device)
    if n == 0 or k == 0:
        return vals, idx
    need = _topk_ws_bytes(n, a)
    if need == 0:
        raise RuntimeError(f"topk_sorted: unsupported size [{n}, {a}]")
    ws = torch.empty((need,), dtype=torch.uint8, device=scores.device)
    with _on(scores.device):
        rc = _L.ovis_topk_sorted_f32(scores.data_ptr(), scores.stride(0) if n > 1 else a, n, a, k, vals.data_ptr(),
                                     idx.data_ptr(), ws.data_ptr(), need, _stream())
    _lib.check(rc, "topk_sorted")
    return vals, idx


def rpn_decode(box_regression, topk_idx, cell_anchors, image_wh, weights, xform_clip, min_size, anchor_stride):
    """Decode + clip + small-box flag of the top-k RPN candidates of a batch in one launch (``ovis_rpn_decode_f32``;
    rpn/inference.py:95-114).  box_regression [N, 4A, H, W] (any strides with stride(2) == W * stride(3): NCHW, or the
    NCHW view of an NHWC GEMM result), topk_idx [N, K] int64 into the (h, w, a) anchor order, cell_anchors [A, 4],
    image_wh [N, 2] float32 (width, height) -> (boxes [N, K, 4], drop [N, K] int32: -1 for boxes below min_size)."""
    if not box_regression.is_cuda or box_regression.dtype != torch.float32:
        raise RuntimeError("rpn_decode: float32 HIP tensor expected (the product path has no CPU fallback)")
    n, c4, h, w = box_regression.shape
    a = cell_anchors.shape[0]
    if c4 != 4 * a or topk_idx.dtype != torch.int64 or topk_idx.dim() != 2 or topk_idx.shape[0] != n:
        raise RuntimeError("rpn_decode: expected box_regression [N,4A,H,W], topk_idx [N,K] int64, cell_anchors [A,4]")
    if h > 1 and box_regression.stride(2) != w * box_regression.stride(3):
        box_regression = box_regression.contiguous()
    topk_idx = topk_idx.contiguous()
    cell_anchors = _dev(cell_anchors, "cell_anchors")
    image_wh = _dev(image_wh, "image_wh")
    k = topk_idx.shape[1]
    boxes = torch.empty((n, k, 4), dtype=torch.float32, device=box_regression.device)
    drop = torch.empty((n, k), dtype=torch.int32, device=box_regression.device)
    if n == 0 or k == 0:
        return boxes, drop
    wx, wy, ww, wh = weights
    with _on(box_regression.device):
        rc = _L.ovis_rpn_decode_f32(box_regression.data_ptr(), box_regression.stride(0), box_regression.stride(3),
                                    box_regression.stride(1), topk_idx.data_ptr(), cell_anchors.data_ptr(),
                                    image_wh.data_ptr(), n, k, a, w, float(anchor_stride), wx, wy, ww, wh, xform_clip,
                                    float(min_size), boxes.data_ptr(), drop.data_ptr(), _stream())
    _lib.check(rc, "rpn_decode")
    return boxes, drop


_BOX_DECODE_MAX_IMAGES = 16  # include/ovis_hip.h: OVIS_BOX_DECODE_MAX_IMAGES


def box_decode(rel_codes, boxes, weights, xform_clip, rows_per_image=None, image_sizes=None):
    """``BoxCoder.decode`` (modeling/box_coder.py:49-95) in one launch (``ovis_box_decode_f32``): rel_codes [R, 4K] (any row
    stride), boxes [R, 4] -> [R, 4K].  With ``rows_per_image`` (ints, image-major rows) and ``image_sizes`` ((width, height)
    per image) every row is also clipped to its image (``BoxList.clip_to_image``, structures/bounding_box.py:214-225)."""
    if not rel_codes.is_cuda or rel_codes.dtype != torch.float32 or rel_codes.dim() != 2 or rel_codes.shape[1] % 4:
        raise RuntimeError("box_decode: rel_codes must be a float32 HIP tensor [R, 4K] (the product path has no CPU fallback)")
    if boxes.shape != (rel_codes.shape[0], 4):
        raise RuntimeError("box_decode: boxes must be [R, 4]")
    boxes = boxes.to(torch.float32)
    if rel_codes.stride(1) != 1:
        rel_codes = rel_codes.contiguous()
    if boxes.stride(1) != 1:
        boxes = boxes.contiguous()
    r, k = rel_codes.shape[0], rel_codes.shape[1] // 4
    out = torch.empty((r, 4 * k), dtype=torch.float32, device=rel_codes.device)
    if r == 0:
        return out
    n = 0 if rows_per_image is None else len(rows_per_image)
    wx, wy, ww, wh_ = weights
    # image sizes travel as kernel arguments, at most _BOX_DECODE_MAX_IMAGES per launch: rows are image-major, so a longer
    # batch is a few launches over consecutive row ranges (as ovis_rois_from_boxes_f32 does internally)
    if n and (len(image_sizes) != n or sum(int(c) for c in rows_per_image) != r):
        raise RuntimeError(f"box_decode: rows_per_image must list {r} rows over {n} images with one (width, height) each")
    i0 = row0 = 0
    with _on(rel_codes.device):
        while True:
            m = min(n - i0, _BOX_DECODE_MAX_IMAGES)
            counts = wh = None
            rows = r
            if n:
                cs = [int(c) for c in rows_per_image[i0:i0 + m]]
                rows = sum(cs)
                counts = (ctypes.c_int32 * m)(*cs)
                wh = (ctypes.c_float * (2 * m))(*[float(v) for size in image_sizes[i0:i0 + m] for v in size])
            if rows:
                rc = _L.ovis_box_decode_f32(rel_codes.data_ptr() + 4 * row0 * rel_codes.stride(0), rel_codes.stride(0),
                                            boxes.data_ptr() + 4 * row0 * boxes.stride(0), boxes.stride(0), rows, k, wx, wy,
                                            ww, wh_, xform_clip, m, counts, wh, out.data_ptr() + 16 * k * row0, _stream())
                _lib.check(rc, "box_decode")
            i0 += m
            row0 += rows
            if i0 >= n:
                break
    return out


def rois_from_boxes(boxes, image_ids=None):
    """``Pooler.convert_to_roi_format`` (modeling/poolers.py:73-86) in one launch (``ovis_rois_from_boxes_f32``): a list of
    per-image [n_i, 4] float32 HIP tensors -> [sum n_i, 5] rows (image index, x1, y1, x2, y2); the image index of list
    entry i is ``image_ids[i]`` (default i)."""
    if not boxes:
        raise RuntimeError("rois_from_boxes: at least one image expected")
    boxes = [_dev(b, "boxes") for b in boxes]
    if any(b.dim() != 2 or b.shape[1] != 4 for b in boxes):
        raise RuntimeError("rois_from_boxes: [n, 4] tensors expected")
    n = len(boxes)
    rois = torch.empty((sum(b.shape[0] for b in boxes), 5), dtype=torch.float32, device=boxes[0].device)
    if rois.shape[0]:
        ptrs = (ctypes.c_void_p * n)(*[b.data_ptr() if b.shape[0] else None for b in boxes])
        counts = (ctypes.c_int32 * n)(*[b.shape[0] for b in boxes])
        ids = None if image_ids is None else (ctypes.c_int32 * n)(*[int(i) for i in image_ids])
        with _on(rois.device):
            rc = _L.ovis_rois_from_boxes_f32(ptrs, counts, ids, n, rois.data_ptr(), _stream())
        _lib.check(rc, "rois_from_boxes")
    return rois


def smooth_l1_picked_fwd_bwd(box_regression, regression_targets, positives, labels, column0, beta, denominator,
                             need_grad=True):
    """Box-regression loss of the box head (roi_heads/box_head/loss.py:147-170) with its gradient in one pass
    (``ovis_smooth_l1_picked_fwd_bwd_f32``): sum over the positives of smooth_l1(box_regression[p, col0 + c] -
    regression_targets[p, c]) / denominator, col0 = 4 * labels[p] (``labels`` given) or ``column0``.
    -> (loss scalar tensor, d loss / d box_regression [R, C] or None)."""
    if not box_regression.is_cuda or box_regression.dtype != torch.float32 or box_regression.dim() != 2:
        raise RuntimeError("smooth_l1_picked: float32 HIP tensor [R, C] expected (the product path has no CPU fallback)")
    if box_regression.stride(1) != 1:
        box_regression = box_regression.contiguous()
    regression_targets = _dev(regression_targets, "regression_targets")
    positives = _dev(positives, "positives", torch.int64)
    labels = None if labels is None else _dev(labels, "labels", torch.int64)
    r, c = box_regression.shape
    loss = torch.empty((1,), dtype=torch.float32, device=box_regression.device)
    grad = torch.empty((r, c), dtype=torch.float32, device=box_regression.device) if need_grad else None
    with _on(box_regression.device):
        rc = _L.ovis_smooth_l1_picked_fwd_bwd_f32(box_regression.data_ptr(), box_regression.stride(0), r, c,
                                                  regression_targets.data_ptr(), regression_targets.stride(0),
                                                  positives.data_ptr(), 0 if labels is None else labels.data_ptr(),
                                                  positives.numel(), column0, beta, float(denominator), loss.data_ptr(),
                                                  0 if grad is None else grad.data_ptr(), _stream())
    _lib.check(rc, "smooth_l1_picked_fwd_bwd")
    return loss[0], grad


# ---- sigmoid focal loss (csrc/SigmoidFocalLoss.h:10-41) ----------------------------------------
def sigmoid_focalloss_forward(logits, targets, num_classes, gamma, alpha):
    logits, targets = _dev(logits, "logits"), _dev(targets, "targets", torch.int32)
    if logits.dim() != 2:
        raise RuntimeError("logits should be NxClass")
    losses = torch.empty_like(logits)
    if losses.numel() == 0:
        return losses
    with _on(logits.device):
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
    with _on(logits.device):
        rc = _L.ovis_sigmoid_focal_loss_backward_f32(logits.data_ptr(), targets.data_ptr(), d_losses.data_ptr(),
                                                     d_logits.data_ptr(), logits.size(0), num_classes, gamma,
                                                     alpha, _stream())
    _lib.check(rc, "sigmoid_focalloss_backward")
    return d_logits


# ---- ROIPool (csrc/ROIPool.h:11-48) ---------------------------------------------------------------
def roi_pool_forward(input, rois, spatial_scale, pooled_height, pooled_width):
    """The reference's `_C.roi_pool_forward`: (output, argmax int32), exact vs cuda/ROIPool_cuda.cu:17-77."""
    input, rois = _dev(input, "input"), _dev(rois, "rois")
    if rois.dim() != 2 or rois.size(1) != 5 or input.dim() != 4:
        raise RuntimeError("roi_pool_forward: expected input [N,C,H,W] and rois [R,5]")
    n, c, h, w = input.shape
    r = rois.size(0)
    out = torch.empty((r, c, pooled_height, pooled_width), dtype=input.dtype, device=input.device)
    argmax = torch.zeros((r, c, pooled_height, pooled_width), dtype=torch.int32, device=input.device)
    if out.numel():
        with _on(input.device):
            rc = _L.ovis_roi_pool_forward_f32(input.data_ptr(), rois.data_ptr(), out.data_ptr(), argmax.data_ptr(), r, n, c, h,
                                              w, pooled_height, pooled_width, spatial_scale, _stream())
        _lib.check(rc, "roi_pool_forward")
    return out, argmax


def roi_pool_backward(grad, input, rois, argmax, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                      height, width):
    """The reference's `_C.roi_pool_backward` (cuda/ROIPool_cuda.cu:80-108); `input` / `spatial_scale` are part of
    the reference signature and unused, as upstream."""
    grad, rois = _dev(grad, "grad"), _dev(rois, "rois")
    argmax = argmax.to(torch.int32).contiguous()
    gin = torch.empty((batch_size, channels, height, width), dtype=grad.dtype, device=grad.device)
    if gin.numel():
        with _on(grad.device):
            rc = _L.ovis_roi_pool_backward_f32(grad.data_ptr(), argmax.data_ptr(), rois.data_ptr(), gin.data_ptr(),
                                               rois.size(0), batch_size, channels, height, width, pooled_height,
                                               pooled_width, _stream())
        _lib.check(rc, "roi_pool_backward")
    return gin


# ---- deformable position-sensitive RoI pooling (csrc/deform_pool.h:11-70) -----------------------------
def _psroi_dims(input, bbox, trans, out, no_trans):
    for t, name in ((input, "input"), (bbox, "bbox"), (out, "out")):
        if not (t.is_cuda and t.dtype == torch.float32):
            raise RuntimeError(f"deform_psroi_pooling: {name} must be a float32 HIP tensor: this package has no CPU implementation")
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")  # deform_pool_cuda.cu:45
    if bbox.size(0) != out.size(0):
        raise RuntimeError(f"Output shape and bbox number wont match: ({out.size(0)} vs {bbox.size(0)}).")
    channels_trans = 2 if no_trans else trans.size(1)
    return input.size(0), input.size(1), input.size(2), input.size(3), channels_trans, bbox.size(0)


def deform_psroi_pooling_forward(input, bbox, trans, out, top_count, no_trans, spatial_scale, output_dim, group_size,
                                 pooled_size, part_size, sample_per_part, trans_std):
    """The reference's `_C.deform_psroi_pooling_forward` (deform_pool.h:11-38): fills ``out`` and ``top_count`` in place."""
    batch, channels, height, width, channels_trans, n = _psroi_dims(input, bbox, trans, out, no_trans)
    bbox = bbox.contiguous()
    tr = 0 if no_trans else _dev(trans, "trans").data_ptr()
    if not (out.is_contiguous() and top_count.is_contiguous() and top_count.dtype == torch.float32):
        raise RuntimeError("deform_psroi_pooling_forward: out / top_count must be contiguous float32")
    with _on(input.device):
        rc = _L.ovis_deform_psroi_pool_forward_f32(input.data_ptr(), bbox.data_ptr(), tr, out.data_ptr(),
                                                   top_count.data_ptr(), n, batch, channels, height, width, channels_trans,
                                                   int(bool(no_trans)), spatial_scale, output_dim, group_size,
                                                   pooled_size, part_size, sample_per_part, trans_std, _stream())
    _lib.check(rc, "deform_psroi_pooling_forward")


def deform_psroi_pooling_backward(out_grad, input, bbox, trans, top_count, input_grad, trans_grad, no_trans,
                                  spatial_scale, output_dim, group_size, pooled_size, part_size, sample_per_part,
                                  trans_std):
    """The reference's `_C.deform_psroi_pooling_backward` (deform_pool.h:40-70): ACCUMULATES into ``input_grad`` and
    ``trans_grad`` (zero-filled by the caller, deform_pool_func.py:68-70)."""
    if not out_grad.is_contiguous():
        raise RuntimeError("out_grad tensor has to be contiguous")  # deform_pool_cuda.cu:71
    batch, channels, height, width, channels_trans, n = _psroi_dims(input, bbox, trans, out_grad, no_trans)
    bbox = bbox.contiguous()
    if not (input_grad.is_cuda and input_grad.is_contiguous() and input_grad.shape == input.shape):
        raise RuntimeError("deform_psroi_pooling_backward: input_grad must be a contiguous HIP tensor shaped like input")
    tr = tg = 0
    if not no_trans:
        tr = _dev(trans, "trans").data_ptr()
        if not (trans_grad.is_cuda and trans_grad.is_contiguous() and trans_grad.shape == trans.shape):
            raise RuntimeError("deform_psroi_pooling_backward: trans_grad must be a contiguous HIP tensor shaped like trans")
        tg = trans_grad.data_ptr()
    with _on(input.device):
        rc = _L.ovis_deform_psroi_pool_backward_f32(out_grad.data_ptr(), top_count.data_ptr(), input.data_ptr(),
                                                    bbox.data_ptr(), tr, input_grad.data_ptr(), tg, n, batch, channels,
                                                    height, width, channels_trans, int(bool(no_trans)), spatial_scale,
                                                    output_dim, group_size, pooled_size, part_size, sample_per_part,
                                                    trans_std, _stream())
    _lib.check(rc, "deform_psroi_pooling_backward")


# ---- cross-modal head + student losses (extensions beyond vision.cpp; include/ovis_hip.h) --------------
def split_bf16x3(x, mode):
    """x [rows, cols] f32 (row-strided view ok) -> [rows, 3*cols] bf16, rows = [hi|hi|lo] (mode 0) / [hi|lo|hi]
    (mode 1).  See include/ovis_hip.h."""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2:
        raise RuntimeError("split_bf16x3: 2-D float32 HIP tensor expected")
    if x.stride(1) != 1:
        x = x.contiguous()
    rows, cols = x.shape
    out = torch.empty((rows, 3 * cols), dtype=torch.bfloat16, device=x.device)
    if out.numel() == 0:
        return out
    with _on(x.device):
        rc = _L.ovis_split_bf16x3_f32(x.data_ptr(), x.stride(0), out.data_ptr(), rows, cols, mode, _stream())
    _lib.check(rc, "split_bf16x3")
    return out


def im2col_split_bf16x3(x, kh, kw, flip=False):
    """x [R, H, W, C] f32 contiguous (NHWC) -> [R*H*W, 3*kh*kw*C] bf16 rows [hi taps | hi taps | lo taps]
    (stride 1, zero "same" padding; flip = taps in reverse order).  See include/ovis_hip.h."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        raise RuntimeError("im2col_split_bf16x3: contiguous [R,H,W,C] float32 HIP tensor expected")
    r, h, w, c = x.shape
    out = torch.empty((r * h * w, 3 * kh * kw * c), dtype=torch.bfloat16, device=x.device)
    if out.numel() == 0:
        return out
    with _on(x.device):
        rc = _L.ovis_im2col_split_bf16x3_f32(x.data_ptr(), out.data_ptr(), r, h, w, c, kh, kw, int(bool(flip)), _stream())
    _lib.check(rc, "im2col_split_bf16x3")
    return out


def match_encode(gt_boxes, gt_labels, proposals, high, low, weights=None, between_keeps_label=False):
    """IoU match + labels (+ box-delta targets when ``weights`` is given) in one launch; see include/ovis_hip.h.
    Returns (matched_idx int64 [P], labels int64 [P], regression_targets f32 [P,4] or None)."""
    gt_boxes, proposals = _dev(gt_boxes, "gt_boxes"), _dev(proposals, "proposals")
    gt_labels = gt_labels.to(torch.int64).contiguous()
    g, p = gt_boxes.shape[0], proposals.shape[0]
    if g == 0:
        raise ValueError("No ground-truth boxes available for one of the images during training")
    idx = torch.empty((p,), dtype=torch.int64, device=proposals.device)
    lab = torch.empty((p,), dtype=torch.int64, device=proposals.device)
    reg = torch.empty((p, 4), dtype=torch.float32, device=proposals.device) if weights is not None else None
    if p:
        wx, wy, ww, wh = weights if weights is not None else (1.0, 1.0, 1.0, 1.0)
        with _on(proposals.device):
            rc = _L.ovis_match_encode_f32(gt_boxes.data_ptr(), gt_labels.data_ptr(), proposals.data_ptr(), g, p, high, low,
                                          int(bool(between_keeps_label)), wx, wy, ww, wh, idx.data_ptr(), lab.data_ptr(),
                                          0 if reg is None else reg.data_ptr(), _stream())
        _lib.check(rc, "match_encode")
    return idx, lab, reg


def rpn_match_encode(gt_boxes, anchors, visibility, high_threshold, low_threshold, allow_low_quality_matches, weights):
    """The RPN's per-image targets in two launches (``ovis_rpn_match_encode_f32``; rpn/loss.py:21-89): gt_boxes [G, 4],
    anchors [A, 4], visibility [A] bool -> (labels [A] int64: 1 / 0 / -1, regression targets [A, 4])."""
    gt_boxes, anchors = _dev(gt_boxes, "gt_boxes"), _dev(anchors, "anchors")
    if not (visibility.is_cuda and visibility.dtype == torch.bool and visibility.shape == (anchors.shape[0],)):
        raise RuntimeError("rpn_match_encode: visibility must be a bool HIP tensor [A]")
    visibility = visibility.contiguous()
    g, a = gt_boxes.shape[0], anchors.shape[0]
    if g == 0:
        raise ValueError("No ground-truth boxes available for one of the images during training")
    labels = torch.empty((a,), dtype=torch.int64, device=anchors.device)
    reg = torch.empty((a, 4), dtype=torch.float32, device=anchors.device)
    scratch = torch.empty((g,), dtype=torch.int32, device=anchors.device)
    wx, wy, ww, wh = weights
    with _on(anchors.device):
        rc = _L.ovis_rpn_match_encode_f32(gt_boxes.data_ptr(), anchors.data_ptr(), visibility.data_ptr(), g, a, high_threshold,
                                          low_threshold, int(bool(allow_low_quality_matches)), wx, wy, ww, wh, scratch.data_ptr(),
                                          labels.data_ptr(), reg.data_ptr(), _stream())
    _lib.check(rc, "rpn_match_encode")
    return labels, reg


def project_masks(masks, gt_index, boxes, resolution):
    """masks [G,H,W] bool / uint8, gt_index [P] int64, boxes [P,4] -> [P, M, M] f32 mask targets (one launch)."""
    if not (masks.is_cuda and masks.dim() == 3 and masks.dtype in (torch.bool, torch.uint8)):
        raise RuntimeError("project_masks: [G,H,W] bool / uint8 HIP tensor expected")
    masks = masks.contiguous()
    boxes = _dev(boxes, "boxes")
    gt_index = gt_index.to(torch.int64).contiguous()
    p = boxes.shape[0]
    out = torch.empty((p, resolution, resolution), dtype=torch.float32, device=boxes.device)
    if p:
        with _on(boxes.device):
            rc = _L.ovis_project_masks_f32(masks.data_ptr(), gt_index.data_ptr(), boxes.data_ptr(), p, masks.shape[1],
                                           masks.shape[2], resolution, int(masks.dtype == torch.bool), out.data_ptr(),
                                           _stream())
        _lib.check(rc, "project_masks")
    return out


def sample_fg_bg(labels, batch_size, max_positives, seed):
    """fg / bg sampling of one image on the device (``ovis_sample_fg_bg``; balanced_positive_negative_sampler.py:19-68):
    labels [P] int64 -> (selected [batch_size] int64: chosen indices ascending, zero padded; positive_slots [batch_size]:
    positions of the positives inside ``selected``; counts [2] int32: number selected, positives among them)."""
    if not labels.is_cuda or labels.dtype != torch.int64 or labels.dim() != 1:
        raise RuntimeError("sample_fg_bg: 1-D int64 HIP tensor expected (the product path has no CPU fallback)")
    labels = labels.contiguous()
    sel = torch.empty((batch_size,), dtype=torch.int64, device=labels.device)
    slots = torch.empty((batch_size,), dtype=torch.int64, device=labels.device)
    counts = torch.empty((2,), dtype=torch.int32, device=labels.device)
    with _on(labels.device):
        rc = _L.ovis_sample_fg_bg(labels.data_ptr(), labels.numel(), int(batch_size), int(max_positives),
                                  int(seed) & 0xFFFFFFFFFFFFFFFF, sel.data_ptr(), slots.data_ptr(), counts.data_ptr(), _stream())
    _lib.check(rc, "sample_fg_bg")
    return sel, slots, counts


def gather_rows(index, boxes_a=None, boxes_b=None, ints_a=None, ints_b=None):
    """``x.index_select(0, index)`` for up to two [P, 4] float32 and two [P] int64 tensors in one launch
    (``ovis_gather_rows``) -> tuple of the gathered tensors (None where the source is None)."""
    index = _dev(index, "index", torch.int64)
    n = index.numel()
    srcs = [None if t is None else _dev(t, "boxes", torch.float32) for t in (boxes_a, boxes_b)] + \
           [None if t is None else _dev(t, "ints", torch.int64) for t in (ints_a, ints_b)]
    for t in srcs[:2]:
        if t is not None and (t.dim() != 2 or t.shape[1] != 4):
            raise RuntimeError("gather_rows: float sources must be [P, 4]")
    for t in srcs[2:]:
        if t is not None and t.dim() != 1:
            raise RuntimeError("gather_rows: int64 sources must be [P]")
    outs = [None if t is None else torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in srcs]
    if n:
        ptr = lambda t: 0 if t is None else t.data_ptr()
        with _on(index.device):
            rc = _L.ovis_gather_rows(index.data_ptr(), n, ptr(srcs[0]), ptr(outs[0]), ptr(srcs[1]), ptr(outs[1]), ptr(srcs[2]),
                                     ptr(outs[2]), ptr(srcs[3]), ptr(outs[3]), _stream())
        _lib.check(rc, "gather_rows")
    return tuple(outs)


def project_pasted_masks(mask_probs, gt_boxes, gt_index, boxes, image_size, resolution, threshold=0.5):
    """Mask targets [P, resolution, resolution] for boxes [P, 4] whose ground truth gt_index[p] has its binary mask
    defined by (mask_probs [G, M, M], gt_boxes [G, 4]) through the Masker paste (``ovis_project_pasted_masks_f32``);
    image_size = (height, width)."""
    mask_probs, gt_boxes, boxes = _dev(mask_probs, "mask_probs"), _dev(gt_boxes, "gt_boxes"), _dev(boxes, "boxes")
    gt_index = _dev(gt_index, "gt_index", torch.int64)
    p = boxes.shape[0]
    out = torch.empty((p, resolution, resolution), dtype=torch.float32, device=boxes.device)
    if p == 0:
        return out
    if mask_probs.dim() != 3 or mask_probs.shape[1] != mask_probs.shape[2] or gt_boxes.shape != (mask_probs.shape[0], 4):
        raise RuntimeError("project_pasted_masks: expected mask_probs [G,M,M] and gt_boxes [G,4]")
    with _on(boxes.device):
        rc = _L.ovis_project_pasted_masks_f32(mask_probs.data_ptr(), gt_boxes.data_ptr(), gt_index.data_ptr(),
                                              boxes.data_ptr(), p, int(image_size[0]), int(image_size[1]),
                                              mask_probs.shape[1], int(resolution), float(threshold), out.data_ptr(), _stream())
    _lib.check(rc, "project_pasted_masks")
    return out


def split_pair(x):
    """x [rows, cols] f32 (row-strided view ok, cols % 32 == 0) -> pair layout [rows, 2*cols] bf16: per 32 values
    [hi(32) | lo(32)].  See include/ovis_hip.h."""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2:
        raise RuntimeError("split_pair: 2-D float32 HIP tensor expected")
    if x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16:
        x = x.contiguous()
    rows, cols = x.shape
    out = torch.empty((rows, 2 * cols), dtype=torch.bfloat16, device=x.device)
    if out.numel() == 0:
        return out
    with _on(x.device):
        rc = _L.ovis_split_pair_f32(x.data_ptr(), x.stride(0), out.data_ptr(), rows, cols, _stream())
    _lib.check(rc, "split_pair")
    return out


def gate_split_pair(dy, gate=None, want_f32=False, pooled=None, pool_rows=0, selected=None, group_slot=None):
    """g = (dy + pooled[row // pool_rows] / pool_rows) * (y > 0) in pair layout (and as fp32 when ``want_f32``); gate =
    y as fp32 [rows, cols] or as its pair form [rows, 2*cols] bf16 (contiguous), None = no gate; ``pooled`` = the
    gradient of a mean over every ``pool_rows`` consecutive rows ([rows / pool_rows, cols] f32) or None; ``dy`` may be
    None when ``pooled`` is given.  ``selected`` [S * pool_rows, cols] f32 with ``group_slot`` [rows / pool_rows] int32
    (index into ``selected`` of a group of pool_rows rows, -1 = none): the gradient of a gather of whole row groups, added
    on the fly instead of being scattered into a zero tensor first.  Returns (g_pair, g_f32 or None)."""
    if dy is None and pooled is None:
        raise RuntimeError("gate_split_pair: dy or pooled must be given")
    if (selected is None) != (group_slot is None):
        raise RuntimeError("gate_split_pair: selected and group_slot are given together")
    if pooled is not None:
        if not (pooled.is_cuda and pooled.dtype == torch.float32 and pooled.dim() == 2):
            raise RuntimeError("gate_split_pair: pooled must be a 2-D float32 HIP tensor")
        pooled = pooled.contiguous()
    if dy is not None:
        if not (dy.is_cuda and dy.dtype == torch.float32 and dy.dim() == 2):
            raise RuntimeError("gate_split_pair: 2-D float32 HIP tensor expected")
        if dy.stride(1) != 1 or dy.stride(0) % 4 or dy.data_ptr() % 16:
            dy = dy.contiguous()
        rows, cols = dy.shape
    else:
        rows, cols = pooled.shape[0] * pool_rows, pooled.shape[1]
    if pooled is not None and (pool_rows <= 0 or pooled.shape != (rows // max(pool_rows, 1), cols) or rows % pool_rows):
        raise RuntimeError("gate_split_pair: pooled must be [rows / pool_rows, cols]")
    dev = dy.device if dy is not None else pooled.device
    is_pair = 0
    if gate is not None:
        is_pair = int(gate.dtype == torch.bfloat16)
        if not (gate.is_cuda and gate.is_contiguous() and gate.shape == (rows, cols * (2 if is_pair else 1))
                and gate.dtype in (torch.bfloat16, torch.float32)):
            raise RuntimeError("gate_split_pair: gate must be the contiguous fp32 or pair form of the forward output")
    out = torch.empty((rows, 2 * cols), dtype=torch.bfloat16, device=dev)
    g32 = torch.empty((rows, cols), dtype=torch.float32, device=dev) if want_f32 else None
    if out.numel() == 0:
        return out, g32
    if selected is not None:
        selected, group_slot = _dev(selected, "selected"), _dev(group_slot, "group_slot", torch.int32)
        if pool_rows <= 0 or rows % pool_rows or selected.dim() != 2 or selected.shape[1] != cols or \
                selected.shape[0] % pool_rows or group_slot.numel() != rows // pool_rows:
            raise RuntimeError("gate_split_pair: selected must be [S * pool_rows, cols] and group_slot [rows / pool_rows]")
    with _on(dev):
        rc = _L.ovis_gate_split_pair_rows_f32(0 if dy is None else dy.data_ptr(), 0 if dy is None else dy.stride(0),
                                              0 if gate is None else gate.data_ptr(), is_pair, out.data_ptr(),
                                              0 if g32 is None else g32.data_ptr(), rows, cols,
                                              0 if pooled is None else pooled.data_ptr(), pool_rows,
                                              0 if selected is None else selected.data_ptr(),
                                              0 if group_slot is None else group_slot.data_ptr(), _stream())
    _lib.check(rc, "gate_split_pair")
    return out, g32


def im2col_pair(xp, h, w, kh, kw):
    """xp [R*h*w, 2*C] pair rows of an NHWC tensor -> [R*h*w, 2*kh*kw*C] pair rows (tap-major, zero rows outside
    the map): the M-contracting operand of a 3x3 weight gradient."""
    if not (xp.is_cuda and xp.dtype == torch.bfloat16 and xp.dim() == 2 and xp.is_contiguous()):
        raise RuntimeError("im2col_pair: contiguous 2-D bfloat16 HIP tensor (pair layout) expected")
    m, c2 = xp.shape
    if m % (h * w):
        raise RuntimeError("im2col_pair: rows must be a multiple of h*w")
    out = torch.empty((m, kh * kw * c2), dtype=torch.bfloat16, device=xp.device)
    if out.numel() == 0:
        return out
    with _on(xp.device):
        rc = _L.ovis_im2col_pair(xp.data_ptr(), out.data_ptr(), m // (h * w), h, w, c2 // 2, kh, kw, _stream())
    _lib.check(rc, "im2col_pair")
    return out


def im2col_nchw_pair(x, kh, kw, stride, pad):
    """x [N, C, H, W] f32 contiguous -> (pair rows [N*Ho*Wo, 2*Kp] of the strided convolution's patches, k =
    (ky*kw + kx)*C + c zero-padded to Kp % 32 == 0, (Ho, Wo)).  See include/ovis_hip.h."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        raise RuntimeError("im2col_nchw_pair: contiguous [N,C,H,W] float32 HIP tensor expected")
    n, c, h, w = x.shape
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    kp = -(-(kh * kw * c) // 32) * 32
    out = torch.empty((n * ho * wo, 2 * kp), dtype=torch.bfloat16, device=x.device)
    if out.numel():
        with _on(x.device):
            rc = _L.ovis_im2col_nchw_pair_f32(x.data_ptr(), out.data_ptr(), n, c, h, w, kh, kw, stride, pad, kp, _stream())
        _lib.check(rc, "im2col_nchw_pair")
    return out, (ho, wo)


_GEMM_TLS = threading.local()


@contextlib.contextmanager
def co_scheduled(on=True):
    """Split-GEMM launches issued inside run BESIDE other work (the frozen half of the student-teacher step on its side stream,
    ``engine/trainer.py``): the library then cuts K for least total work instead of least duration (config bit 16 of
    ``ovis_split_gemm_pair*``).  Per thread: the look-ahead half may be issued by a worker thread."""
    prev = getattr(_GEMM_TLS, "co", False)
    _GEMM_TLS.co = bool(on)
    try:
        yield
    finally:
        _GEMM_TLS.co = prev


# Size queries of the library are pure functions of their arguments (shape + config bits): asked once per distinct
# argument tuple, not once per launch -- 98 ctypes round trips per student step for the split GEMM's workspace alone.
@functools.lru_cache(maxsize=None)
def _gemm_ws_bytes(m, n, ch, ch2, kh, kw, w, config):
    return int(_L.ovis_split_gemm_pair_workspace_bytes_ex(m, n, ch, ch2, kh, kw, w, config))


@functools.lru_cache(maxsize=None)
def _tn_slices(m, n, ch, taps):
    return int(_L.ovis_split_gemm_tn_slices(m, n, ch, taps))


@functools.lru_cache(maxsize=None)
def _gemm_f32_ws_bytes(m, n, k):
    return int(_L.ovis_gemm_f32_workspace_bytes(m, n, k))


@functools.lru_cache(maxsize=None)
def _nms_presorted_ws_bytes(n, k):
    return int(_L.ovis_nms_presorted_workspace_bytes(n, k))


@functools.lru_cache(maxsize=None)
def _topk_ws_bytes(n, a):
    return int(_L.ovis_topk_sorted_workspace_bytes(n, a))


def _gemm_cfg(config):
    return config | (0x10000 if getattr(_GEMM_TLS, "co", False) else 0)


def split_gemm_pair(a_pair, b_pair, bias=None, residual=None, relu=False, out_f32=True, out_pair=False, conv=None,
                    config=0, a2_pair=None, residual_pair=None):
    """act(A @ B^T + bias + residual) with A [M, 2*ch] / B [N, 2*K] in pair layout (``split_pair``); fp32-accurate
    three-term bf16 hi/lo product on the matrix cores (csrc/split_gemm.hip).  conv = (h, w, kh, kw, flip): A is an
    NHWC tensor [M/(h*w), h, w, ch] and B holds [N, kh*kw*ch] tap-major weights -- stride-1 "same" convolution as an
    implicit GEMM.  a2_pair [M, 2*ch2]: a second operand whose products follow A's along K (B = [N, 2*(ch + ch2)]):
    conv3 + projection shortcut of a bottleneck as one product.  ``config``: 0 = choose; test / measurement bits as in
    include/ovis_hip.h.  Returns (C f32 [M, N] or None, C in pair layout [M, 2*N] or None)."""
    for t, name in ((a_pair, "a_pair"), (b_pair, "b_pair")) + (((a2_pair, "a2_pair"),) if a2_pair is not None else ()):
        if not (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1):
            raise RuntimeError(f"split_gemm_pair: {name} must be a 2-D bfloat16 HIP tensor in pair layout")
    m, ch = a_pair.shape[0], a_pair.shape[1] // 2
    n, k = b_pair.shape[0], b_pair.shape[1] // 2
    ch2 = 0
    if a2_pair is not None:
        if conv is not None or a2_pair.shape[0] != m:
            raise RuntimeError("split_gemm_pair: a2_pair needs a plain product and as many rows as a_pair")
        ch2 = a2_pair.shape[1] // 2
    h = w = 0
    kh = kw = 1
    flip = False
    if conv is not None:
        h, w, kh, kw, flip = conv
        if m % (h * w):
            raise RuntimeError("split_gemm_pair: rows must be a multiple of h*w")
    if k != kh * kw * ch + ch2:
        raise RuntimeError(f"split_gemm_pair: contraction mismatch (A has {ch} channels x {kh * kw} taps + {ch2}, B has {k})")
    dev = a_pair.device
    c = torch.empty((m, n), dtype=torch.float32, device=dev) if out_f32 else None
    cp = torch.empty((m, 2 * n), dtype=torch.bfloat16, device=dev) if out_pair else None
    if m == 0 or n == 0:
        return c, cp
    if bias is not None:
        bias = _dev(bias, "bias")
    if residual_pair is not None:
        # shortcut in pair layout (hi + lo, exact in fp32): plain 1x1 products only
        if residual is not None or conv is not None or a2_pair is not None or n % 32 or not (
                residual_pair.is_cuda and residual_pair.dtype == torch.bfloat16 and residual_pair.dim() == 2
                and residual_pair.stride(1) == 1 and residual_pair.shape == (m, 2 * n)):
            raise RuntimeError("split_gemm_pair: residual_pair must be [M, 2N] bfloat16 pair rows of a plain product")
        with _on(dev):
            config = _gemm_cfg(config)
            nbytes = _gemm_ws_bytes(m, n, ch, 0, 1, 1, 0, config) if not (config & 8) else 0
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev) if nbytes else None
            rc = _L.ovis_split_gemm_pair_rp(a_pair.data_ptr(), 2 * a_pair.stride(0), b_pair.data_ptr(), 2 * b_pair.stride(0),
                                            0 if c is None else c.data_ptr(), n, 0 if cp is None else cp.data_ptr(), 4 * n,
                                            0 if bias is None else bias.data_ptr(), residual_pair.data_ptr(),
                                            2 * residual_pair.stride(0), m, n, ch, int(bool(relu)),
                                            0 if ws is None else ws.data_ptr(), nbytes, config, _stream())
        _lib.check(rc, "split_gemm_pair_rp")
        return c, cp
    if residual is not None and not (residual.is_cuda and residual.dtype == torch.float32 and residual.dim() == 2
                                     and residual.stride(1) == 1 and residual.shape == (m, n)):
        raise RuntimeError("split_gemm_pair: residual must be a float32 [M, N] HIP tensor with unit column stride")
    with _on(dev):
        config = _gemm_cfg(config)
        nbytes = _gemm_ws_bytes(m, n, ch, ch2, kh, kw, w, config) if not (config & 8) else 0
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev) if nbytes else None
        rc = _L.ovis_split_gemm_pair(a_pair.data_ptr(), 2 * a_pair.stride(0),
                                     0 if a2_pair is None else a2_pair.data_ptr(),
                                     0 if a2_pair is None else 2 * a2_pair.stride(0),
                                     b_pair.data_ptr(), 2 * b_pair.stride(0),
                                     0 if c is None else c.data_ptr(), n, 0 if cp is None else cp.data_ptr(), 4 * n,
                                     0 if bias is None else bias.data_ptr(),
                                     0 if residual is None else residual.data_ptr(),
                                     0 if residual is None else residual.stride(0), m, n, ch, ch2, kh, kw, h, w,
                                     int(bool(flip)), int(bool(relu)), 0 if ws is None else ws.data_ptr(), nbytes,
                                     config, _stream())
    _lib.check(rc, "split_gemm_pair")
    return c, cp


def split_gemm_pair_pool_supported(m, n, ch, pool_rows):
    return bool(_L.ovis_split_gemm_pair_pool_supported(int(m), int(n), int(ch), int(pool_rows)))


def split_gemm_pair_rp_pool(a_pair, b_pair, bias, residual_pair, relu, out_f32, out_pair, pool_rows):
    """``split_gemm_pair(..., residual_pair=...)`` of the LAST bottleneck of a res5 chain with the head's average pooling in the
    epilogue (``ovis_split_gemm_pair_rp_pool``): also returns the mean of the result over every group of ``pool_rows`` rows
    ([M / pool_rows, N] fp32); with ``out_f32`` and ``out_pair`` both False the result itself is never written.
    -> (C f32 or None, C pair or None, pooled)."""
    for t, name in ((a_pair, "a_pair"), (b_pair, "b_pair")) + (((residual_pair, "residual_pair"),) if residual_pair is not None else ()):
        if not (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1):
            raise RuntimeError(f"split_gemm_pair_rp_pool: {name} must be a 2-D bfloat16 HIP tensor in pair layout")
    m, ch = a_pair.shape[0], a_pair.shape[1] // 2
    n, k = b_pair.shape[0], b_pair.shape[1] // 2
    if k != ch or m % pool_rows or (residual_pair is not None and residual_pair.shape != (m, 2 * n)):
        raise RuntimeError("split_gemm_pair_rp_pool: shape mismatch")
    dev = a_pair.device
    c = torch.empty((m, n), dtype=torch.float32, device=dev) if out_f32 else None
    cp = torch.empty((m, 2 * n), dtype=torch.bfloat16, device=dev) if out_pair else None
    pooled = torch.zeros((m // pool_rows, n), dtype=torch.float32, device=dev)
    if m == 0 or n == 0:
        return c, cp, pooled
    if bias is not None:
        bias = _dev(bias, "bias")
    with _on(dev):
        rc = _L.ovis_split_gemm_pair_rp_pool(a_pair.data_ptr(), 2 * a_pair.stride(0), b_pair.data_ptr(), 2 * b_pair.stride(0),
                                             0 if c is None else c.data_ptr(), n, 0 if cp is None else cp.data_ptr(), 4 * n,
                                             0 if bias is None else bias.data_ptr(),
                                             0 if residual_pair is None else residual_pair.data_ptr(),
                                             0 if residual_pair is None else 2 * residual_pair.stride(0), m, n, ch,
                                             int(bool(relu)), pooled.data_ptr(), int(pool_rows), 1.0 / pool_rows, _stream())
    _lib.check(rc, "split_gemm_pair_rp_pool")
    return c, cp, pooled


def weight_prep_pair(w, scale=None, want_transposed=False):
    """w [N, C, KH, KW] f32 (times scale[N]) -> (pair [N, 2*KH*KW*C] tap-major, transposed pair [C, 2*KH*KW*N] or None)
    in one launch; see include/ovis_hip.h."""
    if not (w.is_cuda and w.dtype == torch.float32 and w.dim() == 4):
        raise RuntimeError("weight_prep_pair: [N,C,KH,KW] float32 HIP tensor expected")
    w = w.detach().contiguous()
    n, c, kh, kw = w.shape
    t = kh * kw
    fwd = torch.empty((n, 2 * t * c), dtype=torch.bfloat16, device=w.device)
    bwd = torch.empty((c, 2 * t * n), dtype=torch.bfloat16, device=w.device) if want_transposed else None
    if scale is not None:
        scale = _dev(scale.detach(), "scale")
    with _on(w.device):
        rc = _L.ovis_weight_prep_pair_f32(w.data_ptr(), 0 if scale is None else scale.data_ptr(), fwd.data_ptr(),
                                          0 if bwd is None else bwd.data_ptr(), n, c, t, _stream())
    _lib.check(rc, "weight_prep_pair")
    return fwd, bwd


def split_gemm_pair_gated(a_pair, b_pair, gate_pair, conv=None, out_f32=False, out_pair=True, config=0):
    """(A @ B^T) * (y > 0) with y given in pair layout (``gate_pair`` [M, 2N]): the data gradient of a layer behind a
    ReLU, gated and split for the next backward GEMM in the epilogue.  Returns (f32 or None, pair or None)."""
    for t, name in ((a_pair, "a_pair"), (b_pair, "b_pair"), (gate_pair, "gate_pair")):
        if not (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1):
            raise RuntimeError(f"split_gemm_pair_gated: {name} must be a 2-D bfloat16 HIP tensor in pair layout")
    m, ch = a_pair.shape[0], a_pair.shape[1] // 2
    n, k = b_pair.shape[0], b_pair.shape[1] // 2
    h = w = 0
    kh = kw = 1
    flip = False
    if conv is not None:
        h, w, kh, kw, flip = conv
    if k != kh * kw * ch or gate_pair.shape != (m, 2 * n):
        raise RuntimeError("split_gemm_pair_gated: shape mismatch")
    dev = a_pair.device
    c = torch.empty((m, n), dtype=torch.float32, device=dev) if out_f32 else None
    cp = torch.empty((m, 2 * n), dtype=torch.bfloat16, device=dev) if out_pair else None
    if m == 0 or n == 0:
        return c, cp
    with _on(dev):
        # under-filled grids take the plan's K slices (the slab reduction applies the gate)
        config = _gemm_cfg(config)
        nbytes = _gemm_ws_bytes(m, n, ch, 0, kh, kw, w, config) if not (config & 8) else 0
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev) if nbytes else None
        rc = _L.ovis_split_gemm_pair_gated_ws(a_pair.data_ptr(), 2 * a_pair.stride(0), b_pair.data_ptr(),
                                              2 * b_pair.stride(0), 0 if c is None else c.data_ptr(), n,
                                              0 if cp is None else cp.data_ptr(), 4 * n, gate_pair.data_ptr(),
                                              2 * gate_pair.stride(0), m, n, ch, kh, kw, h, w, int(bool(flip)),
                                              0 if ws is None else ws.data_ptr(), nbytes, config, _stream())
    _lib.check(rc, "split_gemm_pair_gated")
    return c, cp


def split_gemm_pair_rp_gated(a_pair, b_pair, residual_pair, gate_pair, out_f32=False, out_pair=True, config=0):
    """(A @ B^T + r) * (x > 0) with r (``residual_pair``) and x (``gate_pair``) [M, 2N] in pair layout: the input gradient
    of an identity bottleneck, gated by the ReLU of the block below and split for its backward GEMMs in the epilogue
    (``ovis_split_gemm_pair_rp_gated``).  Returns (f32 or None, pair or None)."""
    for t, name in ((a_pair, "a_pair"), (b_pair, "b_pair"), (residual_pair, "residual_pair"), (gate_pair, "gate_pair")):
        if not (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1):
            raise RuntimeError(f"split_gemm_pair_rp_gated: {name} must be a 2-D bfloat16 HIP tensor in pair layout")
    m, ch = a_pair.shape[0], a_pair.shape[1] // 2
    n, k = b_pair.shape[0], b_pair.shape[1] // 2
    if k != ch or gate_pair.shape != (m, 2 * n) or residual_pair.shape != (m, 2 * n):
        raise RuntimeError("split_gemm_pair_rp_gated: shape mismatch")
    dev = a_pair.device
    c = torch.empty((m, n), dtype=torch.float32, device=dev) if out_f32 else None
    cp = torch.empty((m, 2 * n), dtype=torch.bfloat16, device=dev) if out_pair else None
    if m == 0 or n == 0:
        return c, cp
    with _on(dev):
        rc = _L.ovis_split_gemm_pair_rp_gated(a_pair.data_ptr(), 2 * a_pair.stride(0), b_pair.data_ptr(),
                                              2 * b_pair.stride(0), 0 if c is None else c.data_ptr(), n,
                                              0 if cp is None else cp.data_ptr(), 4 * n, residual_pair.data_ptr(),
                                              2 * residual_pair.stride(0), gate_pair.data_ptr(), 2 * gate_pair.stride(0), m, n,
                                              ch, config, _stream())
    _lib.check(rc, "split_gemm_pair_rp_gated")
    return c, cp


def split_gemm_pair_tn_supported(n, ch, conv=None):
    if n % 128 or ch % 128:
        return False
    if conv is not None:
        h, w, kh, kw = conv
        return kh % 2 == 1 and kw % 2 == 1
    return True


def split_gemm_pair_tn(g_pair, x_pair, conv=None, scale=None, weight_shape=None):
    """dW [N, taps*ch] = G^T X over the rows: G [M, 2N], X [M, 2*ch] in pair layout; conv = (h, w, kh, kw): X is an
    NHWC tensor read shifted by every tap (the 3x3 weight gradient without im2col rows).  With ``weight_shape`` =
    (N, ch, kh, kw) the slabs are reduced straight into that layout, times ``scale[N]`` (the folded FrozenBN scale) --
    the gradient of the raw convolution weight in one extra launch.  csrc/split_gemm.hip."""
    for t, name in ((g_pair, "g_pair"), (x_pair, "x_pair")):
        if not (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1):
            raise RuntimeError(f"split_gemm_pair_tn: {name} must be a 2-D bfloat16 HIP tensor in pair layout")
    m, n, ch = g_pair.shape[0], g_pair.shape[1] // 2, x_pair.shape[1] // 2
    if x_pair.shape[0] != m:
        raise RuntimeError("split_gemm_pair_tn: row counts differ")
    h = w = 0
    kh = kw = 1
    if conv is not None:
        h, w, kh, kw = conv
        if m % (h * w):
            raise RuntimeError("split_gemm_pair_tn: rows must be a multiple of h*w")
    dev = g_pair.device
    if m == 0:
        z = torch.zeros((n, kh * kw * ch), dtype=torch.float32, device=dev)
        return z if weight_shape is None else z.new_zeros(weight_shape)
    slices = _tn_slices(m, n, ch, kh * kw)
    slabs = torch.empty((slices, n, kh * kw * ch), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = _L.ovis_split_gemm_pair_tn(g_pair.data_ptr(), 2 * g_pair.stride(0), x_pair.data_ptr(), 2 * x_pair.stride(0),
                                        slabs.data_ptr(), slices, m, n, ch, kh, kw, h, w, _stream())
        _lib.check(rc, "split_gemm_pair_tn")
        if weight_shape is None:
            return slabs[0] if slices == 1 else slabs.sum(0)
        if tuple(weight_shape) != (n, ch, kh, kw):
            raise RuntimeError("split_gemm_pair_tn: weight_shape must be (N, ch, kh, kw)")
        dw = torch.empty(weight_shape, dtype=torch.float32, device=dev)
        if scale is not None:
            scale = _dev(scale.detach(), "scale")
        rc = _L.ovis_slab_reduce_f32(slabs.data_ptr(), 0 if scale is None else scale.data_ptr(), dw.data_ptr(), slices, n,
                                     ch, kh * kw, _stream())
    _lib.check(rc, "slab_reduce")
    return dw


def bias_act_(y, bias=None, residual=None, relu=True):
    """In place: y[rows, cols] = act(y + bias[col] (+ residual)); contiguous f32, cols % 4 == 0."""
    if not (y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and y.is_contiguous()):
        raise RuntimeError("bias_act_: contiguous 2-D float32 HIP tensor expected")
    if residual is not None and not (residual.is_contiguous() and residual.shape == y.shape):
        raise RuntimeError("bias_act_: residual must be contiguous with y's shape")
    if y.numel() == 0:
        return y
    with _on(y.device):
        rc = _L.ovis_bias_act_f32(y.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                  0 if residual is None else residual.data_ptr(), y.shape[0], y.shape[1],
                                  int(bool(relu)), _stream())
    _lib.check(rc, "bias_act")
    return y


def gemm_nt(a, b, bias=None):
    """a [M,K] @ b[N,K]^T (+ bias[N]) -> [M,N] on the fp32 matrix cores.  a / b may be any 2-D strided views."""
    if not a.is_cuda and not b.is_cuda:
        return _cpu.gemm_nt(a, b, bias)
    if not (a.is_cuda and b.is_cuda):
        raise RuntimeError("gemm_nt: HIP device tensors only")
    if a.dtype != torch.float32 or b.dtype != torch.float32:
        raise RuntimeError("gemm_nt: float32 only")
    m, k = a.shape
    n, k2 = b.shape
    if k != k2:
        raise RuntimeError(f"gemm_nt: inner dimensions differ ({k} vs {k2})")
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    if m == 0 or n == 0:
        return out
    if k == 0:
        return out.zero_() if bias is None else out.copy_(bias.expand(m, n))
    if bias is not None:
        bias = _dev(bias, "bias")
    with _on(a.device):
        _gemm_raw(a.data_ptr(), a.stride(0), a.stride(1), b.data_ptr(), b.stride(0), b.stride(1), out.data_ptr(), n, m, n, k,
                  bias_ptr=0 if bias is None else bias.data_ptr(), device=a.device)
    return out


def region_noun_align(region_emb, noun_emb):
    """-> (raw max score [W], sigmoid score [W], argmax region [W] int64)"""
    if not region_emb.is_cuda:
        return _cpu.region_noun_align(region_emb, noun_emb)
    region_emb, noun_emb = _dev(region_emb, "region_emb"), _dev(noun_emb, "noun_emb")
    p, d = region_emb.shape
    w = noun_emb.shape[0]
    raw = torch.empty((w,), dtype=torch.float32, device=region_emb.device)
    prob = torch.empty_like(raw)
    idx = torch.empty((w,), dtype=torch.int64, device=region_emb.device)
    if w == 0:
        return raw, prob, idx
    with _on(region_emb.device):
        rc = _L.ovis_region_noun_align_f32(region_emb.data_ptr(), noun_emb.data_ptr(), raw.data_ptr(),
                                           prob.data_ptr(), idx.data_ptr(), p, w, d, _stream())
    _lib.check(rc, "region_noun_align")
    return raw, prob, idx


def text_embed(table, input_ids, special_tokens_mask):
    """Word embeddings of tokenised strings (``ovis_text_embed_f32``; language_backbone/transformers.py:27-68 +
    st_generalized_rcnn.py:202-209): table [V, D] float32, input_ids / special_tokens_mask [N, L] integer tensors as the
    tokenizer pads them -> [N, D] unit-norm rows (masked mean of the table rows, then F.normalize)."""
    if not table.is_cuda:  # MODEL.DEVICE cpu: the reference's tensor-op formula on host tensors
        return _cpu.text_embed(table.detach(), input_ids.to(torch.int64), special_tokens_mask.to(torch.int64))
    table = _dev(table, "table")
    if table.dim() != 2 or input_ids.dim() != 2 or input_ids.shape != special_tokens_mask.shape:
        raise RuntimeError("text_embed: expected table [V, D] and input_ids / special_tokens_mask [N, L]")
    if table.requires_grad:
        raise RuntimeError("text_embed: the embedding table is frozen on this path (MODEL.LANGUAGE_BACKBONE.FT_EMB False)")
    ids = input_ids.to(device=table.device, dtype=torch.int32).contiguous()
    sp = special_tokens_mask.to(device=table.device, dtype=torch.int32).contiguous()
    n, l = ids.shape
    out = torch.empty((n, table.shape[1]), dtype=torch.float32, device=table.device)
    if n:
        with _on(table.device):
            rc = _L.ovis_text_embed_f32(table.data_ptr(), table.shape[0], table.shape[1], ids.data_ptr(), sp.data_ptr(), n, l,
                                        out.data_ptr(), _stream())
        _lib.check(rc, "text_embed")
    return out


def project_polygon_masks(coords, polygon_start, instance_start, gt_index, boxes, image_size, resolution):
    """Mask targets [P, M, M] for boxes [P, 4] from POLYGON ground truth (``ovis_project_polygon_masks_f32``;
    mask_head/loss.py:11-42 through PolygonInstance.crop / resize / convert_to_binarymask, segmentation_mask.py:270-334).
    coords float32 [T] (x, y pairs of all polygons), polygon_start int32 [NP + 1] (float offsets), instance_start int32
    [G + 1] (polygon ranges), gt_index [P] int64, image_size = (width, height)."""
    if not boxes.is_cuda:
        return _cpu.project_polygon_masks(coords, polygon_start, instance_start, gt_index, boxes, image_size, resolution)
    boxes = _dev(boxes, "boxes")
    coords = _dev(coords, "coords") if coords.numel() else coords
    polygon_start = _dev(polygon_start, "polygon_start", torch.int32)
    instance_start = _dev(instance_start, "instance_start", torch.int32)
    gt_index = _dev(gt_index, "gt_index", torch.int64)
    p = boxes.shape[0]
    out = torch.empty((p, resolution, resolution), dtype=torch.float32, device=boxes.device)
    if p:
        with _on(boxes.device):
            rc = _L.ovis_project_polygon_masks_f32(coords.data_ptr() if coords.numel() else 0, polygon_start.data_ptr(),
                                                   instance_start.data_ptr(), gt_index.data_ptr(), boxes.data_ptr(), p,
                                                   int(image_size[0]), int(image_size[1]), int(resolution), out.data_ptr(),
                                                   _stream())
        _lib.check(rc, "project_polygon_masks")
    return out


def weighted_ce_fwd_bwd(logits, labels, bg_weight, need_grad=True):
    """-> (loss scalar tensor, dlogits or None)"""
    if not logits.is_cuda:
        return _cpu.weighted_ce_fwd_bwd(logits, labels, bg_weight, need_grad)
    logits, labels = _dev(logits, "logits"), _dev(labels, "labels", torch.int64)
    p, c = logits.shape
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits) if need_grad else None
    scratch = torch.empty((max(p, 1),), dtype=torch.float32, device=logits.device)
    with _on(logits.device):
        rc = _L.ovis_weighted_ce_fwd_bwd_f32(logits.data_ptr(), labels.data_ptr(), bg_weight, loss.data_ptr(),
                                             0 if dlogits is None else dlogits.data_ptr(), scratch.data_ptr(), p, c,
                                             _stream())
    _lib.check(rc, "weighted_ce_fwd_bwd")
    return loss[0], dlogits


def mask_bce_stochastic_fwd_bwd(mu, sigma, eps, pos_index, targets, channel, need_grad=True):
    """mu [P,C,M,M]; sigma [P,1,M,M] or None; eps [P,C,M,M] or None; pos_index [Pp]; targets [Pp,M,M]; channel: int (one
    logit channel for every positive: class-agnostic masks) or int64 tensor [Pp] (the positives' class labels:
    mask_logits[positive_inds, labels_pos], mask_head/loss.py:131-141) -> (loss, dmu or None, dsigma or None)"""
    if not mu.is_cuda:
        return _cpu.mask_bce_stochastic_fwd_bwd(mu, sigma, eps, pos_index, targets, channel, need_grad)
    mu = _dev(mu, "mu")
    p, c = mu.shape[0], mu.shape[1]
    mm = mu.shape[2] * mu.shape[3]
    pos_index, targets = _dev(pos_index, "pos_index", torch.int64), _dev(targets, "targets")
    sigma = None if sigma is None else _dev(sigma, "sigma")
    eps = None if eps is None else _dev(eps, "eps")
    npos = pos_index.numel()
    loss = torch.empty((1,), dtype=torch.float32, device=mu.device)
    dmu = torch.empty_like(mu) if need_grad else None
    dsigma = torch.empty((p, 1, mu.shape[2], mu.shape[3]), dtype=torch.float32, device=mu.device) \
        if (need_grad and sigma is not None) else None
    scratch = torch.empty((max(npos, 1),), dtype=torch.float32, device=mu.device)
    ptr = lambda t: 0 if t is None else t.data_ptr()
    with _on(mu.device):
        if torch.is_tensor(channel):
            channel = _dev(channel, "channel", torch.int64)
            if channel.numel() != npos:
                raise RuntimeError(f"mask_bce_stochastic_fwd_bwd: {channel.numel()} channels for {npos} positives")
            rc = _L.ovis_mask_bce_stochastic_classes_fwd_bwd_f32(mu.data_ptr(), ptr(sigma), ptr(eps), pos_index.data_ptr(),
                                                                 channel.data_ptr(), targets.data_ptr(), loss.data_ptr(),
                                                                 ptr(dmu), ptr(dsigma), scratch.data_ptr(), p, npos, c, mm,
                                                                 _stream())
        else:
            rc = _L.ovis_mask_bce_stochastic_fwd_bwd_f32(mu.data_ptr(), ptr(sigma), ptr(eps), pos_index.data_ptr(),
                                                         targets.data_ptr(), loss.data_ptr(), ptr(dmu), ptr(dsigma),
                                                         scratch.data_ptr(), p, npos, c, mm, channel, _stream())
    _lib.check(rc, "mask_bce_stochastic_fwd_bwd")
    return loss[0], dmu, dsigma


# ---- deformable convolution (csrc/deform_conv.h:11-190; host loops deform_conv_cuda.cu:161-694) ---------
def _gemm_raw(a_ptr, a_rs, a_ks, b_ptr, b_rs, b_ks, c_ptr, c_rs, m, n, k, bias_ptr=0, bias_per_row=0, alpha=1.0,
              accumulate=0, device=None):
    """C = alpha * A . B^T (+ C) + bias on raw pointers.  Products with too few tiles to fill the chip are cut along K
    into slabs of a workspace allocated here (summed in slice order by the library: no atomics)."""
    ws_bytes = _gemm_f32_ws_bytes(m, n, k)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=device or torch.device("cuda", torch.cuda.current_device())) \
        if ws_bytes else None
    rc = _L.ovis_gemm_ex_ws_f32(a_ptr, a_rs, a_ks, b_ptr, b_rs, b_ks, bias_ptr, bias_per_row, alpha, accumulate, c_ptr,
                                c_rs, m, n, k, 0 if ws is None else ws.data_ptr(), ws_bytes, _stream())
    _lib.check(rc, "gemm_ex_ws_f32")


dcn_implicit = True  # False: the column route (im2col + GEMM) also where the implicit GEMM applies (cross-check in the tests)


def _dcn_dims(input, weight, kH, kW, dH, dW, padH, padW, dilH, dilW):
    b, cin, h, w = input.shape
    ho = (h + 2 * padH - (dilH * (kH - 1) + 1)) // dH + 1
    wo = (w + 2 * padW - (dilW * (kW - 1) + 1)) // dW + 1
    if ho <= 0 or wo <= 0:
        raise RuntimeError(f"deform_conv: output size is too small ({ho}x{wo})")
    return b, cin, h, w, ho, wo


def _dcn_check(input, weight, offset, mask, kH, kW, group, dg, ho, wo):
    b, cin = input.shape[0], input.shape[1]
    if weight.dim() != 4 or weight.shape[2] != kH or weight.shape[3] != kW:
        raise RuntimeError("deform_conv: kernel size does not match the weight")
    if cin % group or weight.shape[0] % group or weight.shape[1] != cin // group or cin % dg:
        raise RuntimeError("deform_conv: channels are not divisible by group / deformable_group")
    if tuple(offset.shape) != (b, dg * 2 * kH * kW, ho, wo):
        raise RuntimeError(f"deform_conv: invalid offset shape {tuple(offset.shape)}")
    if mask is not None and tuple(mask.shape) != (b, dg * kH * kW, ho, wo):
        raise RuntimeError(f"deform_conv: invalid mask shape {tuple(mask.shape)}")


def _dcn_forward(input, weight, bias, offset, mask, output, kH, kW, dH, dW, padH, padW, dilH, dilW, group, dg, step):
    input, weight, offset = _dev(input, "input"), _dev(weight, "weight"), _dev(offset, "offset")
    mask = None if mask is None else _dev(mask, "mask")
    b, cin, h, w, ho, wo = _dcn_dims(input, weight, kH, kW, dH, dW, padH, padW, dilH, dilW)
    _dcn_check(input, weight, offset, mask, kH, kW, group, dg, ho, wo)
    cout = weight.shape[0]
    if tuple(output.shape) != (b, cout, ho, wo) or not output.is_contiguous():
        raise RuntimeError("deform_conv: output must be a contiguous [B, C_out, H_out, W_out] tensor")
    K, plane = kH * kW, ho * wo
    cin_g, cout_g = cin // group, cout // group
    if dcn_implicit and group == 1 and (cin // dg) % 32 == 0 and cout % 4 == 0 and b * plane > 0:
        # no column buffer: one implicit GEMM on the pair-layout split GEMM, the A tiles sampled in the kernel
        x_nhwc = input.permute(0, 2, 3, 1).contiguous()
        wp, _ = weight_prep_pair(weight, None)
        rows = torch.empty((b * plane, cout), dtype=torch.float32, device=input.device)
        with _on(input.device):
            rc = _L.ovis_deform_conv_implicit_f32(x_nhwc.data_ptr(), offset.data_ptr(),
                                                  0 if mask is None else mask.data_ptr(), wp.data_ptr(), 2 * wp.stride(0),
                                                  0 if bias is None else _dev(bias, "bias").data_ptr(), rows.data_ptr(),
                                                  cout, b, cin, h, w, cout, ho, wo, kH, kW, dH, dW, padH, padW, dilH, dilW,
                                                  dg, _stream())
        _lib.check(rc, "deform_conv_implicit")
        output.copy_(rows.view(b, ho, wo, cout).permute(0, 3, 1, 2))
        return 1
    step = max(1, min(step, b))
    col = torch.empty((cin * K, step, plane), dtype=torch.float32, device=input.device)
    with _on(input.device):
        for s0 in range(0, b, step):
            nb = min(step, b - s0)
            rc = _L.ovis_deform_im2col_f32(input[s0].data_ptr(), offset[s0].data_ptr(),
                                           0 if mask is None else mask[s0].data_ptr(), col.data_ptr(), nb, cin, h, w,
                                           kH, kW, padH, padW, dH, dW, dilH, dilW, dg, _stream())
            _lib.check(rc, "deform_im2col")
            for i in range(nb):
                for g in range(group):
                    _gemm_raw(weight[g * cout_g].data_ptr(), cin_g * K, 1,
                              col.data_ptr() + 4 * ((g * cin_g * K) * nb * plane + i * plane), 1, nb * plane,
                              output[s0 + i, g * cout_g].data_ptr(), plane, cout_g, plane, cin_g * K,
                              0 if bias is None else bias[g * cout_g:].data_ptr(), 1)
    return 1


def deform_conv_forward(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH, dilationW,
                        dilationH, group, deformable_group, im2col_step):
    """Writes ``output`` in place; ``columns`` / ``ones`` are accepted for signature compatibility (the reference
    re-allocates them internally as well, deform_conv_cuda.cu:207-214).  Note W-before-H argument order."""
    return _dcn_forward(input, weight, None, offset, None, output, kH, kW, dH, dW, padH, padW, dilationH, dilationW,
                        group, deformable_group, im2col_step)


def _dcn_backward(input, weight, offset, mask, grad_output, grad_input, grad_offset, grad_mask, grad_weight, scale,
                  kH, kW, dH, dW, padH, padW, dilH, dilW, group, dg, step):
    input, weight, offset = _dev(input, "input"), _dev(weight, "weight"), _dev(offset, "offset")
    grad_output = _dev(grad_output, "grad_output")
    mask = None if mask is None else _dev(mask, "mask")
    b, cin, h, w, ho, wo = _dcn_dims(input, weight, kH, kW, dH, dW, padH, padW, dilH, dilW)
    _dcn_check(input, weight, offset, mask, kH, kW, group, dg, ho, wo)
    cout = weight.shape[0]
    K, plane = kH * kW, ho * wo
    cin_g, cout_g = cin // group, cout // group
    step = max(1, min(step, b))
    for t in (grad_input, grad_offset, grad_mask, grad_weight):
        if t is not None and not t.is_contiguous():
            raise RuntimeError("deform_conv backward: gradient buffers must be contiguous")
    if dcn_implicit and group == 1 and (cin // dg) % 32 == 0 and cout % 32 == 0 and b * plane > 0:
        return _dcn_backward_rows(input, weight, offset, mask, grad_output, grad_input, grad_offset, grad_mask, grad_weight,
                                  scale, kH, kW, dH, dW, padH, padW, dilH, dilW, dg, b, cin, h, w, ho, wo, cout)
    col = torch.empty((cin * K, step, plane), dtype=torch.float32, device=input.device)
    with _on(input.device):
        for s0 in range(0, b, step):
            nb = min(step, b - s0)
            if grad_input is not None or grad_offset is not None:
                # columns = W^T . grad_output  (per image, per group)
                for i in range(nb):
                    for g in range(group):
                        _gemm_raw(weight[g * cout_g].data_ptr(), 1, cin_g * K,
                                  grad_output[s0 + i, g * cout_g].data_ptr(), 1, plane,
                                  col.data_ptr() + 4 * ((g * cin_g * K) * nb * plane + i * plane), nb * plane,
                                  cin_g * K, plane, cout_g)
                if grad_offset is not None:
                    rc = _L.ovis_deform_col2im_coord_f32(
                        col.data_ptr(), input[s0].data_ptr(), offset[s0].data_ptr(),
                        0 if mask is None else mask[s0].data_ptr(), grad_offset[s0].data_ptr(),
                        0 if grad_mask is None else grad_mask[s0].data_ptr(), nb, cin, h, w, kH, kW, padH, padW, dH,
                        dW, dilH, dilW, dg, _stream())
                    _lib.check(rc, "deform_col2im_coord")
                if grad_input is not None:
                    rc = _L.ovis_deform_col2im_f32(col.data_ptr(), offset[s0].data_ptr(),
                                                   0 if mask is None else mask[s0].data_ptr(),
                                                   grad_input[s0].data_ptr(), nb, cin, h, w, kH, kW, padH, padW, dH, dW,
                                                   dilH, dilW, dg, _stream())
                    _lib.check(rc, "deform_col2im")
            if grad_weight is not None:
                rc = _L.ovis_deform_im2col_f32(input[s0].data_ptr(), offset[s0].data_ptr(),
                                               0 if mask is None else mask[s0].data_ptr(), col.data_ptr(), nb, cin, h,
                                               w, kH, kW, padH, padW, dH, dW, dilH, dilW, dg, _stream())
                _lib.check(rc, "deform_im2col")
                for i in range(nb):
                    for g in range(group):
                        _gemm_raw(grad_output[s0 + i, g * cout_g].data_ptr(), plane, 1,
                                  col.data_ptr() + 4 * ((g * cin_g * K) * nb * plane + i * plane), nb * plane, 1,
                                  grad_weight[g * cout_g].data_ptr(), cin_g * K, cout_g, cin_g * K, plane,
                                  alpha=scale, accumulate=1)


def _dcn_backward_rows(input, weight, offset, mask, grad_output, grad_input, grad_offset, grad_mask, grad_weight, scale,
                       kH, kW, dH, dW, padH, padW, dilH, dilW, dg, b, cin, h, w, ho, wo, cout):
    """The backward without the reference-layout column buffer (csrc/deform_conv_rows.hip): rows = (image, h_out, w_out),
    k = (tap, channel); dcol = dY . W as ONE split GEMM consumed by one scatter / gather pass (dX in NHWC with coalesced
    atomics, offset and mask gradients reduced in the wave), the weight gradient as ONE transpose-read split GEMM over the
    sampled rows written once in pair layout.  Same accumulate / overwrite conventions as the column route."""
    dev = input.device
    T, rows = kH * kW, b * ho * wo
    x_nhwc = input.permute(0, 2, 3, 1).contiguous()
    gy = grad_output.permute(0, 2, 3, 1).reshape(rows, cout)              # NHWC rows of dY
    gyp = split_pair(gy.contiguous())
    geom = (b, cin, h, w, ho, wo, kH, kW, dH, dW, padH, padW, dilH, dilW, dg)
    mptr = 0 if mask is None else mask.data_ptr()
    if grad_input is not None or grad_offset is not None:
        wt = split_pair(weight.permute(2, 3, 1, 0).reshape(T * cin, cout).contiguous())   # B[(t, c), n] = w[n, c, t]
        dcol, _ = split_gemm_pair(gyp, wt)                                                   # [rows, T * cin] f32
        dx = torch.zeros((b, h, w, cin), dtype=torch.float32, device=dev) if grad_input is not None else None
        if grad_offset is not None:
            grad_offset.zero_()
            if grad_mask is not None:
                grad_mask.zero_()
        with _on(dev):
            rc = _L.ovis_deform_col2im_rows_f32(dcol.data_ptr(), dcol.stride(0), x_nhwc.data_ptr(), offset.data_ptr(), mptr,
                                                0 if dx is None else dx.data_ptr(),
                                                0 if grad_offset is None else grad_offset.data_ptr(),
                                                0 if grad_mask is None else grad_mask.data_ptr(), *geom, _stream())
        _lib.check(rc, "deform_col2im_rows")
        if dx is not None:
            grad_input += dx.permute(0, 3, 1, 2)
    if grad_weight is not None:
        colp = torch.empty((rows, 2 * T * cin), dtype=torch.bfloat16, device=dev)
        with _on(dev):
            rc = _L.ovis_deform_im2col_pair_rows_f32(x_nhwc.data_ptr(), offset.data_ptr(), mptr, colp.data_ptr(),
                                                     2 * colp.stride(0), *geom, _stream())
        _lib.check(rc, "deform_im2col_pair_rows")
        if split_gemm_pair_tn_supported(cout, T * cin):
            dwm = split_gemm_pair_tn(gyp, colp)                                              # [cout, T * cin]
        else:  # shapes the transpose-read kernel does not take: one library bf16 product, the four hi / lo quadrants summed
            q = torch.mm(gyp.t(), colp, out_dtype=torch.float32)
            dwm = q.view(cout // 32, 2, 32, T * cin // 32, 2, 32).sum(dim=(1, 4)).reshape(cout, T * cin)
        grad_weight.add_(dwm.view(cout, kH, kW, cin).permute(0, 3, 1, 2), alpha=scale)


def deform_conv_backward_input(input, offset, gradOutput, gradInput, gradOffset, weight, columns, kW, kH, dW, dH, padW,
                               padH, dilationW, dilationH, group, deformable_group, im2col_step):
    """gradInput is accumulated into (callers pass zeros, dcn/deform_conv_func.py:85-86); gradOffset is overwritten."""
    _dcn_backward(input, weight, offset, None, gradOutput, gradInput, gradOffset, None, None, 1.0, kH, kW, dH, dW, padH,
                  padW, dilationH, dilationW, group, deformable_group, im2col_step)
    return 1


def deform_conv_backward_parameters(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW, dH, padW, padH,
                                    dilationW, dilationH, group, deformable_group, scale, im2col_step):
    """gradWeight += scale * d(loss)/d(weight)."""
    _dcn_backward(input, gradWeight, offset, None, gradOutput, None, None, None, gradWeight, float(scale), kH, kW, dH,
                  dW, padH, padW, dilationH, dilationW, group, deformable_group, im2col_step)
    return 1


def modulated_deform_conv_forward(input, weight, bias, ones, offset, mask, output, columns, kernel_h, kernel_w, stride_h,
                                  stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, with_bias):
    _dcn_forward(input, weight, _dev(bias, "bias") if with_bias else None, offset, mask, output, kernel_h, kernel_w,
                 stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, 8)


def modulated_deform_conv_backward(input, weight, bias, ones, offset, mask, columns, grad_input, grad_weight, grad_bias,
                                   grad_offset, grad_mask, grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h,
                                   pad_w, dilation_h, dilation_w, group, deformable_group, with_bias):
    """grad_input / grad_weight / grad_bias are accumulated into (callers pass zeros); grad_offset / grad_mask are
    overwritten (deform_conv_cuda.cu:580-694)."""
    _dcn_backward(input, weight, offset, mask, grad_output, grad_input, grad_offset, grad_mask, grad_weight, 1.0,
                  kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group,
                  8)
    if with_bias:
        grad_bias += grad_output.sum(dim=(0, 2, 3))


# ---- optimizer (engine/solver.py) ------------------------------------------------------------------------------------
_BB_WS = {}


def bottleneck_identity_backward(g3p, xp, o1p, o2p, t1, t2, t3, scales, geom, weight_shapes, side_stream=None):
    """The backward of an identity bottleneck of a pair-only chain in ONE native call (``ovis_bottleneck_identity_backward``:
    the nine launches of the separate wrappers, the three weight gradients on ``side_stream`` beside the data-gradient chain).
    g3p [M, 2C] gated output gradient, xp / o1p / o2p the block's input and conv1 / conv2 outputs, t1 / t2 / t3 the transposed
    weight pair forms, scales (s1, s2, s3) or Nones, geom (h, w, kh, kw), weight_shapes (w1, w2, w3 shapes) ->
    (gx pair [M, 2C], dw1, dw2, dw3).  Pair tensors: 2-D bfloat16 with unit column stride."""
    for t in (g3p, xp, o1p, o2p, t1, t2, t3):
        if not (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1):
            raise RuntimeError("bottleneck_identity_backward: 2-D bfloat16 HIP tensors in pair layout expected")
    m, cin, mid = g3p.shape[0], g3p.shape[1] // 2, o1p.shape[1] // 2
    h, w, kh, kw = geom
    if (xp.shape != (m, 2 * cin) or o1p.shape[0] != m or o2p.shape != (m, 2 * mid) or t1.shape != (cin, 2 * mid)
            or t2.shape != (mid, 2 * kh * kw * mid) or t3.shape != (mid, 2 * cin) or m % (h * w)):
        raise RuntimeError("bottleneck_identity_backward: shape mismatch")
    dev = g3p.device
    g2p = torch.empty((m, 2 * mid), dtype=torch.bfloat16, device=dev)
    g1p = torch.empty((m, 2 * mid), dtype=torch.bfloat16, device=dev)
    gx = torch.empty((m, 2 * cin), dtype=torch.bfloat16, device=dev)
    dw1, dw2, dw3 = (torch.empty(tuple(sh), dtype=torch.float32, device=dev) for sh in weight_shapes)
    config = _gemm_cfg(0)
    key = (m, cin, mid, kh, kw, w, config)
    nbytes = _BB_WS.get(key)
    if nbytes is None:
        nbytes = _BB_WS[key] = int(_L.ovis_bottleneck_identity_backward_workspace_bytes(m, cin, mid, kh, kw, w, config))
    ws = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=dev)
    s1, s2, s3 = (None if s is None else _dev(s.detach(), "scale") for s in scales)
    with _on(dev):
        rc = _L.ovis_bottleneck_identity_backward(
            g3p.data_ptr(), 2 * g3p.stride(0), xp.data_ptr(), 2 * xp.stride(0), o1p.data_ptr(), 2 * o1p.stride(0),
            o2p.data_ptr(), 2 * o2p.stride(0), t1.data_ptr(), 2 * t1.stride(0), t2.data_ptr(), 2 * t2.stride(0),
            t3.data_ptr(), 2 * t3.stride(0), 0 if s1 is None else s1.data_ptr(), 0 if s2 is None else s2.data_ptr(),
            0 if s3 is None else s3.data_ptr(), m, cin, mid, kh, kw, h, w, g2p.data_ptr(), g1p.data_ptr(), gx.data_ptr(),
            dw1.data_ptr(), dw2.data_ptr(), dw3.data_ptr(), ws.data_ptr(), ws.numel(), config, _stream(),
            0 if side_stream is None else side_stream.cuda_stream)
    _lib.check(rc, "bottleneck_identity_backward")
    return gx, dw1, dw2, dw3


def weight_prep_tile():
    return int(_L.ovis_weight_prep_tile())


def weight_prep_pair_multi(items, blocks, max_taps):
    """One launch of the weight preparation over the convolutions described by ``items`` (uint8 device table of 64-byte
    records) and ``blocks`` (int32 [B, 2] = (item, 32 x 32 tile)); ``max_taps`` = the largest KH * KW in the table; see
    include/ovis_hip.h and layers/pair_bottleneck.py::WeightPrepPlan, which builds the tables."""
    if not (items.is_cuda and blocks.is_cuda and items.dtype == torch.uint8 and blocks.dtype == torch.int32):
        raise RuntimeError("weight_prep_pair_multi: uint8 / int32 HIP device tables expected")
    with _on(items.device):
        rc = _L.ovis_weight_prep_pair_multi_f32(items.data_ptr(), blocks.data_ptr(), blocks.size(0), int(max_taps), _stream())
    _lib.check(rc, "weight_prep_pair_multi")


def sgd_chunk_elements():
    return int(_L.ovis_sgd_chunk_elements())


def sgd_momentum_multi(items, blocks, lr, weight_decay, momentum, apply_weight_decay):
    """One launch of the fused SGD update over the tensors described by ``items`` (uint8 device table of 32-byte records
    {p, g, buf, n}) and ``blocks`` (int32 [B, 2] = (item, chunk)); see include/ovis_hip.h."""
    if not (items.is_cuda and blocks.is_cuda):
        raise RuntimeError("sgd_momentum_multi: HIP device tensors only")
    with _on(items.device):
        rc = _L.ovis_sgd_momentum_multi_f32(items.data_ptr(), blocks.data_ptr(), blocks.size(0), lr, weight_decay, momentum,
                                            int(bool(apply_weight_decay)), _stream())
    _lib.check(rc, "sgd_momentum_multi")
