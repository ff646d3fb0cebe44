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


# ---- cross-modal head + student losses (extensions beyond vision.cpp; include/ovis_hip.h) --------------
def gemm_nt(a, b, bias=None):
    """a [M,K] @ b[N,K]^T (+ bias[N]) -> [M,N] on the fp32 matrix cores.  a / b may be any 2-D strided views."""
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
    with torch.cuda.device(a.device):
        rc = _L.ovis_gemm_f32(a.data_ptr(), a.stride(0), a.stride(1), b.data_ptr(), b.stride(0), b.stride(1),
                              0 if bias is None else bias.data_ptr(), out.data_ptr(), n, m, n, k, _stream())
    _lib.check(rc, "gemm_f32")
    return out


def region_noun_align(region_emb, noun_emb):
    """-> (raw max score [W], sigmoid score [W], argmax region [W] int64)"""
    region_emb, noun_emb = _dev(region_emb, "region_emb"), _dev(noun_emb, "noun_emb")
    p, d = region_emb.shape
    w = noun_emb.shape[0]
    raw = torch.empty((w,), dtype=torch.float32, device=region_emb.device)
    prob = torch.empty_like(raw)
    idx = torch.empty((w,), dtype=torch.int64, device=region_emb.device)
    if w == 0:
        return raw, prob, idx
    with torch.cuda.device(region_emb.device):
        rc = _L.ovis_region_noun_align_f32(region_emb.data_ptr(), noun_emb.data_ptr(), raw.data_ptr(),
                                           prob.data_ptr(), idx.data_ptr(), p, w, d, _stream())
    _lib.check(rc, "region_noun_align")
    return raw, prob, idx


def weighted_ce_fwd_bwd(logits, labels, bg_weight, need_grad=True):
    """-> (loss scalar tensor, dlogits or None)"""
    logits, labels = _dev(logits, "logits"), _dev(labels, "labels", torch.int64)
    p, c = logits.shape
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits) if need_grad else None
    scratch = torch.empty((max(p, 1),), dtype=torch.float32, device=logits.device)
    with torch.cuda.device(logits.device):
        rc = _L.ovis_weighted_ce_fwd_bwd_f32(logits.data_ptr(), labels.data_ptr(), bg_weight, loss.data_ptr(),
                                             0 if dlogits is None else dlogits.data_ptr(), scratch.data_ptr(), p, c,
                                             _stream())
    _lib.check(rc, "weighted_ce_fwd_bwd")
    return loss[0], dlogits


def mask_bce_stochastic_fwd_bwd(mu, sigma, eps, pos_index, targets, channel, need_grad=True):
    """mu [P,C,M,M]; sigma [P,1,M,M] or None; eps [P,C,M,M] or None; pos_index [Pp]; targets [Pp,M,M]
    -> (loss, dmu or None, dsigma or None)"""
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
    with torch.cuda.device(mu.device):
        rc = _L.ovis_mask_bce_stochastic_fwd_bwd_f32(mu.data_ptr(), ptr(sigma), ptr(eps), pos_index.data_ptr(),
                                                     targets.data_ptr(), loss.data_ptr(), ptr(dmu), ptr(dsigma),
                                                     scratch.data_ptr(), p, npos, c, mm, channel, _stream())
    _lib.check(rc, "mask_bce_stochastic_fwd_bwd")
    return loss[0], dmu, dsigma
