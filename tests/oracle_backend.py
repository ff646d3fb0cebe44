"""TEST-ONLY: routes the native-op entry points of ``cvpr22_cross_modal_pseudo_labeling_amd._C`` to the CPU oracle for
host tensors.  The product package never does this: a host tensor is served by its own in-package host code
(``_cpu.py`` + ``libovis_cpu.so``, the reference's CPU-only configuration) or, for ops the reference has no host form of,
refused with ``RuntimeError``; nothing in the package imports ``oracle/``.  Tests use this context (a) to produce the CPU
side of GPU-vs-oracle comparisons of whole-model steps with the ORACLE's arithmetic and (b) to check the in-package host
path against the oracle (tests/test_cpu_config.py)."""
import contextlib

import torch

import oracle
from cvpr22_cross_modal_pseudo_labeling_amd import _C


def _nms_padded(dets, scores, thr, ge_mode=False):
    keep = oracle.nms(dets, scores, thr, ge_mode)
    out = torch.zeros(dets.shape[0], dtype=torch.int64)
    out[: keep.numel()] = keep
    return out, torch.tensor([keep.numel()], dtype=torch.int32)


def _gemm_nt(a, b, bias=None):
    y = a @ b.t()
    return y if bias is None else y + bias


def _region_noun_align(emb, nouns):
    raw, idx = torch.max(emb @ nouns.t(), dim=0)
    return raw, torch.sigmoid(raw), idx


@torch.enable_grad()  # called from inside autograd.Function.forward, where grad mode is off
def _weighted_ce(logits, labels, bg_weight, need_grad=True):
    x = logits.detach().clone().requires_grad_(True)
    w = torch.ones(x.shape[1])
    w[0] = bg_weight
    loss = (torch.nn.functional.cross_entropy(x, labels, weight=w, reduction="none") / labels.numel()).sum()
    g = torch.autograd.grad(loss, x)[0] if need_grad else None
    return loss.detach(), g


@torch.enable_grad()
def _mask_bce(mu, sigma, eps, pos_index, targets, channel, need_grad=True):
    m = mu.detach().clone().requires_grad_(True)
    s = None if sigma is None else sigma.detach().clone().requires_grad_(True)
    z = m if s is None else m + eps * (m * 0.0 + s)
    sel = z[pos_index, channel].reshape(pos_index.numel(), -1)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(sel, targets, reduction="none").mean()
    if not need_grad:
        return loss.detach(), None, None
    gs = torch.autograd.grad(loss, [m] + ([s] if s is not None else []))
    return loss.detach(), gs[0], (gs[1] if s is not None else None)


@contextlib.contextmanager
def oracle_ops():
    saved = {k: getattr(_C, k) for k in ("roi_align_forward", "roi_align_forward_mfma", "roi_align_backward", "nms", "nms_padded",
                                         "sigmoid_focalloss_forward", "sigmoid_focalloss_backward", "gemm_nt",
                                         "region_noun_align", "weighted_ce_fwd_bwd", "mask_bce_stochastic_fwd_bwd")}
    _C.roi_align_forward = lambda x, r, s, ph, pw, sr: oracle.roi_align_forward(x, r, s, ph, pw, sr)
    _C.roi_align_forward_mfma = _C.roi_align_forward
    _C.roi_align_backward = lambda g, r, s, ph, pw, n, c, h, w, sr: oracle.roi_align_backward(g, r, s, ph, pw, n, c, h, w, sr)
    _C.nms = lambda d, s, t: oracle.nms(d, s, t)
    _C.nms_padded = _nms_padded
    _C.sigmoid_focalloss_forward = lambda l, t, nc, g, a: oracle.sigmoid_focal_loss_forward(l, t, g, a)
    _C.sigmoid_focalloss_backward = lambda l, t, d, nc, g, a: oracle.sigmoid_focal_loss_backward(l, t, d, g, a)
    # head / loss ops: the reference computes these with plain torch fp32 ops, which is the oracle here
    _C.gemm_nt = _gemm_nt
    _C.region_noun_align = _region_noun_align
    _C.weighted_ce_fwd_bwd = _weighted_ce
    _C.mask_bce_stochastic_fwd_bwd = _mask_bce
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(_C, k, v)
