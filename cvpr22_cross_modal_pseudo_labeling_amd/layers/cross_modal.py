"""Autograd wrappers of the cross-modal head kernels (include/ovis_hip.h, csrc/gemm_f32.hip, csrc/losses.hip).

* ``linear_mfma`` / ``text_logits`` -- the two products of FastRCNNPredictor.forward
  (maskrcnn_benchmark/modeling/roi_heads/box_head/roi_box_predictors.py:66-71) on the fp32 matrix cores,
  forward and both backward products through the same strided GEMM kernel;
* ``weighted_cross_entropy`` -- box_head/loss.py:172-174, loss and d/dlogits in one pass;
* ``stochastic_mask_bce`` -- roi_mask_predictors.py:41-65 + mask_head/loss.py:139-142, loss and the
  gradients w.r.t. the mask logits and the predicted std-dev in one pass.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C


class _LinearMFMA(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _C.gemm_nt(x, weight, bias)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = _C.gemm_nt(dy, weight.t()) if ctx.needs_input_grad[0] else None          # [M,N] x [K,N]^T
        dw = _C.gemm_nt(dy.t(), x.t()) if ctx.needs_input_grad[1] else None          # [N,M] x [K,M]^T
        db = dy.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db


_PAIR_WEIGHT_CACHE = {}


def _pair_weight_padded(weight, n_pad, transposed=False):
    """[N, K] f32 -> pair layout [n_pad, 2K] (zero rows behind N), or of its transpose [K, 2 * n_pad] (the data-gradient
    operand); cached for tensors that do not require grad (the class / vocabulary matrices, frozen emb_pred), keyed on
    identity + version."""
    import torch.nn.functional as F
    frozen = not weight.requires_grad
    slot = (id(weight), transposed)
    key = (id(weight), weight._version, weight.device, n_pad, transposed)
    if frozen:
        hit = _PAIR_WEIGHT_CACHE.get(slot)
        if hit is not None and hit[0] == key and hit[1]() is weight:
            return hit[2]
    w = weight.detach()
    if transposed:
        w = w.t()
        if w.shape[1] != n_pad:
            w = F.pad(w, (0, n_pad - w.shape[1]))
    elif w.shape[0] != n_pad:
        w = F.pad(w, (0, 0, 0, n_pad - w.shape[0]))
    wp = _C.split_pair(w.contiguous())
    if frozen:
        import weakref
        if len(_PAIR_WEIGHT_CACHE) > 64:
            _PAIR_WEIGHT_CACHE.clear()
        _PAIR_WEIGHT_CACHE[slot] = (key, weakref.ref(weight), wp)
    return wp


class _LinearPair(Function):
    """y = x @ weight^T + bias on the pair-layout split GEMM (csrc/split_gemm.hip): the region x text products of the
    cross-modal head (roi_box_predictors.py:62-81) as fp32-accurate three-term bf16 hi/lo products on the bf16 matrix
    cores -- several times the rate of the fp32-input MFMA GEMM they replace, and deterministic (small-M problems are cut
    into K slices whose slabs one kernel reduces; no atomics).  N is padded to a multiple of 128 with zero rows."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        """Returns [M, N] -- a column slice (view) of the padded [M, n_pad] product when N is not a multiple of 128; a
        caller that can build its weight already padded (zero rows) avoids the pad launches."""
        import torch.nn.functional as F
        n = weight.shape[0]
        n_pad = -(-n // 128) * 128
        xp = _C.split_pair(x.detach())   # row-strided views (column slices of a padded product) are read in place
        wp = _pair_weight_padded(weight, n_pad)
        b = None
        if bias is not None:
            b = bias.detach() if n == n_pad else F.pad(bias.detach(), (0, n_pad - n))
        y, _ = _C.split_gemm_pair(xp, wp, b)
        ctx.save_for_backward(xp, weight)
        ctx.has_bias = bias is not None
        ctx.n_pad = n_pad
        return y if n == n_pad else y[:, :n]

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        import torch.nn.functional as F
        xp, weight = ctx.saved_tensors
        n, n_pad = weight.shape[0], ctx.n_pad
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = dw = db = None
        if need_x or need_w:
            g = dy if n == n_pad else F.pad(dy, (0, n_pad - n))
            gp = _C.split_pair(g.contiguous())                                  # [M, 2 n_pad]
            if need_x:
                dx, _ = _C.split_gemm_pair(gp, _pair_weight_padded(weight, n_pad, transposed=True))  # dY W
            if need_w:
                dw = _C.split_gemm_pair_tn(gp, xp)[:n]                          # dY^T X over the rows
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db


def _pair_ok(x, weight):
    return (x.is_cuda and x.dim() == 2 and weight.dim() == 2 and x.dtype == torch.float32 and weight.shape[1] % 128 == 0
            and weight.shape[0] >= 32 and x.shape[0] > 0)


def linear_mfma(x, weight, bias=None):
    """y = x @ weight^T + bias with x [M,K], weight [N,K] (nn.Linear layout): split GEMM on the bf16 matrix cores when
    the shape allows (K % 128 == 0, N >= 32), else the exact-fp32 MFMA GEMM (``_C.gemm_nt``: the mask predictor's
    1- and 2-channel 1x1 heads)."""
    if _pair_ok(x, weight):
        return _LinearPair.apply(x, weight, bias)
    return _LinearMFMA.apply(x, weight, bias)


class _SplitLinearMulti(Function):
    """y_i = x @ w_i^T + b_i for several (w_i, b_i) sharing the operand x, as fp32-accurate GEMMs on the bf16 matrix
    pipe: operands are split into bf16 hi + lo (csrc/split_bf16.hip) and the three products hi.hi + hi.lo + lo.hi
    run as ONE bf16 GEMM with fp32 accumulation over a 3K-long concatenated inner dimension (hipBLASLt through
    torch.mm(out_dtype=float32); 250-390 TFLOP/s fp32-equivalent on the res5 shapes vs 105-145 for the fp32
    GEMM, relative error ~4e-6).  Backward: dX and dW are split products as well; dW contracts over the M rows."""

    @staticmethod
    def forward(ctx, x, *params):
        xs = _C.split_bf16x3(x, 0)                       # [M, 3K] = [hi | hi | lo], shared by every product
        ws = [p for p in params[0::2]]
        outs = []
        for w, b in zip(params[0::2], params[1::2]):
            y = torch.mm(xs, _C.split_bf16x3(w, 1).t(), out_dtype=torch.float32)
            if b is not None:
                if y.shape[1] % 4 == 0:
                    _C.bias_act_(y, b.contiguous(), None, relu=False)
                else:
                    y += b
            outs.append(y)
        ctx.save_for_backward(xs, *ws)
        ctx.has_bias = [b is not None for b in params[1::2]]
        ctx.k = x.shape[1]
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dys):
        xs, *ws = ctx.saved_tensors
        k = ctx.k
        dx = None
        grads = []
        for i, (dy, w) in enumerate(zip(dys, ws)):
            n = w.shape[0]
            dys_ = _C.split_bf16x3(dy.contiguous(), 0)   # [M, 3N] = [hi | hi | lo]
            if ctx.needs_input_grad[0]:
                d = torch.mm(dys_, _C.split_bf16x3(w.t(), 1).t(), out_dtype=torch.float32)   # dY W
                dx = d if dx is None else dx.add_(d)
            dw = None
            if ctx.needs_input_grad[1 + 2 * i]:
                # dY^T X by M-contraction as ONE GEMM: columns N..3N of dys_ are [dY_hi | dY_lo], columns K..3K of the
                # saved operand are [X_hi | X_lo], so [dY_hi | dY_lo]^T [X_hi | X_lo] holds all four hi/lo products
                # as quadrants (the lo.lo one comes for free and is kept)
                q = torch.mm(dys_[:, n:].t(), xs[:, k:], out_dtype=torch.float32)             # [2N, 2K]
                dw = (q[:n, :k] + q[:n, k:]) + (q[n:, :k] + q[n:, k:])
            db = dy.sum(0) if (ctx.has_bias[i] and ctx.needs_input_grad[2 + 2 * i]) else None
            grads += [dw, db]
        return (dx, *grads)


class _SplitConvSame(Function):
    """y[R,H,W,N] = conv(x[R,H,W,C], w[N,C,KH,KW]), stride 1, zero "same" padding, NHWC, as bf16 hi/lo split GEMMs:
    the split + im2col kernel (csrc/split_bf16.hip) lays every pixel's KH*KW neighbourhood out as one row
    [hi taps | hi taps | lo taps], so the convolution is ONE bf16 GEMM with fp32 accumulation over 3*KH*KW*C, the data
    gradient the same thing on dY with the rotated kernel, and the weight gradient three M-contracting GEMMs of dY^T
    against the saved rows.  ~2x MIOpen's fp32 implicit-GEMM kernels on the res5 3x3 (which sit at ~115 TFLOP/s of
    the 157 TFLOP/s fp32 matrix peak), relative error ~4e-6."""

    @staticmethod
    def forward(ctx, x, w):
        r, h, wd, c = x.shape
        n, _, kh, kw = w.shape
        rows = _C.im2col_split_bf16x3(x, kh, kw)                                    # [M, 3*T*C]
        wm = w.permute(0, 2, 3, 1).reshape(n, kh * kw * c)                           # [N, T*C], tap-major like rows
        y = torch.mm(rows, _C.split_bf16x3(wm, 1).t(), out_dtype=torch.float32)
        ctx.save_for_backward(rows, w)
        ctx.shape = (r, h, wd, c, n, kh, kw)
        return y  # [R*H*W, N]: no view is created inside the Function, so callers may finish it in place

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        rows, w = ctx.saved_tensors
        r, h, wd, c, n, kh, kw = ctx.shape
        t = kh * kw
        dy = dy.contiguous().view(r, h, wd, n)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # dX[p, c] = sum_{tap, n} dY[p - tap, n] W[n, c, tap]: the same im2col on dY with the taps reversed
            wt = w.permute(1, 2, 3, 0).reshape(c, t * n)                             # [C, T*N]
            dx = torch.mm(_C.im2col_split_bf16x3(dy, kh, kw, flip=True), _C.split_bf16x3(wt, 1).t(),
                          out_dtype=torch.float32).view(r, h, wd, c)
        if ctx.needs_input_grad[1]:
            dys = _C.split_bf16x3(dy.view(-1, n), 0)                                 # [M, 3N] = [hi | hi | lo]
            tc = t * c
            # one M-contracting GEMM for all four hi/lo products: [dY_hi | dY_lo]^T [rows_hi | rows_lo] -> quadrants
            q = torch.mm(dys[:, n:].t(), rows[:, tc:], out_dtype=torch.float32)      # [2N, 2*T*C]
            dwm = (q[:n, :tc] + q[:n, tc:]) + (q[n:, :tc] + q[n:, tc:])
            dw = dwm.view(n, kh, kw, c).permute(0, 3, 1, 2)
        return dx, dw


def split_conv_same(x, w):
    """NHWC stride-1 "same" convolution (odd kernel, groups = 1, dilation 1) of x [R,H,W,C] as bf16 hi/lo split
    GEMMs; returns the fp32 result as the [R*H*W, N] matrix (view it as [R,H,W,N]); bias / activation are left to
    ``bias_relu_``."""
    return _SplitConvSame.apply(x, w)


class _BiasActInplace(Function):
    """y <- relu(y + bias (+ residual)) in one pass over y (csrc/split_bf16.hip::bias_act_kernel); the backward is the
    ReLU gate on the saved output, shared by y and the residual."""

    @staticmethod
    def forward(ctx, y, bias, residual):
        _C.bias_act_(y, bias, residual, relu=True)
        ctx.mark_dirty(y)
        ctx.save_for_backward(y)
        ctx.has = (bias is not None, residual is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        gi = torch.ops.aten.threshold_backward(g, out, 0.0)
        db = gi.sum(0) if (ctx.has[0] and ctx.needs_input_grad[1]) else None
        return gi, db, (gi if ctx.has[1] else None)


def bias_relu_(y, bias=None, residual=None):
    """In place on the contiguous [rows, cols] f32 tensor y: relu(y + bias[col] (+ residual)).  y must be a tensor
    autograd allows to be modified in place (the fresh output of a GEMM)."""
    return _BiasActInplace.apply(y, bias, residual)


def split_linear(x, *weights_and_biases):
    """(x @ w0^T + b0, x @ w1^T + b1, ...) with x [M,K] and w_i [N_i,K] f32: bf16 hi/lo split GEMMs, fp32 results."""
    return _SplitLinearMulti.apply(x, *weights_and_biases)


def text_logits(region_emb, class_emb):
    """einsum('pe,ce->pc'): region embeddings [P,E] against the class / vocabulary matrix [C,E]."""
    return linear_mfma(region_emb, class_emb, None)


class _WeightedCE(Function):
    @staticmethod
    def forward(ctx, logits, labels, bg_weight):
        loss, dlogits = _C.weighted_ce_fwd_bwd(logits, labels, bg_weight, need_grad=logits.requires_grad)
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None, None


def weighted_cross_entropy(logits, labels, bg_weight):
    """sum_p w[label_p] * CE_p / P with w[0] = bg_weight, w[c>0] = 1 (scalar)."""
    return _WeightedCE.apply(logits, labels, bg_weight)


class _SmoothL1Picked(Function):
    @staticmethod
    def forward(ctx, box_regression, regression_targets, positives, labels, column0, beta, denominator):
        loss, grad = _C.smooth_l1_picked_fwd_bwd(box_regression, regression_targets, positives, labels, column0, beta,
                                                 denominator, need_grad=box_regression.requires_grad)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None, None


def smooth_l1_picked(box_regression, regression_targets, positives, labels, column0, beta, denominator):
    """sum over the rows ``positives`` of smooth_l1(box_regression[p, col0:col0 + 4] - regression_targets[p]; beta) /
    denominator, col0 = 4 * labels[p] or ``column0`` (labels None) -- box_head/loss.py:147-170 as one fused op."""
    return _SmoothL1Picked.apply(box_regression, regression_targets, positives, labels, column0, beta, denominator)


class _StochasticMaskBCE(Function):
    @staticmethod
    def forward(ctx, mu, sigma, eps, pos_index, targets, channel):
        need = mu.requires_grad or (sigma is not None and sigma.requires_grad)
        loss, dmu, dsigma = _C.mask_bce_stochastic_fwd_bwd(mu, sigma, eps, pos_index, targets, channel, need_grad=need)
        ctx.has_sigma = sigma is not None
        ctx.save_for_backward(dmu, dsigma)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        dmu, dsigma = ctx.saved_tensors
        return (dmu * g if dmu is not None else None, dsigma * g if (ctx.has_sigma and dsigma is not None) else None,
                None, None, None, None)


def stochastic_mask_bce(mu, sigma, eps, pos_index, targets, channel=1):
    """mean BCEWithLogits(mu[pos, channel] + eps[pos, channel] * sigma[pos], targets); sigma / eps may be None."""
    return _StochasticMaskBCE.apply(mu, sigma, eps, pos_index, targets, channel)
