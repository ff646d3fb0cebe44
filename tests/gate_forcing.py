"""TEST-ONLY: compare gradients of the GPU path and of the plain-torch CPU path with the ReLU gates FORCED EQUAL.

Why: the whole-model parity tests bound the per-tensor gradient distance at 5e-3 (res5 head) / 2e-2 (trainable trunk, thirty
ReLUs deep) and DESIGN.md attributes all of it to ReLU gates that flip where a pre-activation lies within rounding of zero
(the three-term bf16 product rounds at 2^-17, fp32 at 2^-24).  That claim is testable: record, on the GPU side, the gate every
backward kernel will read -- the sign of the saved activation (fp32 output, or the bf16 hi halves of its pair form:
layers/pair_bottleneck.py) -- and make the CPU side's ReLUs use THOSE masks in their backward (their forward stays the CPU's own
arithmetic).  With equal gates both sides differentiate the same piecewise-linear function and what is left is arithmetic:
the distance must drop to the north_star's 1e-3.  If it did not, the gap would be an arithmetic bug, not gate flips.

``record_gates(sites)`` (GPU run): wraps the pair-route entry points and appends one bool NCHW mask per ReLU, in call order.
``forced_gates(sites)`` (CPU run): ``F.relu_`` / ``F.relu`` consult the recorded masks in the same order (a call whose shape
does not match the next recorded site -- a frozen block's ReLU, which no gradient crosses -- stays an ordinary ReLU); several
CPU passes over row ranges of one GPU pass (the student's two branches) consume the sites cyclically, range by range.
"""
import contextlib

import torch
import torch.nn.functional as F


def _gate(t):
    """The gate the backward kernels derive from a saved activation: > 0 on the fp32 tensor / on the bf16 hi halves of its
    pair form ([rows, 2C]: per 32 values 32 hi | 32 lo)."""
    if t.dtype == torch.bfloat16:
        rows, c = t.shape[0], t.shape[1] // 2
        return t.view(rows, c // 32, 2, 32)[:, :, 0, :].reshape(rows, c) > 0
    return t > 0


@contextlib.contextmanager
def record_gates(sites):
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import pair_bottleneck as pb
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling import backbone, roi_heads

    orig_b, orig_c, orig_up = backbone.bottleneck_pair, pb.conv_same_pair, roi_heads.MaskRCNNC4Predictor._upsampled_rows

    mask_site = [None]  # index of the site the mask head's last call wrote (consecutive calls -- one per branch -- are one site)

    def bneck(x, xp, geom, *a, **k):
        mask_site[0] = None
        res = orig_b(x, xp, geom, *a, **k)
        node = res[0].grad_fn
        if node is not None:  # (a frozen block records nothing: no gradient crosses its ReLUs)
            h, w = geom
            for t in node.saved_tensors[1:4]:  # o1 (pair), o2 (pair), the block output (fp32 or pair)
                g = _gate(t)
                sites.append(g.view(-1, h, w, g.shape[1]).permute(0, 3, 1, 2).cpu())
        return res

    def conv(x2d, geom, w, b=None, relu=False, prepared=None):
        y = orig_c(x2d, geom, w, b, relu, prepared)
        if relu and y.grad_fn is not None:
            g = y.detach() > 0
            sites.append(g.view(-1, geom[0], geom[1], g.shape[1]).permute(0, 3, 1, 2).cpu())
        return y

    def upsampled(self, x):
        n = len(sites)
        rows, shape = orig_up(self, x)
        del sites[n:]  # the raw [P*H*W, 4*C] site of the GEMM form: re-recorded in the transposed convolution's own layout
        if rows.grad_fn is not None:
            p, h2, w2 = shape
            g = (rows.detach() > 0).view(p, h2, w2, -1).permute(0, 3, 1, 2).cpu()
            if mask_site[0] is not None and mask_site[0] == len(sites) - 1 and sites[-1].shape[1:] == g.shape[1:]:
                sites[-1] = torch.cat([sites[-1], g], 0)  # the next branch's positives of the same batched pass
            else:
                sites.append(g)
                mask_site[0] = len(sites) - 1
        return rows, shape

    backbone.bottleneck_pair, pb.conv_same_pair, roi_heads.MaskRCNNC4Predictor._upsampled_rows = bneck, conv, upsampled
    try:
        yield sites
    finally:
        backbone.bottleneck_pair, pb.conv_same_pair, roi_heads.MaskRCNNC4Predictor._upsampled_rows = orig_b, orig_c, orig_up


class _ForcedReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x.clamp(min=0)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


class ForcedGates:
    def __init__(self, sites, plain_relu):
        self.sites, self.off, self.p = sites, [0] * len(sites), 0
        self.plain_relu = plain_relu
        self.forced = self.plain = 0
        self.flipped = 0  # elements whose CPU-side sign differs from the recorded gate
        self.elements = 0

    def relu(self, x, *args, **kwargs):
        if x.dim() == 4 and self.sites and torch.is_grad_enabled():
            s, o = self.sites[self.p], self.off[self.p]
            if s.shape[1:] == x.shape[1:] and o + x.shape[0] <= s.shape[0]:
                m = s[o:o + x.shape[0]]
                self.off[self.p] += x.shape[0]
                self.p = (self.p + 1) % len(self.sites)
                self.forced += 1
                self.flipped += int(((x.detach() > 0) != m).sum())
                self.elements += m.numel()
                return _ForcedReLU.apply(x, m)
        self.plain += 1
        return self.plain_relu(x)

    def all_consumed(self):
        return all(o == s.shape[0] for o, s in zip(self.off, self.sites))


@contextlib.contextmanager
def forced_gates(sites):
    orig = F.relu_, F.relu
    fg = ForcedGates(sites, orig[1])
    F.relu_ = F.relu = fg.relu
    try:
        yield fg
    finally:
        F.relu_, F.relu = orig


def rel_l2(a, b):
    """{name: ||a - b|| / ||b||} over the tensors of b with a non-negligible norm."""
    return {n: (a[n] - v).norm().item() / v.norm().item() for n, v in b.items() if v.norm().item() > 1e-6 and n in a}
