"""One autograd node per ResNet bottleneck, on the pair-layout split GEMM (csrc/split_gemm.hip).

Bottleneck.forward of maskrcnn_benchmark/modeling/backbone/resnet.py:323-344 (conv1 1x1 -> FrozenBN -> ReLU ->
conv2 3x3 -> FrozenBN -> ReLU -> conv3 1x1 -> FrozenBN -> (+ shortcut) -> ReLU, FrozenBN folded into the weights) on
NHWC rows [R*H*W, C], every convolution an fp32-accurate three-term bf16 hi/lo product on the matrix cores:

* activations travel between the layers in PAIR layout only (per 32 values: 64 B hi | 64 B lo, written by the
  producing GEMM's epilogue together with bias / shortcut / ReLU), so there is no separate bias, ReLU, shortcut-add
  or operand-split pass, and no fp32 copy of the two inner activations at all;
* the 3x3 is an implicit GEMM (shifted row reads, zero line outside the map): no im2col matrix in the forward, the
  data gradient or the weight gradient;
* backward: the ReLU gate is fused with the split of the gated gradient (``gate_split_pair``: the gate is read from
  the hi halves of the saved pair activations), dX products are the same kernel against transposed weights with
  the shortcut gradient added in the epilogue, dW products contract over the rows through the LDS transpose-read
  kernel (``split_gemm_pair_tn``; the 3x3 reads its input shifted per tap, so no im2col rows exist in training either).
Saved per row: the pair forms of the input and of the two inner activations + the fp32 output.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C


def pair_weight(w2d):
    """[N, K] f32 weight matrix -> pair layout [N, 2K] bf16."""
    return _C.split_pair(w2d.contiguous())


def conv_weight_matrix(w):
    """[N, C, KH, KW] -> [N, KH*KW*C], tap-major like the implicit GEMM's contraction index."""
    n, c, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(n, kh * kw * c)


def conv_weight_matrix_t(w):
    """[N, C, KH, KW] -> [C, KH*KW*N]: the data-gradient operand (taps are visited with negated offsets)."""
    n, c, kh, kw = w.shape
    return w.permute(1, 2, 3, 0).reshape(c, kh * kw * n)


def dw_pair(gp, xp, conv=None):
    """dW[N, taps*K] = dY^T X over the rows, both operands in pair layout ([M, 2N], [M, 2K]); conv = (h, w, kh, kw):
    X is the NHWC input of a stride-1 "same" convolution and dW is tap-major [N, kh*kw*K].  The transpose-read split
    GEMM (csrc/split_gemm.hip::split_gemm_tn_kernel: three-term product, row slices summed) when the shape allows,
    else one library bf16 GEMM on the pair operands whose four hi/lo quadrants are summed (over pair-layout im2col
    rows for a convolution)."""
    n, k = gp.shape[1] // 2, xp.shape[1] // 2
    if _C.split_gemm_pair_tn_supported(n, k, conv):
        return _C.split_gemm_pair_tn(gp, xp, conv)
    if conv is not None:
        xp = _C.im2col_pair(xp, *conv)
        k = xp.shape[1] // 2
    q = torch.mm(gp.t(), xp, out_dtype=torch.float32)            # [2N, 2K]
    return q.view(n // 32, 2, 32, k // 32, 2, 32).sum(dim=(1, 4)).reshape(n, k)


def _dw(gp, xp, w, scale, conv=None):
    """Gradient of the RAW weight w [N, C, KH, KW] of a convolution evaluated with w * scale[n]: transpose-read GEMM
    slabs reduced, scaled and laid out in one launch; library fallback for shapes the kernel does not take."""
    n, c = w.shape[0], w.shape[1]
    if _C.split_gemm_pair_tn_supported(n, c, conv):
        return _C.split_gemm_pair_tn(gp, xp, conv, scale=scale, weight_shape=tuple(w.shape))
    d = dw_pair(gp, xp, conv).view(n, w.shape[2], w.shape[3], c).permute(0, 3, 1, 2)
    return d if scale is None else d * scale.view(-1, 1, 1, 1)


# Identity bottlenecks of a pair-only chain with at most this many rows run their backward through ONE native call with the
# weight gradients on a second stream (``_C.bottleneck_identity_backward``): the trunk's blocks (M = 8400 / 33400), where a
# block's kernels take 225 us and the host needs 150 us to issue them one by one.  0 = off (the A/B switch; the res5 head's
# blocks, M = 50176 and full-machine GEMMs, gain nothing).
ONE_CALL_BACKWARD_ROWS = 40000

# ---- weights prepared behind the optimizer step -----------------------------------------------------------------------
_WEIGHTS_EPOCH = [0]


def note_weights_written():
    """Called by whatever writes parameters WITHOUT autograd's version counters noticing (the fused SGD launch writes
    through raw pointers): everything a ``WeightPrepPlan`` prepared before is stale from here on."""
    _WEIGHTS_EPOCH[0] += 1


class WeightPrepPlan:
    """The pair-layout operands (forward matrix, transposed matrix) of every trainable bottleneck convolution of a model,
    prepared by ONE launch (``_C.weight_prep_pair_multi``, csrc/weight_prep_multi.hip) into buffers that live as long as the
    plan -- instead of one ``_C.weight_prep_pair`` launch per convolution inside every block's forward (42 launches at the
    head of the teacher step's GEMMs).  ``blocks``: [(key, [(w, scale) for conv1, conv2, conv3, (downsample)])], or
    [(key, [(w, scale)])] for a lone convolution (``conv_same_pair``: the RPN head's 3x3); the conv3
    and downsample forward matrices of a projection block are written side by side into ONE [N, 2 (K3 + Kd)] matrix (the
    operand of the fused conv3 + shortcut product).  ``run()`` after every optimizer step; ``lookup(key, scales)`` hands a
    block its ``wpairs`` dict while nothing has touched its weights since (``_WEIGHTS_EPOCH`` for raw-pointer writers, the
    tensors' version counters for everything else, identity of the FrozenBN scale tensors) -- else None, and the block
    prepares its weights itself as before."""

    def __init__(self, blocks):
        import numpy as np
        tile = _C.weight_prep_tile()
        self.max_taps = 1
        self.entries = {}
        rows, blk = [], []
        dev = blocks[0][1][0][0].device
        for key, convs in blocks:
            ws = [w for w, _ in convs]
            if not all(w.is_cuda and w.dtype == torch.float32 and w.dim() == 4 and w.is_contiguous() and w.device == dev
                       and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0 for w in ws):
                raise RuntimeError("WeightPrepPlan: contiguous float32 HIP weights with channel counts divisible by 32 expected")
            if len(convs) == 1:  # a lone convolution (the RPN head's 3x3): {"w": forward form, "wt": transposed form}
                (w, sc), = convs
                n, c, t = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
                f = torch.empty((n, 2 * t * c), dtype=torch.bfloat16, device=dev)
                tb = torch.empty((c, 2 * t * n), dtype=torch.bfloat16, device=dev)
                if sc is not None and not (sc.is_cuda and sc.dtype == torch.float32 and sc.is_contiguous() and sc.numel() == n):
                    raise RuntimeError("WeightPrepPlan: scale must be a contiguous float32 [N] HIP tensor")
                self.max_taps = max(self.max_taps, t)
                rows.append((w.data_ptr(), 0 if sc is None else sc.data_ptr(), f.data_ptr(), tb.data_ptr(), 2 * f.stride(0),
                             2 * tb.stride(0), n | (c << 32), t))
                blk.extend((len(rows) - 1, ti) for ti in range((n // tile) * (c // tile)))
                self.entries[key] = [{"w": f, "wt": tb}, (w,), (sc,), None]
                continue
            (w1, _), (w2, _), (w3, _) = convs[:3]
            wd = convs[3][0] if len(convs) > 3 else None
            k3 = w3.shape[1] * w3.shape[2] * w3.shape[3]
            kd = wd.shape[1] * wd.shape[2] * wd.shape[3] if wd is not None else 0
            f1 = torch.empty((w1.shape[0], 2 * w1[0].numel()), dtype=torch.bfloat16, device=dev)
            f2 = torch.empty((w2.shape[0], 2 * w2[0].numel()), dtype=torch.bfloat16, device=dev)
            f3d = torch.empty((w3.shape[0], 2 * (k3 + kd)), dtype=torch.bfloat16, device=dev)
            fwd = [(f1, f1.data_ptr()), (f2, f2.data_ptr()), (f3d, f3d.data_ptr())] + ([(f3d, f3d.data_ptr() + 4 * k3)] if wd is not None else [])
            bwd = []
            for (w, sc), (fbuf, fptr) in zip(convs, fwd):
                n, c, t = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
                tb = torch.empty((c, 2 * t * n), dtype=torch.bfloat16, device=dev)
                bwd.append(tb)
                if sc is not None and not (sc.is_cuda and sc.dtype == torch.float32 and sc.is_contiguous() and sc.numel() == n):
                    raise RuntimeError("WeightPrepPlan: scale must be a contiguous float32 [N] HIP tensor")
                self.max_taps = max(self.max_taps, t)
                rows.append((w.data_ptr(), 0 if sc is None else sc.data_ptr(), fptr, tb.data_ptr(), 2 * fbuf.stride(0),
                             2 * tb.stride(0), n | (c << 32), t))
                blk.extend((len(rows) - 1, ti) for ti in range((n // tile) * (c // tile)))
            wp = {"w1": f1, "w2": f2, "w3": f3d[:, :2 * k3], "wd": f3d[:, 2 * k3:] if wd is not None else None,
                  "wts": tuple(bwd) + ((None,) if wd is None else ())}
            if wd is not None:
                wp["w3d"] = f3d
            self.entries[key] = [wp, tuple(w for w, _ in convs), tuple(sc for _, sc in convs), None]
        table = np.zeros((len(rows), 8), dtype=np.int64)
        for i, r in enumerate(rows):
            table[i] = r
        self.items = torch.from_numpy(table.view(np.uint8).reshape(-1)).to(dev)
        self.blocks = torch.tensor(blk, dtype=torch.int32, device=dev).reshape(-1, 2)
        self._weights = [w for e in self.entries.values() for w in e[1]]
        self._pointers = [w.data_ptr() for w in self._weights]
        self.epoch = -1
        self.event = self.stream = None
        self.rebuild = False  # set by a lookup that found other FrozenBN scales than the tables hold

    def describes(self):
        """The device tables still point at the weights' storage (a ``.to()`` / ``.data`` swap moves it) and no block met
        other scale tensors than it was built with."""
        return not self.rebuild and self._pointers == [w.data_ptr() for w in self._weights]

    def run(self):
        """Prepare everything from the weights as they are now (one launch on the current stream)."""
        _C.weight_prep_pair_multi(self.items, self.blocks, self.max_taps)
        self.epoch = _WEIGHTS_EPOCH[0]
        for e in self.entries.values():
            e[3] = [w._version for w in e[1]]
        self.stream = torch.cuda.current_stream()
        self.event = torch.cuda.Event()
        self.event.record(self.stream)

    def lookup(self, key, scales):
        """The block's ``wpairs`` (forward forms, ``"wts"`` = transposed forms), or None: prepare them yourself."""
        ent = self.entries.get(key)
        if ent is None or self.epoch != _WEIGHTS_EPOCH[0]:
            return None
        wp, ws, scs, versions = ent
        if len(scales) != len(scs) or any(a is not b for a, b in zip(scales, scs)):
            self.rebuild = True
            return None
        if versions != [w._version for w in ws]:
            return None
        cur = torch.cuda.current_stream()
        if cur != self.stream:
            cur.wait_event(self.event)
        return wp


_NAN_CELL = {}


def nan_placeholder(device, rows, cols):
    """[rows, cols] fp32 view of ONE NaN element per device (zero strides): the handle autograd routes where a value exists
    in pair layout only.  The element is created once -- a fill launch per placeholder was 24 launches per teacher step."""
    cell = _NAN_CELL.get(device)
    if cell is None:
        cell = _NAN_CELL[device] = torch.full((1,), float("nan"), dtype=torch.float32, device=device)
    return cell.expand(rows, cols)


def is_placeholder(t):
    """A zero-stride stand-in for a block output that exists only in pair layout (see ``_BottleneckPair.forward``)."""
    return t is not None and t.numel() > 1 and all(st == 0 for st in t.stride())


class GradLink:
    """Side channel between two consecutive ``_BottleneckPair`` nodes of a pair-only chain: the block above leaves the
    gradient w.r.t. the block below's output here, in pair layout and ALREADY gated by that output's ReLU
    (``_C.split_gemm_pair_rp_gated``); the fp32 gradient autograd routes between the nodes is then a zero-stride
    placeholder, as the activation was in the forward.  The block below takes it (once) instead of gating and
    splitting an fp32 gradient."""

    __slots__ = ("grad_pair", "claimed")

    def __init__(self):
        self.grad_pair = None
        self.claimed = False  # a pair-only output feeds exactly ONE block (its fp32 handle is a placeholder)


class _BottleneckPair(Function):
    @staticmethod
    def forward(ctx, x, xp, geom, w1, s1, b1, w2, s2, b2, w3, s3, b3, wd, sd, want_pair, wpairs, pool, want_f32=True,
                link_in=None, link_out=None, select=None, pool_only_ok=False):
        """x [M, Cin] f32 rows conv1 reads (may be None when wd is given and no input gradient is wanted), xp its pair
        form or None; geom = (h, w) of the map the rows tile; w1/w2/w3/wd RAW convolution weights (wd None = identity
        shortcut) with their folded FrozenBN scales s1/s2/s3/sd (per output channel, no gradient) and shifts b1/b2,
        b3 = the conv3 (+ shortcut) shift; wpairs: optional cached pair weights of a frozen block.
        want_f32 False (with want_pair): the result is written in PAIR layout only -- the next block reads it as its conv1
        operand AND as its identity shortcut (hi + lo, exact in fp32; ``split_gemm_pair(residual_pair=...)``), so the 4
        bytes per element of the fp32 copy are neither written nor re-read.  The fp32 output slot then carries a
        zero-stride placeholder (never read: it carries the shape and routes the gradient between the autograd nodes of a
        chain)."""
        h, w = geom
        if xp is None:
            xp = _C.split_pair(x)
        kh, kw = w2.shape[2], w2.shape[3]
        need_bwd = any(ctx.needs_input_grad)
        wts = None
        if wpairs is None:
            # fold + matrix form + split (+ the transposed operands of the data gradients) in one launch per weight
            p1, t1 = _C.weight_prep_pair(w1, s1, need_bwd)
            p2, t2 = _C.weight_prep_pair(w2, s2, need_bwd)
            p3, t3 = _C.weight_prep_pair(w3, s3, need_bwd)
            pd, td = _C.weight_prep_pair(wd, sd, need_bwd) if wd is not None else (None, None)
            wpairs = {"w1": p1, "w2": p2, "w3": p3, "wd": pd}
            wts = (t1, t2, t3, td)
        elif need_bwd:
            wts = wpairs.get("wts")  # prepared behind the optimizer step together with the forward forms (WeightPrepPlan)
        f32 = bool(want_f32 or pool or not want_pair)
        x_real = x is not None and not is_placeholder(x)
        _, o1p = _C.split_gemm_pair(xp, wpairs["w1"], b1, None, True, False, True)
        _, o2p = _C.split_gemm_pair(o1p, wpairs["w2"], b2, None, True, False, True, conv=(h, w, kh, kw, False))
        if wd is not None:
            # conv3 and the projection shortcut as ONE product over K = [conv2 output | block input] against
            # [w3 | wd] (pair rows concatenate block-wise): the shortcut tensor is never written or re-read
            w3d = wpairs.get("w3d")
            if w3d is None:
                w3d = torch.cat([wpairs["w3"], wpairs["wd"]], 1)
            out, outp = _C.split_gemm_pair(o2p, w3d, b3, None, True, f32, want_pair, a2_pair=xp)
        elif x_real:
            out, outp = _C.split_gemm_pair(o2p, wpairs["w3"], b3, x, True, f32, want_pair)
        pooled_k = None
        if wd is None and not x_real:  # the block input exists only as its pair form: shortcut = hi + lo
            fused_pool = (pool and 32 <= h * w <= 64
                          and _C.split_gemm_pair_pool_supported(o2p.shape[0], wpairs["w3"].shape[0], o2p.shape[1] // 2, h * w))
            if fused_pool:
                # the head's average pooling in THIS GEMM's epilogue (no second pass over the [rows, 2048] result); a pass
                # that needs neither the backward nor the positives' maps (the no-grad teacher pass) writes the pooled rows only
                keep = need_bwd or want_pair or select is not None or not pool_only_ok
                out, outp, pooled_k = _C.split_gemm_pair_rp_pool(o2p, wpairs["w3"], b3, xp, True, keep, want_pair, h * w)
                if out is None and outp is None:
                    out = nan_placeholder(pooled_k.device, o2p.shape[0], pooled_k.shape[1])  # never read
            else:
                out, outp = _C.split_gemm_pair(o2p, wpairs["w3"], b3, None, True, f32, want_pair, residual_pair=xp)
        # pool: also return the mean over the h*w rows of every map (the head's average pooling) as an output of THIS
        # node, so that its gradient is broadcast inside the fused gate + split kernel of the backward instead of
        # being materialised ([rows, C] expand) and added to the dense gradient by two tensor ops
        if pooled_k is not None:
            pooled = pooled_k
        else:
            pooled = out.view(-1, h * w, out.shape[1]).mean(dim=1) if pool else None
        # select [S] int64: also return the rows of the maps `select` ([S, h*w, C]; what the mask head reads: the res5
        # features of the positive RoIs) as an output of THIS node, so that their gradient reaches the backward's first
        # kernel as S dense maps -- an index backward would scatter it into a zero [rows, C] tensor (822 MB written and
        # read back at the step's size) first
        out_sel = out.view(-1, h * w, out.shape[1]).index_select(0, select) if (select is not None and f32) else None
        gate_src = out if out is not None else outp  # the last ReLU's gate: the fp32 result or the hi halves of its pair form
        if out is None:  # NaN-filled: any consumer outside the one-block contract shows up in the loss instead of reading garbage
            out = nan_placeholder(outp.device, outp.shape[0], outp.shape[1] // 2)
        ctx.save_for_backward(xp, o1p, o2p, gate_src, w1, w2, w3, wd, s1, s2, s3, sd, select if out_sel is not None else None)
        ctx.wts = wts
        ctx.geom = (h, w, kh, kw)
        # link_in: this block's input exists only in pair layout and comes from a block that reads its gradient from the
        # link; link_out: this block's own output is pair-only and the block above may leave its gradient there
        ctx.link_in = link_in if (wd is None and not x_real) else None
        ctx.link_out = link_out if not f32 else None
        ctx.set_materialize_grads(False)  # no zero tensors for absent / non-differentiable gradient slots
        if outp is not None:
            ctx.mark_non_differentiable(outp)
        return out, outp, pooled, out_sel

    @staticmethod
    @once_differentiable
    def backward(ctx, dout, _dpair, dpooled, dsel):
        link_out, link_in = ctx.link_out, ctx.link_in
        linked = link_out.grad_pair if link_out is not None else None
        if link_out is not None:
            link_out.grad_pair = None
        if dout is None and dpooled is None and linked is None and dsel is None:
            return (None,) * 22
        xp, o1p, o2p, out, w1, w2, w3, wd, s1, s2, s3, sd, select = ctx.saved_tensors   # out: fp32 result or its pair form (gate)
        h, w, kh, kw = ctx.geom
        need = ctx.needs_input_grad
        need_x, need_w1, need_w2, need_w3, need_wd = need[0], need[3], need[6], need[9], need[12]
        wts = ctx.wts
        if wts is None:  # cached (frozen) weights in the forward: only the input gradient can be wanted
            wts = tuple(_C.weight_prep_pair(t, sc, True)[1] if t is not None else None
                        for t, sc in ((w1, s1), (w2, s2), (w3, s3), (wd, sd)))
        t1, t2, t3, td = wts
        n3 = w3.shape[0]
        # the input gradient goes to the block below through the link when that block left one and the fused form applies
        to_link = (need_x and wd is None and link_in is not None and xp.shape[1] // 2 >= 128)
        if linked is not None:
            if dout is not None and not is_placeholder(dout):
                raise RuntimeError("_BottleneckPair: a pair-only output received a real fp32 gradient next to the linked one "
                                   "(a second consumer of the placeholder: hook, retain_grad, slicing ...)")
            g3p, g3 = linked, None  # gated and split by the block above (dout is only a placeholder)
        else:
            # gate of the block's last ReLU, fused with the split; the identity shortcut needs the gated gradient too: in
            # fp32, or (linked form) in the pair layout just written
            sel_grad = slot = None
            if dsel is not None:
                if dout is None and dpooled is None:
                    dpooled = dsel.new_zeros((out.shape[0] // (h * w), n3))
                slot = torch.full((out.shape[0] // (h * w),), -1, dtype=torch.int32, device=dsel.device)
                slot[select] = torch.arange(select.numel(), dtype=torch.int32, device=dsel.device)
                sel_grad = dsel.reshape(-1, n3)
            g3p, g3 = _C.gate_split_pair(None if dout is None else dout.reshape(-1, n3), out,
                                         want_f32=(wd is None and need_x and not to_link), pooled=dpooled, pool_rows=h * w,
                                         selected=sel_grad, group_slot=slot)
        if (ONE_CALL_BACKWARD_ROWS and linked is not None and wd is None and to_link and need_w1 and need_w2 and need_w3
                and 0 < g3p.shape[0] <= ONE_CALL_BACKWARD_ROWS and t1 is not None and t2 is not None and t3 is not None
                and n3 % 128 == 0 and w1.shape[0] % 128 == 0 and kh % 2 == 1 and kw % 2 == 1):
            # an identity block in the middle of a small pair-only chain (the trunk): its nine launches behind one native
            # call, the three weight gradients on the second stream beside the data-gradient chain
            from ..engine.trainer import branch_stream
            gx, dw1, dw2, dw3 = _C.bottleneck_identity_backward(g3p, xp, o1p, o2p, t1, t2, t3, (s1, s2, s3), (h, w, kh, kw),
                                                               (w1.shape, w2.shape, w3.shape), branch_stream())
            link_in.grad_pair = gx
            dx = nan_placeholder(gx.device, xp.shape[0], xp.shape[1] // 2)
            return (dx, None, None, dw1, None, None, dw2, None, None, dw3, None, None, None, None, None, None, None, None, None,
                    None, None, None)
        dw3 = _dw(g3p, o2p, w3, s3) if need_w3 else None
        _, g2p = _C.split_gemm_pair_gated(g3p, t3, o2p)                  # (dY W3) gated by relu(o2), split: one kernel
        dw2 = _dw(g2p, o1p, w2, s2, (h, w, kh, kw)) if need_w2 else None
        dx = dw1 = dwd = None
        if need_x or need_w1:
            _, g1p = _C.split_gemm_pair_gated(g2p, t2, o1p, conv=(h, w, kh, kw, True))
            if need_w1:
                dw1 = _dw(g1p, xp, w1, s1)
            if need_x:
                if to_link:
                    # (dY1 W1 + shortcut gradient) gated by the block input's ReLU, in pair layout only: what the block
                    # below would compute from an fp32 gradient with one more pass over it
                    _, link_in.grad_pair = _C.split_gemm_pair_rp_gated(g1p, t1, g3p, xp)
                    dx = nan_placeholder(g1p.device, xp.shape[0], xp.shape[1] // 2)
                else:
                    if wd is not None:
                        res, _ = _C.split_gemm_pair(g3p, td)
                        dx, _ = _C.split_gemm_pair(g1p, t1, None, res)
                    elif g3 is not None:
                        dx, _ = _C.split_gemm_pair(g1p, t1, None, g3)
                    else:  # linked gradient from above, fp32 gradient wanted below: the shortcut term from its pair form
                        dx, _ = _C.split_gemm_pair(g1p, t1, None, None, False, True, False, residual_pair=g3p)
        if wd is not None and need_wd:
            dwd = _dw(g3p, xp, wd, sd)
        return (dx, None, None, dw1, None, None, dw2, None, None, dw3, None, None, dwd, None, None, None, None, None, None,
                None, None, None)


def bottleneck_pair(x, xp, geom, w1, b1, w2, b2, w3, b3, wd=None, want_pair=False, wpairs=None, pool=False,
                    scales=(None, None, None, None), want_f32=True, select=None, pool_only_ok=False):
    """(out f32 [M, Cout], out in pair layout or None[, mean of out over the h*w rows of every map when ``pool``]) of
    one bottleneck on the rows x [M, Cin] of an (h, w) map.  w1/w2/w3/wd are the convolution weights as the model
    stores them; ``scales`` = their folded FrozenBN scales (None = weights already folded).  ``pool_only_ok``: the caller
    reads nothing but the pooled rows of a no-grad pass, so the [M, Cout] result itself need not be written (its handle is
    then a NaN placeholder)."""
    s1, s2, s3, sd = scales
    pair_only = want_pair and not want_f32 and not pool
    link_in = getattr(xp, "_ovis_grad_link", None) if (xp is not None and (x is None or is_placeholder(x))) else None
    if link_in is not None and torch.is_grad_enabled():
        if link_in.claimed:
            raise RuntimeError("bottleneck_pair: a block output kept in pair layout only can feed one block -- its fp32 "
                               "handle is a placeholder, so the gradients of two consumers could not be summed")
        link_in.claimed = True
    link_out = GradLink() if pair_only else None
    out, outp, pooled, out_sel = _BottleneckPair.apply(x, xp, geom, w1, s1, b1, w2, s2, b2, w3, s3, b3, wd, sd, want_pair,
                                                       wpairs, pool, want_f32, link_in, link_out, select, pool_only_ok)
    if out_sel is not None:
        out._ovis_selected = (select, out_sel)  # [S, h*w, C] rows of the maps `select`, an output of the same node
    if link_out is not None and outp is not None:
        outp._ovis_grad_link = link_out  # travels with the pair tensor to the block that consumes it
    return (out, outp, pooled) if pool else (out, outp)


class _ConvSamePair(Function):
    """y = relu?(conv(x, w) + b) for a stride-1 "same" odd-kernel convolution on NHWC rows, trainable: forward = implicit
    split GEMM with the bias / ReLU epilogue, backward = fused gate + split, implicit GEMM with negated taps for dX,
    transpose-read GEMM for dW (csrc/split_gemm.hip).  Used for the RPN head's 3x3 (rpn.py:74-106) when it trains."""

    @staticmethod
    def forward(ctx, x2d, geom, w, b, relu, prepared=None):
        h, wd_ = geom
        n, c, kh, kw = w.shape
        xp = _C.split_pair(x2d)
        if prepared is not None:  # {"w", "wt"} of a WeightPrepPlan: written behind the optimizer step
            wp, wt = prepared["w"], prepared["wt"]
        else:
            wp, wt = _C.weight_prep_pair(w, None, ctx.needs_input_grad[0])
        y, _ = _C.split_gemm_pair(xp, wp, b, None, relu, True, False, conv=(h, wd_, kh, kw, False))
        ctx.wt = wt
        ctx.save_for_backward(xp, y if relu else None, w)
        ctx.geom = (h, wd_, kh, kw)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        xp, y, w = ctx.saved_tensors
        h, wd_, kh, kw = ctx.geom
        n, c = w.shape[0], w.shape[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[3]
        gp, g32 = _C.gate_split_pair(dy.reshape(-1, n), y, want_f32=need_b)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx, _ = _C.split_gemm_pair(gp, ctx.wt, conv=(h, wd_, kh, kw, True))
        if ctx.needs_input_grad[2]:
            dw = _dw(gp, xp, w, None, (h, wd_, kh, kw))
        if need_b:
            db = g32.sum(0)
        return dx, None, dw, db, None, None


def conv_same_pair(x2d, geom, w, b=None, relu=False, prepared=None):
    """x2d [N*H*W, C] f32 NHWC rows of (H, W) maps -> [N*H*W, Cout] f32 (C, Cout % 32 == 0).  ``prepared``: the weight's
    pair forms from a ``WeightPrepPlan`` lookup, or None (prepared here)."""
    return _ConvSamePair.apply(x2d, geom, w, b, relu, prepared)
