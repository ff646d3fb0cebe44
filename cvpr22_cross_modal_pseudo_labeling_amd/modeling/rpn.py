"""Region proposal network: anchors, head, proposal selection (top-k -> decode -> clip -> NMS), loss.

Counterpart of maskrcnn_benchmark/modeling/rpn/anchor_generator.py:34-128,200-290,
rpn/rpn.py:74-197, rpn/inference.py:15-205 and rpn/loss.py:21-131 for the single-level (C4)
case every shipped config uses.  The proposal path calls the HIP NMS through ``layers.nms``.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..layers import smooth_l1_loss
from .box_coder import BoxCoder
from .matcher import BalancedPositiveNegativeSampler, Matcher
from .structures import BoxList, box_iou, boxlist_nms, cat_boxlist, remove_small_boxes


# ---- anchors (anchor_generator.py:200-290; the classic Faster R-CNN enumeration) ------------------------
def _whctrs(a):
    w, h = a[2] - a[0] + 1, a[3] - a[1] + 1
    return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)


def _mkanchors(ws, hs, x_ctr, y_ctr):
    ws, hs = ws[:, None], hs[:, None]
    return np.hstack((x_ctr - 0.5 * (ws - 1), y_ctr - 0.5 * (hs - 1), x_ctr + 0.5 * (ws - 1), y_ctr + 0.5 * (hs - 1)))


def generate_cell_anchors(stride, sizes, aspect_ratios):
    base = np.array([1, 1, stride, stride], dtype=np.float64) - 1
    w, h, x_ctr, y_ctr = _whctrs(base)
    ratios = np.array(aspect_ratios, dtype=np.float64)
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    ratio_anchors = _mkanchors(ws, hs, x_ctr, y_ctr)
    scales = np.array(sizes, dtype=np.float64) / stride
    out = []
    for a in ratio_anchors:
        w, h, x_ctr, y_ctr = _whctrs(a)
        out.append(_mkanchors(w * scales, h * scales, x_ctr, y_ctr))
    return torch.from_numpy(np.vstack(out)).float()


class AnchorGenerator(nn.Module):
    def __init__(self, sizes, aspect_ratios, stride, straddle_thresh=0):
        super().__init__()
        self.stride = stride
        self.straddle_thresh = straddle_thresh
        self.register_buffer("cell_anchors", generate_cell_anchors(stride, sizes, aspect_ratios), persistent=False)
        self._grid_cache = {}

    def num_anchors_per_location(self):
        return self.cell_anchors.shape[0]

    def grid_anchors(self, grid_h, grid_w):
        key = (grid_h, grid_w, self.cell_anchors.device)
        if key not in self._grid_cache:
            dev = self.cell_anchors.device
            sx = torch.arange(0, grid_w * self.stride, step=self.stride, dtype=torch.float32, device=dev)
            sy = torch.arange(0, grid_h * self.stride, step=self.stride, dtype=torch.float32, device=dev)
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
            self._grid_cache[key] = (shifts.view(-1, 1, 4) + self.cell_anchors.view(1, -1, 4)).reshape(-1, 4)
        return self._grid_cache[key]

    def image_wh(self, sizes, device):
        """[N, 2] float32 (width, height) of the images (BoxList.size order), cached per size tuple: an operand of the
        device-side proposal decode."""
        key = (tuple(sizes), device)
        if key not in self._grid_cache:
            self._grid_cache[key] = torch.tensor([[float(w), float(h)] for (w, h) in sizes], dtype=torch.float32,
                                                 device=device)
        return self._grid_cache[key]

    def visibility(self, anchors, image_w, image_h):
        if self.straddle_thresh < 0:
            return torch.ones(anchors.shape[0], dtype=torch.bool, device=anchors.device)
        t = self.straddle_thresh
        return ((anchors[:, 0] >= -t) & (anchors[:, 1] >= -t) & (anchors[:, 2] < image_w + t)
                & (anchors[:, 3] < image_h + t))

    def forward(self, image_sizes, feature):
        """-> list (one per image) of BoxList with a 'visibility' field."""
        anchors = self.grid_anchors(feature.shape[-2], feature.shape[-1])
        out = []
        for (h, w) in image_sizes:
            b = BoxList(anchors, (w, h))
            key = ("visibility", int(w), int(h), feature.shape[-2], feature.shape[-1], anchors.device)
            if key not in self._grid_cache:  # a function of the image size and the grid only
                self._grid_cache[key] = self.visibility(anchors, w, h)
            b.add_field("visibility", self._grid_cache[key])
            out.append(b)
        return out


class RPNHead(nn.Module):  # rpn.py:74-106
    split_gemm = True  # False = plain convolutions (cross-check in the tests)

    def __init__(self, in_channels, num_anchors):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self.cls_logits = nn.Conv2d(in_channels, num_anchors, kernel_size=1, stride=1)
        self.bbox_pred = nn.Conv2d(in_channels, num_anchors * 4, kernel_size=1, stride=1)
        for l in (self.conv, self.cls_logits, self.bbox_pred):
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)

    def forward(self, feature):
        if self._gemm_ok(feature):
            return self._forward_gemm(feature)
        c = self.conv
        if (feature.is_cuda and c.in_channels % 128 == 0 and c.out_channels % 128 == 0 and c.kernel_size == (3, 3)
                and c.padding == (1, 1) and c.stride == (1, 1) and self.split_gemm):
            # trainable head (teacher configuration): the 3x3 through the split-GEMM autograd node, the two small 1x1
            # predictors as linear maps of its NHWC rows (layers/cross_modal.py::linear_mfma: the 60-channel box
            # predictor on the split GEMM, the 15-channel objectness on the exact-fp32 MFMA GEMM) -- no library
            # convolution in the step; NCHW views of the results
            from ..layers.cross_modal import linear_mfma
            from ..layers.pair_bottleneck import conv_same_pair
            n, ch, h, w = feature.shape
            plan = self.__dict__.get("_prep_plan")  # the 3x3's GEMM operands, prepared behind the optimizer step if still fresh
            prepared = plan.lookup(id(self), (None,)) if plan is not None else None
            t = conv_same_pair(feature.permute(0, 2, 3, 1).reshape(-1, ch), (h, w), c.weight, c.bias, True, prepared)
            a = self.cls_logits.out_channels
            cls = linear_mfma(t, self.cls_logits.weight.view(a, -1), self.cls_logits.bias)
            box = linear_mfma(t, self.bbox_pred.weight.view(4 * a, -1), self.bbox_pred.bias)
            return cls.view(n, h, w, a).permute(0, 3, 1, 2), box.view(n, h, w, 4 * a).permute(0, 3, 1, 2)
        t = F.relu(self.conv(feature))
        return self.cls_logits(t), self.bbox_pred(t)

    def prep_plan_convs(self):
        """[(weight, scale)] for ``modeling/backbone.py::prepare_weights_ahead``: the trainable 3x3 on the split GEMM."""
        c = self.conv
        if (self.split_gemm and c.weight.requires_grad and c.weight.is_cuda and c.in_channels % 128 == 0
                and c.out_channels % 128 == 0 and c.kernel_size == (3, 3) and c.padding == (1, 1) and c.stride == (1, 1)):
            return [(c.weight, None)]
        return None

    def _gemm_ok(self, feature):
        c = self.conv
        no_grad = not torch.is_grad_enabled() or not (feature.requires_grad or any(p.requires_grad for p in self.parameters()))
        return (feature.is_cuda and no_grad and c.in_channels % 32 == 0 and c.out_channels % 32 == 0
                and c.kernel_size == (3, 3) and c.padding == (1, 1) and c.stride == (1, 1) and self.split_gemm)

    def _forward_gemm(self, feature):
        """Frozen head on the split GEMM (csrc/split_gemm.hip): the 3x3 as an implicit GEMM with bias + ReLU in the
        epilogue and its result only in pair layout, the two 1x1 predictors as ONE GEMM over the concatenated (zero-
        padded) weight; returns NCHW views of the NHWC result.  MIOpen's fp32 Winograd kernel for the 3x3
        (1024 -> 1024 on 50x84) takes 1.6 ms per batch."""
        from .. import _C
        from ..layers.pair_bottleneck import conv_weight_matrix, pair_weight
        n, c, h, w = feature.shape
        ws = (self.conv.weight, self.conv.bias, self.cls_logits.weight, self.cls_logits.bias, self.bbox_pred.weight,
              self.bbox_pred.bias)
        key = tuple((id(t), t._version) for t in ws)
        if getattr(self, "_gemm_cache", None) is None or self._gemm_cache[0] != key:
            a = self.cls_logits.out_channels
            n11 = -(-(5 * a) // 4) * 4
            w11 = feature.new_zeros((n11, c))
            b11 = feature.new_zeros((n11,))
            w11[:a] = self.cls_logits.weight.detach().view(a, c)
            w11[a:5 * a] = self.bbox_pred.weight.detach().view(4 * a, c)
            b11[:a] = self.cls_logits.bias.detach()
            b11[a:5 * a] = self.bbox_pred.bias.detach()
            self._gemm_cache = (key, pair_weight(conv_weight_matrix(self.conv.weight.detach())),
                                self.conv.bias.detach().contiguous(), pair_weight(w11), b11)
        _, w3p, b3, w11p, b11 = self._gemm_cache
        a = self.cls_logits.out_channels
        xp = _C.split_pair(feature.detach().permute(0, 2, 3, 1).contiguous().view(-1, c))
        _, tp = _C.split_gemm_pair(xp, w3p, b3, None, True, False, True, conv=(h, w, 3, 3, False))
        y, _ = _C.split_gemm_pair(tp, w11p, b11)
        y = y.view(n, h, w, -1)
        return y[..., :a].permute(0, 3, 1, 2), y[..., a:5 * a].permute(0, 3, 1, 2)


def permute_and_flatten(layer, n, a, c, h, w):  # rpn/utils.py
    return layer.reshape(n, -1, c, h, w).permute(0, 3, 4, 1, 2).reshape(n, -1, c)


class RPNPostProcessor(nn.Module):  # inference.py:15-140 (single feature map)
    """Proposal selection.  On the device (``anchor_generator`` attached by ``RPNModule``, HIP tensors) the chain after
    the top-k is three launches for the whole batch: decode + clip + small-box flag (``_C.rpn_decode``), then the NMS
    mask and reduce kernels on the already score-sorted candidates (``_C.nms_presorted_batched``), then one gather; the
    host reads back ONE small tensor of survivor counts.  The tensor-op formulation below (``forward_tensor_ops``: the
    reference's sequence, ~40 launches and 2 host syncs per image) serves CPU tensors and is the cross-check."""

    def __init__(self, pre_nms_top_n, post_nms_top_n, nms_thresh, min_size, box_coder):
        super().__init__()
        self.pre_nms_top_n = pre_nms_top_n
        self.post_nms_top_n = post_nms_top_n
        self.nms_thresh = nms_thresh
        self.min_size = min_size
        self.box_coder = box_coder
        self.device_pipeline = True  # False = tensor ops also on the device (cross-check in the tests)

    def _on_device(self, objectness):
        return (self.device_pipeline and objectness.is_cuda and self.nms_thresh > 0
                and self.__dict__.get("anchor_generator") is not None)

    def _candidates(self, anchors, objectness, box_regression, pre):
        """sigmoid -> top-k (sorted) -> decoded, clipped candidate boxes [N, pre, 4] + drop flags [N, pre]."""
        from .. import _C
        n, a, h, w = objectness.shape
        obj = permute_and_flatten(objectness, n, a, 1, h, w).view(n, -1).sigmoid()
        scores, topk_idx = _C.topk_sorted(obj, pre)  # (torch.topk: ~80 launches for 12000 of 63000)
        ag = self.__dict__["anchor_generator"]
        boxes, drop = _C.rpn_decode(box_regression, topk_idx, ag.cell_anchors, ag.image_wh([b.size for b in anchors], obj.device),
                                    self.box_coder.weights, self.box_coder.bbox_xform_clip, self.min_size, ag.stride)
        return scores, boxes, drop

    @staticmethod
    def _gather(boxes, scores, keep, post):
        sel = keep[:, :post]
        return torch.gather(boxes, 1, sel.unsqueeze(-1).expand(-1, -1, 4)), torch.gather(scores, 1, sel)

    @staticmethod
    def _with_gt(boxlist, target):  # inference.py:51-74
        gt = BoxList(target.bbox, target.size)
        gt.add_field("objectness", torch.ones(len(gt), device=gt.bbox.device))
        return cat_boxlist((boxlist, gt))

    def forward(self, anchors, objectness, box_regression, targets=None, add_gt=False):
        if not self._on_device(objectness):
            return self.forward_tensor_ops(anchors, objectness, box_regression, targets, add_gt)
        return self.finish(self.launch(anchors, objectness, box_regression), targets, add_gt)

    def launch(self, anchors, objectness, box_regression):
        """Device pipeline, first half: every launch of the selection, no host read -- the caller may issue independent work
        (the RPN loss and its backward, on another stream) before ``finish`` waits for the survivor counts."""
        from .. import _C
        n, a, h, w = objectness.shape
        scores, boxes, drop = self._candidates(anchors, objectness, box_regression, min(self.pre_nms_top_n, a * h * w))
        keep, counts = _C.nms_presorted_batched(boxes, drop, self.nms_thresh)
        pb, ps = self._gather(boxes, scores, keep, self.post_nms_top_n)
        return anchors, pb, ps, counts

    def finish(self, launched, targets=None, add_gt=False):
        anchors, pb, ps, counts = launched
        n = pb.shape[0]
        cnt = counts[:, 0].tolist()  # the one host read of the selection
        result = []
        for i in range(n):
            m = min(cnt[i], self.post_nms_top_n)
            boxlist = BoxList(pb[i, :m], anchors[i].size)
            boxlist.add_field("objectness", ps[i, :m])
            if add_gt and targets is not None:
                boxlist = self._with_gt(boxlist, targets[i])
            result.append(boxlist)
        return result

    def forward_tensor_ops(self, anchors, objectness, box_regression, targets=None, add_gt=False):
        n, a, h, w = objectness.shape
        objectness = permute_and_flatten(objectness, n, a, 1, h, w).view(n, -1).sigmoid()
        box_regression = permute_and_flatten(box_regression, n, a, 4, h, w)
        pre_nms_top_n = min(self.pre_nms_top_n, a * h * w)
        objectness, topk_idx = objectness.topk(pre_nms_top_n, dim=1, sorted=True)
        batch_idx = torch.arange(n, device=objectness.device)[:, None]
        box_regression = box_regression[batch_idx, topk_idx]
        concat_anchors = torch.stack([b.bbox for b in anchors], 0)[batch_idx, topk_idx]
        proposals = self.box_coder.decode(box_regression.reshape(-1, 4), concat_anchors.reshape(-1, 4)).view(n, -1, 4)
        result = []
        for i in range(n):
            boxlist = BoxList(proposals[i], anchors[i].size)
            boxlist.add_field("objectness", objectness[i])
            boxlist = boxlist.clip_to_image(remove_empty=False)
            boxlist = remove_small_boxes(boxlist, self.min_size)
            boxlist = boxlist_nms(boxlist, self.nms_thresh, max_proposals=self.post_nms_top_n,
                                  score_field="objectness")
            if add_gt and targets is not None:
                boxlist = self._with_gt(boxlist, targets[i])
            result.append(boxlist)
        return result

    def forward_with(self, other, anchors, objectness, box_regression, targets=None, add_gt=False):
        """Proposals of THIS selector and of ``other`` (same threshold / min size, other.pre_nms_top_n <= ours) from ONE
        decode + NMS pass: candidates are sorted by score and greedy NMS only lets higher-scored boxes suppress, so the
        survivors among the first ``other.pre_nms_top_n`` candidates ARE ``other``'s NMS result -- and, survivors being
        listed by ascending rank, a PREFIX of ours.  Returns (ours, theirs) -- what two separate ``forward`` calls
        return (ours with the ground truth appended)."""
        assert other.pre_nms_top_n <= self.pre_nms_top_n and other.nms_thresh == self.nms_thresh \
            and other.min_size == self.min_size
        n, a, h, w = objectness.shape
        pre = min(self.pre_nms_top_n, a * h * w)
        pre_other = min(other.pre_nms_top_n, a * h * w)
        if self._on_device(objectness):
            from .. import _C
            scores, boxes, drop = self._candidates(anchors, objectness, box_regression, pre)
            keep, counts = _C.nms_presorted_batched(boxes, drop, self.nms_thresh, below=pre_other)
            pb, ps = self._gather(boxes, scores, keep, max(self.post_nms_top_n, other.post_nms_top_n))
            cnt = counts.tolist()  # the one host read: [survivors, survivors among the first pre_other] per image
            ours, theirs = [], []
            for i in range(n):
                m, t = min(cnt[i][0], self.post_nms_top_n), min(cnt[i][1], other.post_nms_top_n)
                sub = BoxList(pb[i, :t], anchors[i].size)
                sub.add_field("objectness", ps[i, :t])
                theirs.append(sub)
                mine = BoxList(pb[i, :m], anchors[i].size)
                mine.add_field("objectness", ps[i, :m])
                if add_gt and targets is not None:
                    mine = self._with_gt(mine, targets[i])
                ours.append(mine)
            return ours, theirs
        objectness = permute_and_flatten(objectness, n, a, 1, h, w).view(n, -1).sigmoid()
        box_regression = permute_and_flatten(box_regression, n, a, 4, h, w)
        objectness, topk_idx = objectness.topk(pre, dim=1, sorted=True)
        batch_idx = torch.arange(n, device=objectness.device)[:, None]
        box_regression = box_regression[batch_idx, topk_idx]
        concat_anchors = torch.stack([b.bbox for b in anchors], 0)[batch_idx, topk_idx]
        proposals = self.box_coder.decode(box_regression.reshape(-1, 4), concat_anchors.reshape(-1, 4)).view(n, -1, 4)
        rank = torch.arange(pre, device=objectness.device)
        ours, theirs = [], []
        for i in range(n):
            boxlist = BoxList(proposals[i], anchors[i].size)
            boxlist.add_field("objectness", objectness[i])
            boxlist.add_field("rank", rank)
            boxlist = boxlist.clip_to_image(remove_empty=False)
            boxlist = remove_small_boxes(boxlist, self.min_size)
            kept = boxlist_nms(boxlist, self.nms_thresh, max_proposals=-1, score_field="objectness")
            sub = kept[torch.nonzero(kept.get_field("rank") < pre_other).squeeze(1)[:other.post_nms_top_n]]
            theirs.append(sub.copy_with_fields(["objectness"]))
            mine = kept[:self.post_nms_top_n].copy_with_fields(["objectness"])
            if add_gt and targets is not None:
                mine = self._with_gt(mine, targets[i])
            ours.append(mine)
        return ours, theirs


class RPNLossComputation:  # loss.py:21-131
    def __init__(self, matcher, sampler, box_coder):
        self.matcher, self.sampler, self.box_coder = matcher, sampler, box_coder
        self.device_targets = True  # False = the tensor-op sequence also on the device (cross-check in the tests)

    def __call__(self, anchors, objectness, box_regression, targets):
        if objectness.is_cuda and self.device_targets:
            return self._call_device(anchors, objectness, box_regression, targets)
        return self._call_tensor_ops(anchors, objectness, box_regression, targets)

    def _call_device(self, anchors, objectness, box_regression, targets):
        """The same two losses without the ~140 tensor-op launches and the six host syncs per step of the sequence below:
        per image two native calls -- ``_C.rpn_match_encode`` (IoU, Matcher with low-quality matches, the three label
        rules, delta targets; csrc/targets.hip) and the device fg / bg sampler -- then fixed-shape gathers over the padded
        [N, batch_size] selections; the number of sampled anchors stays a device scalar (no host read).  The sampler
        draws its uniformly random subsets from its own key stream (see ``BalancedPositiveNegativeSampler.sample_device``),
        not from ``torch.randperm``."""
        from .. import _C
        n, a, h, w = objectness.shape
        b = self.sampler.batch_size_per_image
        sels, slots, counts, labs, regs = [], [], [], [], []
        for anc, tgt in zip(anchors, targets):
            lab, reg = _C.rpn_match_encode(tgt.bbox, anc.bbox, anc.get_field("visibility"), self.matcher.high_threshold,
                                           self.matcher.low_threshold, self.matcher.allow_low_quality_matches,
                                           self.box_coder.weights)
            sel, slot, cnt = self.sampler.sample_device(lab)
            sels.append(sel)
            slots.append(slot)
            counts.append(cnt)
            labs.append(lab)
            regs.append(reg)
        sel, slot, cnt = torch.stack(sels), torch.stack(slots), torch.stack(counts)      # [N, B], [N, B], [N, 2]
        lab, regt = torch.stack(labs), torch.stack(regs)                                 # [N, A], [N, A, 4]
        ar = torch.arange(b, device=sel.device)[None]
        valid, pos_valid = ar < cnt[:, 0:1], ar < cnt[:, 1:2]
        total = valid.sum().to(torch.float32)                                            # sampled_inds.numel()
        obj = permute_and_flatten(objectness, n, a, 1, h, w).reshape(n, -1)
        reg = permute_and_flatten(box_regression, n, a, 4, h, w).reshape(n, -1, 4)
        bce = F.binary_cross_entropy_with_logits(obj.gather(1, sel), lab.gather(1, sel).to(torch.float32), reduction="none")
        objectness_loss = (bce * valid).sum() / total
        pos_idx = sel.gather(1, slot.clamp(min=0, max=b - 1))[..., None].expand(-1, -1, 4)  # anchors of the sampled positives
        d = (reg.gather(1, pos_idx) - regt.gather(1, pos_idx)).abs()
        beta = 1.0 / 9
        l1 = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)                    # layers/smooth_l1_loss.py:6-16
        box_loss = (l1 * pos_valid[..., None]).sum() / total
        return objectness_loss, box_loss

    def _call_tensor_ops(self, anchors, objectness, box_regression, targets):
        labels, reg_targets = [], []
        for anc, tgt in zip(anchors, targets):
            matched = self.matcher(box_iou(tgt.bbox, anc.bbox))
            lab = (matched >= 0).to(torch.float32)
            lab[matched == Matcher.BELOW_LOW_THRESHOLD] = 0
            lab[~anc.get_field("visibility")] = -1
            lab[matched == Matcher.BETWEEN_THRESHOLDS] = -1
            labels.append(lab)
            reg_targets.append(self.box_coder.encode(tgt.bbox[matched.clamp(min=0)], anc.bbox))
        pos, neg = self.sampler(labels)
        pos = torch.nonzero(torch.cat(pos, 0)).squeeze(1)
        neg = torch.nonzero(torch.cat(neg, 0)).squeeze(1)
        sampled = torch.cat([pos, neg], 0)
        n, a, h, w = objectness.shape
        obj = permute_and_flatten(objectness, n, a, 1, h, w).reshape(-1)
        reg = permute_and_flatten(box_regression, n, a, 4, h, w).reshape(-1, 4)
        labels, reg_targets = torch.cat(labels, 0), torch.cat(reg_targets, 0)
        box_loss = smooth_l1_loss(reg[pos], reg_targets[pos], beta=1.0 / 9, size_average=False) / sampled.numel()
        objectness_loss = F.binary_cross_entropy_with_logits(obj[sampled], labels[sampled])
        return objectness_loss, box_loss


class _BranchRunAhead(torch.autograd.Function):
    """Joins a branch whose backward was already run (``RPNModule.forward_ahead``) to the graph: identity on the feature and on
    the branch's (detached) loss values in the forward; the backward hands on the feature gradient PLUS the branch's
    pre-computed one, and the branch's parameter gradients -- each times the gradient the loss values receive, which a sum
    of loss terms hands to every term alike (``losses = sum(loss_dict.values())``, engine/trainer.py:96, also under
    GRADIENT_ACCUMULATION_STEPS' 1 / k).  Two loss terms that receive DIFFERENT gradients cannot be served from one
    pre-computed pass: the gradients turn NaN rather than silently wrong."""

    @staticmethod
    def forward(ctx, done, pre, feature, *loss_and_params):
        # done = (event behind the branch's loss forward, event behind its backward) on the branch's stream;
        # pre = [d/d feature, d/d parameter ...] for unit loss gradients
        ctx.done, ctx.pre = done[1], pre
        if done[0] is not None:  # (None, None): everything on one stream / host tensors -- the algebra alone, as the CPU test uses it
            torch.cuda.current_stream().wait_event(done[0])  # whoever reads the loss values from here on finds them written
        return (feature.view_as(feature),) + tuple(t.view_as(t) for t in loss_and_params[:2])

    @staticmethod
    def backward(ctx, g_feature, g_lo, g_lb):
        pre = ctx.pre
        if ctx.done is not None:
            main = torch.cuda.current_stream()
            main.wait_event(ctx.done)
            for t in pre:
                if t is not None:
                    t.record_stream(main)
        if all(t is None for t in pre):  # the branch reaches neither the feature nor a parameter
            return (None, None, g_feature if ctx.needs_input_grad[2] else None, None, None) + (None,) * (len(pre) - 1)
        ref = next(t for t in pre if t is not None)
        zero = ref.new_zeros(())
        g_lo = zero if g_lo is None else g_lo.reshape(())
        g_lb = zero if g_lb is None else g_lb.reshape(())
        s = torch.where(g_lo == g_lb, g_lo, torch.full_like(g_lo, float("nan")))
        out_feature = None
        if ctx.needs_input_grad[2]:
            if pre[0] is None:
                out_feature = g_feature
            elif g_feature is None:
                out_feature = pre[0] * s
            else:
                out_feature = torch.addcmul(g_feature, pre[0], s)
        live = [t for t in pre[1:] if t is not None]
        scaled = iter(torch._foreach_mul(live, s) if live else [])
        return (None, None, out_feature, None, None) + tuple(None if t is None else next(scaled) for t in pre[1:])


class RPNModule(nn.Module):  # rpn.py:109-197
    def __init__(self, cfg, in_channels):
        super().__init__()
        r = cfg.MODEL.RPN
        if r.USE_FPN or len(r.ANCHOR_STRIDE) != 1:
            raise NotImplementedError("single-level RPN only (no shipped config uses FPN)")
        self.anchor_generator = AnchorGenerator(r.ANCHOR_SIZES, r.ASPECT_RATIOS, r.ANCHOR_STRIDE[0], r.STRADDLE_THRESH)
        self.head = RPNHead(in_channels, self.anchor_generator.num_anchors_per_location())
        coder = BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        self.box_selector_train = RPNPostProcessor(r.PRE_NMS_TOP_N_TRAIN, r.POST_NMS_TOP_N_TRAIN, r.NMS_THRESH,
                                                   r.MIN_SIZE, coder)
        self.box_selector_test = RPNPostProcessor(r.PRE_NMS_TOP_N_TEST, r.POST_NMS_TOP_N_TEST, r.NMS_THRESH,
                                                  r.MIN_SIZE, coder)
        for sel in (self.box_selector_train, self.box_selector_test):
            sel.__dict__["anchor_generator"] = self.anchor_generator  # a reference, not a registered sub-module
        self.loss_evaluator = RPNLossComputation(
            Matcher(r.FG_IOU_THRESHOLD, r.BG_IOU_THRESHOLD, allow_low_quality_matches=True),
            BalancedPositiveNegativeSampler(r.BATCH_SIZE_PER_IMAGE, r.POSITIVE_FRACTION), coder)
        self.loss_beside_selection = True  # False = one stream (the A/B switch of the measurement, and of the test)
        self.backward_ahead = True         # ``forward_ahead``; False = the branch's backward inside the caller's

    def forward(self, images, features, targets=None, compute_loss=True, head_out=None):
        """``head_out``: (objectness, box_regression) of ``self.head`` on the same features, when the caller already
        has them (the student-teacher step selects test-mode AND train-mode proposals from one frozen head pass; the
        reference runs the head twice, st_generalized_rcnn.py:296-309, with identical results)."""
        feature = features[0]
        objectness, box_regression = self.head(feature) if head_out is None else head_out
        anchors = self.anchor_generator(images.image_sizes, feature)
        if self.training:
            # Loss and selection both start from the head's outputs and neither reads the other's result, and both are
            # chains of small launches around single-workgroup kernels (fg / bg sampler 2 x 113 us, NMS reduce 329 us):
            # on a device the loss goes to a second stream and runs beside the selection (and its backward beside the box
            # head's: autograd runs a node on the stream of its forward); the selection's launches are issued first, its one
            # host read last.
            sel = self.box_selector_train
            beside = compute_loss and objectness.is_cuda and self.loss_beside_selection and sel._on_device(objectness)
            if beside:
                from ..engine.trainer import branch_stream
                main, side = torch.cuda.current_stream(), branch_stream()
                side.wait_stream(main)
                with torch.no_grad():
                    launched = sel.launch(anchors, objectness, box_regression)  # its launches first: they are the critical path
                with torch.cuda.stream(side):
                    lo, lb = self.loss_evaluator(anchors, objectness, box_regression, targets)
                with torch.no_grad():
                    boxes = sel.finish(launched, targets, add_gt=True)
                main.wait_stream(side)
                lo.record_stream(main)
                lb.record_stream(main)
                return boxes, {"loss_objectness": lo, "loss_rpn_box_reg": lb}
            with torch.no_grad():
                boxes = sel(anchors, objectness, box_regression, targets, add_gt=True)
            if not compute_loss:
                return boxes, {}
            lo, lb = self.loss_evaluator(anchors, objectness, box_regression, targets)
            return boxes, {"loss_objectness": lo, "loss_rpn_box_reg": lb}
        return self.box_selector_test(anchors, objectness, box_regression), {}

    def forward_ahead(self, images, features, targets):
        """Training step of a TRAINABLE RPN on a device, with the branch's backward run ahead: head, loss and the whole
        backward of the two RPN losses (0.96 ms of 3x3 data / weight gradient GEMMs) are issued on a second stream as soon
        as the trunk's feature exists, and run beside the proposal selection, its host read and the box head's sampling --
        a stretch of single-workgroup kernels (NMS reduce 0.33 ms) and host round trips that leaves the machine empty.
        Returns (proposals, losses, features): the caller continues from the RETURNED features -- the same values, joined
        to the graph through ``_BranchRunAhead`` so that the trunk receives box-head + RPN gradient and the head's parameters
        theirs when (and only if) the caller's backward arrives.  Same kernels on the same operands as ``forward``:
        losses and gradients are bit-identical (tests/test_model_gpu.py)."""
        from ..engine.trainer import branch_stream
        feature = features[0]
        main, side = torch.cuda.current_stream(), branch_stream()
        anchors = self.anchor_generator(images.image_sizes, feature)
        params = [p for p in self.head.parameters() if p.requires_grad]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            branch_in = feature.detach().requires_grad_(feature.requires_grad)
            objectness, box_regression = self.head(branch_in)
            head_done = torch.cuda.Event()
            head_done.record(side)
        main.wait_event(head_done)
        objectness.record_stream(main)
        box_regression.record_stream(main)
        sel = self.box_selector_train
        on_device = sel._on_device(objectness)
        if on_device:  # the selection's launches first (they are the critical path), its host read after the branch's launches
            with torch.no_grad():
                launched = sel.launch(anchors, objectness.detach(), box_regression.detach())
        with torch.cuda.stream(side):
            lo, lb = self.loss_evaluator(anchors, objectness, box_regression, targets)
            loss_done = torch.cuda.Event()
            loss_done.record(side)
            wrt = ([branch_in] if branch_in.requires_grad else []) + params
            grads = list(torch.autograd.grad([lo + lb], wrt, allow_unused=True))
            pre = grads if branch_in.requires_grad else [None] + grads
            done = torch.cuda.Event()
            done.record(side)
        lo.record_stream(main)
        lb.record_stream(main)
        with torch.no_grad():
            boxes = (sel.finish(launched, targets, add_gt=True) if on_device else
                     sel(anchors, objectness.detach(), box_regression.detach(), targets, add_gt=True))
        joined, lo, lb = _BranchRunAhead.apply((loss_done, done), pre, feature, lo.detach(), lb.detach(), *params)
        return boxes, {"loss_objectness": lo, "loss_rpn_box_reg": lb}, [joined] + list(features[1:])

    def runs_ahead(self, features):
        """Whether ``forward_ahead`` serves this call: training with a loss to compute, on a device, something to train."""
        return (self.training and self.backward_ahead and features[0].is_cuda and torch.is_grad_enabled()
                and any(p.requires_grad for p in self.head.parameters()))

    @torch.no_grad()
    def proposals_train_and_test(self, images, features, targets, head_out=None):
        """(train-mode proposals with the ground truth appended, test-mode proposals) from one head pass and one
        NMS per image (``RPNPostProcessor.forward_with``); falls back to two selections when the selectors differ in
        more than their top-n counts."""
        feature = features[0]
        objectness, box_regression = self.head(feature) if head_out is None else head_out
        anchors = self.anchor_generator(images.image_sizes, feature)
        tr, te = self.box_selector_train, self.box_selector_test
        if te.pre_nms_top_n <= tr.pre_nms_top_n and te.nms_thresh == tr.nms_thresh and te.min_size == tr.min_size:
            return tr.forward_with(te, anchors, objectness, box_regression, targets, add_gt=True)
        return (tr(anchors, objectness, box_regression, targets, add_gt=True), te(anchors, objectness, box_regression))
