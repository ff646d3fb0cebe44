"""RoI heads: pooler, embedding-based box head, class-agnostic C4 mask head with the uncertainty branch.

Counterparts (maskrcnn_benchmark/modeling/...):
  poolers.py:45-121                              -> ``Pooler``
  roi_heads/box_head/roi_box_feature_extractors.py:13-46 -> ``ResNet50Conv5ROIFeatureExtractor``
  roi_heads/box_head/roi_box_predictors.py:8-92  -> ``FastRCNNPredictor``   (region x text head)
  roi_heads/box_head/loss.py:15-185              -> ``FastRCNNLossComputation``
  roi_heads/box_head/inference.py:12-163         -> ``PostProcessor``
  roi_heads/box_head/box_head.py:11-79           -> ``ROIBoxHead``
  roi_heads/mask_head/roi_mask_predictors.py:11-65 -> ``MaskRCNNC4Predictor``
  roi_heads/mask_head/loss.py:11-148             -> ``project_masks_on_boxes`` / ``MaskRCNNLossComputation``
  roi_heads/mask_head/inference.py:12-66,124-205 -> ``MaskPostProcessor`` / ``Masker``
  roi_heads/mask_head/mask_head.py:13-106        -> ``ROIMaskHead``
  roi_heads/roi_heads.py:11-102                  -> ``CombinedROIHeads``

Deviations, all stated in DESIGN.md: everything stays on the device (no ``.cpu()`` in the mask-target
projection, no per-mask python loop there), the fg/bg class weights are built on the logits'
device instead of a hard-coded ``.cuda()``, and the mask head handles any number of
images per process.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import _C
from ..layers import (Conv2d, ConvTranspose2d, ROIAlign, linear_mfma, smooth_l1_loss, smooth_l1_picked, stochastic_mask_bce,
                      text_logits, weighted_cross_entropy)
from .backbone import ResNetHead
from .box_coder import BoxCoder
from .matcher import BalancedPositiveNegativeSampler, Matcher
from .structures import BoxList, PastedMasks, PolygonMasks, box_iou, boxlist_nms, cat_boxlist


def _cat(tensors, dim=0):
    return tensors[0] if len(tensors) == 1 else torch.cat(tensors, dim)


class Pooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio):
        super().__init__()
        if len(scales) != 1:
            raise NotImplementedError("single-level pooler only (C4)")
        self.pooler = ROIAlign(output_size, spatial_scale=scales[0], sampling_ratio=sampling_ratio)
        self.output_size = output_size

    @staticmethod
    def convert_to_roi_format(boxes):
        if boxes and all(b.bbox.is_cuda and b.bbox.dtype == torch.float32 for b in boxes):
            return _C.rois_from_boxes([b.bbox for b in boxes])  # one launch
        parts = []
        for i, b in enumerate(boxes):
            ids = torch.full((len(b), 1), float(i), dtype=b.bbox.dtype, device=b.bbox.device)
            parts.append(torch.cat([ids, b.bbox], dim=1))
        return _cat(parts, 0)

    def forward(self, x, boxes):
        return self.pooler(x[0], self.convert_to_roi_format(boxes))

    def forward_strided_nhwc(self, x, boxes, bin_stride):
        return self.pooler.forward_strided_nhwc(x[0], self.convert_to_roi_format(boxes), bin_stride)


class ResNet50Conv5ROIFeatureExtractor(nn.Module):
    def __init__(self, cfg, head_cfg):
        super().__init__()
        res = head_cfg.POOLER_RESOLUTION
        self.pooler = Pooler((res, res), head_cfg.POOLER_SCALES, head_cfg.POOLER_SAMPLING_RATIO)
        self.head = ResNetHead(cfg)
        self.out_channels = self.head.out_channels

    def forward(self, x, proposals, pooled_only=False):
        return self.forward_rois(x, self.pooler.convert_to_roi_format(proposals), pooled_only=pooled_only)

    def forward_rois(self, x, rois, select=None, pooled_only=False):
        """``rois`` [R, 5] = (image index into x[0], x1, y1, x2, y2): the same pass on an explicit RoI tensor (lets a
        caller pool RoIs of several proposal lists / image subsets in one go)."""
        p = self.pooler.pooler
        s = self.head.pooler_stride() if x[0].is_cuda else 0
        if s:  # the head's first 1x1 has stride s: pool only the bins it reads, straight into NHWC
            need_grad = torch.is_grad_enabled() and x[0].requires_grad
            if not need_grad and x[0].shape[1] % 32 == 0 and self.head.pooled_pair_ok():
                # frozen features: the bins go to the first GEMM in pair layout, no fp32 copy / split pass in between
                ph, pw = p.output_size
                yp, (oh, ow) = _C.roi_align_forward_strided_pair(x[0], rois, p.spatial_scale, ph, pw, p.sampling_ratio, s)
                return self.head.forward_pooled_nhwc(None, yp, (rois.shape[0], oh, ow), select=select, pooled_only=pooled_only)
            return self.head.forward_pooled_nhwc(p.forward_strided_nhwc(x[0], rois, s), select=select, pooled_only=pooled_only)
        return self.head(p(x[0], rois))


# ------------------------------------------------------------------------------------------------
# box head
# ------------------------------------------------------------------------------------------------
class FastRCNNPredictor(nn.Module):
    """avgpool -> emb_pred Linear -> dot product with the class-embedding matrix; agnostic box deltas (EMBEDDING_BASED, every
    shipped config).  EMBEDDING_BASED False = the plain Fast R-CNN predictor (roi_box_predictors.py:33-40): a learned
    ``cls_score`` Linear over NUM_CLASSES and per-class box deltas unless CLS_AGNOSTIC_BBOX_REG."""

    def __init__(self, cfg, in_channels, is_teacher=False):
        super().__init__()
        bh = cfg.MODEL.ROI_BOX_HEAD
        self.embedding_based = bh.EMBEDDING_BASED
        if self.embedding_based:
            assert cfg.MODEL.CLS_AGNOSTIC_BBOX_REG
            self.emb_dim = bh.EMB_DIM
            self.emb_pred = nn.Linear(in_channels, self.emb_dim)
            nn.init.normal_(self.emb_pred.weight, mean=0, std=0.01)
            nn.init.constant_(self.emb_pred.bias, 0)
            num_bbox_reg_classes = 2
            self.num_classes = None
            self.cls_score = None  # [C, emb_dim], set by set_class_embeddings after the optimizer is made
            if bh.FREEZE_EMB_PRED:
                self.emb_pred.weight.requires_grad = False
                self.emb_pred.bias.requires_grad = False
        else:
            self.num_classes = bh.NUM_CLASSES
            num_bbox_reg_classes = 2 if cfg.MODEL.CLS_AGNOSTIC_BBOX_REG else self.num_classes
            self.cls_score = nn.Linear(in_channels, self.num_classes)
            nn.init.normal_(self.cls_score.weight, mean=0, std=0.01)
            nn.init.constant_(self.cls_score.bias, 0)
        self.bbox_pred = nn.Linear(in_channels, num_bbox_reg_classes * 4)
        nn.init.normal_(self.bbox_pred.weight, mean=0, std=0.001)
        nn.init.constant_(self.bbox_pred.bias, 0)

    def pooled(self, x):
        if x.dim() != 4:
            return x
        p = getattr(x, "_ovis_pooled", None)
        if p is not None and p.shape == (x.shape[0], x.shape[1]):
            return p  # the res5 head's last autograd node already produced the pooled map (backbone.py::_forward_pair)
        xl = x.permute(0, 2, 3, 1)
        if xl.is_contiguous():
            # the res5 head hands over an NCHW view of NHWC memory: reduce over the contiguous [R, H*W, C] form, so
            # that the gradient (an expand over H*W) is born contiguous in the layout the head's backward reads --
            # mean(dim=(2, 3)) on the view made autograd clone 411 MB through a strided copy (0.47 ms per pass)
            return xl.reshape(x.shape[0], -1, x.shape[1]).mean(dim=1)
        return x.mean(dim=(2, 3))

    def forward(self, x):
        x = self.pooled(x)
        # both Linear layers share the pooled operand: one GEMM over the concatenated [768 + 8, 2048] weight, built
        # already padded with zero rows to the split GEMM's 128-column tiles (one cat, no separate pad)
        first = self.emb_pred if self.embedding_based else self.cls_score
        n1 = first.out_features
        n = n1 + self.bbox_pred.out_features
        pad = (-n) % 128 if x.is_cuda else 0
        zw, zb = self._zero_rows(pad, x)
        w = torch.cat([first.weight, self.bbox_pred.weight] + ([zw] if pad else []), 0)
        b = torch.cat([first.bias, self.bbox_pred.bias] + ([zb] if pad else []), 0)
        y = linear_mfma(x, w, b)
        head, bbox = y[:, :n1], y[:, n1:n]
        if not self.embedding_based:
            return head, bbox  # cls_score(x), bbox_pred(x): roi_box_predictors.py:70-72
        cls_logit = text_logits(head, self.cls_score)  # einsum('pe,ce->pc')
        return cls_logit, bbox

    def _zero_rows(self, pad, x):
        z = getattr(self, "_zero_pad", None)
        if pad and (z is None or z[0].shape[0] != pad or z[0].device != x.device):
            z = (x.new_zeros((pad, self.bbox_pred.in_features)), x.new_zeros((pad,)))
            self._zero_pad = z
        return z if pad else (None, None)

    def embed(self, x):
        """region embeddings only (teacher alignment pass)"""
        if not self.embedding_based:
            raise RuntimeError("FastRCNNPredictor.embed: MODEL.ROI_BOX_HEAD.EMBEDDING_BASED is False -- there is no embedding "
                               "projection to align with text")
        return linear_mfma(self.pooled(x), self.emb_pred.weight, self.emb_pred.bias)

    def set_class_embeddings(self, embs):
        if not self.embedding_based:  # (the reference would overwrite its Linear with the matrix and fail in forward)
            raise RuntimeError("set_class_embeddings: MODEL.ROI_BOX_HEAD.EMBEDDING_BASED is False -- the classifier is a "
                               "learned Linear over NUM_CLASSES")
        self.num_classes = embs.shape[0]
        self.cls_score = embs.to(self.emb_pred.weight.device)


class FastRCNNLossComputation:
    def __init__(self, cfg):
        rh = cfg.MODEL.ROI_HEADS
        self.matcher = Matcher(rh.FG_IOU_THRESHOLD, rh.BG_IOU_THRESHOLD, allow_low_quality_matches=False)
        self.sampler = BalancedPositiveNegativeSampler(rh.BATCH_SIZE_PER_IMAGE, rh.POSITIVE_FRACTION)
        self.box_coder = BoxCoder(weights=rh.BBOX_REG_WEIGHTS)
        self.cls_agnostic_bbox_reg = cfg.MODEL.CLS_AGNOSTIC_BBOX_REG
        self.bg_weight = cfg.MODEL.ROI_BOX_HEAD.LOSS_WEIGHT_BACKGROUND
        self.generator = None  # optional torch.Generator for reproducible sampling in tests
        self.device_sampler = True  # False = the tensor-op sampler also on the device (cross-check in the tests)

    def subsample(self, proposals, targets):
        return self.subsample_many([(proposals, targets)])[0]

    def subsample_many(self, groups):
        """``subsample`` for several (proposals, targets) groups -- the student's pseudo-label and ground-truth branches --
        at once.  On the device every image costs two launches (IoU match + labels + delta targets, csrc/targets.hip
        ``match_encode``; fg / bg sampling, ``sample_fg_bg``) and the host reads the survivor counts of ALL images back in
        ONE copy; the sampled lists carry ``pos_index`` (where the positives sit), so neither the box loss nor the mask
        head runs a ``nonzero``.  The reference's sequence (box_head/loss.py:89-123: IoU matrix, Matcher, randperm
        sampler, boolean indexing: ~60 launches and 3 host syncs per image) serves CPU tensors."""
        if not all(p.bbox.is_cuda for props, _ in groups for p in props) or self.matcher.allow_low_quality_matches \
                or not self.device_sampler:
            return [self._subsample_tensor_ops(props, tgts) for props, tgts in groups]
        pending = []
        for props, tgts in groups:
            for prop, tgt in zip(props, tgts):
                idx, labels, reg = _C.match_encode(tgt.bbox, tgt.get_field("labels"), prop.bbox, self.matcher.high_threshold,
                                                   self.matcher.low_threshold, self.box_coder.weights)
                sel, slots, counts = self.sampler.sample_device(labels, self.generator)
                pending.append((prop, idx, labels, reg, sel, slots, counts))
        cnt = torch.stack([p[-1] for p in pending]).tolist() if pending else []  # the one host read
        out, k = [], 0
        for props, _ in groups:
            group = []
            for _p in props:
                prop, idx, labels, reg, sel, slots, _c = pending[k]
                n, npos = cnt[k]
                k += 1
                bbox_s, reg_s, labels_s, idx_s = _C.gather_rows(sel[:n], prop.bbox, reg, labels, idx)  # one launch
                b = BoxList(bbox_s, prop.size)
                b.add_field("labels", labels_s)
                b.add_field("regression_targets", reg_s)
                b.add_field("matched_gt", idx_s)
                b.pos_index = slots[:npos]
                group.append(b)
            out.append(group)
        self._proposals = out[-1]
        return out

    def _subsample_tensor_ops(self, proposals, targets):
        out = []
        labels_all = []
        for prop, tgt in zip(proposals, targets):
            if prop.bbox.is_cuda and not self.matcher.allow_low_quality_matches:
                # IoU -> match -> labels -> delta targets in one launch (csrc/targets.hip)
                idx, labels, reg = _C.match_encode(tgt.bbox, tgt.get_field("labels"), prop.bbox,
                                                   self.matcher.high_threshold, self.matcher.low_threshold,
                                                   self.box_coder.weights)
            else:
                matched = self.matcher(box_iou(tgt.bbox, prop.bbox))
                idx = matched.clamp(min=0)
                labels = tgt.get_field("labels")[idx].to(torch.int64)
                labels[matched == Matcher.BELOW_LOW_THRESHOLD] = 0
                labels[matched == Matcher.BETWEEN_THRESHOLDS] = -1
                reg = self.box_coder.encode(tgt.bbox[idx], prop.bbox)
            prop.add_field("labels", labels)
            prop.add_field("regression_targets", reg)
            prop.add_field("matched_gt", idx)
            labels_all.append(labels)
        pos, neg = self.sampler(labels_all, generator=self.generator)
        for prop, p, n in zip(proposals, pos, neg):
            out.append(prop[torch.nonzero(p | n).squeeze(1)])
        self._proposals = out
        return out

    def __call__(self, class_logits, box_regression):
        proposals = self._proposals
        labels = _cat([p.get_field("labels") for p in proposals], 0)
        reg_targets = _cat([p.get_field("regression_targets") for p in proposals], 0)
        pos = positives_index(proposals)
        if pos is None:
            pos = torch.nonzero(labels > 0).squeeze(1)
        if box_regression.is_cuda and labels.numel():
            # gather + smooth L1 + its gradient in one pass (csrc/boxes.hip)
            box_loss = smooth_l1_picked(box_regression, reg_targets, pos, None if self.cls_agnostic_bbox_reg else labels, 4,
                                        1.0, labels.numel())
        else:
            if self.cls_agnostic_bbox_reg:
                picked = box_regression.index_select(0, pos)[:, 4:8]
            else:
                map_inds = 4 * labels[pos][:, None] + torch.tensor([0, 1, 2, 3], device=class_logits.device)
                picked = box_regression[pos[:, None], map_inds]
            box_loss = smooth_l1_loss(picked, reg_targets.index_select(0, pos), size_average=False, beta=1) / labels.numel()
        cls_loss = weighted_cross_entropy(class_logits, labels, self.bg_weight)
        return cls_loss, box_loss


def positives_index(proposals):
    """Row indices of the positives in the concatenation of sampled proposal lists, from the ``pos_index`` the device
    sampler attached to every list (None when a list has none: the caller falls back to ``nonzero``)."""
    parts, off = [], 0
    for p in proposals:
        pi = getattr(p, "pos_index", None)
        if pi is None:
            return None
        parts.append(pi + off if off else pi)
        off += len(p)
    return _cat(parts, 0) if parts else None


def positive_proposals(p):
    """The positives of a sampled list as a light BoxList (boxes, labels, matched ground-truth index), flagged
    ``all_positive`` so that the mask loss neither re-matches them nor searches them again."""
    pi = p.pos_index
    if p.bbox.is_cuda:
        bbox, _, labels, matched = _C.gather_rows(pi, p.bbox, None, p.get_field("labels"), p.get_field("matched_gt"))
    else:
        bbox, labels, matched = (t.index_select(0, pi) for t in (p.bbox, p.get_field("labels"), p.get_field("matched_gt")))
    out = BoxList(bbox, p.size)
    out.add_field("labels", labels)
    out.add_field("matched_gt", matched)
    out.all_positive = True
    return out


class PostProcessor(nn.Module):
    def __init__(self, cfg, is_teacher=False):
        super().__init__()
        rh = cfg.MODEL.ROI_HEADS
        self.score_thresh, self.nms, self.detections_per_img = rh.SCORE_THRESH, rh.NMS, rh.DETECTIONS_PER_IMG
        self.box_coder = BoxCoder(weights=rh.BBOX_REG_WEIGHTS)
        self.cls_agnostic_bbox_reg = cfg.MODEL.CLS_AGNOSTIC_BBOX_REG
        self.is_teacher = is_teacher
        # MODEL.GT_BOX_EVAL (inference.py:177-181): ground-truth boxes are the proposals of the evaluation pass, every one
        # is kept (no score cut below 1, no suppression) and scored on its own class only
        self.gt_box_eval = cfg.MODEL.GT_BOX_EVAL
        if self.gt_box_eval:
            self.score_thresh = self.nms = 1.0

    def forward(self, x, boxes):
        class_logits, box_regression = x
        class_prob = F.softmax(class_logits, -1)
        if self.gt_box_eval and not self.is_teacher:  # inference.py:82-89: prob of the box's own class + 1.1, zero elsewhere
            own = torch.zeros_like(class_prob)
            rows = [i for i, b in enumerate(boxes) if b.has_field("labels")]
            if rows:
                offsets = [0]
                for b in boxes:
                    offsets.append(offsets[-1] + len(b))
                r = _cat([torch.arange(offsets[i], offsets[i + 1], device=class_prob.device) for i in rows], 0)
                c = _cat([boxes[i].get_field("labels").long() for i in rows], 0)
                own[r, c] = class_prob[r, c] + 1.1
            class_prob = own
        per_img = [len(b) for b in boxes]
        concat = _cat([b.bbox for b in boxes], 0)
        if self.cls_agnostic_bbox_reg:
            box_regression = box_regression[:, -4:]
        # decode + clip_to_image in one launch on the device (BoxCoder.decode with the image sizes)
        proposals = self.box_coder.decode(box_regression.reshape(sum(per_img), -1), concat, per_img, [b.size for b in boxes])
        num_classes = class_prob.shape[1]
        if self.cls_agnostic_bbox_reg:
            proposals = proposals.repeat(1, num_classes)
        results = []
        for prob, prop, b in zip(class_prob.split(per_img, 0), proposals.split(per_img, 0), boxes):
            boxlist = BoxList(prop.reshape(-1, 4), b.size)  # already clipped (remove_empty=False: nothing else to do)
            boxlist.add_field("scores", prob.reshape(-1))
            if not self.is_teacher:
                boxlist = self.filter_results(boxlist, num_classes)
            results.append(boxlist)
        return results

    def filter_results(self, boxlist, num_classes):  # inference.py:121-163
        """Score threshold -> per-class NMS -> keep the DETECTIONS_PER_IMG best.  On the device the per-class loop of the
        reference (one ``nonzero`` + one NMS + host syncs per class) is ONE grouped NMS over the candidates of every
        class (``_C.nms_grouped``: a box only suppresses boxes of its class); same detections in the same order
        (class-major, proposal index ascending inside a class)."""
        if not boxlist.bbox.is_cuda:
            return self.filter_results_per_class(boxlist, num_classes)
        boxes = boxlist.bbox.reshape(-1, num_classes, 4)
        scores = boxlist.get_field("scores").reshape(-1, num_classes)
        cls, row = (scores.t()[1:] > self.score_thresh).nonzero(as_tuple=True)  # class-major candidate order
        cls = cls + 1
        cand_boxes, cand_scores = boxes[row, cls], scores[row, cls]
        keep = _C.nms_grouped(cand_boxes, cand_scores, cls, self.nms) if cls.numel() else cls
        result = BoxList(cand_boxes[keep], boxlist.size)
        result.add_field("scores", cand_scores[keep])
        result.add_field("labels", cls[keep])
        n = len(result)
        if n > self.detections_per_img > 0:
            s = result.get_field("scores")
            thresh = torch.kthvalue(s, n - self.detections_per_img + 1).values
            result = result[torch.nonzero(s >= thresh).squeeze(1)]
        return result

    def filter_results_per_class(self, boxlist, num_classes):
        """The reference's loop as written (inference.py:137-150); the oracle-backed CPU runs of the tests use it."""
        boxes = boxlist.bbox.reshape(-1, num_classes * 4)
        scores = boxlist.get_field("scores").reshape(-1, num_classes)
        inds_all = scores > self.score_thresh
        result = []
        for j in range(1, num_classes):
            inds = inds_all[:, j].nonzero().squeeze(1)
            if inds.numel() == 0:
                continue
            cls_boxes = BoxList(boxes[inds, j * 4:(j + 1) * 4], boxlist.size)
            cls_boxes.add_field("scores", scores[inds, j])
            cls_boxes = boxlist_nms(cls_boxes, self.nms)
            cls_boxes.add_field("labels", torch.full((len(cls_boxes),), j, dtype=torch.int64, device=scores.device))
            result.append(cls_boxes)
        if not result:
            empty = BoxList(boxes.new_zeros((0, 4)), boxlist.size)
            empty.add_field("scores", scores.new_zeros((0,)))
            empty.add_field("labels", torch.zeros((0,), dtype=torch.int64, device=scores.device))
            return empty
        result = cat_boxlist(result)
        n = len(result)
        if n > self.detections_per_img > 0:
            s = result.get_field("scores")
            thresh = torch.kthvalue(s, n - self.detections_per_img + 1).values
            result = result[torch.nonzero(s >= thresh).squeeze(1)]
        return result


class ROIBoxHead(nn.Module):
    def __init__(self, cfg, in_channels, is_teacher=False):
        super().__init__()
        self.feature_extractor = ResNet50Conv5ROIFeatureExtractor(cfg, cfg.MODEL.ROI_BOX_HEAD)
        self.predictor = FastRCNNPredictor(cfg, self.feature_extractor.out_channels, is_teacher)
        self.post_processor = PostProcessor(cfg, is_teacher)
        self.loss_evaluator = FastRCNNLossComputation(cfg)
        if cfg.MODEL.ROI_BOX_HEAD.FREEZE_FEATURE_EXTRACTOR:
            for p in self.feature_extractor.parameters():
                p.requires_grad = False
        self.is_teacher = is_teacher

    def forward(self, features, proposals, targets=None):
        if self.training:
            with torch.no_grad():
                proposals = self.loss_evaluator.subsample(proposals, targets)
        # evaluation / no-grad passes read nothing but the pooled rows of the box features (the predictor here, the teacher's
        # ``predictor.embed`` in generate_pseudo_label; the mask head pools its own features then): the res5 head's last
        # block may leave its [R*49, 2048] result unwritten
        pooled_only = not self.training and not torch.is_grad_enabled()
        x = self.feature_extractor(features, proposals, pooled_only=pooled_only)
        if pooled_only and getattr(x, "_ovis_pooled", None) is not None:
            # the pass may have produced NOTHING but the pooled rows (the [R, 2048, 7, 7] view is then a NaN placeholder):
            # hand the [R, 2048] rows themselves on as the box features -- ``predictor.pooled`` / ``embed`` take 2-D input --
            # so that no consumer of ``package_x['bbox']`` can read the placeholder
            x = x._ovis_pooled
        class_logits, box_regression = self.predictor(x)
        if not self.training:
            return x, self.post_processor((class_logits, box_regression), proposals), {}
        loss_classifier, loss_box_reg = self.loss_evaluator(class_logits, box_regression)
        return x, proposals, dict(loss_classifier=loss_classifier, loss_box_reg=loss_box_reg)


# ------------------------------------------------------------------------------------------------
# mask head
# ------------------------------------------------------------------------------------------------
class MaskRCNNC4Predictor(nn.Module):
    split_gemm = True  # False = the module's plain convolutions (cross-check in tests/test_split_gemm_pair.py)

    def __init__(self, cfg, in_channels):
        super().__init__()
        num_classes = 2 if cfg.MODEL.CLS_AGNOSTIC_MASK else cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES
        dim_reduced = cfg.MODEL.ROI_MASK_HEAD.CONV_LAYERS[-1]
        self.conv5_mask = ConvTranspose2d(in_channels, dim_reduced, 2, 2, 0)
        self.mask_fcn_logits = Conv2d(dim_reduced, num_classes, 1, 1, 0)
        self.uncertainty = cfg.MODEL.UNCERTAINTY
        if self.uncertainty:
            self.uncertain_pred = Conv2d(dim_reduced, 1, 1, 1, 0)
        for name, param in self.named_parameters():
            if "bias" in name:
                nn.init.constant_(param, 0)
            elif "weight" in name:
                nn.init.kaiming_normal_(param, mode="fan_out", nonlinearity="relu")
        if self.uncertainty:
            nn.init.normal_(self.uncertain_pred.weight, mean=0, std=0.001)
            nn.init.constant_(self.uncertain_pred.bias, 1)

    def _gemm_ok(self, x):
        c = self.conv5_mask
        return (x.is_cuda and x.dim() == 4 and c.kernel_size == (2, 2) and c.stride == (2, 2) and c.padding == (0, 0)
                and c.output_padding == (0, 0) and c.groups == 1 and c.in_channels % 128 == 0
                and (4 * c.out_channels) % 128 == 0 and self.split_gemm)

    def _upsampled_rows(self, x):
        """relu(conv5_mask(x)) as NHWC rows [P*2H*2W, dim_reduced]: the 2x2 / stride-2 transposed convolution is ONE
        GEMM over the input pixels with the 4 sub-pixel kernels stacked along N (split-GEMM autograd node, bias + ReLU
        in its epilogue), followed by the pixel shuffle.  Keeps MIOpen -- whose kernels are compiled per input shape,
        and the number of positives changes every step -- out of the training step."""
        from ..layers.pair_bottleneck import conv_same_pair
        c = self.conv5_mask
        p, ci, h, w = x.shape
        co = c.out_channels
        rows = x.permute(0, 2, 3, 1).reshape(-1, ci)
        wm = c.weight.permute(2, 3, 1, 0).reshape(4 * co, ci, 1, 1)          # n = (dy, dx, co)
        y = conv_same_pair(rows, (h, w), wm, c.bias.repeat(4) if c.bias is not None else None, True)
        return y.view(p, h, w, 2, 2, co).permute(0, 1, 3, 2, 4, 5).reshape(p * 2 * h * 2 * w, co), (p, 2 * h, 2 * w)

    def _rows_conv1x1(self, rows, shape, conv):
        p, h, w = shape
        y = linear_mfma(rows, conv.weight.view(conv.out_channels, -1), conv.bias)
        return y.view(p, h, w, conv.out_channels).permute(0, 3, 1, 2)

    def forward_parts(self, x):
        """-> (mask logits mu [P,C,M,M], predicted std-dev sigma [P,1,M,M]) for the fused stochastic BCE."""
        if self._gemm_ok(x):
            rows, shape = self._upsampled_rows(x)
            mu = self._rows_conv1x1(rows, shape, self.mask_fcn_logits)
            sigma = torch.exp(0.5 * self._rows_conv1x1(rows.detach(), shape, self.uncertain_pred)) if self.uncertainty else None
            return mu, sigma
        x_ = F.relu(self.conv5_mask(x))
        mu = self.mask_fcn_logits(x_)
        sigma = torch.exp(0.5 * self.uncertain_pred(x_.detach())) if self.uncertainty else None
        return mu, sigma

    def forward(self, x, compute_uncertain=False, eps=None):
        """``eps`` (standard-normal noise, [1,P,C,M,M] -- the reference draws it with the shape of
        ``mask_logits*0+scale``, i.e. independently per logit channel, roi_mask_predictors.py:47-53,62)
        can be injected for reproducible tests; by default it is drawn on the device (the reference
        draws on the host and copies)."""
        if self._gemm_ok(x):
            rows, shape = self._upsampled_rows(x)
            mask_logits = self._rows_conv1x1(rows, shape, self.mask_fcn_logits)
            unc = (lambda: self._rows_conv1x1(rows.detach(), shape, self.uncertain_pred))
        else:
            x_ = F.relu(self.conv5_mask(x))
            mask_logits = self.mask_fcn_logits(x_)
            unc = (lambda: self.uncertain_pred(x_.detach()))
        if self.uncertainty and compute_uncertain:
            scale = torch.exp(0.5 * unc())  # [P,1,M,M] std-dev
            if self.training:
                std = mask_logits * 0.0 + scale  # [P,C,M,M]
                if eps is None:
                    eps = torch.randn((1, *std.shape), device=std.device, dtype=std.dtype)
                mask_logits = mask_logits[None] + eps * std[None]  # [1,P,C,M,M]
            return mask_logits, scale
        return mask_logits


def project_masks_on_boxes(masks, gt_index, boxes, M):
    """Device-side counterpart of mask_head/loss.py:11-42 for binary ('mask' mode) targets.

    masks [G,H,W] (bool or uint8), gt_index [P] (which mask each box uses), boxes [P,4] xyxy.
    Crop (rounded, clamped like BinaryMaskList.crop, segmentation_mask.py:117-136) and bilinear-resize
    (F.interpolate align_corners=False semantics, :138-156) to MxM, then cast back to the mask dtype
    (bool: any positive weight on a set pixel -> 1; uint8: truncation) and to float32."""
    P = boxes.shape[0]
    if P == 0:
        return torch.empty(0, dtype=torch.float32, device=boxes.device)
    H, W = masks.shape[-2:]
    b = torch.round(boxes)  # python round() in the reference = half-to-even = torch.round
    xmin = b[:, 0].clamp(0, W - 1)
    ymin = b[:, 1].clamp(0, H - 1)
    xmax = torch.maximum(b[:, 2].clamp(0, W), xmin + 1)
    ymax = torch.maximum(b[:, 3].clamp(0, H), ymin + 1)
    w, h = xmax - xmin, ymax - ymin  # crop is [ymin:ymax, xmin:xmax]
    dst = torch.arange(M, device=boxes.device, dtype=torch.float32)

    m_t = torch.full((1, 1), float(M), device=boxes.device)

    def axis(size, lo):
        # in / out as a tensor-by-tensor division: correctly rounded on every device (tensor / python-scalar is a
        # multiplication by the reciprocal on the GPU, an ulp off where the crop size is a multiple of M)
        src = ((dst[None, :] + 0.5) * (size[:, None] / m_t) - 0.5).clamp(min=0)  # [P,M]
        i0 = src.floor()
        lam = src - i0
        i0 = torch.minimum(i0, size[:, None] - 1)
        i1 = torch.minimum(i0 + 1, size[:, None] - 1)
        return (i0 + lo[:, None]).long(), (i1 + lo[:, None]).long(), lam

    y0, y1, ly = axis(h, ymin)
    x0, x1, lx = axis(w, xmin)
    g = gt_index[:, None, None]
    mf = masks
    v00 = mf[g, y0[:, :, None], x0[:, None, :]].float()
    v01 = mf[g, y0[:, :, None], x1[:, None, :]].float()
    v10 = mf[g, y1[:, :, None], x0[:, None, :]].float()
    v11 = mf[g, y1[:, :, None], x1[:, None, :]].float()
    ly, lx = ly[:, :, None], lx[:, None, :]
    out = (1 - ly) * ((1 - lx) * v00 + lx * v01) + ly * ((1 - lx) * v10 + lx * v11)
    if masks.dtype == torch.bool:
        return (out != 0).float()
    return out.to(masks.dtype).float()


class MaskRCNNLossComputation:
    def __init__(self, cfg):
        rh = cfg.MODEL.ROI_HEADS
        self.matcher = Matcher(rh.FG_IOU_THRESHOLD, rh.BG_IOU_THRESHOLD, allow_low_quality_matches=False)
        self.discretization_size = cfg.MODEL.ROI_MASK_HEAD.RESOLUTION
        self.cls_agnostic_mask = cfg.MODEL.CLS_AGNOSTIC_MASK

    def prepare_targets(self, proposals, targets):
        labels, masks = [], []
        for prop, tgt in zip(proposals, targets):
            if len(prop) == 0:  # an image without positive proposals contributes nothing
                labels.append(torch.zeros(0, dtype=torch.int64, device=prop.bbox.device))
                masks.append(torch.empty(0, dtype=torch.float32, device=prop.bbox.device))
                continue
            gt_masks = tgt.get_field("masks")
            if getattr(prop, "all_positive", False) and prop.has_field("matched_gt") and prop.bbox.is_cuda:
                # sampled positives of the box head: same thresholds, so the match is the one already made
                labels.append(prop.get_field("labels"))
                masks.append(self._project(gt_masks, prop.get_field("matched_gt"), prop.bbox))
                continue
            if isinstance(gt_masks, PastedMasks):
                gt_masks = gt_masks.materialize()
            poly = isinstance(gt_masks, PolygonMasks)
            fused = (prop.bbox.is_cuda and not self.matcher.allow_low_quality_matches
                     and (poly or (gt_masks.dim() == 3 and gt_masks.dtype in (torch.bool, torch.uint8))))
            if fused:
                idx, lab, _ = _C.match_encode(tgt.bbox, tgt.get_field("labels"), prop.bbox, self.matcher.high_threshold,
                                              self.matcher.low_threshold, None, between_keeps_label=True)
            else:
                matched = self.matcher(box_iou(tgt.bbox, prop.bbox))
                idx = matched.clamp(min=0)
                lab = tgt.get_field("labels")[idx].to(torch.int64)
                lab[matched == Matcher.BELOW_LOW_THRESHOLD] = 0
            pos = torch.nonzero(lab > 0).squeeze(1)
            if poly:   # crop + resize + rasterise every positive's polygons in one launch (csrc/polygons.hip)
                masks.append(self._project(gt_masks, idx[pos], prop.bbox[pos]))
            elif fused:  # crop + bilinear resize of every positive's mask in one launch (csrc/targets.hip)
                masks.append(_C.project_masks(gt_masks, idx[pos], prop.bbox[pos], self.discretization_size))
            else:
                masks.append(project_masks_on_boxes(gt_masks, idx[pos], prop.bbox[pos], self.discretization_size))
            labels.append(lab)
        return labels, masks

    def _project(self, gt_masks, gt_index, boxes):
        m = self.discretization_size
        if isinstance(gt_masks, PastedMasks):  # pseudo labels: targets straight from the probability maps
            return _C.project_pasted_masks(gt_masks.probs, gt_masks.boxes, gt_index, boxes, gt_masks.image_size, m,
                                           gt_masks.threshold)
        if isinstance(gt_masks, PolygonMasks):  # COCO polygon ground truth (SegmentationMask mode 'poly')
            if gt_masks.coords.device != boxes.device:
                gt_masks = gt_masks.to(boxes.device)
            return _C.project_polygon_masks(gt_masks.coords, gt_masks.polygon_start, gt_masks.instance_start, gt_index, boxes,
                                            gt_masks.size, m)
        if gt_masks.dim() == 3 and gt_masks.dtype in (torch.bool, torch.uint8):
            return _C.project_masks(gt_masks, gt_index, boxes, m)
        return project_masks_on_boxes(gt_masks, gt_index, boxes, m)

    def fused(self, proposals, mu, sigma, eps, targets):
        """Same value as ``__call__`` on ``mu[None] + eps * sigma`` (one noise sample), through the fused HIP
        forward+backward kernel; ``sigma`` / ``eps`` None = deterministic logits."""
        labels, mask_targets = self.prepare_targets(list(proposals), list(targets))
        labels = _cat(labels, 0)
        mask_targets = _cat([m for m in mask_targets if m.numel() > 0] or mask_targets[:1], 0)
        if all(getattr(p, "all_positive", False) for p in proposals):
            pos = torch.arange(labels.numel(), device=labels.device)
        else:
            pos = torch.nonzero(labels > 0).squeeze(1)
        if mask_targets.numel() == 0:
            return mu.sum() * 0
        self.mask_targets, self.positive_inds = mask_targets, pos
        e = None if eps is None else eps.reshape(mu.shape)
        # class-agnostic: every positive reads logit channel 1 (labels_pos * 0 + 1); class-specific: the channel of its
        # label (mask_logits[positive_inds, labels_pos], mask_head/loss.py:131-141) -- an index into the same kernel
        channel = 1 if self.cls_agnostic_mask else labels.index_select(0, pos)
        return stochastic_mask_bce(mu, sigma, e, pos, mask_targets.reshape(pos.numel(), -1), channel)

    def __call__(self, proposals, mask_logits, targets):
        repeat = 1
        if mask_logits.dim() == 5:  # [n_samples, P, C, M, M]
            repeat = mask_logits.shape[0]
            mask_logits = torch.flatten(mask_logits, 0, 1)
        labels, mask_targets = self.prepare_targets(list(proposals) * repeat, list(targets) * repeat)
        labels = _cat(labels, 0)
        mask_targets = _cat([m for m in mask_targets if m.numel() > 0] or mask_targets[:1], 0)
        pos = torch.nonzero(labels > 0).squeeze(1)
        labels_pos = labels[pos]
        if self.cls_agnostic_mask:
            labels_pos = labels_pos * 0 + 1
        if mask_targets.numel() == 0:
            return mask_logits.sum() * 0
        self.mask_targets, self.positive_inds = mask_targets, pos
        return F.binary_cross_entropy_with_logits(mask_logits[pos, labels_pos], mask_targets, reduction="none").mean()


def paste_mask_in_image(mask, box, im_h, im_w, thresh=0.5, padding=1):
    """mask [M,M] probabilities, box [4] -> bool [im_h, im_w] (mask_head/inference.py:100-160), on device."""
    M = mask.shape[-1]
    scale = float(M + 2 * padding) / M
    padded = mask.new_zeros((M + 2 * padding, M + 2 * padding))
    padded[padding:-padding, padding:-padding] = mask
    w_half = (box[2] - box[0]) * 0.5 * scale
    h_half = (box[3] - box[1]) * 0.5 * scale
    x_c, y_c = (box[2] + box[0]) * 0.5, (box[3] + box[1]) * 0.5
    bx = torch.stack((x_c - w_half, y_c - h_half, x_c + w_half, y_c + h_half)).to(torch.int32).tolist()
    w = max(bx[2] - bx[0] + 1, 1)
    h = max(bx[3] - bx[1] + 1, 1)
    resized = F.interpolate(padded[None, None].float(), size=(h, w), mode="bilinear", align_corners=False)[0, 0]
    resized = resized > thresh
    im_mask = torch.zeros((im_h, im_w), dtype=torch.bool, device=mask.device)
    x0, x1 = max(bx[0], 0), min(bx[2] + 1, im_w)
    y0, y1 = max(bx[1], 0), min(bx[3] + 1, im_h)
    if x1 > x0 and y1 > y0:
        im_mask[y0:y1, x0:x1] = resized[(y0 - bx[1]):(y1 - bx[1]), (x0 - bx[0]):(x1 - bx[0])]
    return im_mask


class Masker:
    def __init__(self, threshold=0.5, padding=1):
        self.threshold, self.padding = threshold, padding

    def __call__(self, masks, boxlist):
        """masks [P,1,M,M], boxlist -> bool [P,1,H,W].  Same arithmetic per mask as ``paste_mask_in_image`` (the
        reference's loop, mask_head/inference.py:124-205), restructured for the device: the expanded integer boxes of ALL
        masks come back in one host read instead of one per mask, padding and the image-size canvas are one fill each,
        and only the resize to the box size (a data-dependent shape) stays per mask."""
        im_w, im_h = boxlist.size
        P = masks.shape[0]
        if P == 0:
            return masks.new_empty((0, 1, im_h, im_w), dtype=torch.bool)
        M, pad = masks.shape[-1], self.padding
        scale = float(M + 2 * pad) / M
        b = boxlist.bbox
        w_half = (b[:, 2] - b[:, 0]) * 0.5 * scale
        h_half = (b[:, 3] - b[:, 1]) * 0.5 * scale
        x_c, y_c = (b[:, 2] + b[:, 0]) * 0.5, (b[:, 3] + b[:, 1]) * 0.5
        boxes = torch.stack((x_c - w_half, y_c - h_half, x_c + w_half, y_c + h_half), 1).to(torch.int32).tolist()
        padded = masks.new_zeros((P, 1, M + 2 * pad, M + 2 * pad), dtype=torch.float32)
        padded[:, 0, pad:-pad, pad:-pad] = masks[:, 0]
        out = torch.zeros((P, 1, im_h, im_w), dtype=torch.bool, device=masks.device)
        for i, bx in enumerate(boxes):
            w = max(bx[2] - bx[0] + 1, 1)
            h = max(bx[3] - bx[1] + 1, 1)
            x0, x1 = max(bx[0], 0), min(bx[2] + 1, im_w)
            y0, y1 = max(bx[1], 0), min(bx[3] + 1, im_h)
            if x1 > x0 and y1 > y0:
                resized = F.interpolate(padded[i:i + 1], size=(h, w), mode="bilinear", align_corners=False)[0, 0]
                out[i, 0, y0:y1, x0:x1] = resized[(y0 - bx[1]):(y1 - bx[1]), (x0 - bx[0]):(x1 - bx[0])] > self.threshold
        return out


class ROIMaskHead(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        self.cfg = cfg
        self.share = cfg.MODEL.ROI_MASK_HEAD.SHARE_BOX_FEATURE_EXTRACTOR
        self.feature_extractor = ResNet50Conv5ROIFeatureExtractor(cfg, cfg.MODEL.ROI_MASK_HEAD)
        self.predictor = MaskRCNNC4Predictor(cfg, self.feature_extractor.out_channels)
        self.loss_evaluator = MaskRCNNLossComputation(cfg)
        self.cls_agnostic_mask = cfg.MODEL.CLS_AGNOSTIC_MASK
        # MODEL.ROI_MASK_HEAD.POSTPROCESS_MASKS (mask_head/inference.py:207-213): the evaluation pass hands out the masks
        # pasted into the image at POSTPROCESS_MASKS_THRESHOLD instead of the 14 x 14 probabilities
        mh = cfg.MODEL.ROI_MASK_HEAD
        self.masker = Masker(threshold=mh.POSTPROCESS_MASKS_THRESHOLD, padding=1) if mh.POSTPROCESS_MASKS else None
        if self.masker is not None and self.masker.threshold < 0:
            raise NotImplementedError("POSTPROCESS_MASKS_THRESHOLD < 0 (the reference's un-thresholded debugging paste)")
        self.log = "N/A"
        self.avg_uncertain = "N/A"

    def forward(self, features, proposals, targets=None, compute_uncertain=False, eps=None):
        if self.training:
            sel = positives_index(proposals)
            if sel is not None:  # device sampler: the positives are known by index
                proposals = [positive_proposals(p) for p in proposals]
            else:
                positive_inds = [p.get_field("labels") > 0 for p in proposals]
                proposals = [p[i] for p, i in zip(proposals, positive_inds)]
                sel = _cat(positive_inds, 0)
        if self.training and self.share:
            fl = features.permute(0, 2, 3, 1) if features.dim() == 4 else None
            if fl is not None and fl.is_contiguous() and not features.is_contiguous():
                # box-head features arrive as an NCHW view of NHWC memory: select the positives in that layout, so the
                # gradient autograd scatters back (zeros + index_put) is NHWC-dense like the pooled gradient it is
                # added to -- the NCHW zeros of a plain features[sel] made that add strided and forced a 411 MB
                # relayout copy in front of the res5 backward (0.8 ms per pass)
                x = fl[sel].permute(0, 3, 1, 2)
            else:
                x = features[sel]
        else:
            x = self.feature_extractor(features, proposals)
        if self.training:
            return x, proposals, dict(loss_mask=self.fused_training_loss(x, proposals, targets, compute_uncertain, eps))
        if compute_uncertain:
            mask_logits, scale = self.predictor(x, True, eps=eps)
            self.log, self.avg_uncertain = scale.max(), scale.mean()
        else:
            mask_logits = self.predictor(x)
        if not self.training:
            prob = mask_logits.sigmoid()
            if self.cls_agnostic_mask:
                prob = prob[:, 1][:, None]
            else:
                labels = _cat([b.get_field("labels") for b in proposals], 0)
                prob = prob[torch.arange(prob.shape[0], device=prob.device), labels][:, None]
            results = []
            for p, b in zip(prob.split([len(b) for b in proposals], 0), proposals):
                out = b.copy_with_fields(b.fields())
                out.add_field("mask", p if self.masker is None else self.masker(p, b))
                results.append(out)
            return x, results, {}
        loss_mask = self.loss_evaluator(proposals, mask_logits, targets)
        return x, proposals, dict(loss_mask=loss_mask)


def _mask_fused_training_loss(self, x, proposals, targets, compute_uncertain=False, eps=None):
    """Training loss of the mask head (class-agnostic or class-specific logits) on the positives' features x (fused
    stochastic BCE; noise drawn on the device unless injected)."""
    mu, sigma = self.predictor.forward_parts(x)
    if compute_uncertain and sigma is not None:
        self.log, self.avg_uncertain = sigma.max(), sigma.mean()
        if eps is None:
            eps = torch.randn((1, *mu.shape), device=mu.device, dtype=mu.dtype)
        else:  # injected noise (tests): a pool at least as large as the positives
            eps = eps[:, : mu.shape[0]].to(mu.device)
        return self.loss_evaluator.fused(proposals, mu, sigma, eps, targets)
    return self.loss_evaluator.fused(proposals, mu, None, None, targets)


ROIMaskHead.fused_training_loss = _mask_fused_training_loss


class CombinedROIHeads(nn.ModuleDict):
    batch_branches = True  # False = one head pass per branch, the reference's order (cross-check in tests/test_model_gpu.py)

    def __init__(self, cfg, in_channels, is_teacher=False):
        heads = [("box", ROIBoxHead(cfg, in_channels, is_teacher))]
        if cfg.MODEL.MASK_ON:
            heads.append(("mask", ROIMaskHead(cfg, in_channels)))
        super().__init__(heads)
        self.cfg = cfg
        self.mask_on = cfg.MODEL.MASK_ON
        if self.mask_on and cfg.MODEL.ROI_MASK_HEAD.SHARE_BOX_FEATURE_EXTRACTOR:
            self.mask.feature_extractor = self.box.feature_extractor

    def branches_batchable(self, feat):
        return (self.training and feat.is_cuda and (not self.mask_on or (self.mask.share and self.mask.cls_agnostic_mask))
                and self.batch_branches)

    def forward_branches(self, feat, branches):
        """Training losses of several independent branches (the student's pseudo-label branch and ground-truth branch,
        st_generalized_rcnn.py:284-408) that share these heads' weights, with ONE pooler + res5 pass over the RoIs of
        all of them: the per-RoI work does not depend on the branch, only the class-embedding matrix of the predictor,
        the targets and the losses do.  ``branches``: dicts with ``image_ids`` (rows of ``feat`` the proposals belong
        to), ``proposals``, ``targets``, ``cls_embs``, ``compute_uncertain``, ``eps``.  Returns one loss dict per
        branch -- the values the sequential calls give (same sampling order), with half the launches and GEMMs of twice
        the height."""
        box = self.box
        with torch.no_grad():
            sampled = box.loss_evaluator.subsample_many([(br["proposals"], br["targets"]) for br in branches])
        rois = _C.rois_from_boxes([p.bbox for props in sampled for p in props],
                                  [img for br in branches for img in br["image_ids"]])  # RoIs of every branch, one launch
        # the mask head reads the res5 features of the positive RoIs only: their indices are known from the sampled lists,
        # so the last res5 block hands those maps out itself (their gradient then enters its backward as dense maps
        # instead of being scattered into a zero tensor of all RoIs first)
        sel_pos = positives_index([p for props in sampled for p in props]) if self.mask_on else None
        x = box.feature_extractor.forward_rois([feat], rois, select=sel_pos)
        pooled = box.predictor.pooled(x)
        counts = [sum(len(p) for p in props) for props in sampled]
        out, off = [], 0
        for br, props, cnt in zip(branches, sampled, counts):
            box.predictor.set_class_embeddings(br["cls_embs"])
            class_logits, box_regression = box.predictor(pooled[off:off + cnt])
            box.loss_evaluator._proposals = props
            lc, lb = box.loss_evaluator(class_logits, box_regression)
            out.append(dict(loss_classifier=lc, loss_box_reg=lb))
            off += cnt
        if self.mask_on:
            mask = self.mask
            sel = sel_pos
            if sel is not None:
                pos_all = [[positive_proposals(p) for p in props] for props in sampled]
            else:
                pos_masks = [[p.get_field("labels") > 0 for p in props] for props in sampled]
                sel = _cat([m for ms in pos_masks for m in ms], 0)
                pos_all = [[p[m] for p, m in zip(props, ms)] for props, ms in zip(sampled, pos_masks)]
            handed = getattr(x, "_ovis_selected", None)
            fl = x.permute(0, 2, 3, 1)
            if handed is not None and handed[0] is sel:
                xs = handed[1].view(sel.numel(), x.shape[2], x.shape[3], x.shape[1]).permute(0, 3, 1, 2)
            else:
                xs = fl[sel].permute(0, 3, 1, 2) if (fl.is_contiguous() and not x.is_contiguous()) else x[sel]  # ONE index op
            off = 0
            for br, pos_props, losses in zip(branches, pos_all, out):
                k = sum(len(p) for p in pos_props)
                losses["loss_mask"] = mask.fused_training_loss(xs[off:off + k], pos_props, br["targets"],
                                                               br.get("compute_uncertain", False), br.get("eps"))
                off += k
        return out

    def forward(self, features, proposals, targets=None, bbox_only=False, compute_uncertain=False, eps=None,
                is_eval_func=False):
        losses, package_x = {}, {}
        # MODEL.GT_BOX_EVAL (roi_heads.py:31-49): the evaluation pass of a detector -- not the teacher's passes inside
        # generate_pseudo_label, which leave ``is_eval_func`` False -- classifies and segments the ground-truth boxes
        gt_boxes = self.cfg.MODEL.GT_BOX_EVAL and is_eval_func and not self.training
        if gt_boxes:
            device = features[0].device if isinstance(features, (list, tuple)) else features.device
            proposals = []
            for t in targets:
                det = t.copy_with_fields(["labels"])
                det.add_field("objectness", t.get_field("labels") * 0.0 + 1.0)
                proposals.append(det.to(device))
        x, detections, loss_box = self.box(features, proposals, targets)
        if gt_boxes:
            for det, tar in zip(detections, targets):
                assert len(det) == len(tar), "GT_BOX_EVAL keeps one detection per ground-truth box"
        package_x["bbox"] = x
        losses.update(loss_box)
        if self.mask_on and not bbox_only:
            mask_features = features
            if self.training and self.cfg.MODEL.ROI_MASK_HEAD.SHARE_BOX_FEATURE_EXTRACTOR:
                mask_features = x
            x, detections, loss_mask = self.mask(mask_features, detections, targets, compute_uncertain, eps=eps)
            package_x["mask"] = x
            losses.update(loss_mask)
        return package_x, detections, losses
