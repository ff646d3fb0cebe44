"""Box <-> regression-delta codec (maskrcnn_benchmark/modeling/box_coder.py:7-95)."""
import math

import torch

TO_REMOVE = 1.0


class BoxCoder:
    def __init__(self, weights, bbox_xform_clip=math.log(1000.0 / 16)):
        self.weights = weights
        self.bbox_xform_clip = bbox_xform_clip

    def encode(self, reference_boxes, proposals):
        ex_w = proposals[:, 2] - proposals[:, 0] + TO_REMOVE
        ex_h = proposals[:, 3] - proposals[:, 1] + TO_REMOVE
        ex_cx = proposals[:, 0] + 0.5 * ex_w
        ex_cy = proposals[:, 1] + 0.5 * ex_h
        gt_w = reference_boxes[:, 2] - reference_boxes[:, 0] + TO_REMOVE
        gt_h = reference_boxes[:, 3] - reference_boxes[:, 1] + TO_REMOVE
        gt_cx = reference_boxes[:, 0] + 0.5 * gt_w
        gt_cy = reference_boxes[:, 1] + 0.5 * gt_h
        wx, wy, ww, wh = self.weights
        return torch.stack((wx * (gt_cx - ex_cx) / ex_w, wy * (gt_cy - ex_cy) / ex_h,
                            ww * torch.log(gt_w / ex_w), wh * torch.log(gt_h / ex_h)), dim=1)

    def decode(self, rel_codes, boxes, rows_per_image=None, image_sizes=None):
        """``rows_per_image`` / ``image_sizes`` ((width, height) per image; extension): also clip every row to its image, as
        ``BoxList.clip_to_image`` would afterwards.  Device tensors: one launch (``_C.box_decode``)."""
        if rel_codes.is_cuda and rel_codes.dtype == torch.float32:
            from .. import _C
            return _C.box_decode(rel_codes, boxes, self.weights, self.bbox_xform_clip, rows_per_image, image_sizes)
        pred = self._decode_tensor_ops(rel_codes, boxes)
        if rows_per_image is not None:
            at = 0
            for n, (w, h) in zip(rows_per_image, image_sizes):
                rows = pred[at:at + n]
                rows[:, 0::4].clamp_(min=0, max=w - TO_REMOVE)
                rows[:, 1::4].clamp_(min=0, max=h - TO_REMOVE)
                rows[:, 2::4].clamp_(min=0, max=w - TO_REMOVE)
                rows[:, 3::4].clamp_(min=0, max=h - TO_REMOVE)
                at += n
        return pred

    def _decode_tensor_ops(self, rel_codes, boxes):
        boxes = boxes.to(rel_codes.dtype)
        widths = boxes[:, 2] - boxes[:, 0] + TO_REMOVE
        heights = boxes[:, 3] - boxes[:, 1] + TO_REMOVE
        ctr_x = boxes[:, 0] + 0.5 * widths
        ctr_y = boxes[:, 1] + 0.5 * heights
        wx, wy, ww, wh = self.weights
        dx = rel_codes[:, 0::4] / wx
        dy = rel_codes[:, 1::4] / wy
        dw = torch.clamp(rel_codes[:, 2::4] / ww, max=self.bbox_xform_clip)
        dh = torch.clamp(rel_codes[:, 3::4] / wh, max=self.bbox_xform_clip)
        pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
        pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
        pred_w = torch.exp(dw) * widths[:, None]
        pred_h = torch.exp(dh) * heights[:, None]
        pred = torch.zeros_like(rel_codes)
        pred[:, 0::4] = pred_ctr_x - 0.5 * pred_w
        pred[:, 1::4] = pred_ctr_y - 0.5 * pred_h
        pred[:, 2::4] = pred_ctr_x + 0.5 * pred_w - 1  # "- 1" is the reference's x2 convention
        pred[:, 3::4] = pred_ctr_y + 0.5 * pred_h - 1
        return pred
