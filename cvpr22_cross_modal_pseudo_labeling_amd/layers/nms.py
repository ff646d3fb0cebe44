"""``nms(boxes, scores, threshold)`` -- maskrcnn_benchmark/layers/nms.py:8 (fp32-only hint kept)."""
from .. import _C
from ._fp32 import float_function


@float_function
def nms(boxes, scores, threshold):
    return _C.nms(boxes, scores, threshold)


@float_function
def nms_padded(boxes, scores, threshold, ge_mode=False):
    return _C.nms_padded(boxes, scores, threshold, ge_mode)
