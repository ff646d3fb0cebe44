"""``nms(boxes, scores, threshold)`` -- maskrcnn_benchmark/layers/nms.py:8 (fp32-only hint kept)."""
from .. import _C
from ._fp32 import float_function

nms = float_function(_C.nms)
nms_padded = float_function(_C.nms_padded)
