"""fp32-only op guard: the role apex's ``amp.float_function`` plays in the reference
(maskrcnn_benchmark/layers/roi_align.py:57, layers/nms.py:8) without depending on apex."""
import functools

import torch


def float_function(fn):
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        def cast(a):
            if torch.is_tensor(a) and a.is_floating_point() and a.dtype != torch.float32:
                return a.float()
            return a

        args = [cast(a) for a in args]
        kwargs = {k: cast(v) for k, v in kwargs.items()}
        with torch.autocast(device_type="cuda", enabled=False):
            return fn(*args, **kwargs)

    return wrapper
