"""``FrozenBatchNorm2d`` -- maskrcnn_benchmark/layers/batch_norm.py:6-31 (note: no epsilon).

Same four buffers (``weight, bias, running_mean, running_var``) so reference checkpoints load
unchanged.  ``fold()`` returns the per-channel (scale, shift) pair so a caller can fold the
affine into the preceding convolution at load time instead of running it as a separate
memory-bound pass."""
import torch
from torch import nn


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def fold_key(self):
        """Identity + version of the four buffers: what any cache of a value derived from ``fold()`` must be keyed on."""
        return tuple((id(b), b._version) for b in (self.weight, self.bias, self.running_mean, self.running_var))

    def fold(self):
        # the four buffers are constants of the step: the pair is computed once per buffer state (in-place updates such
        # as load_state_dict bump the version counters, .to() replaces the tensors)
        key = self.fold_key()
        cached = getattr(self, "_fold_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1], cached[2]
        scale = self.weight * self.running_var.rsqrt()
        shift = self.bias - self.running_mean * scale
        self._fold_cache = (key, scale, shift)
        return scale, shift

    def forward(self, x):
        scale, shift = self.fold()
        if x.dtype != scale.dtype:
            scale, shift = scale.to(x.dtype), shift.to(x.dtype)
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)
