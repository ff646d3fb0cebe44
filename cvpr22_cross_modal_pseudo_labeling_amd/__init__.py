"""MI355X-native hot path of hbdat/cvpr22_cross_modal_pseudo_labeling.

Drop-in for the reference's native-op surface:

* ``cvpr22_cross_modal_pseudo_labeling_amd._C``      <->  ``maskrcnn_benchmark._C``
  (maskrcnn_benchmark/csrc/vision.cpp:9-25), backed by ``libovis_hip.so`` -- hand-written HIP
  kernels for gfx950 behind the C ABI declared in ``include/ovis_hip.h``.
* ``cvpr22_cross_modal_pseudo_labeling_amd.layers``  <->  ``maskrcnn_benchmark.layers``
  (maskrcnn_benchmark/layers/__init__.py:23-46).

There is no CPU implementation in this package: ops raise ``RuntimeError`` for CPU tensors and
``ImportError`` when the HIP library has not been built (``python -c "import __graft_entry__ as g;
g.build()"`` or ``make -C cvpr22_cross_modal_pseudo_labeling_amd/csrc``).
"""
__version__ = "0.1.0"
