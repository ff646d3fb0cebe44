"""MI355X-native hot path of hbdat/cvpr22_cross_modal_pseudo_labeling.

Drop-in for the reference's native-op surface:

* ``cvpr22_cross_modal_pseudo_labeling_amd._C``      <->  ``maskrcnn_benchmark._C``
  (maskrcnn_benchmark/csrc/vision.cpp:9-25), backed by ``libovis_hip.so`` -- hand-written HIP
  kernels for gfx950 behind the C ABI declared in ``include/ovis_hip.h``.
* ``cvpr22_cross_modal_pseudo_labeling_amd.layers``  <->  ``maskrcnn_benchmark.layers``
  (maskrcnn_benchmark/layers/__init__.py:23-46).

Device tensors are only ever served by the HIP library (``ImportError`` when it has not been built: ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C cvpr22_cross_modal_pseudo_labeling_amd/csrc``) -- there is no fallback.  Like the
reference's module, the entry points of its CPU-only configuration (``MODEL.DEVICE cpu``) dispatch on the tensor's device: host
tensors go to in-package host code (``_cpu.py``, ``libovis_cpu.so``: RoIAlign, NMS, the heads' torch formulas); every other op
raises ``RuntimeError`` on host tensors, as the reference raises "Not implemented on the CPU".
"""
__version__ = "0.1.0"
