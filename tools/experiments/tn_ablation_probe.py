"""3x3 weight-gradient GEMM of the res5 head at the step's size with ablation builds of the kernel (build_variants.sh
"nomask:-DOVIS_TN_ABL_NOMASK" "noshift:-DOVIS_TN_ABL_NOMASK -DOVIS_TN_ABL_NOSHIFT"; wrong results, timing only) next to the 1x1
weight gradients.  python tools/experiments/tn_ablation_probe.py [variant]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
if len(sys.argv) > 1 and sys.argv[1] != "base":
    from cvpr22_cross_modal_pseudo_labeling_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "experiments", "variants", f"libovis_hip_{sys.argv[1]}.so")
from bench_ops import timeit  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd import _C  # noqa: E402

for (m, n, ch, conv, tag) in ((2048 * 49, 512, 512, (7, 7, 3, 3), "3x3 dW"), (2048 * 49, 2048, 512, None, "conv3 dW"),
                              (2048 * 49, 512, 2048, None, "conv1 dW")):
    gp = _C.split_pair(torch.randn(m, n, device="cuda"))
    xp = _C.split_pair(torch.randn(m, ch, device="cuda"))
    taps = 9 if conv else 1
    ms = timeit(lambda: _C.split_gemm_pair_tn(gp, xp, conv), 30)
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'base':8s} {tag:10s} {ms * 1e3:8.1f} us  {6.0 * m * n * ch * taps / ms / 1e9:7.1f} TFLOP/s")
    del gp, xp
