"""Weight-gradient GEMM of the res5 3x3 at the step's size (M = 100352 rows) for several row-slice counts: time per launch
(HIP events, 20 launches).  python tools/experiments/tn_slices_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import timeit  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd import _C  # noqa: E402

L = _C._L
orig = L.ovis_split_gemm_tn_slices
BIG = ((2048 * 49, 512, 512, (7, 7, 3, 3), "3x3 dW"), (2048 * 49, 2048, 512, None, "conv3 dW"),
       (2048 * 49, 512, 2048, None, "conv1 b1 dW"), (2048 * 49, 2048, 1024, None, "shortcut dW"))
SMALL = ((8400, 256, 256, (50, 84, 3, 3), "layer3 3x3 dW"), (8400, 1024, 256, None, "layer3 conv3 dW"), (8400, 256, 1024, None, "layer3 conv1 dW"),
         (33400, 128, 128, (100, 167, 3, 3), "layer2 3x3 dW"), (33400, 512, 128, None, "layer2 conv3 dW"), (33400, 128, 512, None, "layer2 conv1 dW"),
         (8400, 1024, 1024, (50, 84, 3, 3), "RPN head dW"))   # the trainable trunk of the teacher step (--small)
for (m, n, ch, conv, tag) in (SMALL if "--small" in sys.argv else BIG):
    gp = _C.split_pair(torch.randn(m, n, device="cuda"))
    xp = _C.split_pair(torch.randn(m, ch, device="cuda"))
    taps = 9 if conv else 1
    fl = 6.0 * m * n * ch * taps
    ms0 = timeit(lambda: _C.split_gemm_pair_tn(gp, xp, conv), 20)
    print(tag, "default slices", orig(m, n, ch, taps), f"{ms0 * 1e3:8.1f} us")
    for s in ((1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32) if "--small" in sys.argv else (2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 21, 28)):
        L.ovis_split_gemm_tn_slices = lambda *a, s=s: s
        try:
            ms = timeit(lambda: _C.split_gemm_pair_tn(gp, xp, conv), 20)
        finally:
            L.ovis_split_gemm_tn_slices = orig
        tiles = (n // 128) * (taps * ch // 128)
        print(f"  slices {s:3d}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  ({s * tiles} workgroups = {s * tiles / 512:.2f} rounds)")
    del gp, xp
