"""Split GEMM on the under-filled grids of the step (layer3 / RPN head / layer2 shapes): time with the plan's K slices
(config 0) and without (config 8).  python tools/experiments/small_grid_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import timeit  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd import _C  # noqa: E402

for (m, n, ch, conv) in ((8400, 1024, 1024, (50, 84, 3, 3, False)), (8400, 256, 256, (50, 84, 3, 3, False)), (8400, 256, 1024, None),
                         (33400, 128, 128, (100, 167, 3, 3, False)), (8400, 1024, 256, None), (8400, 1024, 768, None), (33400, 128, 512, None),
                         (8400, 256, 512, None), (1372, 2048, 1024, None)):
    k = ch * (9 if conv else 1)
    a = _C.split_pair(torch.randn(m, ch, device="cuda"))
    b = _C.split_pair(torch.randn(n, k, device="cuda") * 0.05)
    bias = torch.randn(n, device="cuda")
    fl = 6.0 * m * n * k
    t = {}
    for cfgv in (0, 8):
        t[cfgv] = timeit(lambda: _C.split_gemm_pair(a, b, bias, None, True, False, True, conv=conv, config=cfgv), 30)
    y0, _ = _C.split_gemm_pair(a, b, bias, None, True, True, False, conv=conv, config=0)
    y8, _ = _C.split_gemm_pair(a, b, bias, None, True, True, False, conv=conv, config=8)
    print(f"M={m} N={n} K={k} conv={bool(conv)}: K slices {t[0] * 1e3:7.1f} us ({fl / t[0] / 1e9:6.1f} TF)   un-split {t[8] * 1e3:7.1f} us   "
          f"max diff {float((y0 - y8).abs().max()):.2e}")
