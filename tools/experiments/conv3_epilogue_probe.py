"""conv3 of a res5 bottleneck (M = 100352, K = 512 -> N = 2048) with different epilogue traffic: which outputs are
written and whether a shortcut is read.  python tools/experiments/conv3_epilogue_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import timeit  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd import _C  # noqa: E402

for m, k, n in ((2048 * 49, 512, 2048), (2000 * 49, 512, 2048), (2048 * 49, 1536, 2048), (133600, 64, 256), (33400, 128, 512), (8400, 256, 1024)):
    a = _C.split_pair(torch.randn(m, k, device="cuda"))
    b = _C.split_pair(torch.randn(n, k, device="cuda") * 0.05)
    bias = torch.randn(n, device="cuda")
    res = torch.randn(m, n, device="cuda")
    fl = 6.0 * m * n * k
    print(f"M={m} K={k} N={n}")
    for tag, kw in (("f32 + pair out, f32 shortcut (12 B/elem)", dict(residual=res, out_f32=True, out_pair=True)),
                    ("pair out, f32 shortcut          ( 8 B/elem)", dict(residual=res, out_f32=False, out_pair=True)),
                    ("f32 out, f32 shortcut           ( 8 B/elem)", dict(residual=res, out_f32=True, out_pair=False)),
                    ("f32 + pair out, no shortcut     ( 8 B/elem)", dict(residual=None, out_f32=True, out_pair=True)),
                    ("pair out only                   ( 4 B/elem)", dict(residual=None, out_f32=False, out_pair=True)),
                    ("f32 out only                    ( 4 B/elem)", dict(residual=None, out_f32=True, out_pair=False))):
        ms = timeit(lambda: _C.split_gemm_pair(a, b, bias, kw["residual"], True, kw["out_f32"], kw["out_pair"]), 20)
        print(f"  {tag}: {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
    del a, b, res
