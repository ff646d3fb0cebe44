"""MFMA-only ablation of the split GEMM loop (tile code 12128: LDS-DMA only for the first k-step, then fragment reads + MFMAs
on the resident stage) on zero-filled vs random operands: how much of the distance to the 2.5 PFLOP/s spec figure is the
data-dependent power / clock behaviour of the matrix cores rather than the kernel's schedule."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cvpr22_cross_modal_pseudo_labeling_amd import _C  # noqa: E402


def t(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


m, k, n = 100352, 1024, 2048
fl = 6.0 * m * n * k
for name, mk in (("zeros", torch.zeros), ("randn", torch.randn), ("zeros again", torch.zeros)):
    ap, bp = _C.split_pair(mk(m, k, device="cuda")), _C.split_pair(mk(n, k, device="cuda"))
    ms0 = t(lambda: _C.split_gemm_pair(ap, bp, tile_m=12128))
    ms1 = t(lambda: _C.split_gemm_pair(ap, bp))
    print(f"{name:12s} MFMA-only {ms0:.3f} ms = {fl / ms0 / 1e9:.0f} TFLOP/s bf16 | shipped kernel {ms1:.3f} ms = {fl / ms1 / 1e9:.0f} TFLOP/s")
