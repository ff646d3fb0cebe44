"""Host-side cost of the step's native calls: each wrapper issued N times back to back on small operands (layer3 shapes: M = 8400)
without waiting for the GPU -- microseconds of HOST time per call (Python wrapper + allocator + ctypes + launch).
python tools/experiments/host_cost_probe.py"""
import os
import sys
import time

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd import _C  # noqa: E402

dev = torch.device("cuda", 0)
M, C, CM = 8400, 1024, 256
x = torch.randn(M, C, device=dev)
xp = _C.split_pair(x)
w1 = torch.randn(CM, C, 1, 1, device=dev) * 0.05
w2 = torch.randn(CM, CM, 3, 3, device=dev) * 0.05
p1, t1 = _C.weight_prep_pair(w1, None, True)
p2, t2 = _C.weight_prep_pair(w2, None, True)
b1 = torch.zeros(CM, device=dev)
_, o1p = _C.split_gemm_pair(xp, p1, b1, None, True, False, True)
_, o2p = _C.split_gemm_pair(o1p, p2, b1, None, True, False, True, conv=(50, 84, 3, 3, False))
g = torch.randn(M, CM, device=dev)
gp = _C.split_pair(g)


def host_us(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return t


rows = [
    ("torch.empty (1 MB)", lambda: torch.empty(262144, device=dev)),
    ("torch add (tiny)", lambda: b1 + b1),
    ("split_pair", lambda: _C.split_pair(g)),
    ("weight_prep_pair 1x1", lambda: _C.weight_prep_pair(w1, None, True)),
    ("weight_prep_pair 3x3", lambda: _C.weight_prep_pair(w2, None, True)),
    ("split_gemm_pair 1x1 (pair out)", lambda: _C.split_gemm_pair(xp, p1, b1, None, True, False, True)),
    ("split_gemm_pair 3x3 (pair out)", lambda: _C.split_gemm_pair(o1p, p2, b1, None, True, False, True, conv=(50, 84, 3, 3, False))),
    ("split_gemm_pair_gated 3x3 dX", lambda: _C.split_gemm_pair_gated(gp, t2, o1p, conv=(50, 84, 3, 3, True))),
    ("split_gemm_pair_tn 1x1 + slab sum", lambda: _C.split_gemm_pair_tn(gp, xp, None, scale=None, weight_shape=(CM, C, 1, 1))),
    ("split_gemm_pair_tn 3x3 + slab sum", lambda: _C.split_gemm_pair_tn(gp, o1p, (50, 84, 3, 3), scale=None, weight_shape=(CM, CM, 3, 3))),
    ("gate_split_pair", lambda: _C.gate_split_pair(g, o1p)),
]
for name, fn in rows:
    print(f"{name:40s} {host_us(fn):7.1f} us host per call")
