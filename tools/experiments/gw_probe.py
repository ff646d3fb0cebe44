"""Column-group width of the split GEMM's tile order by K (variant libraries of build_variants.sh with -DOVIS_SG_GW_SHORTK=8 /
-DOVIS_SG_GW_LONGK=2): the step's large 1x1 products, time per call.  python tools/experiments/gw_probe.py [variant]"""
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

torch.manual_seed(0)
for (m, n, k, shortcut) in ((100352, 2048, 512, True), (100352, 2048, 512, False), (100352, 512, 2048, False), (100352, 512, 1024, False),
                            (100352, 2048, 1536, False), (100352, 1024, 512, False)):
    a = _C.split_pair(torch.randn(m, k, device="cuda"))
    b = _C.split_pair(torch.randn(n, k, device="cuda") * 0.05)
    rp = _C.split_pair(torch.randn(m, n, device="cuda")) if shortcut else None
    bias = torch.randn(n, device="cuda")
    t = timeit(lambda: _C.split_gemm_pair(a, b, bias, None, True, False, True, residual_pair=rp), 40)
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'base':8s} M={m} N={n} K={k} shortcut={shortcut}: {t * 1e3:7.1f} us  {6.0 * m * n * k / t / 1e9:7.1f} TF")
    del a, b, rp
