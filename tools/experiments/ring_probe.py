"""Split GEMM on the under-filled grids of the step: the plan's form (two LDS stages, two workgroups per CU, K slices) against the
ring forms -- config bits 16 (128x128 tile, ring of 4 stages, one workgroup per CU), 32 (64x128 tile, ring of 5, one per CU),
48 (64x128 tile, ring of 3, two per CU) -- each with 1..4 forced K slices (config >> 8).
python tools/experiments/ring_probe.py [--quick]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import timeit  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd import _C  # noqa: E402

SHAPES = ((8400, 256, 256, (50, 84, 3, 3, False)), (8400, 256, 1024, None), (8400, 1024, 256, None), (8400, 256, 512, None),
          (8400, 512, 1024, None), (8400, 1024, 768, None), (8400, 1024, 1024, (50, 84, 3, 3, False)),
          (33400, 128, 128, (100, 167, 3, 3, False)), (33400, 128, 512, None), (33400, 512, 128, None), (33400, 512, 384, None),
          (1024, 2048, 896, None), (1024, 896, 2048, None), (1372, 1024, 2048, None), (1372, 2048, 1024, None),
          (490, 512, 512, (7, 7, 3, 3, False)), (490, 2048, 512, None), (490, 512, 2048, None))
quick = "--quick" in sys.argv
for (m, n, ch, conv) in SHAPES[:4] if quick else SHAPES:
    k = ch * (9 if conv else 1)
    a = _C.split_pair(torch.randn(m, ch, device="cuda"))
    b = _C.split_pair(torch.randn(n, k, device="cuda") * 0.05)
    bias = torch.randn(n, device="cuda")
    fl = 6.0 * m * n * k
    ref, _ = _C.split_gemm_pair(a, b, bias, None, True, True, False, conv=conv, config=8)
    res = {}
    for ring in (0, 16, 32, 48):
        for sl in ((0,) if ring == 0 else (1, 2, 3, 4)):
            cfg = ring | (sl << 8) | (8 if sl == 1 else 0)
            if sl > 1 and k // 32 // sl < 4:
                continue
            try:
                y, _ = _C.split_gemm_pair(a, b, bias, None, True, True, False, conv=conv, config=cfg)
            except RuntimeError as e:  # e.g. the halo form has no ring
                res[(ring, sl)] = (float("nan"), str(e)[:40])
                continue
            d = float((y - ref).abs().max())
            t = timeit(lambda: _C.split_gemm_pair(a, b, bias, None, True, False, True, conv=conv, config=cfg), 30)
            res[(ring, sl)] = (t * 1e3, d)
    base = res[(0, 0)][0]
    best = min((v[0], kk) for kk, v in res.items() if v[0] == v[0])
    print(f"M={m} N={n} K={k} conv={bool(conv)}: plan {base:6.1f} us ({fl / base / 1e6:6.1f} TF)  best {best[1]} {best[0]:6.1f} us "
          f"({fl / best[0] / 1e6:6.1f} TF)")
    print("    " + "  ".join(f"r{kk[0]}/s{kk[1]}:{v[0]:5.1f}({v[1] if isinstance(v[1], str) else f'{v[1]:.0e}'})" for kk, v in sorted(res.items())))
