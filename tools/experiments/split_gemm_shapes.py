"""Times the split GEMM on every distinct (M, N, K, taps) of one student-teacher step (profiles/r2_split_gemm_per_shape.csv)
under the launcher's choice (config 0) and the alternative launch forms (include/ovis_hip.h: 1 / 2 = one / two LDS
stages, 4 = shifted-row 3x3 instead of the halo tile, 8 = no K slices).
    python tools/experiments/split_gemm_shapes.py [--csv profiles/r2_split_gemm_per_shape.csv] [--iters 10]
"""
import argparse
import csv
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd import _C  # noqa: E402

MAPS = {133600: (200, 334), 33400: (100, 167), 8400: (50, 84)}  # rows -> (H, W) of the trunk's maps (2 images)


def t(fn, iters):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def best_of(fns, iters, rounds=3):
    """Interleaved rounds over the variants (same clocks / cache state for all), minimum per variant."""
    for fn in fns:
        if fn is not None:
            fn()
    best = [None] * len(fns)
    for _ in range(rounds):
        for i, fn in enumerate(fns):
            if fn is not None:
                us = t(fn, iters)
                best[i] = us if best[i] is None else min(best[i], us)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--csv", default=os.path.join(ROOT, "profiles", "r2_split_gemm_per_shape.csv"))
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--min-us", type=float, default=0.0, help="skip shapes below this many us per step in the csv")
    args = ap.parse_args()
    rows = [r for r in csv.DictReader(open(args.csv)) if r["op"] in ("split_gemm_pair", "split_gemm_pair_gated")]
    seen = set()
    print("op,M,N,K,taps,csv_us," + ",".join(f"cfg{c}_us" for c in (0, 1, 2, 4, 5, 8)) + ",cfg0_TFLOPs")
    for r in rows:
        m, n, k, taps = int(r["M"]), int(r["N"]), int(r["K"]), int(r["taps"])
        key = (r["op"], m, n, k, taps)
        if key in seen or float(r["us_per_step"]) < args.min_us:
            continue
        seen.add(key)
        ch = k // taps
        a = torch.randn(m, ch, device="cuda")
        b = torch.randn(n, k, device="cuda") * 0.05
        ap_, bp = _C.split_pair(a), _C.split_pair(b)
        conv = None
        if taps > 1:
            h, w = MAPS.get(m, (7, 7))
            conv = (h, w, 3, 3, r["op"].endswith("gated"))
        gate = _C.split_pair(torch.randn(m, n, device="cuda")) if (r["op"].endswith("gated") and n % 32 == 0) else None
        del a, b
        fns = []
        for cfg in (0, 1, 2, 4, 5, 8):
            if (cfg in (4, 5) and taps == 1):
                fns.append(None)
            elif gate is not None:
                fns.append(lambda cfg=cfg: _C.split_gemm_pair_gated(ap_, bp, gate, conv=conv, config=cfg))
            else:
                f32 = n >= 1024 or n % 32 != 0
                fns.append(lambda cfg=cfg: _C.split_gemm_pair(ap_, bp, None, None, True, f32, not f32, conv=conv, config=cfg))
        out = ["" if v is None else f"{v:.1f}" for v in best_of(fns, args.iters)]
        tf = 6.0 * m * n * k / float(out[0]) / 1e6
        print(f"{r['op']},{m},{n},{k},{taps},{float(r['avg_us']):.1f}," + ",".join(out) + f",{tf:.0f}", flush=True)


if __name__ == "__main__":
    main()
