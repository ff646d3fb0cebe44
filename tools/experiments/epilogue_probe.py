"""Epilogue / k-loop decomposition of the one-stage 128x128 split GEMM on the res5 shapes, for the regular library and the
experiment builds of tools/experiments/build_variants.sh:

    python tools/experiments/epilogue_probe.py [variant ...]        (variant = name of variants/libovis_hip_<name>.so)

Per library: time of the plain 1x1 product at M = 100352, N = 2048 for K = 512 / 1024 / 2048 with (a) pair output only, (b) pair
output + pair shortcut (conv3 of an identity block), (c) the gated linked form (rp_gated), (d) fp32 output only; the fit
t = a + b * (K / 32) per epilogue kind separates the per-tile fixed cost (epilogue + pipeline fill) from the k-loop."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(variant):
    import torch
    from cvpr22_cross_modal_pseudo_labeling_amd import _lib
    if variant != "base":
        _lib.LIB_PATH = os.path.join(ROOT, "tools", "experiments", "variants", f"libovis_hip_{variant}.so")
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    torch.manual_seed(0)
    m, n = 100352, 2048
    rp = _C.split_pair(torch.randn(m, n, device="cuda"))
    gate = _C.split_pair(torch.randn(m, n, device="cuda"))
    bias = torch.randn(n, device="cuda")

    def t(fn, it=12):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record()
        for _ in range(it):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / it * 1e3

    rows = {}
    for k in (512, 1024, 2048):
        a = _C.split_pair(torch.randn(m, k, device="cuda"))
        b = _C.split_pair(torch.randn(n, k, device="cuda") * 0.05)
        rows[k] = {
            "pair": t(lambda: _C.split_gemm_pair(a, b, bias, None, True, False, True)),
            "pair+rp": t(lambda: _C.split_gemm_pair(a, b, bias, None, True, False, True, residual_pair=rp)),
            "rp_gated": t(lambda: _C.split_gemm_pair_rp_gated(a, b, rp, gate)),
            "f32": t(lambda: _C.split_gemm_pair(a, b, None, None, False, True, False)),
        }
    rounds = (m // 128) * (n // 128) / 1024.0
    print(f"== {variant}: us per launch (TFLOP/s issued) at M={m} N={n}; {rounds:.2f} rounds of 1024 resident workgroups")
    for kind in ("pair", "pair+rp", "rp_gated", "f32"):
        ts = [rows[k][kind] for k in (512, 1024, 2048)]
        b_ = (ts[2] - ts[0]) / (64 - 16)          # us per k-step of the whole launch
        a_ = ts[0] - 16 * b_
        print(f"  {kind:9s} " + "  ".join(f"K={k}: {rows[k][kind]:7.1f} ({6.0 * m * n * k / rows[k][kind] / 1e6:5.0f})" for k in (512, 1024, 2048))
              + f"   fit: fixed {a_:6.1f} us = {a_ / rounds:5.1f} us per round, {b_ / rounds:5.2f} us per k-step and round")


if __name__ == "__main__":
    variants = sys.argv[1:] or ["base"]
    if len(variants) == 1:
        run(variants[0])
    else:
        for v in variants:  # one process per library (a process binds one libovis_hip.so)
            subprocess.run([sys.executable, os.path.abspath(__file__), v], check=False)
