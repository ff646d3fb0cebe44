"""Weight-gradient GEMM (split_gemm_tn_kernel) tile order inside a row slice: shipped = X column
tile (tap, channel block) major / G column tile minor; variants `tn_gmajor` / `tn_xmajor` (tools/experiments/patches/tn_tile_order.patch with -DOVIS_TN_FORCE_GMAJOR=1 / 0) = G
column tile major, so that an XCD's contiguous run of 64 workgroups covers 1-2 of G's four column tiles instead of all four.

    python tools/experiments/tn_order_ab.py                  # same-box alternation of the two libraries, three rounds
    python tools/experiments/tn_order_ab.py child [variant]  # one library, few launches (the program rocprofv3 --pmc runs)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [("res5 3x3 dW", 100352, 512, 512, (7, 7, 3, 3)), ("res5 conv3 dW  N=2048 K=512", 100352, 2048, 512, None),
          ("res5 conv1 dW  N=512 K=2048", 100352, 512, 2048, None)]


def child(variant, iters):
    sys.path.insert(0, ROOT)
    if variant:
        from cvpr22_cross_modal_pseudo_labeling_amd import _lib
        _lib.LIB_PATH = os.path.join(ROOT, "tools", "experiments", "variants", f"libovis_hip_{variant}.so")
    import torch
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_ops import timeit
    g = torch.Generator(device="cuda").manual_seed(0)
    out = {}
    for tag, m, n, k, conv in SHAPES:
        gp = _C.split_pair(torch.randn(m, n, device="cuda", generator=g))
        xp = _C.split_pair(torch.randn(m, k, device="cuda", generator=g))
        fn = (lambda: _C.split_gemm_pair_tn(gp, xp, conv)) if conv else (lambda: _C.split_gemm_pair_tn(gp, xp))
        if iters:
            out[tag] = round(1e3 * timeit(fn, iters), 1)
        else:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
        del gp, xp
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "shipped" else "", int(os.environ.get("TN_AB_ITERS", "0")))
    else:
        env = dict(os.environ, TN_AB_ITERS="50")
        for i in range(3):
            for v in sys.argv[1:] or ("shipped", "tn_gmajor"):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", v], capture_output=True, text=True, env=env)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                print(i, f"{v:9s}", line[-1] if line else r.stderr[-500:], flush=True)
