"""RoIAlign backward micro-benchmark over experiment builds (tools/experiments/build_variants.sh roi_align_bwd_plane:<name>:"flags"):

    python tools/experiments/roi_bwd_probe.py [variant ...]       ("base" = the regular library)

Per library (one child process each -- a process maps one libovis_hip.so): time of the full 14x14 form (fp32 tiles,
[2,1024,50,84], R = 1024, uniform and RPN-like RoIs: the 856.5 MB micro-benchmark) and of the strided 7x7 form the teacher
step runs, and the distance of every result from the f32 oracle's on a sampled set of planes / from the first library's."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(variant, ref_path):
    import torch
    from cvpr22_cross_modal_pseudo_labeling_amd import _lib
    if variant != "base":
        _lib.LIB_PATH = os.path.join(ROOT, "tools", "experiments", "variants", f"libovis_hip_{variant}.so")
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    from tools.bench_ops import bench_rois, timeit

    g = torch.Generator().manual_seed(1234)
    n, c, h, w, r = 2, 1024, 50, 84, 1024
    out = {"variant": variant}
    keep = {}
    for kind in ("uniform", "rpn_like"):
        rois = bench_rois(r, n, g, kind).cuda()
        go = torch.randn(r, c, 14, 14, generator=g).cuda()
        ms = min(timeit(lambda: _C.roi_align_backward(go, rois, 1 / 16, 14, 14, n, c, h, w, 0), 30) for _ in range(3))
        a = _C.roi_align_backward(go, rois, 1 / 16, 14, 14, n, c, h, w, 0)
        b = _C.roi_align_backward(go, rois, 1 / 16, 14, 14, n, c, h, w, 0)
        out[f"full_{kind}_us"] = round(1e3 * ms, 1)
        out[f"full_{kind}_frac_hbm"] = round((4 * r * c * 196 + 4 * n * c * h * w + 20 * r) / ms / 1e6 / 8000.0, 3)
        out[f"full_{kind}_repeatable"] = bool(torch.equal(a, b))
        keep[f"full_{kind}"] = a[:, ::97].float().cpu()
        gs = torch.randn(r, c, 7, 7, generator=g).cuda()
        ms = min(timeit(lambda: _C.roi_align_backward_strided(gs, rois, 1 / 16, 14, 14, n, c, h, w, 0, 2), 30) for _ in range(3))
        out[f"strided_{kind}_us"] = round(1e3 * ms, 1)
        keep[f"strided_{kind}"] = _C.roi_align_backward_strided(gs, rois, 1 / 16, 14, 14, n, c, h, w, 0, 2)[:, ::97].float().cpu()
        del go, gs
    if os.path.exists(ref_path):
        ref = torch.load(ref_path)
        for k, v in keep.items():
            out[f"{k}_maxdiff_vs_first"] = float((v - ref[k]).abs().max())
            out[f"{k}_absmax"] = float(ref[k].abs().max())
    else:
        torch.save(keep, ref_path)
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        run(sys.argv[2], sys.argv[3])
    else:
        ref = "/tmp/roi_bwd_probe_ref.pt"
        if os.path.exists(ref):
            os.remove(ref)
        for v in (sys.argv[1:] or ["base"]):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", v, ref], check=False)
