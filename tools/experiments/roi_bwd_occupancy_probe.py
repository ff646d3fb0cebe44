"""Does the LDS-plane RoIAlign backward gain from 16 instead of 8 waves per CU?  A 50 x 42 map (half the C4 map's width: the planes
of 8 channels take 67 KB) with twice the channels: the shipped build (24 KB table ring) fits ONE workgroup per CU, a build with a
12 KB ring (build_variants.sh roi_align_bwd_plane:kri3:"-DOVIS_ROI_KRI=3 -DOVIS_ROI_KRING=2") fits TWO.
    python tools/experiments/roi_bwd_occupancy_probe.py [--lib tools/experiments/variants/libovis_hip_kri3.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if "--lib" in sys.argv:
    from cvpr22_cross_modal_pseudo_labeling_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
import torch
from cvpr22_cross_modal_pseudo_labeling_amd import _C

g = torch.Generator().manual_seed(1)
for (n, c, h, w, r) in ((2, 2048, 50, 42, 1024), (2, 1024, 50, 84, 1024)):
    b = torch.randint(0, n, (r, 1), generator=g).float()
    x1 = torch.rand(r, 1, generator=g) * (w * 16 - 16 - 300 * w / 84); y1 = torch.rand(r, 1, generator=g) * 640
    ww = torch.rand(r, 1, generator=g) * 300 * w / 84 + 16; hh = torch.rand(r, 1, generator=g) * 300 + 16
    rois = torch.cat([b, x1, y1, (x1 + ww).clamp(max=w * 16 - 1), (y1 + hh).clamp(max=799)], 1).cuda()
    go = torch.randn(r, c, 14, 14, device="cuda")
    for _ in range(5):
        _C.roi_align_backward(go, rois, 1 / 16, 14, 14, n, c, h, w, 0)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        _C.roi_align_backward(go, rois, 1 / 16, 14, 14, n, c, h, w, 0)
    e.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(e) / 20
    alg = 4 * r * c * 196 + 4 * n * c * h * w
    print(f"map {h}x{w} C={c}: {ms:.4f} ms  {alg / ms / 1e6:.0f} GB/s  frac {alg / ms / 1e6 / 8000:.3f}")
