"""Reads the per-wave shader-clock sums an OVIS_ROI_PROBE_TIME build of csrc/roi_align_bwd_plane.hip leaves in grad_input[:, :, 0, 0:5]
(round start -> DMA issued | items of the round | round-end wait + barrier | barrier -> next round start | item count)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "experiments", "variants", f"libovis_hip_{sys.argv[1]}.so")
from cvpr22_cross_modal_pseudo_labeling_amd import _C  # noqa: E402
from tools.bench_ops import bench_rois, timeit  # noqa: E402

g = torch.Generator().manual_seed(1234)
n, c, h, w, r = 2, 1024, 50, 84, 1024
for kind in ("uniform", "rpn_like"):
    rois = bench_rois(r, n, g, kind).cuda()
    go = torch.randn(r, c, 14, 14, generator=g).cuda()
    ms = timeit(lambda: _C.roi_align_backward(go, rois, 1 / 16, 14, 14, n, c, h, w, 0), 20)
    out = _C.roi_align_backward(go, rois, 1 / 16, 14, 14, n, c, h, w, 0)
    t = out[:, :, 0, :5].double().cpu()            # [n, c, 5]
    for img in range(n):
        items = t[img, :, 4].mean().item()
        ab, bc, cd, da = (t[img, :, i].mean().item() / items for i in range(4))
        print(f"{kind} image {img}: {ms * 1e3:.1f} us/call, {items:.0f} items (padded); clocks per item and wave: "
              f"round start+DMA {ab:.1f} | items {bc:.1f} | end wait+barrier {cd:.1f} | barrier->next {da:.1f} | sum {ab + bc + cd + da:.1f}")
