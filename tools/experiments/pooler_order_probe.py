"""Does the order in which the NHWC strided pooler walks the RoIs matter (L2 locality)?  Same RoIs in score order (as the
RPN hands them over: spatially random), sorted by (image, y1, x1), and sorted by area.  python tools/experiments/pooler_order_probe.py"""
import os
import sys

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_ops import bench_rois, timeit
from cvpr22_cross_modal_pseudo_labeling_amd import _C

g = torch.Generator().manual_seed(1234)
n, c, h, w, r = 2, 1024, 50, 84, 2048
x = torch.randn(n, h, w, c, generator=g).cuda().permute(0, 3, 1, 2)  # channels-last map
for kind in ("uniform", "rpn_like"):
    rois = bench_rois(r, n, g, kind).cuda()
    area = (rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])
    orders = {"as given": torch.arange(r, device="cuda"),
              "by (image, y1, x1)": torch.argsort(rois[:, 0] * 1e8 + rois[:, 2] * 1e4 + rois[:, 1]),
              "by area": torch.argsort(area), "by (image, area)": torch.argsort(rois[:, 0] * 1e12 + area)}
    for name, o in orders.items():
        rr = rois[o].contiguous()
        ms = timeit(lambda: _C.roi_align_forward_strided_pair(x, rr, 1 / 16, 14, 14, 0, 2), 20)
        print(f"{kind:9s} {name:22s} {1e3 * ms:7.1f} us")
