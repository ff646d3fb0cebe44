"""Same-box A/B of the NMS suppression-mask stores: the shipped library (four mask words per lane as
one aligned 32-byte store, padded row stride) against a variant library built from the previous nms.hip (one 8-byte store per
lane).  Times ``nms_padded`` at K = 6000 / 12000 and the batched score-sorted form of the RPN (2 x 12000) after 0.3 s of the
same op (sustained clock), alternating the two libraries three times.

    python tools/experiments/nms_store_ab.py [variant-name, default nmsold]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import sys, time, json
sys.path.insert(0, %(root)r)
%(patch)s
import torch
from cvpr22_cross_modal_pseudo_labeling_amd import _C
sys.path.insert(0, %(root)r + "/tools")
from bench_ops import timeit
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
out = {}
for k in (6000, 12000):
    xy = torch.rand(k, 2, generator=g) * torch.tensor([1200.0, 720.0]); wh = torch.rand(k, 2, generator=g) * 200 + 8
    boxes = torch.cat([xy, xy + wh], 1).to(dev); scores = torch.rand(k, generator=g).to(dev)
    out["nms_padded_K%%d_us" %% k] = 1e3 * timeit(lambda: _C.nms_padded(boxes, scores, 0.7), 50)
    if k == 12000:
        order = scores.argsort(descending=True)
        b2 = boxes[order][None].repeat(2, 1, 1).contiguous()
        drop = torch.zeros(2, k, dtype=torch.int32, device=dev)
        out["nms_presorted_batched_2x12000_us"] = 1e3 * timeit(lambda: _C.nms_presorted_batched(b2, drop, 0.7, below=k), 50)
print(json.dumps(out))
"""


def run(variant):
    patch = ""
    if variant:
        path = os.path.join(ROOT, "tools", "experiments", "variants", f"libovis_hip_{variant}.so")
        patch = f"from cvpr22_cross_modal_pseudo_labeling_amd import _lib; _lib.LIB_PATH = {path!r}"
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "patch": patch}], capture_output=True, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        raise SystemExit(r.stderr[-2000:])
    return json.loads(lines[-1])


if __name__ == "__main__":
    variant = sys.argv[1] if len(sys.argv) > 1 else "nmsold"
    for i in range(3):
        for name, v in (("shipped (32-byte stores)", ""), (f"variant {variant}", variant)):
            print(i, f"{name:28s}", {k: round(x, 1) for k, x in run(v).items()}, flush=True)
