"""Where the RPN's NMS survivors sit among the 12000 score-sorted candidates of the bench's teacher model and data: survivors
per image, index of the 1000th / 2000th (what a scan that stops at post_nms_top_n could skip).
python tools/experiments/nms_survivor_probe.py"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd import _C
from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import calibrate_stem_bn, make_batch, make_embeddings
from cvpr22_cross_modal_pseudo_labeling_amd.modeling.detector import build_detection_model
dev = torch.device("cuda", 0)
cfg = get_defaults(); cfg.merge_from_file(os.path.join(ROOT, "configs/coco_cap_det/zeroshot_mask.yaml")); cfg.freeze()
torch.manual_seed(1234)
model = build_detection_model(cfg).to(dev)
images, targets = make_batch(2, device=dev, seed=1234)
calibrate_stem_bn(model, images); model.train()
orig = _C.nms_presorted_batched
def probe(boxes, drop, thr, **k):
    keep, counts = orig(boxes, drop, thr, **k)
    c = counts.tolist()
    for i in range(boxes.shape[0]):
        n = c[i][0]
        print("image", i, "candidates", boxes.shape[1], "survivors", n, "index of the 2000th survivor", int(keep[i, 1999]) if n >= 2000 else None, "of the 1000th", int(keep[i, 999]) if n >= 1000 else None)
    return keep, counts
_C.nms_presorted_batched = probe
from cvpr22_cross_modal_pseudo_labeling_amd.modeling.structures import to_image_list
with torch.no_grad():
    il = to_image_list(images); feats = model.backbone(il.tensors)
    model.rpn(il, feats, targets, compute_loss=False)
