"""Where the host time of one sequential student-teacher step goes: cProfile over a few steps (host reads show up as the
time spent inside .tolist() / .item()).  python tools/experiments/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import calibrate_stem_bn, make_batch, make_embeddings
from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer
from cvpr22_cross_modal_pseudo_labeling_amd.modeling.detector import build_detection_model

dev = torch.device("cuda", 0)
cfg = get_defaults()
cfg.merge_from_file(os.path.join(ROOT, "configs", "coco_cap_det", "student_teacher_mask_rcnn_uncertainty.yaml"))
cfg.merge_from_list(["SOLVER.BASE_LR", 1e-6, "SOLVER.IMS_PER_BATCH", 2])
cfg.freeze()
torch.manual_seed(1234)
model = build_detection_model(cfg).to(dev)
e_vocab, e_seen = make_embeddings(cfg.MODEL.ROI_BOX_HEAD.EMB_DIM, seed=1234, device=dev)
model.set_class_embeddings(e_seen)
model.set_caption_vocab(e_vocab)
images, targets = make_batch(2, device=dev, seed=1234)
calibrate_stem_bn(model, images)
model.train()
optimizer = solver.make_optimizer(cfg, model)
reducer = comm.BucketedGradReducer(model)
pipe = trainer.PipelinedTrainer(model, optimizer, reducer, solver.make_lr_scheduler(cfg, optimizer))
pipe.enabled = False
for _ in range(5):
    pipe.step(images, targets, (images, targets))
torch.cuda.synchronize()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    pipe.step(images, targets, (images, targets))
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print(f"--- {steps} sequential steps; times are totals over them (divide by {steps})")
st.print_stats(45)
st.sort_stats("cumulative")
st.print_stats(60)
