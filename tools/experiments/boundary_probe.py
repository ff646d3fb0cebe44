"""What the step boundary of the pipelined teacher step costs: host time of its parts (perf_counter, no syncs) and the GPU time
between the last kernel of a backward and the first kernel of the next forward (event pair on the main stream), which holds the
fused SGD launch, the weight preparation and whatever idle the host leaves.  python tools/experiments/boundary_probe.py"""
import os
import sys
import time

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import calibrate_stem_bn, make_batch, make_embeddings
from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer
from cvpr22_cross_modal_pseudo_labeling_amd.modeling.detector import build_detection_model

dev = torch.device("cuda", 0)
cfg = get_defaults()
cfg.merge_from_file(os.path.join(ROOT, "configs", "coco_cap_det", "zeroshot_mask.yaml"))
cfg.merge_from_list(["SOLVER.BASE_LR", 1e-6, "SOLVER.IMS_PER_BATCH", 2])
cfg.freeze()
torch.manual_seed(1234)
model = build_detection_model(cfg).to(dev)
_, e_seen = make_embeddings(cfg.MODEL.ROI_BOX_HEAD.EMB_DIM, seed=1234, device=dev)
model.set_class_embeddings(e_seen)
images, targets = make_batch(2, device=dev, seed=1234)
calibrate_stem_bn(model, images)
model.train()
opt = solver.make_optimizer(cfg, model)
sched = solver.make_lr_scheduler(cfg, opt)
red = comm.BucketedGradReducer(model)
pipe = trainer.PipelinedTrainer(model, opt, red, sched)
pol = pipe.policy
marks, evs = [], []
end0, begin0 = pol.end, pol.begin


def end(*a, **k):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    t0 = time.perf_counter()
    out = end0(*a, **k)
    marks.append(["end", time.perf_counter() - t0])
    evs.append([e, None])
    return out


def begin(*a, **k):
    t0 = time.perf_counter()
    out = begin0(*a, **k)
    if evs and evs[-1][1] is None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs[-1][1] = e
    marks.append(["begin", time.perf_counter() - t0])
    return out


pol.end, pol.begin = end, begin
step_t = []
for i in range(30):
    t0 = time.perf_counter()
    pipe.step(images, targets, (images, targets))
    step_t.append(time.perf_counter() - t0)
torch.cuda.synchronize()
tail = lambda xs: xs[10:]
ends = [m[1] for m in marks if m[0] == "end"]
begins = [m[1] for m in marks if m[0] == "begin"]
gaps = [a.elapsed_time(b) for a, b in evs if b is not None]
print(f"host: policy.end {1e3 * sum(tail(ends)) / len(tail(ends)):.3f} ms, policy.begin {1e3 * sum(tail(begins)) / len(tail(begins)):.3f} ms per step")
print(f"GPU, main stream: end of backward -> first forward launch {sum(tail(gaps)) / len(tail(gaps)):.3f} ms (SGD 0.12 + weight preparation 0.09-0.16 + idle)")
print(f"host time of step() {1e3 * sum(tail(step_t)) / len(tail(step_t)):.3f} ms")
