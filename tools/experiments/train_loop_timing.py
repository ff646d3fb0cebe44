"""Where the wall time of tools/train_net.py's loop goes when every iteration sees a NEW synthetic batch (varying proposal
counts -> varying tensor sizes), unlike bench.py's resident batch: batch generation vs step, allocator statistics."""
import os
import sys
import time

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import calibrate_stem_bn, make_batch, make_embeddings  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd.modeling.detector import build_detection_model  # noqa: E402

cfg = get_defaults()
cfg.merge_from_file(os.path.join(ROOT, "configs/coco_cap_det/student_teacher_mask_rcnn_uncertainty.yaml"))
cfg.merge_from_list(["SOLVER.BASE_LR", float(os.environ.get("LR", "1e-5"))])
cfg.freeze()
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_detection_model(cfg).to(dev)
e_vocab, e_seen = make_embeddings(cfg.MODEL.ROI_BOX_HEAD.EMB_DIM, device=dev)
model.set_class_embeddings(e_seen)
model.set_caption_vocab(e_vocab)
calibrate_stem_bn(model, make_batch(1, device=dev, seed=7)[0])
model.train()
opt = solver.make_optimizer(cfg, model)
red = comm.BucketedGradReducer(model)
pipe = trainer.PipelinedTrainer(model, opt, red, None)
mode = sys.argv[1] if len(sys.argv) > 1 else "new"
n = int(os.environ.get("N", "40"))
batches = [make_batch(2, device=dev, seed=1000 + i) for i in range(n if mode != "same" else 1)]
torch.cuda.synchronize()
get = (lambda i: batches[i % len(batches)])
t_step = []
torch.cuda.reset_peak_memory_stats()
s0 = torch.cuda.memory_stats()
for i in range(n):
    b = get(i)
    nx = get(i + 1) if i + 1 < n else None
    torch.cuda.synchronize()
    t = time.perf_counter()
    pipe.step(b[0], b[1], nx)
    torch.cuda.synchronize()
    t_step.append((time.perf_counter() - t) * 1e3)
pipe.drain()
s1 = torch.cuda.memory_stats()
print(mode, "step ms (every 10th):", " ".join(f"{x:.0f}" for x in t_step[::10]))
print(mode, "step ms (last 20):", " ".join(f"{x:.0f}" for x in t_step[-20:]))
print("mean of last 20:", round(sum(t_step[-20:]) / 20, 1), "ms;  hipMalloc calls during loop:",
      s1["num_device_alloc"] - s0["num_device_alloc"], " frees:", s1["num_device_free"] - s0["num_device_free"],
      " alloc retries:", s1["num_alloc_retries"] - s0["num_alloc_retries"],
      " reserved GB:", round(s1["reserved_bytes.all.peak"] / 2**30, 1))

# the library loop itself on the pre-generated batches, without and with the LR scheduler
import logging  # noqa: E402

logging.basicConfig(level=logging.WARNING)
for use_sched in (False, True):
    sched = solver.make_lr_scheduler(cfg, opt) if use_sched else None
    torch.cuda.synchronize()
    t = time.perf_counter()
    if use_sched:
        trainer.do_train(cfg, model, iter(batches * (1 if mode != "same" else n)), opt, sched, n, log_period=1000)
    else:
        p2 = trainer.PipelinedTrainer(model, opt, comm.BucketedGradReducer(model), None)
        for i in range(n):
            b = get(i)
            p2.step(b[0], b[1], get(i + 1) if i + 1 < n else None)
        p2.drain()
    torch.cuda.synchronize()
    print("do_train with scheduler" if use_sched else "pipe.step loop, no per-step sync", round((time.perf_counter() - t) * 1e3 / n, 1), "ms / it")
