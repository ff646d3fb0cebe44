"""Which host-side ops issue the step's ~3000 kernel launches: torch.profiler table by call count."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import calibrate_stem_bn, make_batch, make_embeddings
from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer
from cvpr22_cross_modal_pseudo_labeling_amd.modeling.detector import build_detection_model
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = get_defaults(); cfg.merge_from_file(os.path.join(ROOT, "configs", "coco_cap_det", "student_teacher_mask_rcnn_uncertainty.yaml"))
cfg.merge_from_list(["SOLVER.BASE_LR", 1e-6, "SOLVER.IMS_PER_BATCH", 2]); cfg.freeze()
dev = torch.device("cuda", 0); torch.manual_seed(1234)
model = build_detection_model(cfg).to(dev)
e_vocab, e_seen = make_embeddings(cfg.MODEL.ROI_BOX_HEAD.EMB_DIM, seed=1234, device=dev)
model.set_class_embeddings(e_seen); model.set_caption_vocab(e_vocab)
images, targets = make_batch(2, device=dev, seed=1234); calibrate_stem_bn(model, images)
model.train(); opt = solver.make_optimizer(cfg, model); sch = solver.make_lr_scheduler(cfg, opt); red = comm.BucketedGradReducer(model)
for _ in range(3): trainer.train_step(model, opt, red, images, targets, sch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    trainer.train_step(model, opt, red, images, targets, sch)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=50))
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=40, max_src_column_width=110))
