"""Tensor-library ops of one training step that touch a LARGE tensor (>= 8 M elements: fills, copies, adds, relayouts ...), with the
package line that issued them -- the byte passes that are not hand-written kernels.  python tools/experiments/big_passes.py [student|teacher]"""
import collections
import os
import sys

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import calibrate_stem_bn, make_batch, make_embeddings
from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer
from cvpr22_cross_modal_pseudo_labeling_amd.modeling.detector import build_detection_model

NO_LAUNCH = ("view", "reshape", "permute", "expand", "slice", "select", "as_strided", "t.", "transpose", "unsqueeze", "squeeze",
             "detach", "alias", "_unsafe_view", "empty", "new_empty", "unbind", "split", "_local_scalar_dense", "sym_", "stride",
             "size", "is_", "_to_copy_noop", "lift_fresh", "chunk", "narrow", "unfold", "resize_", "set_", "record_stream")
BIG = 8 << 20
seen = collections.Counter()


def _numel(x):
    if torch.is_tensor(x):
        return x.numel()
    if isinstance(x, (list, tuple)):
        return max((_numel(t) for t in x), default=0)
    return 0


class Big(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if func.__name__.startswith(NO_LAUNCH):
            return out
        n = max(_numel(out), max((_numel(a) for a in args), default=0))
        if n >= BIG:
            try:
                f = sys._getframe(1)
            except ValueError:
                f = None
            while f is not None and "cvpr22_cross_modal_pseudo_labeling_amd" not in f.f_code.co_filename:
                f = f.f_back
            where = "<autograd thread>" if f is None else f"{f.f_code.co_filename.split('_amd/')[-1]}:{f.f_lineno} {f.f_code.co_name}"
            shapes = [tuple(a.shape) for a in args if torch.is_tensor(a)]
            seen[(func.__name__, where, str(shapes)[:110])] += 1
        return out


WL = sys.argv[1] if len(sys.argv) > 1 else "student"
dev = torch.device("cuda", 0)
cfg = get_defaults()
cfg.merge_from_file(os.path.join(ROOT, "configs", "coco_cap_det",
                                 ("student_teacher_mask_rcnn_uncertainty" if WL == "student" else "zeroshot_mask") + ".yaml"))
cfg.merge_from_list(["SOLVER.BASE_LR", 1e-6, "SOLVER.IMS_PER_BATCH", 2])
cfg.freeze()
torch.manual_seed(1234)
model = build_detection_model(cfg).to(dev)
e_vocab, e_seen = make_embeddings(cfg.MODEL.ROI_BOX_HEAD.EMB_DIM, seed=1234, device=dev)
model.set_class_embeddings(e_seen)
if hasattr(model, "set_caption_vocab"):
    model.set_caption_vocab(e_vocab)
images, targets = make_batch(2, device=dev, seed=1234)
calibrate_stem_bn(model, images)
model.train()
optimizer = solver.make_optimizer(cfg, model)
reducer = comm.BucketedGradReducer(model)
pipe = trainer.PipelinedTrainer(model, optimizer, reducer, None)
pipe.enabled = False
for _ in range(2):
    pipe.step(images, targets, (images, targets))
torch.cuda.synchronize()
with Big():
    pipe.step(images, targets, (images, targets))
torch.cuda.synchronize()
for (op, where, shapes), c in sorted(seen.items(), key=lambda kv: kv[0][1]):
    print(f"{c:3d}  {op:28s} {where:60s} {shapes}")
