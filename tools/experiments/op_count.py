"""Count aten ops (~ kernel launches) of one sequential student-teacher step, attributed to the innermost tagged
module / function (TorchDispatchMode + forward hooks), with host wall time per tag.  python tools/experiments/op_count.py"""
import collections
import os
import sys
import time

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import calibrate_stem_bn, make_batch, make_embeddings
from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer
from cvpr22_cross_modal_pseudo_labeling_amd.modeling.detector import build_detection_model

stack = ["top"]
counts = collections.Counter()
opnames = collections.defaultdict(collections.Counter)
wall = collections.Counter()


NO_LAUNCH = ("view", "reshape", "permute", "expand", "slice", "select", "as_strided", "t.", "transpose", "unsqueeze", "squeeze",
             "detach", "alias", "_unsafe_view", "empty", "new_empty", "unbind", "split", "_local_scalar_dense", "sym_", "stride",
             "size", "is_", "_to_copy_noop", "lift_fresh", "chunk", "narrow", "unfold", "resize_", "set_", "record_stream")
sites = collections.Counter()
site_ops = collections.defaultdict(collections.Counter)


class Counter(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        counts[stack[-1]] += 1
        opnames[stack[-1]][func.__name__] += 1
        if not func.__name__.startswith(NO_LAUNCH):
            try:
                f = sys._getframe(1)
            except ValueError:  # an op the autograd thread runs without any Python frame
                f = None
            while f is not None and "cvpr22_cross_modal_pseudo_labeling_amd" not in f.f_code.co_filename:
                f = f.f_back
            where = f"<autograd thread> {func.__name__}" if f is None else f"{f.f_code.co_filename.split('_amd/')[-1]}:{f.f_lineno} {f.f_code.co_name}"
            sites[where] += 1
            site_ops[where][func.__name__] += 1
        return func(*args, **(kwargs or {}))


def tag_module(m, name):
    def pre(mod, inp):
        stack.append(name)
        mod._t0 = time.perf_counter()

    def post(mod, inp, out):
        wall[name] += time.perf_counter() - mod._t0
        stack.pop()

    m.register_forward_pre_hook(pre)
    m.register_forward_hook(post)


def tag_fn(obj, attr, name):
    orig = getattr(obj, attr)

    def wrapped(*a, **k):
        stack.append(name)
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            wall[name] += time.perf_counter() - t0
            stack.pop()

    setattr(obj, attr, wrapped)


WL = sys.argv[1] if len(sys.argv) > 1 else "student"   # student | teacher
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
scheduler = solver.make_lr_scheduler(cfg, optimizer)
reducer = comm.BucketedGradReducer(model)
pipe = trainer.PipelinedTrainer(model, optimizer, reducer, scheduler)
pipe.enabled = False
for _ in range(3):
    pipe.step(images, targets, (images, targets))
torch.cuda.synchronize()

tag_module(model.backbone, "backbone")
tag_module(model.rpn, "rpn")
for hn, heads in ((("teacher", model.roi_heads), ("student", model.roi_heads_student)) if WL == "student"
                  else (("heads", model.roi_heads),)):
    for k in ("box", "mask"):
        h = heads[k]
        tag_module(h, f"{hn}.{k}")
        tag_module(h.feature_extractor, f"{hn}.{k}.feature_extractor")
        tag_module(h.predictor, f"{hn}.{k}.predictor")
        if hasattr(h, "loss_evaluator"):
            le = h.loss_evaluator
            if hasattr(le, "subsample"):
                tag_fn(le, "subsample", f"{hn}.{k}.subsample")
            tag_fn(le, "__call__", f"{hn}.{k}.loss")  # instance attribute is not used by (); handled below
if WL == "student":
    tag_fn(model, "generate_pseudo_label", "generate_pseudo_label")
    tag_fn(model, "compute_dummy_loss", "dummy_loss")
    tag_fn(model, "forward_frozen", "forward_frozen(other)")
    tag_fn(model, "forward_student", "forward_student(other)")
else:
    tag_fn(model.rpn, "loss_evaluator", "rpn.loss")
    tag_module(model.rpn.box_selector_train, "rpn.select")
tag_fn(optimizer, "step", "optimizer.step")

class _CountingLib:
    """Proxy of the ctypes library handle: counts the native-library calls (each launches 1-3 kernels)."""

    def __init__(self, lib):
        self._lib, self.calls = lib, collections.Counter()

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("ovis_"):
            return fn

        def call(*a):
            self.calls[name] += 1
            return fn(*a)
        return call


from cvpr22_cross_modal_pseudo_labeling_amd import _C as _ops
_ops._L = native = _CountingLib(_ops._L)

t0 = time.perf_counter()
with Counter():
    stack.append("step(other: backward, reducer, ...)")
    pipe.step(images, targets, (images, targets))
    stack.pop()
torch.cuda.synchronize()
print(f"step wall {1e3 * (time.perf_counter() - t0):.1f} ms, {sum(counts.values())} aten ops (backward ops run on the autograd thread and are not counted)")
for k, v in counts.most_common():
    print(f"{v:6d} ops  {1e3 * wall[k]:8.2f} ms host wall  {k}")
    print("         " + ", ".join(f"{n}:{c}" for n, c in opnames[k].most_common(8)))
launching_native = {k: v for k, v in native.calls.items() if not k.endswith(("_bytes", "_slices", "_supported", "_version"))}
print(f"--- device work of the step: {sum(sites.values())} launching aten ops (both threads) + {sum(launching_native.values())} "
      "native-library calls; the kernel-trace count of launches per step is in profiles/r2_gap_report_student_nopipe.txt")
print("    native calls: " + ", ".join(f"{n[5:]}:{c}" for n, c in sorted(launching_native.items(), key=lambda kv: -kv[1])))
print("--- launching aten ops by source line (ops of the autograd thread have no Python frame)")
print(sum(sites.values()), "ops")
for k, v in sites.most_common(150):
    print(f"{v:5d}  {k}   " + ", ".join(f"{n}:{c}" for n, c in site_ops[k].most_common(4)))
