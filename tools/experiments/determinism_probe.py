"""Which part of a training step is not bit-reproducible run to run?  (VERDICT r5 missing-3)

    python tools/experiments/determinism_probe.py [zeroshot_mask|student_teacher_mask_rcnn_uncertainty] [--warn]

Two runs of three optimisation steps of the tiny model (tests/tiny_model.py) from identical weights, seeds and batches: per
step the losses that differ, after step 0 the gradients that differ (name, max |delta|), at the end the parameters that differ.
--warn: one extra step under torch.use_deterministic_algorithms(True, warn_only=True) to list the torch ops it flags.
"""
import copy
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.tiny_model import build_tiny  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer  # noqa: E402

name = next((a for a in sys.argv[1:] if not a.startswith("--")), "zeroshot_mask")
model, e_vocab, e_seen, images, targets = build_tiny(name)
cfg = get_defaults()
cfg.merge_from_file(os.path.join(ROOT, f"configs/coco_cap_det/{name}.yaml"))
cfg.merge_from_list(["SOLVER.BASE_LR", 1e-4])
cfg.freeze()
images = images.cuda()
tg = [t.to("cuda") for t in targets]
batches = [(images, tg), (images.flip(-1).contiguous(), tg), (images * 0.5, tg)]


def run(grads_after_first=False):
    m = copy.deepcopy(model).cuda()
    m.set_class_embeddings(e_seen.cuda())
    if hasattr(m, "set_caption_vocab"):
        m.set_caption_vocab(e_vocab.cuda())
    m.train()
    opt = solver.make_optimizer(cfg, m)
    red = comm.BucketedGradReducer(m)
    pipe = trainer.PipelinedTrainer(m, opt, red)
    pipe.enabled = False
    losses, grads = [], None
    for i, (im, t) in enumerate(batches):
        torch.manual_seed(100 + i)
        losses.append({k: float(v) for k, v in pipe.step(im, t, None).items()})
        if i == 0 and grads_after_first:
            grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    pipe.drain()
    red.remove()
    return losses, grads, {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}


a_l, a_g, a_w = run(True)
for trial in range(3):
    b_l, b_g, b_w = run(True)
    for i, (x, y) in enumerate(zip(a_l, b_l)):
        d = {k: abs(x[k] - y[k]) for k in x if x[k] != y[k]}
        print(f"trial {trial} step {i}: losses differing: {d if d else 'none (bit-identical)'}")
    gd = {n: float((a_g[n] - b_g[n]).abs().max()) for n in a_g if not torch.equal(a_g[n], b_g[n])}
    print(f"trial {trial} gradients after step 0 differing: {len(gd)} of {len(a_g)}")
    for n, v in sorted(gd.items(), key=lambda kv: -kv[1])[:12]:
        print(f"    {n:70s} {v:.3e}  (|g|max {float(a_g[n].abs().max()):.3e})")
    wd = [n for n in a_w if not torch.equal(a_w[n], b_w[n])]
    print(f"trial {trial} parameters differing after 3 steps: {len(wd)} of {len(a_w)}")

if "--warn" in sys.argv:
    torch.use_deterministic_algorithms(True, warn_only=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        run()
    for msg in sorted({str(x.message)[:200] for x in w}):
        print("WARN", msg)
