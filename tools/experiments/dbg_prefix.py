import copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.tiny_model import build_tiny
from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer

model, e_vocab, e_seen, images, targets = build_tiny("zeroshot_mask")
cfg = get_defaults()
cfg.merge_from_file(os.path.join(ROOT, "configs/coco_cap_det/zeroshot_mask.yaml"))
cfg.merge_from_list(["SOLVER.BASE_LR", 1e-4])
cfg.freeze()
images = images.cuda()
tg = [t.to("cuda") for t in targets]
batches = [(images, tg), (images.flip(-1).contiguous(), tg), (images * 0.5, tg), (images, tg)]

def run(kind):
    m = copy.deepcopy(model).cuda()
    m.set_class_embeddings(e_seen.cuda())
    m.train()
    opt = solver.make_optimizer(cfg, m)
    red = comm.BucketedGradReducer(m)
    pipe = trainer.PipelinedTrainer(m, opt, red, threaded=(kind == "threaded"))
    pipe.enabled = not kind.startswith("plain")
    out = []
    for i, (im, t) in enumerate(batches):
        torch.manual_seed(100 + i)
        nxt = batches[i + 1] if i + 1 < len(batches) else None
        out.append({k: float(v) for k, v in pipe.step(im, t, nxt).items()})
    pipe.drain()
    red.remove()
    return out

ref = run("plain")
for kind in ("plain2", "serial", "threaded", "plain3"):
    got = run(kind)
    for i, (a, b) in enumerate(zip(got, ref)):
        print(kind, i, {k: f"{abs(a[k] - b[k]) / max(abs(b[k]), 1e-3):.1e}" for k in b})
