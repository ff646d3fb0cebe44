"""Under which stream do the gradient reducer's accumulate hooks fire?  Tiny teacher model, two steps; prints, per probed
parameter, "main" or the other stream -- the RPN head's (forward on the side stream since round 6) arrive under the side stream,
which is why engine/comm.py orders a bucket's collective behind every stream its hooks fired under.
python tools/experiments/hook_stream_probe.py"""
import os, sys, torch, warnings
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT))
from tests.tiny_model import build_tiny
from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer
from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
model, e_vocab, e_seen, images, targets = build_tiny("zeroshot_mask")
model = model.cuda(); model.set_class_embeddings(e_seen.cuda()); model.train()
images = images.cuda(); tg = [t.to("cuda") for t in targets]
main = torch.cuda.current_stream()
seen = {}
def mk(name):
    def hook(p):
        cur = torch.cuda.current_stream()
        seen.setdefault(name, []).append("main" if cur == main else f"OTHER({cur})")
    return hook
red = comm.BucketedGradReducer(model)
for n, p in model.named_parameters():
    if p.requires_grad and (n.startswith("rpn.head") or n.endswith("layer3.5.conv3.weight") or n.endswith("layer4.2.conv3.weight") or "bbox_pred" in n):
        p.register_post_accumulate_grad_hook(mk(n))
warnings.simplefilter("always")
for it in range(2):
    red.zero_grad()
    loss = trainer.total_loss(model(images, tg))
    loss.backward()
    red.finish()
torch.cuda.synchronize()
for k, v in seen.items():
    print(k, v)
