"""Debug aid: one RoI, constant gradient, HIP vs oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle
if "--lib" in sys.argv:
    from cvpr22_cross_modal_pseudo_labeling_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from cvpr22_cross_modal_pseudo_labeling_amd import _C
torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)
n, c, h, w, p = 1, 8, 50, 84, 14
for box in ([0, 100.0, 100.0, 300.0, 260.0],):
    rois = torch.tensor([box])
    go = torch.ones(1, c, p, p)
    go[:, :, 3:, :] *= 2.0
    go[:, :, :, 5:] *= 3.0
    want = oracle.roi_align_backward(go, rois, 1 / 16, p, p, n, c, h, w, 0)
    got = _C.roi_align_backward(go.cuda(), rois.cuda(), 1 / 16, p, p, n, c, h, w, 0).cpu()
    print("box", box, "max err", (got - want).abs().max().item(), "sum got/want", got.sum().item(), want.sum().item())
    bs_w = want[0, 0].reshape(-1, w).unfold(0, 2, 2).sum(-1) if False else None
    def blocks(t):
        return torch.stack([torch.stack([t[rb * 16:(rb + 1) * 16, cb * 16:(cb + 1) * 16].sum() for cb in range(6)]) for rb in range(4)])
    print("want block sums\n", blocks(want[0, 0])); print("got block sums\n", blocks(got[0, 0]))
    print("channel sums got", got.sum(dim=(0, 2, 3)).tolist())
