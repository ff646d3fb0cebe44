"""Forward + backward of the benchmark DCN layer ([2,512,100,168], 3x3 -> 512) a few times: run under
rocprofv3 --kernel-trace --stats to see where the backward's time goes."""
import sys, torch
sys.path.insert(0, "/root/repo")
from cvpr22_cross_modal_pseudo_labeling_amd.layers import deform_conv
g = torch.Generator().manual_seed(0)
x = torch.randn(2, 512, 100, 168, generator=g).cuda().requires_grad_(True)
w = (torch.randn(512, 512, 3, 3, generator=g) * 0.02).cuda().requires_grad_(True)
off = (torch.randn(2, 18, 100, 168, generator=g) * 2).cuda().requires_grad_(True)
gy = torch.randn(2, 512, 100, 168, generator=g).cuda()
for _ in range(6):
    for t in (x, w, off):
        t.grad = None
    deform_conv(x, off, w, 1, 1, 1, 1, 1, 2).backward(gy)
torch.cuda.synchronize()
