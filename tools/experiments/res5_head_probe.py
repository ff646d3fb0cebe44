"""Probe: ResNetHead NCHW conv path vs NHWC/GEMM path -- parity and fwd+bwd time at R RoIs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
from cvpr22_cross_modal_pseudo_labeling_amd.modeling.backbone import ResNetHead
cfg = get_defaults(); cfg.freeze()
torch.manual_seed(0)
head = ResNetHead(cfg).cuda()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.randn(R, 1024, 14, 14, device="cuda")
def run(nhwc, train):
    head.nhwc = nhwc
    xx = x.clone().requires_grad_(train)
    y = head(xx)
    if train:
        y.square().mean().backward()
    return y
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
with torch.no_grad():
    ya, yb = run(False, False), run(True, False)
print("max |diff| / max |y|:", ((ya - yb).abs().max() / ya.abs().max()).item(), "layout", yb.stride())
head.zero_grad(); run(False, True); ga = [p.grad.clone() for p in head.parameters() if p.grad is not None]
head.zero_grad(); run(True, True); gb = [p.grad.clone() for p in head.parameters() if p.grad is not None]
print("grad rel diff:", max(((a - b).abs().max() / a.abs().max()).item() for a, b in zip(ga, gb)))
for b in head.layer4:
    b.split_gemm = False
    b.split_conv = False
for nhwc, c33, sg in ((False, True, False), (True, True, False), (True, True, True), (True, None, True)):
    for b in head.layer4:
        b.conv3x3_nchw = c33
        b.split_gemm = sg
    with torch.no_grad():
        f = t(lambda: run(nhwc, False))
    fb = t(lambda: run(nhwc, True))
    print(f"nhwc={nhwc} conv3x3_nchw={c33} split_gemm={sg} split_conv={head.layer4[0].split_conv}: fwd {f:.2f} ms  fwd+bwd {fb:.2f} ms  (R={R})")

for b in head.layer4:
    b.split_conv = True
with torch.no_grad():
    yc = run(True, False)
    f = t(lambda: run(True, False))
fb = t(lambda: run(True, True))
print("split conv: max |diff| / max |y|:", ((ya - yc).abs().max() / ya.abs().max()).item(), f"fwd {f:.2f} ms  fwd+bwd {fb:.2f} ms")
