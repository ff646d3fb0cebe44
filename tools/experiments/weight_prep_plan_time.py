"""Duration of the one-launch weight preparation (WeightPrepPlan) over the teacher's 43 trainable convolutions.
python tools/experiments/weight_prep_plan_time.py"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd.layers.pair_bottleneck import WeightPrepPlan
g = torch.Generator().manual_seed(0)
def conv(n, c, k): return (torch.randn(n, c, k, k, generator=g) * 0.1).cuda().requires_grad_(), (torch.rand(n, generator=g) + 0.5).cuda()
blocks = []
for i in range(4): blocks.append((f"l2{i}", [conv(128, 512, 1), conv(128, 128, 3), conv(512, 128, 1)]))
for i in range(6): blocks.append((f"l3{i}", [conv(256, 1024, 1), conv(256, 256, 3), conv(1024, 256, 1)]))
for i in range(3): blocks.append((f"l4{i}", [conv(512, 2048, 1), conv(512, 512, 3), conv(2048, 512, 1)]))
blocks.append(("rpn", [(conv(1024, 1024, 3)[0], None)]))
plan = WeightPrepPlan(blocks)
for _ in range(5): plan.run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): plan.run()
b.record(); torch.cuda.synchronize()
n = sum(w.numel() for _, convs in blocks for w, _ in convs)
print(f"{n/1e6:.1f} M weights, {a.elapsed_time(b)/50*1e3:.1f} us per launch, {12*n/(a.elapsed_time(b)/50*1e-3)/1e12:.2f} TB/s")
