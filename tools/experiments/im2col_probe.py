import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cvpr22_cross_modal_pseudo_labeling_amd import _C
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for shp in [(2,200,334,64),(2,100,167,128),(2,50,84,256),(1024,7,7,512),(2000,7,7,512)]:
    x = torch.randn(*shp, device="cuda")
    ms = t(lambda: _C.im2col_split_bf16x3(x, 3, 3))
    print(shp, f"{ms:.3f} ms", f"{58*x.numel()/ms/1e6:.0f} GB/s")
