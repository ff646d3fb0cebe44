import sys, torch
sys.path.insert(0, "/root/repo")
from cvpr22_cross_modal_pseudo_labeling_amd import _C
torch.manual_seed(0)
m, n, k = 100352, 2048, 512
a = _C.split_pair(torch.randn(m, k, device="cuda"))
b = _C.split_pair(torch.randn(n, k, device="cuda") * 0.05)
rp = _C.split_pair(torch.randn(m, n, device="cuda"))
bias = torch.randn(n, device="cuda")
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for cfg in (0, 2):
    us = t(lambda: _C.split_gemm_pair(a, b, bias, None, True, False, True, config=cfg, residual_pair=rp))
    print("conv3+pair shortcut config", cfg, "%.1f us" % us, "%.0f TF/s" % (6.0 * m * n * k / us / 1e6))
    us = t(lambda: _C.split_gemm_pair(a, b, bias, None, True, False, True, config=cfg))
    print("   no shortcut      config", cfg, "%.1f us" % us, "%.0f TF/s" % (6.0 * m * n * k / us / 1e6))
    us = t(lambda: _C.split_gemm_pair(a, b, None, None, False, True, False, config=cfg))
    print("   f32 out only     config", cfg, "%.1f us" % us, "%.0f TF/s" % (6.0 * m * n * k / us / 1e6))
