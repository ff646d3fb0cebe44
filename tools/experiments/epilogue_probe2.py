import sys, torch
sys.path.insert(0, "/root/repo")
from cvpr22_cross_modal_pseudo_labeling_amd import _C
torch.manual_seed(0)
m, n, k = 100352, 2048, 512
a = _C.split_pair(torch.randn(m, k, device="cuda"))
b = _C.split_pair(torch.randn(n, k, device="cuda") * 0.05)
bias = torch.randn(n, device="cuda")
def t(fn, it=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
cases = [
    ("pair, bias+relu", lambda: _C.split_gemm_pair(a, b, bias, None, True, False, True)),
    ("pair, no bias no relu", lambda: _C.split_gemm_pair(a, b, None, None, False, False, True)),
    ("f32, bias+relu", lambda: _C.split_gemm_pair(a, b, bias, None, True, True, False)),
    ("f32, plain", lambda: _C.split_gemm_pair(a, b, None, None, False, True, False)),
    ("f32+pair, bias+relu", lambda: _C.split_gemm_pair(a, b, bias, None, True, True, True)),
    ("pair, bias+relu, two-stage (config 2)", lambda: _C.split_gemm_pair(a, b, bias, None, True, False, True, config=2)),
    ("f32 plain two-stage (config 2)", lambda: _C.split_gemm_pair(a, b, None, None, False, True, False, config=2)),
    ("pair, bias only", lambda: _C.split_gemm_pair(a, b, bias, None, False, False, True)),
    ("pair, relu only", lambda: _C.split_gemm_pair(a, b, None, None, True, False, True)),
]
for rep in range(2):   # both orders: clocks drift with temperature / power over a run
    for name, fn in (cases if rep == 0 else cases[::-1]):
        us = t(fn)
        print(f"{name:40s} {us:7.1f} us  {6.0*m*n*k/us/1e6:5.0f} TF/s")
