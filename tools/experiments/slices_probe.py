"""K-slice count x LDS stages of the split GEMM on the step's sub-round / tail-paying grids (VERDICT r5 item 3 ii):
    python tools/experiments/slices_probe.py [--lib variant.so]
For each shape: the launcher's own choice (config 0) against forced slice counts (config bits 8..15) with one / two LDS stages
(bits 1 / 2), same box, 0.3 s of the same op before each timing.  Prints us and bf16 TFLOP/s issued (6 M N K)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if "--lib" in sys.argv:
    from cvpr22_cross_modal_pseudo_labeling_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from cvpr22_cross_modal_pseudo_labeling_amd import _C, _lib  # noqa: E402

_L = _lib.load()
SHAPES = [  # tag, images x h x w, N, channels, taps
    ("RPN head 3x3 9x1024 -> 1024", (2, 50, 84), 1024, 1024, 3),
    ("layer3 3x3 9x256 -> 256", (2, 50, 84), 256, 256, 3),
    ("layer3 1x1 1024 -> 256", (2, 50, 84), 256, 1024, 1),
    ("layer3 1x1 256 -> 1024", (2, 50, 84), 1024, 256, 1),
    ("layer3 first 1x1 512 -> 256 (100x167 rows)", (2, 100, 167), 256, 512, 1),
    ("layer2 3x3 9x128 -> 128", (2, 100, 167), 128, 128, 3),
    ("layer2 1x1 128 -> 512", (2, 100, 167), 512, 128, 1),
    ("layer2 1x1 512 -> 128", (2, 100, 167), 128, 512, 1),
    ("mask-head rows 1470 x 1024 -> 2048", (1, 1, 1470), 2048, 1024, 1),
    ("mask-head rows 2205 x 2048 -> 1024", (1, 1, 2205), 1024, 2048, 1),
    ("mask-head rows 2695 x 2048 -> 1024", (1, 1, 2695), 1024, 2048, 1),
    ("mask-head rows 1274 x 1024 -> 2048", (1, 1, 1274), 2048, 1024, 1),
]


def timeit(fn, iters=60, warm=0.3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters  # us


g = torch.Generator().manual_seed(3)
for tag, (nimg, h, w), n, ch, k in SHAPES:
    m = nimg * h * w
    a = _C.split_pair(torch.randn(m, ch, generator=g).cuda())
    b = _C.split_pair(torch.randn(n, k * k * ch, generator=g).cuda())
    c = torch.empty(m, n, device="cuda")
    ws = torch.empty(16 * m * n * 4, dtype=torch.uint8, device="cuda")
    ref, _ = _C.split_gemm_pair(a, b, conv=(h, w, k, k, False) if k > 1 else None)

    def run(cfg):
        rc = _L.ovis_split_gemm_pair(a.data_ptr(), 2 * a.stride(0), 0, 0, b.data_ptr(), 2 * b.stride(0), c.data_ptr(), n, 0, 4 * n,
                                     0, 0, 0, m, n, ch, 0, k, k, h if k > 1 else 0, w if k > 1 else 0, 0, 0, ws.data_ptr(), ws.numel(),
                                     cfg, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc

    fl = 6.0 * m * n * ch * k * k
    row = []
    for cfg_tag, cfg in [("auto", 0)] + [(f"s{sl}/{st}st", (sl << 8) | st) for sl in (1, 2, 3, 4, 6, 8) for st in (1, 2)]:
        if (ch * k * k) // 32 < (cfg >> 8) * 4:
            continue
        run(cfg)
        assert (c - ref).abs().max().item() <= 1e-3 * ref.abs().max().item(), (tag, cfg_tag)
        us = timeit(lambda: run(cfg))
        row.append(f"{cfg_tag} {us:.1f}us/{fl / us / 1e6:.0f}TF")
    print(f"{tag:46s} M={m:6d}: " + " | ".join(row), flush=True)
