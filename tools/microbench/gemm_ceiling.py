"""A practical ceiling for the split GEMM's roofline fraction: for every large product shape of the
step, the library's tuned bf16 GEMM (hipBLASLt through ``torch.matmul``, MEASUREMENT ONLY -- never on the product path) next to
the split GEMM, on the same box, with the same clock protocol (the op keeps the GPU busy for 0.3 s before it is timed).

The split GEMM issues THREE bf16 products per fp32-accurate product (hi*hi, hi*lo, lo*hi), so its issued bf16 flops are
6*M*N*K; the library runs ONE product (2*M*N*K).  Both rates are bf16 TFLOP/s issued on the matrix pipe, i.e. directly
comparable: what fraction of the silicon's tuned-kernel rate under the power limit does the hand-written kernel reach, rather
than what fraction of the 2.5 PFLOP/s spec figure.  The library figure reads/writes bf16 (2 bytes per element); the split
GEMM reads hi+lo pairs (4 bytes) and writes fp32 and/or pairs, so the short-K shapes also differ in bytes per flop.

    python tools/microbench/gemm_ceiling.py [--out profiles/r5_gemm_ceiling.txt]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# (tag, M, N, K, taps) of the student / teacher steps' large products (profiles/r4_split_gemm_per_shape_*.csv)
SHAPES = [
    ("res5 3x3 (implicit, 9 taps x 512)", 100352, 512, 4608, 9),
    ("res5 conv1  K=2048 -> N=512", 100352, 512, 2048, 1),
    ("res5 conv3  K=512 -> N=2048", 100352, 2048, 512, 1),
    ("res5 conv3 + shortcut K=1536 -> N=2048", 100352, 2048, 1536, 1),
    ("res5 conv1  K=1024 -> N=512", 100352, 512, 1024, 1),
    ("dX conv3  K=2048 -> N=512", 100352, 512, 2048, 1),
    ("trunk layer3 3x3 (9 x 256)", 8400, 256, 2304, 9),
    ("trunk layer3 K=1024 -> N=256", 8400, 256, 1024, 1),
    ("trunk layer3 K=256 -> N=1024", 8400, 1024, 256, 1),
    ("RPN head 3x3 (9 x 1024)", 8400, 1024, 9216, 9),
    ("trunk layer2 3x3 (9 x 128)", 33400, 128, 1152, 9),
    ("trunk layer1 3x3 (9 x 64)", 133600, 64, 576, 9),
]


def timeit(fn, iters=50, warm_seconds=0.3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_seconds:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters  # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    lines = ["shape | M x N x K | hipBLASLt bf16: us, TFLOP/s | split GEMM (3 products, fp32 out): us, TFLOP/s issued | "
             "split / library | split vs 2500 spec | library vs 2500 spec"]
    for tag, m, n, k, taps in SHAPES:
        a = torch.randn(m, k, device=dev, generator=g).bfloat16()
        b = torch.randn(n, k, device=dev, generator=g).bfloat16()
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        bt = b.t()
        ms_lib = timeit(lambda: torch.matmul(a, bt, out=out), args.iters)
        lib = 2.0 * m * n * k / ms_lib / 1e9
        af = torch.randn(m, k // taps if taps > 1 else k, device=dev, generator=g)
        bf = torch.randn(n, k, device=dev, generator=g) * 0.05
        ap_, bp_ = _C.split_pair(af), _C.split_pair(bf)
        conv = None
        if taps > 1:  # implicit 3x3: M = images * h * w rows of an NHWC map
            h, w = {100352: (7, 7), 8400: (50, 84), 33400: (100, 167), 133600: (200, 334)}[m]
            conv = (h, w, 3, 3, False)
        ms_split = timeit(lambda: _C.split_gemm_pair(ap_, bp_, conv=conv), args.iters)
        split = 6.0 * m * n * k / ms_split / 1e9
        lines.append(f"{tag} | {m} x {n} x {k} | {1e3 * ms_lib:.1f} us, {lib:.0f} | {1e3 * ms_split:.1f} us, {split:.0f} | "
                     f"{split / lib:.2f} | {split / 2500:.3f} | {lib / 2500:.3f}")
        del a, b, out, af, bf, ap_, bp_
    # weight gradients: dW[N, K] = G[M, N]^T X[M, K], contraction over the M rows (split_gemm_tn_kernel)
    lines.append("")
    lines.append("weight gradients (contraction over M rows) | M x N x K | hipBLASLt bf16 (G^T X): us, TFLOP/s | split TN: us, TFLOP/s issued | split / library")
    for tag, m, n, k, conv in [("res5 3x3 dW", 100352, 512, 4608, (7, 7, 3, 3)), ("res5 conv3 dW", 100352, 2048, 512, None),
                               ("res5 conv1 dW", 100352, 512, 2048, None), ("res5 conv3 + shortcut dW", 100352, 2048, 1024, None)]:
        gm = torch.randn(m, n, device=dev, generator=g).bfloat16()
        xm = torch.randn(m, k, device=dev, generator=g).bfloat16()
        out = torch.empty(n, k, device=dev, dtype=torch.bfloat16)
        gt = gm.t()
        ms_lib = timeit(lambda: torch.matmul(gt, xm, out=out), args.iters)
        lib = 2.0 * m * n * k / ms_lib / 1e9
        gp = _C.split_pair(torch.randn(m, n, device=dev, generator=g))
        xp = _C.split_pair(torch.randn(m, k // 9 if conv else k, device=dev, generator=g))
        ms_split = timeit((lambda: _C.split_gemm_pair_tn(gp, xp, conv)) if conv else (lambda: _C.split_gemm_pair_tn(gp, xp)), args.iters)
        split = 6.0 * m * n * k / ms_split / 1e9
        lines.append(f"{tag} | {m} x {n} x {k} | {1e3 * ms_lib:.1f} us, {lib:.0f} | {1e3 * ms_split:.1f} us, {split:.0f} | {split / lib:.2f}")
        del gm, xm, out, gp, xp
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(__doc__.split("\n\n")[0] + "\n\n" + text + "\n")


if __name__ == "__main__":
    main()
