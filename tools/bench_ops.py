"""Per-op micro-benchmarks on the shapes of SURVEY.md 8(d) / BASELINE.md 3: achieved GB/s (or
TFLOP/s) of each hand-written kernel against its algorithmic byte / flop count.

    python tools/bench_ops.py [--ops roi_fwd,roi_bwd,nms,focal] [--iters 20]
Prints one JSON object per op.  Timing: HIP events on torch's current stream (the stream the
kernels are launched on), `iters` launches after 3 warm-ups.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:  # an experiment build (tools/experiments/*_variants.sh) instead of the regular library
    from cvpr22_cross_modal_pseudo_labeling_amd import _lib  # noqa: E402
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from cvpr22_cross_modal_pseudo_labeling_amd import _C  # noqa: E402

HBM_PEAK_GBS = 8000.0


def timeit(fn, iters, warmup=3, warm_seconds=0.3):
    """Mean ms per call over `iters` launches after the GPU has been kept busy with the same op for `warm_seconds` (an idle part
    needs a few hundred ms of load to reach its sustained clock: 20 launches from idle read 4-20 % slow, box dependent)."""
    import time
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_seconds:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters  # ms


def bench_rois(r, n_img, g, kind="uniform"):
    b = torch.randint(0, n_img, (r, 1), generator=g).float()
    if kind == "uniform":  # x1~U[0,1066], y1~U[0,640], w,h~U[16,316], clipped
        x1 = torch.rand(r, 1, generator=g) * 1066
        y1 = torch.rand(r, 1, generator=g) * 640
        w = torch.rand(r, 1, generator=g) * 300 + 16
        h = torch.rand(r, 1, generator=g) * 300 + 16
    else:  # RPN-like: log-uniform areas 32^2..512^2, ratios {.5,1,2}
        area = torch.exp(torch.rand(r, 1, generator=g) * (2 * torch.log(torch.tensor(512.0 / 32))) + 2 * torch.log(torch.tensor(32.0)))
        ratio = torch.tensor([0.5, 1.0, 2.0])[torch.randint(0, 3, (r, 1), generator=g)]
        w, h = torch.sqrt(area / ratio), torch.sqrt(area * ratio)
        x1 = torch.rand(r, 1, generator=g) * (1333 - w).clamp(min=1)
        y1 = torch.rand(r, 1, generator=g) * (800 - h).clamp(min=1)
    return torch.cat([b, x1, y1, (x1 + w).clamp(max=1332), (y1 + h).clamp(max=799)], 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="roi_fwd,roi_bwd,roi_bwd_strided,nms,focal,gemm,dcn,res5,split_gemm")
    ap.add_argument("--lib", default="", help="path of an experiment build of libovis_hip.so (handled before the import)")
    ap.add_argument("--roi-kinds", default="uniform,rpn_like", help="RoI distributions of the RoIAlign micro-benchmarks")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    ops = args.ops.split(",")
    g = torch.Generator().manual_seed(1234)
    dev = "cuda"
    res = []
    if "roi_fwd" in ops or "roi_bwd" in ops or "roi_bwd_strided" in ops:
        n, c, h, w, r = 2, 1024, 50, 84, 1024
        x = torch.randn(n, c, h, w, generator=g).to(dev)
        for kind in args.roi_kinds.split(","):
            rois = bench_rois(r, n, g, kind).to(dev)
            alg = 4 * r * c * 196 + 4 * n * c * h * w + 20 * r
            if "roi_fwd" in ops:
                ms = timeit(lambda: _C.roi_align_forward_mfma(x, rois, 1 / 16, 14, 14, 0), args.iters)
                res.append({"op": "roi_align_forward_mfma", "rois": kind, "ms": ms, "alg_MB": alg / 1e6,
                            "GBps": alg / ms / 1e6, "frac_hbm": alg / ms / 1e6 / HBM_PEAK_GBS})
                ms = timeit(lambda: _C.roi_align_forward(x, rois, 1 / 16, 14, 14, 0), args.iters)
                res.append({"op": "roi_align_forward", "rois": kind, "ms": ms, "alg_MB": alg / 1e6,
                            "GBps": alg / ms / 1e6, "frac_hbm": alg / ms / 1e6 / HBM_PEAK_GBS})
            if "roi_fwd" in ops:
                ms = timeit(lambda: _C.roi_align_forward_strided_nhwc(x, rois, 1 / 16, 14, 14, 0, 2), args.iters)
                alg_s = 4 * r * c * 49 + 4 * n * c * h * w + 20 * r
                res.append({"op": "roi_align_forward_strided_nhwc(s=2)", "rois": kind, "ms": ms, "alg_MB": alg_s / 1e6,
                            "GBps": alg_s / ms / 1e6, "frac_hbm": alg_s / ms / 1e6 / HBM_PEAK_GBS})
                x_cl = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)  # channels-last map: pooled in place
                ms = timeit(lambda: _C.roi_align_forward_strided_pair(x_cl, rois, 1 / 16, 14, 14, 0, 2), args.iters)
                res.append({"op": "roi_align_forward_strided_pair(s=2, channels-last map)", "rois": kind, "ms": ms,
                            "alg_MB": alg_s / 1e6, "GBps": alg_s / ms / 1e6, "frac_hbm": alg_s / ms / 1e6 / HBM_PEAK_GBS})
                del x_cl
            if "roi_bwd" in ops:
                go = torch.randn(r, c, 14, 14, generator=g).to(dev)
                ms = timeit(lambda: _C.roi_align_backward(go, rois, 1 / 16, 14, 14, n, c, h, w, 0), args.iters)
                res.append({"op": "roi_align_backward", "rois": kind, "ms": ms, "alg_MB": alg / 1e6,
                            "GBps": alg / ms / 1e6, "frac_hbm": alg / ms / 1e6 / HBM_PEAK_GBS})
                del go
            if "roi_bwd_strided" in ops:
                # the teacher step's form: gradient of the 7 x 7 computed bins, NHWC as the res5 data-gradient GEMM leaves it
                gs = torch.randn(r, 7, 7, c, generator=g).to(dev)
                alg_s = 4 * r * c * 49 + 4 * n * c * h * w + 20 * r
                ms = timeit(lambda: _C.roi_align_backward_strided(gs.permute(0, 3, 1, 2).contiguous(), rois, 1 / 16, 14, 14, n, c, h, w, 0, 2), args.iters)
                res.append({"op": "roi_align_backward_strided(s=2) after permute+contiguous", "rois": kind, "ms": ms,
                            "alg_MB": alg_s / 1e6, "GBps": alg_s / ms / 1e6, "frac_hbm": alg_s / ms / 1e6 / HBM_PEAK_GBS})
                if hasattr(_C, "roi_align_backward_strided_nhwc"):
                    ms = timeit(lambda: _C.roi_align_backward_strided_nhwc(gs, rois, 1 / 16, 14, 14, n, c, h, w, 0, 2), args.iters)
                    res.append({"op": "roi_align_backward_strided_nhwc(s=2, pre-split tiles)", "rois": kind, "ms": ms,
                                "alg_MB": alg_s / 1e6, "GBps": alg_s / ms / 1e6, "frac_hbm": alg_s / ms / 1e6 / HBM_PEAK_GBS})
                del gs
        del x
    if "res5" in ops:  # byte kernels of the NHWC res5 head at the student pass's shapes (R = 1024 -> M = 50176 rows)
        m = 1024 * 49
        for cols in (512, 1024, 2048):
            xm = torch.randn(m, cols, generator=g).to(dev)
            ms = timeit(lambda: _C.split_bf16x3(xm, 0), args.iters)
            res.append({"op": "split_bf16x3", "shape": f"{m}x{cols}", "ms": ms, "alg_MB": 10 * xm.numel() / 1e6,
                        "GBps": 10 * xm.numel() / ms / 1e6, "frac_hbm": 10 * xm.numel() / ms / 1e6 / HBM_PEAK_GBS})
            ym = torch.randn(m, cols, generator=g).to(dev)
            bm = torch.randn(cols, generator=g).to(dev)
            ms = timeit(lambda: _C.bias_act_(ym, bm, xm, True), args.iters)
            res.append({"op": "bias_act(+shortcut)", "shape": f"{m}x{cols}", "ms": ms, "alg_MB": 12 * xm.numel() / 1e6,
                        "GBps": 12 * xm.numel() / ms / 1e6, "frac_hbm": 12 * xm.numel() / ms / 1e6 / HBM_PEAK_GBS})
            del xm, ym
        xi = torch.randn(1024, 7, 7, 512, generator=g).to(dev)
        ms = timeit(lambda: _C.im2col_split_bf16x3(xi, 3, 3), args.iters)
        res.append({"op": "im2col_split_bf16x3", "shape": "[1024,7,7,512] 3x3", "ms": ms, "alg_MB": 58 * xi.numel() / 1e6,
                    "GBps": 58 * xi.numel() / ms / 1e6, "frac_hbm": 58 * xi.numel() / ms / 1e6 / HBM_PEAK_GBS})
        del xi
    if "split_gemm" in ops:  # pair-layout split GEMMs of the res5 head at R = 1024 (M = 50176 rows): forward / dX, 3x3, dW
        m = 1024 * 49
        peak = 2500.0  # dense bf16 TFLOP/s; 3 bf16 products per fp32-accurate product
        for tag, k, n in (("conv1 b0", 1024, 512), ("shortcut", 1024, 2048), ("conv3", 512, 2048), ("conv1 b1", 2048, 512)):
            ap = _C.split_pair(torch.randn(m, k, generator=g).to(dev))
            bp = _C.split_pair(torch.randn(n, k, generator=g).to(dev))
            ms = timeit(lambda: _C.split_gemm_pair(ap, bp), args.iters)
            fl = 6.0 * m * n * k
            res.append({"op": "split_gemm_pair", "shape": f"{tag} [{m}x{k}]x[{n}x{k}]^T", "ms": ms, "TFLOPs_bf16": fl / ms / 1e9,
                        "TFLOPs_fp32_equiv": fl / 3 / ms / 1e9, "frac_mfma": fl / ms / 1e9 / peak})
            gp = _C.split_pair(torch.randn(m, n, generator=g).to(dev))
            ms = timeit(lambda: _C.split_gemm_pair_tn(gp, ap), args.iters)
            res.append({"op": "split_gemm_pair_tn", "shape": f"{tag} dW [{m}x{n}]^T[{m}x{k}]", "ms": ms,
                        "TFLOPs_bf16": fl / ms / 1e9, "TFLOPs_fp32_equiv": fl / 3 / ms / 1e9, "frac_mfma": fl / ms / 1e9 / peak})
            del ap, bp, gp
        xp = _C.split_pair(torch.randn(m, 512, generator=g).to(dev))
        wp = _C.split_pair(torch.randn(512, 9 * 512, generator=g).to(dev))
        gp = _C.split_pair(torch.randn(m, 512, generator=g).to(dev))
        fl = 6.0 * m * 512 * 9 * 512
        ms = timeit(lambda: _C.split_gemm_pair(xp, wp, conv=(7, 7, 3, 3, False)), args.iters)
        res.append({"op": "split_gemm_pair(3x3 implicit)", "shape": "[1024,7,7,512]->512", "ms": ms, "TFLOPs_bf16": fl / ms / 1e9,
                    "TFLOPs_fp32_equiv": fl / 3 / ms / 1e9, "frac_mfma": fl / ms / 1e9 / peak})
        ms = timeit(lambda: _C.split_gemm_pair_tn(gp, xp, (7, 7, 3, 3)), args.iters)
        res.append({"op": "split_gemm_pair_tn(3x3 dW)", "shape": "[1024,7,7,512]->512", "ms": ms, "TFLOPs_bf16": fl / ms / 1e9,
                    "TFLOPs_fp32_equiv": fl / 3 / ms / 1e9, "frac_mfma": fl / ms / 1e9 / peak})
        for cols in (512, 2048):
            xm = torch.randn(m, cols, generator=g).to(dev)
            ms = timeit(lambda: _C.split_pair(xm), args.iters)
            res.append({"op": "split_pair", "shape": f"{m}x{cols}", "ms": ms, "alg_MB": 8 * xm.numel() / 1e6,
                        "GBps": 8 * xm.numel() / ms / 1e6, "frac_hbm": 8 * xm.numel() / ms / 1e6 / HBM_PEAK_GBS})
            ym = torch.randn(m, cols, generator=g).to(dev)
            ms = timeit(lambda: _C.gate_split_pair(xm, ym, want_f32=True), args.iters)
            res.append({"op": "gate_split_pair(f32 gate, +f32 out)", "shape": f"{m}x{cols}", "ms": ms, "alg_MB": 16 * xm.numel() / 1e6,
                        "GBps": 16 * xm.numel() / ms / 1e6, "frac_hbm": 16 * xm.numel() / ms / 1e6 / HBM_PEAK_GBS})
            del xm, ym
        img = torch.randn(2, 3, 800, 1333, generator=g).to(dev)
        ms = timeit(lambda: _C.im2col_nchw_pair(img, 7, 7, 2, 3), args.iters)
        rows = 2 * 400 * 667
        res.append({"op": "im2col_nchw_pair(stem 7x7/2)", "shape": "[2,3,800,1333]", "ms": ms, "alg_MB": (img.numel() * 4 + rows * 640) / 1e6,
                    "GBps": (img.numel() * 4 + rows * 640) / ms / 1e6, "frac_hbm": (img.numel() * 4 + rows * 640) / ms / 1e6 / HBM_PEAK_GBS})
    if "nms" in ops:
        for k in (6000, 12000):
            xy = torch.rand(k, 2, generator=g) * torch.tensor([1200.0, 720.0])
            wh = torch.rand(k, 2, generator=g) * 200 + 8
            boxes = torch.cat([xy, xy + wh], 1).to(dev)
            scores = torch.rand(k, generator=g).to(dev)
            ms = timeit(lambda: _C.nms_padded(boxes, scores, 0.7), args.iters)
            nb = (k + 63) // 64
            alg = 20 * k + 2 * 8 * k * nb + 8 * k
            keep, num = _C.nms_padded(boxes, scores, 0.7)
            res.append({"op": "nms", "K": k, "kept": int(num), "ms": ms, "alg_MB": alg / 1e6, "GBps": alg / ms / 1e6,
                        "MIoU_per_s": k * (k - 1) / 2 / ms / 1e3})
    if "topk" in ops:  # the RPN's sorted top-k: 12000 of 50 * 84 * 15 = 63000 scores per image, two images
        sc = torch.rand(2, 63000, generator=g).to(dev)
        ms = timeit(lambda: _C.topk_sorted(sc, 12000), args.iters)
        ms_t = timeit(lambda: sc.topk(12000, dim=1, sorted=True), args.iters)
        res.append({"op": "topk_sorted", "shape": "[2,63000] k=12000", "ms": ms, "torch_topk_ms": ms_t})
    if "focal" in ops:
        m, c = 400000, 80
        logits = (torch.randn(m, c, generator=g) * 2).to(dev)
        targets = torch.where(torch.rand(m, generator=g) < 0.01, torch.randint(1, c + 1, (m,), generator=g),
                              torch.zeros(m, dtype=torch.int64)).int().to(dev)
        d = torch.rand(m, c, generator=g).to(dev)
        ms = timeit(lambda: _C.sigmoid_focalloss_forward(logits, targets, c, 2.0, 0.25), args.iters)
        alg = 8 * m * c + 4 * m
        res.append({"op": "sigmoid_focal_forward", "ms": ms, "alg_MB": alg / 1e6, "GBps": alg / ms / 1e6,
                    "frac_hbm": alg / ms / 1e6 / HBM_PEAK_GBS})
        ms = timeit(lambda: _C.sigmoid_focalloss_backward(logits, targets, d, c, 2.0, 0.25), args.iters)
        alg = 12 * m * c + 4 * m
        res.append({"op": "sigmoid_focal_backward", "ms": ms, "alg_MB": alg / 1e6, "GBps": alg / ms / 1e6,
                    "frac_hbm": alg / ms / 1e6 / HBM_PEAK_GBS})
    if "gemm" in ops:
        for (m, n, k, what) in ((1024, 776, 2048, "emb_pred+bbox_pred fwd"), (1024, 1203, 768, "text logits LVIS"),
                                (2000, 776, 2048, "teacher pass"), (1024, 49, 768, "text logits COCO")):
            a, b = torch.randn(m, k, generator=g).to(dev), torch.randn(n, k, generator=g).to(dev)
            ms = timeit(lambda: _C.gemm_nt(a, b), args.iters)
            res.append({"op": "gemm_f32_mfma", "shape": f"{m}x{n}x{k}", "what": what, "ms": ms,
                        "TFLOPs": 2.0 * m * n * k / ms / 1e9, "frac_fp32_mfma_peak": 2.0 * m * n * k / ms / 1e9 / 157.3})
            # the route the cross-modal head takes (layers/cross_modal.py::_LinearPair): operand split + pair-layout split
            # GEMM on the bf16 matrix cores; "gemm only" = operands already in pair layout (cached class matrix)
            from cvpr22_cross_modal_pseudo_labeling_amd.layers import linear_mfma
            with torch.no_grad():
                ms = timeit(lambda: linear_mfma(a, b), args.iters)
                n_pad = -(-n // 128) * 128
                ap = _C.split_pair(a)
                bp = _C.split_pair(torch.nn.functional.pad(b, (0, 0, 0, n_pad - n)))
                ms_g = timeit(lambda: _C.split_gemm_pair(ap, bp), args.iters)
            res.append({"op": "linear_pair(split GEMM)", "shape": f"{m}x{n}x{k}", "what": what, "ms": ms, "ms_gemm_only": ms_g,
                        "TFLOPs_fp32_equiv": 2.0 * m * n * k / ms / 1e9, "TFLOPs_fp32_equiv_gemm_only": 2.0 * m * n * k / ms_g / 1e9,
                        "TFLOPs_bf16_issued_gemm_only": 6.0 * m * n_pad * k / ms_g / 1e9})
    if "dcn" in ops:
        from cvpr22_cross_modal_pseudo_labeling_amd.layers import deform_conv
        x = torch.randn(2, 512, 100, 168, generator=g).to(dev)
        w = (torch.randn(512, 512, 3, 3, generator=g) * 0.02).to(dev)
        off = (torch.randn(2, 18, 100, 168, generator=g) * 2).to(dev)
        ms = timeit(lambda: deform_conv(x, off, w, 1, 1, 1, 1, 1, 2), max(3, args.iters // 4))
        fl = 2.0 * 512 * 512 * 9 * 2 * 100 * 168
        from cvpr22_cross_modal_pseudo_labeling_amd import _C as _ops
        _ops.dcn_implicit = False
        ms_col = timeit(lambda: deform_conv(x, off, w, 1, 1, 1, 1, 1, 2), max(3, args.iters // 4))
        _ops.dcn_implicit = True
        res.append({"op": "deform_conv_forward (column route: im2col + fp32 GEMM)", "shape": "[2,512,100,168] 3x3 -> 512",
                    "ms": ms_col, "TFLOPs": fl / ms_col / 1e9})
        res.append({"op": "deform_conv_forward", "shape": "[2,512,100,168] 3x3 -> 512", "ms": ms, "TFLOPs": fl / ms / 1e9,
                    "frac_fp32_mfma_peak": fl / ms / 1e9 / 157.3, "col_MB": 4 * 512 * 9 * 2 * 100 * 168 / 1e6})
    if "dcn" in ops:
        # backward (dX, dOffset, dW) of the same layer: rows route (split GEMMs, no reference-layout column buffer) vs the
        # column route (fp32-MFMA GEMMs around im2col / col2im / col2im_coord)
        from cvpr22_cross_modal_pseudo_labeling_amd import _C as _ops
        xg, og, wg = x.clone().requires_grad_(True), off.clone().requires_grad_(True), w.clone().requires_grad_(True)
        gy = torch.randn(2, 512, 100, 168, generator=g).to(dev)

        def fwd_bwd():
            for t in (xg, og, wg):
                t.grad = None
            deform_conv(xg, og, wg, 1, 1, 1, 1, 1, 2).backward(gy)

        ms_fb = timeit(fwd_bwd, max(3, args.iters // 4))
        _ops.dcn_implicit = False
        ms_fb_col = timeit(fwd_bwd, max(3, args.iters // 4))
        _ops.dcn_implicit = True
        res.append({"op": "deform_conv forward+backward (column route)", "shape": "[2,512,100,168] 3x3 -> 512", "ms": ms_fb_col,
                    "backward_ms": ms_fb_col - ms_col})
        res.append({"op": "deform_conv forward+backward (rows route)", "shape": "[2,512,100,168] 3x3 -> 512", "ms": ms_fb,
                    "backward_ms": ms_fb - ms, "backward_over_forward": (ms_fb - ms) / ms})
    for r_ in res:
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r_.items()}))


if __name__ == "__main__":
    main()
