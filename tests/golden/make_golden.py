"""Generates tests/golden/*.npz from the REFERENCE itself (run in the build container only).

Sources of truth:
  * oracle/_ref/ref_C.so   -- the reference's own CPU kernels (roi_align_forward, nms), built by
                              oracle/build_ref.py from /root/reference.
  * /root/reference Python -- imported with two sys.modules stubs (apex.amp.float_function =
                              identity, maskrcnn_benchmark._C = ref_C) exactly as SURVEY.md 8(c)
                              describes; used for sigmoid_focal_loss_cpu, smooth_l1_loss,
                              FrozenBatchNorm2d, the box / mask predictors and losses.
Only inputs and expected outputs are stored; no reference source text is copied.

    python tests/golden/make_golden.py          # rewrites the fixtures in place
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF = "/root/reference"


def import_reference():
    ref_c = oracle.ref_module()
    assert ref_c is not None, "run `python oracle/build_ref.py` first"
    apex = types.ModuleType("apex")
    amp = types.ModuleType("apex.amp")
    amp.float_function = lambda f: f
    apex.amp = amp
    sys.modules["apex"] = apex
    sys.modules["apex.amp"] = amp
    sys.path.insert(0, REF)
    import maskrcnn_benchmark  # noqa: F401

    sys.modules["maskrcnn_benchmark._C"] = ref_c
    maskrcnn_benchmark._C = ref_c
    return ref_c


def rand_rois(g, r, n_img, img_w, img_h, wmin, wmax):
    b = torch.randint(0, n_img, (r, 1), generator=g).float()
    x1 = torch.rand(r, 1, generator=g) * (img_w * 0.8)
    y1 = torch.rand(r, 1, generator=g) * (img_h * 0.8)
    w = torch.rand(r, 1, generator=g) * (wmax - wmin) + wmin
    h = torch.rand(r, 1, generator=g) * (wmax - wmin) + wmin
    return torch.cat([b, x1, y1, (x1 + w).clamp(max=img_w - 1), (y1 + h).clamp(max=img_h - 1)], 1)


def gen_roi_align(ref_c):
    g = torch.Generator().manual_seed(1234)
    n, c, h, w = 2, 8, 25, 42
    x = torch.randn(n, c, h, w, generator=g)
    rois = rand_rois(g, 56, n, 672, 400, 8, 330)
    edge = torch.tensor([
        [0, -40.0, -30.0, 20.0, 25.0],      # sticks out top-left
        [1, 600.0, 350.0, 800.0, 500.0],    # sticks out bottom-right
        [0, 100.0, 100.0, 100.5, 100.2],    # degenerate -> forced 1x1
        [1, 0.0, 0.0, 671.0, 399.0],        # whole image (gh=2, gw=3)
        [0, 900.0, 900.0, 950.0, 950.0],    # fully outside: every sample invalid -> zeros
        [1, -500.0, -500.0, 1500.0, 1200.0],  # far larger than the map (gh=8, gw=9)
        [0, 333.3, 111.1, 340.9, 390.7],    # thin and tall
        [1, 15.99, 16.01, 239.99, 240.01],  # cell-boundary coordinates
    ])
    rois = torch.cat([rois, edge], 0)
    out = {"input": x.numpy(), "rois": rois.numpy(), "scale": np.float32(1 / 16)}
    for sr in (0, 2):
        out[f"out_sr{sr}"] = ref_c.roi_align_forward(x, rois, 1 / 16, 14, 14, sr).numpy()
    out["out_7x7_sr0"] = ref_c.roi_align_forward(x, rois, 1 / 16, 7, 7, 0).numpy()
    np.savez_compressed(os.path.join(HERE, "roi_align_forward.npz"), **out)


def gen_nms(ref_c):
    g = torch.Generator().manual_seed(4321)
    out = {}
    for name, k, thr in (("rpn_like", 1500, 0.7), ("dense", 700, 0.5), ("tiny", 5, 0.5), ("one", 1, 0.3)):
        xy = torch.rand(k, 2, generator=g) * torch.tensor([600.0, 360.0])
        wh = torch.rand(k, 2, generator=g) * 200 + 8
        if name == "dense":  # heavy overlap: many suppressions per survivor
            xy = xy * 0.15 + 100
        boxes = torch.cat([xy, xy + wh], 1)
        scores = torch.rand(k, generator=g)
        assert scores.unique().numel() == k  # no ties: sort order is unambiguous
        keep = ref_c.nms(boxes, scores, thr)
        out[f"{name}_boxes"], out[f"{name}_scores"] = boxes.numpy(), scores.numpy()
        out[f"{name}_thr"], out[f"{name}_keep"] = np.float32(thr), keep.numpy()
    np.savez_compressed(os.path.join(HERE, "nms.npz"), **out)


def gen_focal():
    from maskrcnn_benchmark.layers.sigmoid_focal_loss import sigmoid_focal_loss_cpu

    g = torch.Generator().manual_seed(77)
    num, c = 257, 80
    logits = (torch.randn(num, c, generator=g) * 3).requires_grad_(True)
    targets = torch.randint(-1, c + 1, (num,), generator=g, dtype=torch.int32)
    d_losses = torch.rand(num, c, generator=g)
    # The reference's Python formula takes log(1 - sigmoid(x)) literally, which loses precision in
    # fp32 for x >~ 8; its CUDA kernel uses the stable form.  The fixture therefore stores the
    # reference formula evaluated in fp64 (fp32 inputs promoted), which both agree with.
    logits = logits.detach().double().requires_grad_(True)
    d_losses = d_losses.double()
    loss = sigmoid_focal_loss_cpu(logits, targets, 2.0, 0.25)
    (grad,) = torch.autograd.grad(loss, logits, d_losses)
    # second case: non-default gamma/alpha and a class count that is not a multiple of 4
    logits2 = (torch.randn(64, 7, generator=g) * 2).requires_grad_(True)
    targets2 = torch.randint(-1, 8, (64,), generator=g, dtype=torch.int32)
    d2 = torch.rand(64, 7, generator=g)
    logits2 = logits2.detach().double().requires_grad_(True)
    d2 = d2.double()
    loss2 = sigmoid_focal_loss_cpu(logits2, targets2, 1.5, 0.4)
    (grad2,) = torch.autograd.grad(loss2, logits2, d2)
    np.savez_compressed(os.path.join(HERE, "sigmoid_focal_loss.npz"),
                        logits=logits.detach().float().numpy(), targets=targets.numpy(),
                        d_losses=d_losses.float().numpy(), loss=loss.detach().numpy(), grad=grad.numpy(),
                        logits2=logits2.detach().float().numpy(), targets2=targets2.numpy(),
                        d_losses2=d2.float().numpy(), loss2=loss2.detach().numpy(), grad2=grad2.numpy())


def main():
    torch.set_num_threads(1)
    ref_c = import_reference()
    gen_roi_align(ref_c)
    gen_nms(ref_c)
    gen_focal()
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
