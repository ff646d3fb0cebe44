"""CPU: the oracle (oracle/ovis_oracle.c) against the golden vectors produced by the reference
itself (tests/golden/make_golden.py) and against analytic pins where the reference has no
runnable implementation."""
import os

import numpy as np
import pytest
import torch


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_roi_align_forward_bit_exact_vs_reference(oracle_mod, golden_dir):
    z = _load(golden_dir, "roi_align_forward.npz")
    x, rois, scale = torch.from_numpy(z["input"]), torch.from_numpy(z["rois"]), float(z["scale"])
    for key, (ph, pw, sr) in {"out_sr0": (14, 14, 0), "out_sr2": (14, 14, 2), "out_7x7_sr0": (7, 7, 0)}.items():
        got = oracle_mod.roi_align_forward(x, rois, scale, ph, pw, sr)
        assert torch.equal(got, torch.from_numpy(z[key])), key


def test_roi_align_backward_is_adjoint_of_forward_fp64(oracle_mod, golden_dir):
    # The reference has no CPU RoIAlign backward (csrc/ROIAlign.h:44): pin <fwd(x), g> == <x, bwd(g)>.
    z = _load(golden_dir, "roi_align_forward.npz")
    rois, scale = torch.from_numpy(z["rois"]), float(z["scale"])
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8, 25, 42, generator=g, dtype=torch.float64)
    for sr in (0, 2):
        go = torch.randn(rois.shape[0], 8, 14, 14, generator=g, dtype=torch.float64)
        f = oracle_mod.roi_align_forward(x, rois, scale, 14, 14, sr, dtype=torch.float64)
        b = oracle_mod.roi_align_backward(go, rois, scale, 14, 14, 2, 8, 25, 42, sr, dtype=torch.float64)
        lhs, rhs = (f * go).sum().item(), (x * b).sum().item()
        assert abs(lhs - rhs) <= 1e-10 * max(1.0, abs(lhs))
    # the f32 restatement satisfies the same identity with ITS OWN f32 weights (round-off only) ...
    go = torch.randn(rois.shape[0], 8, 14, 14, generator=g)
    x32 = x.float()
    f32 = oracle_mod.roi_align_forward(x32, rois, scale, 14, 14, 0)
    b32 = oracle_mod.roi_align_backward(go, rois, scale, 14, 14, 2, 8, 25, 42, 0)
    lhs, rhs = (f32.double() * go.double()).sum().item(), (x32.double() * b32.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * (f32.double() * go.double()).abs().sum().item()
    # ... and agrees with the f64 one up to the f32 rounding of the sample coordinates
    # (weights move by ~eps*coordinate, i.e. ~1e-5 absolute).
    b64 = oracle_mod.roi_align_backward(go.double(), rois, scale, 14, 14, 2, 8, 25, 42, 0, dtype=torch.float64)
    assert torch.allclose(b32.double(), b64, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("name", ["rpn_like", "dense", "tiny", "one"])
def test_nms_exact_vs_reference(oracle_mod, golden_dir, name):
    z = _load(golden_dir, "nms.npz")
    boxes, scores = torch.from_numpy(z[f"{name}_boxes"]), torch.from_numpy(z[f"{name}_scores"])
    want = torch.from_numpy(z[f"{name}_keep"])
    thr = float(z[f"{name}_thr"])
    # reference CPU kernel suppresses on >= (cpu/nms_cpu.cpp:60); CUDA on > (cuda/nms.cu:60):
    # identical unless some IoU equals thr exactly -- the fixtures have no such pair.
    assert torch.equal(oracle_mod.nms(boxes, scores, thr, ge_mode=True), want)
    assert torch.equal(oracle_mod.nms(boxes, scores, thr, ge_mode=False), want)


def test_nms_comparison_modes_differ_only_on_exact_ties(oracle_mod):
    # two boxes with IoU exactly 0.5: kept by `>` (CUDA), suppressed by `>=` (CPU)
    boxes = torch.tensor([[0.0, 0.0, 9.0, 9.0], [0.0, 0.0, 9.0, 4.0]])
    scores = torch.tensor([0.9, 0.8])
    assert oracle_mod.nms(boxes, scores, 0.5, ge_mode=False).tolist() == [0, 1]
    assert oracle_mod.nms(boxes, scores, 0.5, ge_mode=True).tolist() == [0]
    assert oracle_mod.nms(torch.zeros(0, 4), torch.zeros(0), 0.5).numel() == 0


def test_focal_vs_reference_python_formula(oracle_mod, golden_dir):
    z = _load(golden_dir, "sigmoid_focal_loss.npz")
    for sfx, gamma, alpha in (("", 2.0, 0.25), ("2", 1.5, 0.4)):
        logits, targets = torch.from_numpy(z["logits" + sfx]), torch.from_numpy(z["targets" + sfx])
        d = torch.from_numpy(z["d_losses" + sfx])
        loss = oracle_mod.sigmoid_focal_loss_forward(logits, targets, gamma, alpha)
        grad = oracle_mod.sigmoid_focal_loss_backward(logits, targets, d, gamma, alpha)
        # expected = the reference's Python formula evaluated in fp64 (see make_golden.py)
        assert torch.allclose(loss.double(), torch.from_numpy(z["loss" + sfx]), rtol=2e-5, atol=1e-7)
        assert torch.allclose(grad.double(), torch.from_numpy(z["grad" + sfx]), rtol=2e-5, atol=1e-7)


def _roi_pool_tensor_formulation(x, rois, scale, PH, PW):
    """Independent restatement of ROIPool_cuda.cu:17-77 with numpy float32 scalars + torch slicing (no shared code with
    the C oracle): the reference has neither a CPU kernel nor tests for this op."""
    import math
    R = rois.shape[0]
    N, C, H, W = x.shape
    out = torch.zeros(R, C, PH, PW)
    arg = torch.full((R, C, PH, PW), -1, dtype=torch.int32)

    def cuda_round(v):  # round half away from zero
        return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)

    for n in range(R):
        b = int(rois[n, 0])
        sw, sh, ew, eh = (cuda_round(float(np.float32(rois[n, k]) * np.float32(scale))) for k in (1, 2, 3, 4))
        rw, rh = max(ew - sw + 1, 1), max(eh - sh + 1, 1)
        bh, bw = np.float32(rh) / np.float32(PH), np.float32(rw) / np.float32(PW)
        for ph in range(PH):
            for pw in range(PW):
                hs, he = int(np.floor(np.float32(ph) * bh)), int(np.ceil(np.float32(ph + 1) * bh))
                ws, we = int(np.floor(np.float32(pw) * bw)), int(np.ceil(np.float32(pw + 1) * bw))
                hs, he = min(max(hs + sh, 0), H), min(max(he + sh, 0), H)
                ws, we = min(max(ws + sw, 0), W), min(max(we + sw, 0), W)
                if he <= hs or we <= ws:
                    continue
                reg = x[b, :, hs:he, ws:we].reshape(C, -1)
                v, i = reg.max(1)
                # first maximum in scan order
                i = (reg == v[:, None]).float().argmax(1)
                out[n, :, ph, pw] = v
                arg[n, :, ph, pw] = ((hs + i // (we - ws)) * W + ws + i % (we - ws)).int()
    return out, arg


def test_roi_pool_oracle_vs_independent_formulation(oracle_mod):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 10, 12, generator=g)
    x[0, 0, 2:4, 2:5] = 7.0  # ties: the first maximum in (h, w) order wins
    rois = torch.tensor([[0, 0.0, 0.0, 40.0, 30.0], [1, 8.0, 4.0, 100.0, 80.0], [0, -10.0, -5.0, 5.0, 6.0],
                         [1, 200.0, 200.0, 210.0, 220.0], [0, 10.0, 10.0, 9.0, 9.0], [1, 2.0, 2.0, 2.5, 2.5]])
    for (ph, pw) in ((3, 3), (7, 7), (2, 5)):
        out, arg = oracle_mod.roi_pool_forward(x, rois, 0.25, ph, pw)
        o2, a2 = _roi_pool_tensor_formulation(x, rois, 0.25, ph, pw)
        assert torch.equal(out, o2) and torch.equal(arg, a2)
        gout = torch.randn(out.shape, generator=g)
        gin = oracle_mod.roi_pool_backward(gout, arg, rois, 2, 3, 10, 12)
        # adjoint identity of the (piecewise linear) max-pool selection: <out, g> == <x, gin> for the selected elements
        assert abs(float((out * gout)[arg >= 0].sum()) - float((x * gin).sum())) < 1e-4
