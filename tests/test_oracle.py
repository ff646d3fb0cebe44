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
