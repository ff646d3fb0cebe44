"""GPU: the RPN proposal chain as device kernels (csrc/rpn.hip decode + clip + small-box flag, csrc/nms.hip batched
score-sorted NMS with drop flags) against (a) the fixture produced by the reference's RPNPostProcessor
(tests/golden/heads.npz, rpn/inference.py:76-123), (b) the tensor-op formulation of the same module on the same
device, and (c) the CPU oracle's NMS on the non-dropped candidates.  Index outputs are compared exactly."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "heads.npz"))


def _module(pre_train, post_train, pre_test, post_test, min_size):
    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.rpn import RPNModule

    cfg = get_defaults()
    cfg.merge_from_list(["MODEL.RPN.PRE_NMS_TOP_N_TRAIN", pre_train, "MODEL.RPN.POST_NMS_TOP_N_TRAIN", post_train,
                         "MODEL.RPN.PRE_NMS_TOP_N_TEST", pre_test, "MODEL.RPN.POST_NMS_TOP_N_TEST", post_test,
                         "MODEL.RPN.MIN_SIZE", min_size])
    cfg.freeze()
    return RPNModule(cfg, 64).cuda()


def test_rpn_device_pipeline_matches_reference_fixture(z):
    """Same inputs as tests/test_components.py::test_anchors_and_rpn_proposals_match_reference, through the device
    kernels: the reference module's proposals and scores."""
    rpn = _module(600, 50, 600, 50, 0)
    pp = rpn.box_selector_test
    H, W = 9, 12
    sizes = [(H * 16, W * 16), (H * 16 - 10, W * 16 - 7)]
    anchors = rpn.anchor_generator(sizes, torch.zeros(2, 1, H, W, device="cuda"))
    assert pp._on_device(T(z["rpn_obj"]).cuda())
    res = pp(anchors, T(z["rpn_obj"]).cuda(), T(z["rpn_reg"]).cuda())
    for i, r in enumerate(res):
        want_b, want_s = T(z[f"rpn_boxes{i}"]), T(z[f"rpn_scores{i}"])
        assert r.bbox.shape == want_b.shape
        assert torch.allclose(r.bbox.cpu(), want_b, rtol=1e-5, atol=1e-4)
        assert torch.allclose(r.get_field("objectness").cpu(), want_s)


@pytest.mark.parametrize("min_size", [0, 24])
@pytest.mark.parametrize("layout", ["nchw", "nhwc_view"])
def test_rpn_device_pipeline_equals_tensor_ops_full_size(min_size, layout):
    """BASELINE size (two 800x1333 images: 15 x 50 x 84 anchors, 12000 -> 2000 and 6000 -> 1000): the device pipeline
    returns exactly the boxes and scores of the tensor-op sequence, for one selector and for the shared
    train + test selection; min_size 24 exercises the drop flags (boxes removed in front of the NMS)."""
    torch.manual_seed(3)
    rpn = _module(12000, 2000, 6000, 1000, min_size)
    n, a, h, w = 2, 15, 50, 84
    # scores without ties (63000 random floats hold ~100 equal pairs, and the order of equal scores is the one thing the two
    # top-k routes may differ in: ascending index here, unspecified for torch.topk): a shuffled ramp of logits
    ramp = torch.stack([torch.linspace(-6, 6, a * h * w, device="cuda")[torch.randperm(a * h * w, device="cuda")] for _ in range(n)])
    assert all(len(torch.unique(r.sigmoid())) == a * h * w for r in ramp)
    if layout == "nchw":
        obj = ramp.view(n, a, h, w).contiguous()
        reg = torch.randn(n, 4 * a, h, w, device="cuda") * 0.3
    else:  # what the frozen GEMM head hands over: NCHW views of one NHWC [N, H, W, 76] result
        y = torch.randn(n, h, w, 76, device="cuda")
        y[..., :a] = ramp.view(n, h, w, a)
        y[..., a:] *= 0.3
        obj, reg = y[..., :a].permute(0, 3, 1, 2), y[..., a:5 * a].permute(0, 3, 1, 2)
    sizes = [(800, 1333), (790, 1301)]
    anchors = rpn.anchor_generator(sizes, obj)
    gt = []
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.structures import BoxList
    for s in sizes:
        gt.append(BoxList(torch.tensor([[10.0, 20.0, 200.0, 300.0], [400.0, 100.0, 700.0, 500.0]], device="cuda"), (s[1], s[0])))
    tr, te = rpn.box_selector_train, rpn.box_selector_test

    def both(device_pipeline):
        tr.device_pipeline = te.device_pipeline = device_pipeline
        single = tr(anchors, obj, reg, gt, add_gt=True)
        ours, theirs = tr.forward_with(te, anchors, obj, reg, gt, add_gt=True)
        test_only = te(anchors, obj, reg)
        return single, ours, theirs, test_only

    got, want = both(True), both(False)
    for g_list, w_list in zip(got, want):
        for g, w_ in zip(g_list, w_list):
            assert len(g) == len(w_) and len(g) > 100
            assert torch.equal(g.bbox, w_.bbox)
            assert torch.equal(g.get_field("objectness"), w_.get_field("objectness"))
    if min_size:
        ws = got[0][0].bbox[:, 2] - got[0][0].bbox[:, 0] + 1
        assert float(ws[:-2].min()) >= min_size  # the appended ground truth aside


@pytest.mark.parametrize("n,a,k", [(2, 63000, 12000), (1, 100, 100), (3, 1000, 1), (5, 4097, 600)])
def test_topk_sorted_is_the_stable_descending_sort(n, a, k):
    """``_C.topk_sorted`` (one device radix sort of the batch; rpn/inference.py:95 ``objectness.topk(..., sorted=True)``):
    values AND indices equal the first k columns of a stable descending sort of every row -- scores quantised to force
    thousands of ties (equal scores come in ascending index order), -0 == +0, NaN largest, +-inf at the ends."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    g = torch.Generator().manual_seed(n * 1000 + k)
    x = (torch.randn(n, a, generator=g) * 64).round() / 64
    x[:, 3] = -0.0
    x[:, 7] = 0.0
    if a > 50:
        x[0, 11], x[0, 13], x[-1, 17], x[-1, 19] = float("nan"), float("inf"), float("-inf"), float("nan")
    want_v, want_i = torch.sort(x, dim=1, descending=True, stable=True)
    got_v, got_i = _C.topk_sorted(x.cuda(), k)
    assert got_i.dtype == torch.int64 and got_v.shape == (n, k)
    assert torch.equal(got_i.cpu(), want_i[:, :k])
    assert torch.equal(got_v.cpu().view(torch.int32), want_v[:, :k].contiguous().view(torch.int32))  # bit for bit, NaNs included
    # rows of a wider matrix (row stride > row length), and the tensor library's own top-k on tie-free scores
    wide = torch.rand(n, a + 37, generator=g).cuda()
    v2, i2 = _C.topk_sorted(wide[:, :a], k)
    tv, ti = wide[:, :a].topk(k, dim=1, sorted=True)
    assert torch.equal(v2, tv)
    same = i2 == ti
    assert bool((same | (v2 == torch.roll(v2, 1, 1)) | (v2 == torch.roll(v2, -1, 1))).all())  # they differ at ties only


def test_topk_sorted_rejects_what_it_cannot_serve():
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    x = torch.rand(2, 10, device="cuda")
    with pytest.raises(RuntimeError):
        _C.topk_sorted(x, 11)
    with pytest.raises(RuntimeError):
        _C.topk_sorted(x.cpu(), 3)
    with pytest.raises(RuntimeError):
        _C.topk_sorted(x.double(), 3)
    v, i = _C.topk_sorted(x, 0)
    assert v.shape == (2, 0) and i.shape == (2, 0)


def test_rpn_decode_keeps_non_finite_deltas_non_finite():
    """A diverged head (NaN / inf regression deltas) must not turn into valid-looking proposals: ``torch.clamp`` propagates
    NaN (rpn/inference.py:104-114 through box_coder.py:75-76 and bounding_box.py:214-225) and so does the kernel; finite
    candidates of the same launch are untouched."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    torch.manual_seed(2)
    n, a, h, w, k = 1, 3, 4, 5, 16
    reg = torch.randn(n, 4 * a, h, w, device="cuda") * 0.2
    idx = torch.randperm(a * h * w, device="cuda")[:k].view(1, k)
    cell = torch.tensor([[-8.0, -8, 23, 23], [-24, -8, 39, 23], [-8, -24, 23, 39]], device="cuda")
    wh = torch.tensor([[80.0, 64.0]], device="cuda")
    args = (cell, wh, (1.0, 1.0, 1.0, 1.0), 4.135, 0.0, 16)
    clean, _ = _C.rpn_decode(reg, idx, *args)
    bad = reg.clone()
    for j, (comp, val) in enumerate([(0, float("nan")), (2, float("nan")), (3, float("nan")), (1, float("inf"))]):
        i = int(idx[0, j])
        aa, pos = i % a, i // a
        bad[0, 4 * aa + comp, pos // w, pos % w] = val
    got, _ = _C.rpn_decode(bad, idx, *args)
    assert torch.isnan(got[0, 0, [0, 2]]).all()      # dx NaN -> x1, x2
    assert torch.isnan(got[0, 1, [0, 2]]).all()      # dw NaN -> x1, x2 (fminf would have returned xform_clip)
    assert torch.isnan(got[0, 2, [1, 3]]).all()      # dh NaN -> y1, y2
    assert torch.equal(got[0, 3, [1, 3]], torch.tensor([63.0, 63.0], device="cuda"))  # +inf clamps to the border, as torch
    assert torch.equal(got[0, 4:], clean[0, 4:])


def test_nms_presorted_batched_vs_oracle(oracle_mod):
    """Score-sorted batched NMS with drop flags == the oracle's NMS on the non-dropped boxes of every image; the second
    count is the number of survivors among the first `below` candidates, and they are a prefix of the list."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    g = torch.Generator().manual_seed(5)
    n, k = 3, 3000
    xy = torch.rand(n, k, 2, generator=g) * torch.tensor([900.0, 600.0])
    wh = torch.rand(n, k, 2, generator=g) * 250 + 4
    boxes = torch.cat([xy, xy + wh], 2)
    drop = -(torch.rand(n, k, generator=g) < 0.2).to(torch.int32)
    scores = torch.sort(torch.rand(n, k, generator=g), dim=1, descending=True).values  # distinct, descending
    below = 1100
    keep, counts = _C.nms_presorted_batched(boxes.cuda(), drop.cuda(), 0.6, below=below)
    keep, counts = keep.cpu(), counts.cpu()
    for i in range(n):
        alive = torch.nonzero(drop[i] == 0).squeeze(1)
        want = alive[oracle_mod.nms(boxes[i][alive], scores[i][alive], 0.6)]
        c, cb = int(counts[i, 0]), int(counts[i, 1])
        assert c == want.numel()
        assert torch.equal(keep[i, :c], want)
        assert bool((keep[i, c:] == 0).all())
        assert cb == int((want < below).sum()) and bool((keep[i, :cb] < below).all())
    keep2, counts2 = _C.nms_presorted_batched(boxes.cuda(), None, 0.6)
    for i in range(n):
        want = oracle_mod.nms(boxes[i], scores[i], 0.6)
        assert torch.equal(keep2[i, : int(counts2[i, 0])].cpu(), want)


def test_rpn_targets_and_loss_on_device_match_reference_fixture(z):
    """``_C.rpn_match_encode`` (IoU, Matcher WITH low-quality matches, visibility / between rules, delta targets) reproduces the
    reference's per-anchor labels exactly and its regression targets to rounding; the device form of the RPN loss (native
    targets + device sampler + padded gathers, no host read) gives the reference's two losses, and the tensor-op form run on
    the device gives the same."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    from tests.test_components import _rpn_loss_case

    loss, anchors, obj, reg, targets = _rpn_loss_case(z, "cuda")
    for i in range(2):
        lab, tgt = _C.rpn_match_encode(targets[i].bbox, anchors[i].bbox, anchors[i].get_field("visibility"), 0.7, 0.3, True,
                                       (1.0, 1.0, 1.0, 1.0))
        assert torch.equal(lab.cpu().float(), T(z[f"rpnloss_labels{i}"]))
        assert torch.allclose(tgt.cpu(), T(z[f"rpnloss_targets{i}"]), rtol=1e-6, atol=1e-6)
        plain, _ = _C.rpn_match_encode(targets[i].bbox, anchors[i].bbox, anchors[i].get_field("visibility"), 0.7, 0.3, False,
                                       (1.0, 1.0, 1.0, 1.0))
        assert int((plain == 1).sum()) <= int((lab == 1).sum())  # the low-quality rule only ever adds positives
    obj.requires_grad_(True)
    reg.requires_grad_(True)
    lo, lb = loss(anchors, obj, reg, targets)
    assert abs(float(lo) - float(z["rpnloss_objectness"])) <= 1e-5 * float(z["rpnloss_objectness"])
    assert abs(float(lb) - float(z["rpnloss_box"])) <= 1e-5 * float(z["rpnloss_box"])
    g_dev = torch.autograd.grad(lo + lb, [obj, reg])
    loss.device_targets = False
    lo2, lb2 = loss(anchors, obj, reg, targets)
    g_ops = torch.autograd.grad(lo2 + lb2, [obj, reg])
    assert abs(float(lo) - float(lo2)) <= 1e-6 * float(lo2) and abs(float(lb) - float(lb2)) <= 1e-6 * float(lb2)
    for a, b in zip(g_dev, g_ops):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-8)


def test_rpn_loss_on_device_at_baseline_size_with_sampling():
    """15 x 50 x 84 anchors per image, 256 sampled per image (the shipped quotas): the device form samples with its own key
    stream, so the losses are compared as statistics -- every sampled anchor carries label 0 / 1, the counts are the
    reference's quotas, and the objectness loss over the device's sample equals the tensor-op formula evaluated on exactly
    that sample."""
    import torch.nn.functional as F
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.structures import BoxList

    torch.manual_seed(4)
    rpn = _module(12000, 2000, 6000, 1000, 0)
    n, a, h, w = 2, 15, 50, 84
    obj = torch.randn(n, a, h, w, device="cuda")
    reg = torch.randn(n, 4 * a, h, w, device="cuda") * 0.1
    sizes = [(800, 1333), (790, 1301)]
    anchors = rpn.anchor_generator(sizes, obj)
    targets = []
    for s in sizes:
        xy = torch.rand(7, 2, device="cuda") * torch.tensor([900.0, 500.0], device="cuda")
        targets.append(BoxList(torch.cat([xy, xy + torch.rand(7, 2, device="cuda") * 300 + 30], 1), (s[1], s[0])))
    le = rpn.loss_evaluator
    calls = []
    orig = le.sampler.sample_device
    le.sampler.sample_device = lambda lab, generator=None: (calls.append((lab, orig(lab, generator))), calls[-1][1])[1]
    lo, lb = le(anchors, obj, reg, targets)
    assert torch.isfinite(lo) and torch.isfinite(lb) and len(calls) == 2
    total, bce_sum = 0, 0.0
    for i, (lab, (sel, slots, cnt)) in enumerate(calls):
        nsel, npos = (int(v) for v in cnt.tolist())
        assert nsel == 256 and npos == min(int((lab == 1).sum()), 128)
        picked = lab[sel[:nsel]]
        assert bool(((picked == 0) | (picked == 1)).all()) and int((picked == 1).sum()) == npos
        o = obj[i].permute(1, 2, 0).reshape(-1)[sel[:nsel]]
        bce_sum += float(F.binary_cross_entropy_with_logits(o, picked.float(), reduction="sum"))
        total += nsel
    assert abs(float(lo) - bce_sum / total) <= 1e-5 * (bce_sum / total)
