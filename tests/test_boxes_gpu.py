"""GPU: the box-head box arithmetic kernels (csrc/boxes.hip, csrc/targets.hip ``gather_rows``) against the reference-pinned
host forms: BoxCoder.decode (fixture produced by the reference's BoxCoder, tests/golden/heads.npz; and the tensor-op
formulation, bit for bit on the same device), clip_to_image, Pooler.convert_to_roi_format, the box-regression loss
(layers/smooth_l1_loss.py + box_head/loss.py:147-170 through autograd) and the per-field row gathers."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def _coder():
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.box_coder import BoxCoder

    return BoxCoder(weights=(10.0, 10.0, 5.0, 5.0))


def test_box_decode_matches_reference_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "heads.npz"))
    got = _coder().decode(T(z["coder_codes"]).cuda(), T(z["coder_prop"]).cuda())
    assert torch.allclose(got.cpu(), T(z["coder_dec"]), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("k", [1, 3])
@pytest.mark.parametrize("clip", [False, True])
@pytest.mark.parametrize("weights", [(1.0, 1.0, 1.0, 1.0), (10.0, 10.0, 5.0, 5.0)])
def test_box_decode_equals_tensor_ops(k, clip, weights):
    """Random boxes / deltas incl. deltas beyond bbox_xform_clip and boxes far outside the image, rows of a wider matrix
    (row stride != 4k) vs the tensor-op sequence (decode, then clip_to_image per image) on the same device.  Unit weights:
    the same bits.  Other weights: torch divides a tensor by a Python scalar as a multiplication by its reciprocal on the
    device, the kernel divides (as the CPU reference and the fixtures do): the deltas differ by an ulp, the boxes by
    ~1e-4 pixels."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.structures import BoxList

    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.box_coder import BoxCoder

    torch.manual_seed(5)
    coder = BoxCoder(weights=weights)
    per_img, sizes = [700, 0, 1301], [(1333, 800), (640, 480), (901, 777)]
    r = sum(per_img)
    xy = torch.rand(r, 2, device="cuda") * 1200 - 100
    boxes = torch.cat([xy, xy + torch.rand(r, 2, device="cuda") * 500], 1)
    wide = torch.randn(r, 4 * k + 8, device="cuda") * 4
    wide[::7, 2] = 60.0  # far above log(1000 / 16) after the weight
    codes = wide[:, 4:4 * k + 4]
    want = coder._decode_tensor_ops(codes, boxes)
    if clip:
        at = 0
        for n, size in zip(per_img, sizes):
            for j in range(k):  # BoxList.clip_to_image clamps in place through the views
                BoxList(want[at:at + n, 4 * j:4 * j + 4], size).clip_to_image(remove_empty=False)
            at += n
        got = coder.decode(codes, boxes, per_img, sizes)
    else:
        got = coder.decode(codes, boxes)
    assert got.shape == want.shape
    if weights[0] == 1.0:
        assert torch.equal(got, want)
    else:
        assert torch.allclose(got, want, rtol=2e-6, atol=1e-3)
    assert _C.box_decode(codes[:0], boxes[:0], coder.weights, coder.bbox_xform_clip).shape == (0, 4 * k)


def test_box_decode_more_images_than_one_launch_carries():
    """40 images in one call (an eval batch larger than the 16 image sizes a launch carries as kernel arguments, some images
    without rows): the same bits as decode + clip_to_image image by image."""
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.box_coder import BoxCoder
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.structures import BoxList

    torch.manual_seed(11)
    coder = BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
    per_img = [int(v) for v in torch.randint(0, 60, (40,))]
    per_img[16] = 0
    sizes = [(int(400 + 23 * i), int(300 + 17 * i)) for i in range(40)]
    r = sum(per_img)
    xy = torch.rand(r, 2, device="cuda") * 900 - 100
    boxes = torch.cat([xy, xy + torch.rand(r, 2, device="cuda") * 500], 1)
    codes = torch.randn(r, 8, device="cuda") * 2
    want = coder._decode_tensor_ops(codes, boxes)
    at = 0
    for n, size in zip(per_img, sizes):
        for j in range(2):
            BoxList(want[at:at + n, 4 * j:4 * j + 4], size).clip_to_image(remove_empty=False)
        at += n
    got = coder.decode(codes, boxes, per_img, sizes)
    assert torch.equal(got, want)


def test_box_decode_argument_errors():
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    c, b = torch.zeros(4, 4, device="cuda"), torch.zeros(4, 4, device="cuda")
    with pytest.raises(RuntimeError):
        _C.box_decode(c.cpu(), b.cpu(), (1, 1, 1, 1), 4.0)
    with pytest.raises(RuntimeError):
        _C.box_decode(c, b[:3], (1, 1, 1, 1), 4.0)
    with pytest.raises(RuntimeError):  # rows per image must add up
        _C.box_decode(c, b, (1, 1, 1, 1), 4.0, [1, 2], [(10, 10), (10, 10)])


def test_rois_from_boxes_equals_convert_to_roi_format():
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.roi_heads import Pooler
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.structures import BoxList

    torch.manual_seed(2)
    lists = [BoxList(torch.rand(n, 4) * 300, (400, 300)) for n in (5, 0, 513, 1)]
    want = Pooler.convert_to_roi_format(lists)  # CPU tensors: the reference's full + cat sequence
    got = Pooler.convert_to_roi_format([b.to("cuda") for b in lists])
    assert torch.equal(got.cpu(), want)
    ids = [3, 1, 0, 2]
    got = _C.rois_from_boxes([b.bbox.cuda() for b in lists], ids)
    want[:, 0] = torch.cat([torch.full((len(b),), float(i)) for b, i in zip(lists, ids)])
    assert torch.equal(got.cpu(), want)
    many = [torch.rand(3, 4, device="cuda") for _ in range(37)]  # more images than one launch carries
    got = _C.rois_from_boxes(many)
    assert torch.equal(got[:, 1:], torch.cat(many)) and got[:, 0].tolist() == [float(i) for i in range(37) for _ in range(3)]


def test_gather_rows_equals_index_select():
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    torch.manual_seed(4)
    p = 2000
    a, b = torch.randn(p, 4, device="cuda"), torch.randn(p, 4, device="cuda")
    ia, ib = torch.randint(-5, 90, (p,), device="cuda"), torch.randint(0, 7, (p,), device="cuda")
    idx = torch.randperm(p, device="cuda")[:512].sort().values
    got = _C.gather_rows(idx, a, b, ia, ib)
    for g, src in zip(got, (a, b, ia, ib)):
        assert torch.equal(g, src.index_select(0, idx))
    got = _C.gather_rows(idx[:7], a, None, ia, None)
    assert got[1] is None and got[3] is None and torch.equal(got[0], a[idx[:7]]) and torch.equal(got[2], ia[idx[:7]])
    assert _C.gather_rows(idx[:0], a, b, ia, ib)[0].shape == (0, 4)


@pytest.mark.parametrize("agnostic", [True, False])
def test_smooth_l1_picked_loss_and_gradient_vs_autograd(agnostic):
    """box_head/loss.py:147-170 through torch autograd (index, smooth_l1_loss(beta=1, size_average=False) / numel) vs
    the fused kernel: loss to 1e-6 relative, gradient element-wise (both are a handful of fp32 operations per entry)."""
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import smooth_l1_loss, smooth_l1_picked

    torch.manual_seed(8)
    r, classes = 1024, 2 if agnostic else 49
    wide = (torch.randn(r, 4 * classes + 24, device="cuda") * 1.5)
    reg = wide[:, 8:8 + 4 * classes].detach().requires_grad_(True)  # a column-slice view, as the predictor returns
    tgt = torch.randn(r, 4, device="cuda")
    labels = torch.randint(0, classes, (r,), device="cuda")
    pos = torch.nonzero(labels > 0).squeeze(1)
    if agnostic:
        picked = reg.index_select(0, pos)[:, 4:8]
    else:
        picked = reg[pos[:, None], 4 * labels[pos][:, None] + torch.arange(4, device="cuda")]
    want = smooth_l1_loss(picked, tgt.index_select(0, pos), size_average=False, beta=1) / labels.numel()
    (gw,) = torch.autograd.grad(want * 3.0, reg)
    got = smooth_l1_picked(reg, tgt, pos, None if agnostic else labels, 4, 1.0, labels.numel())
    (gg,) = torch.autograd.grad(got * 3.0, reg)
    assert abs(got.item() - want.item()) <= 1e-6 * abs(want.item())
    assert gg.shape == gw.shape and torch.allclose(gg, gw, rtol=1e-6, atol=1e-9)
    # no positives: zero loss, zero gradient
    got = smooth_l1_picked(reg, tgt, pos[:0], None if agnostic else labels, 4, 1.0, labels.numel())
    (gg,) = torch.autograd.grad(got, reg)
    assert got.item() == 0.0 and not gg.any()
    # deterministic
    a = smooth_l1_picked(reg, tgt, pos, None if agnostic else labels, 4, 1.0, labels.numel())
    b = smooth_l1_picked(reg, tgt, pos, None if agnostic else labels, 4, 1.0, labels.numel())
    assert torch.equal(a, b)
