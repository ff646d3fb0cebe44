"""GPU: fused training-target kernels (csrc/targets.hip) against the tensor-op formulations they replace (which are
pinned to the reference by tests/test_components.py fixtures).  Integer outputs exact; deltas 1e-6 (device logf)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _boxes(n, g, w=1333.0, h=800.0):
    xy = torch.rand(n, 2, generator=g) * torch.tensor([w - 60, h - 60])
    wh = torch.rand(n, 2, generator=g) * 300 + 8
    return torch.cat([xy, torch.minimum(xy + wh, torch.tensor([w - 1, h - 1]))], 1)


@pytest.mark.parametrize("G,P", [(1, 7), (7, 1000), (20, 2007), (3, 1)])
def test_match_encode_vs_tensor_ops(G, P):
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.box_coder import BoxCoder
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.matcher import Matcher
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.structures import box_iou
    g = torch.Generator().manual_seed(G * 1000 + P)
    gt = _boxes(G, g).cuda()
    prop = _boxes(P, g)
    prop[: min(P, G)] = gt[: min(P, G)].cpu() + torch.randn(min(P, G), 4, generator=g) * 6   # some real positives
    prop = torch.cat([prop, gt.cpu()], 0).cuda()   # the ground truth itself is appended in training
    labels = torch.randint(1, 49, (G,), generator=g).cuda()
    matcher, coder = Matcher(0.5, 0.3), BoxCoder((10.0, 10.0, 5.0, 5.0))
    iou = box_iou(gt, prop)
    matched = matcher(iou)
    idx_ref = matched.clamp(min=0)
    for keep in (False, True):
        idx, lab, reg = _C.match_encode(gt, labels, prop, 0.5, 0.3, coder.weights, between_keeps_label=keep)
        lab_ref = labels[idx_ref].clone()
        lab_ref[matched == Matcher.BELOW_LOW_THRESHOLD] = 0
        if not keep:
            lab_ref[matched == Matcher.BETWEEN_THRESHOLDS] = -1
        assert torch.equal(idx, idx_ref)
        assert torch.equal(lab, lab_ref)
        reg_ref = coder.encode(gt[idx_ref], prop)
        assert (reg - reg_ref).abs().max().item() <= 1e-6 * max(1.0, reg_ref.abs().max().item())
    assert (lab_ref > 0).any()
    idx2, lab2, reg2 = _C.match_encode(gt, labels, prop, 0.5, 0.3)
    assert reg2 is None and torch.equal(idx2, idx_ref)
    with pytest.raises(ValueError):
        _C.match_encode(gt[:0], labels[:0], prop, 0.5, 0.5)


@pytest.mark.parametrize("dtype", [torch.bool, torch.uint8])
def test_project_masks_vs_tensor_ops(dtype):
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.roi_heads import project_masks_on_boxes
    g = torch.Generator().manual_seed(11)
    H, W, G, P = 200, 333, 5, 300
    masks = torch.zeros(G, H, W, dtype=torch.uint8)
    gtb = _boxes(G, g, W, H)
    for i, b in enumerate(gtb.round().long()):
        masks[i, b[1]:b[3] + 1, b[0]:b[2] + 1] = 1
        masks[i, ::7, ::5] = 0   # holes, so the interpolation sees edges everywhere
    boxes = _boxes(P, g, W, H)
    boxes[:8] = torch.tensor([[-5.0, -3.0, 10.5, 7.5], [W - 3.0, H - 2.0, W + 9.0, H + 4.0], [10.5, 10.5, 10.6, 10.7],
                              [0.5, 1.5, 2.5, 3.5], [0.0, 0.0, W - 1.0, H - 1.0], [30.0, 40.0, 30.0, 40.0],
                              [2.5, 3.5, 100.5, 7.5], [50.2, 60.8, 51.1, 199.9]])
    idx = torch.randint(0, G, (P,), generator=g)
    masks = masks.to(dtype).cuda()
    ref = project_masks_on_boxes(masks, idx.cuda(), boxes.cuda(), 14)
    got = _C.project_masks(masks, idx.cuda(), boxes.cuda(), 14)
    assert got.shape == (P, 14, 14)
    assert torch.equal(got, ref)
    assert _C.project_masks(masks, idx[:0].cuda(), boxes[:0].cuda(), 14).shape == (0, 14, 14)
