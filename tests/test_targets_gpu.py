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


def _interpolate_project(masks, idx, boxes, M):
    """project_masks_on_boxes as the reference spells it (mask_head/loss.py:31-42 on a BinaryMaskList,
    segmentation_mask.py:117-156): python-rounded crop, F.interpolate(bilinear, align_corners=False), type_as."""
    out = []
    H, W = masks.shape[-2:]
    for b, g in zip(boxes.tolist(), idx.tolist()):
        xmin, ymin, xmax, ymax = [round(float(v)) for v in b]
        xmin, ymin = min(max(xmin, 0), W - 1), min(max(ymin, 0), H - 1)
        xmax, ymax = max(min(max(xmax, 0), W), xmin + 1), max(min(max(ymax, 0), H), ymin + 1)
        crop = masks[g, ymin:ymax, xmin:xmax]
        r = torch.nn.functional.interpolate(crop[None, None].float(), size=(M, M), mode="bilinear", align_corners=False)[0, 0]
        out.append(r.type_as(masks).float())
    return torch.stack(out)


@pytest.mark.parametrize("on_gpu", [False, True])
def test_project_masks_on_crops_that_are_multiples_of_the_resolution(on_gpu):
    """Crops of 14, 28, 42, 56 ... pixels put every source position on a pixel centre (or exactly between two): the second
    tap's weight is exactly 0 there, and 'any weight on a set pixel' (bool masks) must not count it.  The scale has to be
    the correctly rounded in / out of F.interpolate -- size * (1 / 14) is an ulp high for 42 and 84 (found by the
    reference-made whole-step fixture, tests/test_step_golden.py).  Against the reference's own formulation on the CPU."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.roi_heads import project_masks_on_boxes
    g = torch.Generator().manual_seed(5)
    H, W, G = 128, 160, 3
    masks = torch.rand(G, H, W, generator=g) > 0.35        # salt-and-pepper: every tap matters
    boxes = []
    for k in range(1, 9):
        for j in range(1, 7):
            x0, y0 = float(torch.randint(0, W - 14 * k + 1, (1,), generator=g)), float(torch.randint(0, H - 14 * j + 1, (1,), generator=g))
            boxes.append([x0 + 0.2, y0 - 0.3, x0 + 14 * k - 0.4, y0 + 14 * j + 0.3])
    boxes = torch.tensor(boxes)
    idx = torch.randint(0, G, (boxes.shape[0],), generator=g)
    want = _interpolate_project(masks, idx, boxes, 14)
    dev = "cuda" if on_gpu else "cpu"
    got_formula = project_masks_on_boxes(masks.to(dev), idx.to(dev), boxes.to(dev), 14).cpu()
    assert torch.equal(got_formula, want), int((got_formula != want).sum())
    if on_gpu:
        got = _C.project_masks(masks.cuda(), idx.cuda(), boxes.cuda(), 14).cpu()
        assert torch.equal(got, want), int((got != want).sum())


# ---- device fg / bg sampler (csrc/targets.hip::sample_fg_bg_kernel) ------------------------------------------------------
def _labels(p, n_pos, n_ign, g):
    lab = torch.zeros(p, dtype=torch.int64)
    perm = torch.randperm(p, generator=g)
    lab[perm[:n_pos]] = torch.randint(1, 49, (n_pos,), generator=g)
    lab[perm[n_pos:n_pos + n_ign]] = -1
    return lab


@pytest.mark.parametrize("p,n_pos,n_ign,batch,frac", [(2007, 300, 40, 512, 0.25), (2007, 900, 0, 512, 1.0), (1007, 3, 5, 512, 0.25),
                                                      (300, 20, 10, 512, 0.25), (63000, 700, 30000, 256, 0.5), (5, 0, 5, 512, 0.25)])
def test_sample_fg_bg_counts_and_membership(p, n_pos, n_ign, batch, frac):
    """Counts follow balanced_positive_negative_sampler.py:39-47; the selection is a subset of the right classes, ascending,
    zero padded; positive_slots point at the positives; the same seed repeats, another seed differs when there is a choice."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    g = torch.Generator().manual_seed(p + n_pos)
    lab = _labels(p, n_pos, n_ign, g)
    n_neg = p - n_pos - n_ign
    want_pos = min(n_pos, int(batch * frac))
    want_neg = min(n_neg, batch - want_pos)
    sel, slots, counts = _C.sample_fg_bg(lab.cuda(), batch, int(batch * frac), 12345)
    sel, slots, (n, npos) = sel.cpu(), slots.cpu(), counts.tolist()
    assert (n, npos) == (want_pos + want_neg, want_pos)
    s = sel[:n]
    assert bool((s[1:] > s[:-1]).all()) and bool((sel[n:] == 0).all())
    assert int((lab[s] >= 1).sum()) == want_pos and int((lab[s] == 0).sum()) == want_neg
    assert torch.equal(torch.nonzero(lab[s] >= 1).squeeze(1), slots[:npos])
    sel2, _, c2 = _C.sample_fg_bg(lab.cuda(), batch, int(batch * frac), 12345)
    assert torch.equal(sel2.cpu(), sel) and c2.tolist() == [n, npos]
    if want_neg < n_neg or want_pos < n_pos:
        sel3, _, _ = _C.sample_fg_bg(lab.cuda(), batch, int(batch * frac), 999)
        assert not torch.equal(sel3.cpu(), sel)


def test_sample_fg_bg_vs_reference_fixture(golden_dir):
    """Fixture made by the reference's BalancedPositiveNegativeSampler (tests/golden/make_golden.py): the masks when the
    quotas cover every candidate (deterministic), the counts when they do not."""
    import os

    import numpy as np

    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    z = np.load(os.path.join(golden_dir, "heads.npz"))
    lab = torch.from_numpy(z["sampler_labels"]).to(torch.int64)
    sel, slots, counts = _C.sample_fg_bg(lab.cuda(), 4096, int(4096 * 0.25), 7)
    n, npos = counts.tolist()
    want = torch.from_numpy(z["sampler_pos_all"] | z["sampler_neg_all"])
    assert torch.equal(sel[:n].cpu(), torch.nonzero(want).squeeze(1))
    assert torch.equal(sel[:n].cpu()[slots[:npos].cpu()], torch.nonzero(torch.from_numpy(z["sampler_pos_all"])).squeeze(1))
    for batch, frac, key in ((512, 0.25, "sampler_counts_512_025"), (256, 1.0, "sampler_counts_256_100")):
        _, _, c = _C.sample_fg_bg(lab.cuda(), batch, int(batch * frac), 11)
        n, npos = c.tolist()
        assert [npos, n - npos] == z[key].tolist()


def test_sample_fg_bg_is_uniform():
    """Every negative is picked with probability k / n: over 400 seeds the per-element pick counts of 64-of-256 stay
    inside 5 sigma of the binomial, and so do the counts of pairs of neighbours (no index-correlated keys)."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    lab = torch.zeros(256, dtype=torch.int64).cuda()
    hits = torch.zeros(256)
    pair = torch.zeros(255)
    trials = 400
    for seed in range(trials):
        sel, _, counts = _C.sample_fg_bg(lab, 64, 0, seed * 7919 + 1)
        m = torch.zeros(256)
        m[sel[: counts.tolist()[0]].cpu()] = 1
        hits += m
        pair += m[1:] * m[:-1]
    pr = 64 / 256
    sd = (trials * pr * (1 - pr)) ** 0.5
    assert float((hits - trials * pr).abs().max()) < 5 * sd
    pp = pr * 63 / 255
    assert float((pair - trials * pp).abs().max()) < 5 * (trials * pp * (1 - pp)) ** 0.5 + 1


# ---- mask targets straight from (probability map, box) pairs -----------------------------------------------------------
@pytest.mark.parametrize("G,P,size", [(6, 300, (800, 1333)), (1, 7, (160, 192)), (3, 50, (97, 61))])
def test_project_pasted_masks_equals_paste_then_project(G, P, size):
    """_C.project_pasted_masks == Masker paste (mask_head/inference.py:124-205) followed by _C.project_masks, bit for bit,
    including pseudo boxes and proposals that stick out of the image."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.structures import PastedMasks
    h, w = size
    g = torch.Generator().manual_seed(G * 100 + P)
    gt = _boxes(G, g, float(w), float(h))
    gt[0] = torch.tensor([-7.3, -4.1, w * 0.6, h * 0.7])           # sticks out at the top left
    gt[-1] = torch.tensor([w * 0.5, h * 0.4, w + 9.0, h + 5.5])     # ... and at the bottom right
    probs = torch.rand(G, 14, 14, generator=g)
    idx = torch.randint(0, G, (P,), generator=g)
    prop = gt[idx] + torch.randn(P, 4, generator=g) * 9
    pm = PastedMasks(probs.cuda(), gt.cuda(), (h, w))
    full = pm.materialize()
    assert full.shape == (G, h, w) and full.dtype == torch.bool and bool(full.any())
    want = _C.project_masks(full, idx.cuda(), prop.cuda(), 14)
    got = _C.project_pasted_masks(probs.cuda(), gt.cuda(), idx.cuda(), prop.cuda(), (h, w), 14)
    assert got.shape == want.shape == (P, 14, 14)
    assert torch.equal(got, want), int((got != want).sum())
    assert 0.05 < float(got.mean()) < 0.95


def _sampler_model(lab, batch, max_pos, seed):
    """NumPy statement of the sampler: per class the k smallest splitmix64 keys of (seed, index), ties by index."""
    import numpy as np
    lab = lab.numpy()
    with np.errstate(over="ignore"):
        i = np.arange(lab.shape[0], dtype=np.uint64)
        z = np.uint64(seed) + (i + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        key = ((z ^ (z >> np.uint64(31))) >> np.uint64(32)).astype(np.int64)
    pos, neg = np.nonzero(lab >= 1)[0], np.nonzero(lab == 0)[0]
    k_pos = min(len(pos), max_pos)
    k_neg = min(len(neg), batch - k_pos)
    take = lambda idx, k: idx[np.lexsort((idx, key[idx]))[:k]]
    sel = np.sort(np.concatenate([take(pos, k_pos), take(neg, k_neg)]))
    return torch.from_numpy(sel), torch.from_numpy(np.nonzero(lab[sel] >= 1)[0])


@pytest.mark.parametrize("p,n_pos,n_ign,batch,frac,seed", [(63000, 40, 300, 256, 0.5, 5), (63000, 700, 30000, 256, 0.5, 77),
                                                           (2007, 900, 0, 512, 0.25, 3), (90000, 5000, 100, 512, 0.25, 9),
                                                           (76792, 10, 0, 256, 0.5, 1), (76793, 10, 0, 256, 0.5, 1), (70, 30, 5, 16, 0.5, 2)])
def test_sample_fg_bg_is_the_k_smallest_keys_per_class(p, n_pos, n_ign, batch, frac, seed):
    """Exact selection against the NumPy model, in the LDS-cached form (P <= 76 792: class + top 14 key bits per element in
    LDS) and in the uncached one (larger P) -- the two are the same function."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    g = torch.Generator().manual_seed(p + seed)
    lab = _labels(p, n_pos, n_ign, g)
    sel, slots, counts = _C.sample_fg_bg(lab.cuda(), batch, int(batch * frac), seed)
    n, npos = counts.tolist()
    want_sel, want_slots = _sampler_model(lab, batch, int(batch * frac), seed)
    assert n == want_sel.numel() and npos == want_slots.numel()
    assert torch.equal(sel[:n].cpu(), want_sel) and torch.equal(slots[:npos].cpu(), want_slots)


def test_sample_fg_bg_ties_follow_the_index():
    """All-equal keys cannot be forced through the hash, but duplicated THRESHOLD digits can: with 60 000 negatives and 256
    wanted, several elements share the threshold's top 14 bits (the cached part) and are told apart by the full key."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    lab = torch.zeros(60000, dtype=torch.int64)
    for seed in range(20):
        sel, _, counts = _C.sample_fg_bg(lab.cuda(), 256, 0, seed)
        want, _ = _sampler_model(lab, 256, 0, seed)
        assert counts.tolist() == [256, 0] and torch.equal(sel.cpu(), want)
