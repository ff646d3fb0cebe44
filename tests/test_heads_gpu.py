"""GPU parity of the cross-modal head kernels (MFMA GEMM, region<->noun alignment, fused losses): against
fixtures produced by the reference's own Python modules (tests/golden/heads.npz) and against plain fp32/fp64
torch formulas (what the reference computes these with) on seeded inputs."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "heads.npz"))


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_box_predictor_fixture(z):
    from tests.test_components import small_cfg
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling import roi_heads as RH

    pred = RH.FastRCNNPredictor(small_cfg(), 96).cuda()
    pred.load_state_dict({k[5:]: T(z[k]) for k in z.files if k.startswith("pred_") and k[5:] in pred.state_dict()})
    x = T(z["pred_x"]).cuda()
    for c in (1, 49, 1203):
        pred.set_class_embeddings(T(z[f"pred_cls{c}"]).cuda())
        logits, box = pred(x)
        assert torch.allclose(logits.cpu(), T(z[f"pred_logits{c}"]), rtol=1e-4, atol=1e-5)  # north_star: 1e-3 rel
    assert torch.allclose(box.cpu(), T(z["pred_box"]), rtol=1e-4, atol=1e-6)


def test_box_loss_fixture(z):
    from tests.test_components import small_cfg
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling import roi_heads as RH
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.structures import BoxList

    ev = RH.FastRCNNLossComputation(small_cfg())
    P = z["boxloss_labels"].shape[0]
    prop = BoxList(torch.zeros(P, 4), (100, 100))
    prop.add_field("labels", T(z["boxloss_labels"]).cuda())
    prop.add_field("regression_targets", T(z["boxloss_targets"]).cuda())
    ev._proposals = [prop]
    lc, lb = ev(T(z["boxloss_logits"]).cuda(), T(z["boxloss_reg"]).cuda())
    assert torch.allclose(lc.cpu(), T(z["boxloss_cls"]), rtol=1e-5)
    assert torch.allclose(lb.cpu(), T(z["boxloss_box"]), rtol=1e-5)


def test_mask_predictor_and_fused_loss_fixture(z):
    from tests.test_components import small_cfg
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import stochastic_mask_bce
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling import roi_heads as RH

    mp = RH.MaskRCNNC4Predictor(small_cfg(), 64).cuda()
    mp.load_state_dict({k[9:]: T(z[k]) for k in z.files if k.startswith("maskpred_")})
    x = T(z["mask_x"]).cuda()
    mp.train()
    eps = T(z["mask_eps"]).cuda()
    logits5, scale = mp(x, True, eps=eps)
    assert torch.allclose(logits5.cpu(), T(z["mask_logits5"]), rtol=1e-4, atol=1e-5)
    mu, sigma = mp.forward_parts(x)
    loss = stochastic_mask_bce(mu, sigma, eps[0], torch.arange(6, device="cuda"), T(z["mask_targets"]).cuda().reshape(6, -1), 1)
    assert torch.allclose(loss.cpu(), T(z["mask_loss"]), rtol=1e-5)


@pytest.mark.parametrize("m,n,k", [(512, 776, 2048), (1000, 1203, 768), (1024, 49, 768), (7, 5, 768), (65, 130, 33),
                                   (1, 1, 1)])
def test_gemm_and_linear_autograd_vs_torch(m, n, k):
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import linear_mfma

    g = torch.Generator().manual_seed(m * 7 + n)
    a, b, bias = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g), torch.randn(n, generator=g)
    want = (a.double() @ b.double().t() + bias.double())
    got = _C.gemm_nt(a.cuda(), b.cuda(), bias.cuda()).cpu().double()
    tol = 2e-6 * (a.double().abs() @ b.double().abs().t() + bias.double().abs())  # fp32 round-off class bound
    assert bool(((got - want).abs() <= tol + 1e-30).all())
    # strided operands (the backward products) and autograd
    ad, bd, biasd = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True), bias.cuda().requires_grad_(True)
    y = linear_mfma(ad, bd, biasd)
    gy = torch.randn(m, n, generator=g)
    y.backward(gy.cuda())
    ar, br, biasr = a.double().requires_grad_(True), b.double().requires_grad_(True), bias.double().requires_grad_(True)
    (ar @ br.t() + biasr).backward(gy.double())
    for got_g, want_g in ((ad.grad, ar.grad), (bd.grad, br.grad), (biasd.grad, biasr.grad)):
        scale = want_g.abs().max().item() + 1e-12
        assert (got_g.cpu().double() - want_g).abs().max().item() <= 1e-5 * scale * max(1.0, (m + n + k) ** 0.5 / 8)


@pytest.mark.parametrize("m,n,k", [(1024, 776, 2048), (2000, 768, 2048), (1024, 1203, 768), (1024, 49, 768), (3, 776, 2048)])
def test_linear_pair_split_gemm_vs_fp64_and_deterministic(m, n, k):
    """The cross-modal head's products through the pair-layout split GEMM (layers.cross_modal._LinearPair): inside the
    three-term bf16 split's bound 3e-5 * sum|a||b| of the fp64 product (a single product can reach 2^-17 + 2^-17 + 2^-18 =
    1.9e-5 plus the fp32 accumulation; long sums measure ~1e-6), forward and both gradients,
    and bit-identical run to run (K slices are reduced in a fixed order, no atomics)."""
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import cross_modal

    g = torch.Generator().manual_seed(m + n + k)
    a, b, bias = torch.randn(m, k, generator=g) * 0.5, torch.randn(n, k, generator=g) * 0.05, torch.randn(n, generator=g)
    gy = torch.randn(m, n, generator=g)
    assert cross_modal._pair_ok(a.cuda(), b.cuda())

    def run():
        ad, bd, biasd = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True), bias.cuda().requires_grad_(True)
        y = cross_modal.linear_mfma(ad, bd, biasd)
        y.backward(gy.cuda())
        return y.detach(), ad.grad, bd.grad, biasd.grad

    y, da, db, dbias = run()
    a64, b64, g64 = a.double(), b.double(), gy.double()
    for got, want, bound in ((y, a64 @ b64.t() + bias.double(), a64.abs() @ b64.abs().t() + bias.double().abs()),
                             (da, g64 @ b64, g64.abs() @ b64.abs()), (db, g64.t() @ a64, g64.abs().t() @ a64.abs())):
        assert got.shape == want.shape
        assert bool(((got.cpu().double() - want).abs() <= 3e-5 * bound + 1e-30).all())
    assert torch.allclose(dbias.cpu().double(), g64.sum(0), rtol=1e-5, atol=1e-5)
    for t1, t2 in zip(run(), (y, da, db, dbias)):
        assert torch.equal(t1, t2)


def test_region_noun_align_vs_torch():
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    g = torch.Generator().manual_seed(3)
    emb = torch.randn(1000, 768, generator=g) * 0.05
    nouns = F.normalize(torch.randn(9, 768, generator=g), dim=-1)
    raw, idx = torch.max(emb.double() @ nouns.double().t(), dim=0)
    r, p, i = _C.region_noun_align(emb.cuda(), nouns.cuda())
    assert torch.equal(i.cpu(), idx)  # indices exact (the fixture has no near-ties)
    assert torch.allclose(r.cpu().double(), raw, rtol=1e-5, atol=1e-6)
    assert torch.allclose(p.cpu().double(), torch.sigmoid(raw), rtol=1e-5)
    # exact ties: lowest region index wins, like torch.max
    emb2 = torch.zeros(10, 8)
    emb2[3, 0] = emb2[7, 0] = 2.0
    _, _, i2 = _C.region_noun_align(emb2.cuda(), torch.eye(8)[:1].cuda())
    assert i2.tolist() == [3]
    r0, p0, i0 = _C.region_noun_align(emb.cuda(), torch.zeros(0, 768).cuda())
    assert r0.numel() == 0 and i0.numel() == 0


@pytest.mark.parametrize("p,c", [(1024, 1203), (512, 49), (3, 2), (1, 1)])
def test_weighted_ce_vs_torch(p, c):
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import weighted_cross_entropy

    g = torch.Generator().manual_seed(p + c)
    x = torch.randn(p, c, generator=g) * 3
    lab = torch.randint(0, c, (p,), generator=g)
    lab[::3] = 0
    w = torch.ones(c, dtype=torch.float64)
    w[0] = 0.2
    xr = x.double().requires_grad_(True)
    want = (F.cross_entropy(xr, lab, weight=w, reduction="none") / p).sum()
    want.backward()
    xd = x.cuda().requires_grad_(True)
    got = weighted_cross_entropy(xd, lab.cuda(), 0.2)
    (got * 1.7).backward()
    assert abs(got.item() - want.item()) <= 1e-5 * max(abs(want.item()), 1e-3)
    assert torch.allclose(xd.grad.cpu().double(), 1.7 * xr.grad, rtol=1e-4, atol=1e-8)


def test_stochastic_mask_bce_vs_torch():
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import stochastic_mask_bce

    g = torch.Generator().manual_seed(11)
    P, M = 37, 14
    mu = torch.randn(P, 2, M, M, generator=g) * 2
    sigma = torch.rand(P, 1, M, M, generator=g) + 0.3
    eps = torch.randn(P, 2, M, M, generator=g)
    pos = torch.tensor([0, 3, 4, 9, 20, 36])
    tgt = (torch.rand(pos.numel(), M * M, generator=g) > 0.5).float()
    mr, sr = mu.double().requires_grad_(True), sigma.double().requires_grad_(True)
    z_ = mr + eps.double() * (mr * 0.0 + sr)
    want = F.binary_cross_entropy_with_logits(z_[pos, 1].reshape(pos.numel(), -1), tgt.double(), reduction="none").mean()
    want.backward()
    md, sd = mu.cuda().requires_grad_(True), sigma.cuda().requires_grad_(True)
    got = stochastic_mask_bce(md, sd, eps.cuda(), pos.cuda(), tgt.cuda(), 1)
    got.backward()
    assert abs(got.item() - want.item()) <= 1e-5 * abs(want.item())
    assert torch.allclose(md.grad.cpu().double(), mr.grad, rtol=1e-4, atol=1e-9)
    assert torch.allclose(sd.grad.cpu().double(), sr.grad, rtol=1e-4, atol=1e-9)
    # deterministic variant (no sigma / eps) and the empty-positives case
    got2 = stochastic_mask_bce(mu.cuda(), None, None, pos.cuda(), tgt.cuda(), 1)
    want2 = F.binary_cross_entropy_with_logits(mu[pos, 1].reshape(pos.numel(), -1), tgt)
    assert abs(got2.item() - want2.item()) <= 1e-5 * abs(want2.item())


# ---------------------------------------------------------------- bf16 hi/lo split GEMM + the NHWC res5 head
def test_split_bf16x3_layout_and_precision():
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    g = torch.Generator().manual_seed(4)
    x = (torch.randn(37, 64, generator=g) * torch.logspace(-3, 3, 64)).cuda()
    for mode in (0, 1):
        s = _C.split_bf16x3(x, mode)
        assert s.shape == (37, 192) and s.dtype == torch.bfloat16
        hi = x.to(torch.bfloat16)
        lo = (x - hi.float()).to(torch.bfloat16)
        parts = (hi, hi, lo) if mode == 0 else (hi, lo, hi)
        for i, p in enumerate(parts):
            assert torch.equal(s[:, 64 * i: 64 * (i + 1)], p)  # round-to-nearest-even, exact remainder
        assert ((hi.double() + lo.double() - x.double()).abs() <= 2.0 ** -16 * x.double().abs()).all()
    # a row-strided view is read in place
    v = x[:, :32]
    assert torch.equal(_C.split_bf16x3(v, 0)[:, :32], v.to(torch.bfloat16))


@pytest.mark.parametrize("m,k,ns", [(4096, 1024, (512, 2048)), (1000, 512, (2048,)), (77, 64, (12,))])
def test_split_linear_vs_fp64(m, k, ns):
    """Forward and all three gradient products within the stated split tolerance: 2e-5 of sum_k |a_k b_k|
    (measured ~4e-6 of the result's magnitude; an fp32 GEMM sits at ~2e-6)."""
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import split_linear

    g = torch.Generator().manual_seed(m + k)
    x = torch.randn(m, k, generator=g).cuda().requires_grad_(True)
    ws = [(torch.randn(n, k, generator=g) / k ** 0.5).cuda().requires_grad_(True) for n in ns]
    bs = [torch.randn(n, generator=g).cuda().requires_grad_(True) for n in ns]
    ys = split_linear(x, *[t for wb in zip(ws, bs) for t in wb])
    gs = [torch.randn(m, n, generator=g).cuda() for n in ns]
    sum((y * gy).sum() for y, gy in zip(ys, gs)).backward()
    xd = x.detach().double()
    dx_ref = torch.zeros_like(xd)
    for y, w, b, gy in zip(ys, ws, bs, gs):
        wd, gd = w.detach().double(), gy.double()
        ref = xd @ wd.t() + b.detach().double()
        bound = 2e-5 * (xd.abs() @ wd.abs().t()) + 1e-6
        assert ((y.detach().double() - ref).abs() <= bound).all()
        dw_ref = gd.t() @ xd
        assert ((w.grad.double() - dw_ref).abs() <= 2e-5 * (gd.abs().t() @ xd.abs()) + 1e-6).all()
        assert torch.allclose(b.grad.double(), gd.sum(0), rtol=1e-5, atol=1e-4)
        dx_ref += gd @ wd
    dx_bound = 2e-5 * sum(gy.double().abs() @ w.detach().double().abs() for gy, w in zip(gs, ws)) + 1e-6
    assert ((x.grad.double() - dx_ref).abs() <= dx_bound).all()


@pytest.mark.parametrize("r,h,w,c,n,k", [(9, 7, 7, 64, 32, 3), (3, 5, 6, 8, 12, 3), (2, 4, 4, 16, 8, 5), (4, 7, 7, 32, 16, 1)])
def test_split_conv_same_vs_fp64(r, h, w, c, n, k):
    """Stride-1 "same" convolution, its data gradient and weight gradient as split GEMMs over im2col rows, against
    fp64 conv2d; tolerance 2e-5 of the |x| * |w| convolution (the split bound)."""
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import split_conv_same

    g = torch.Generator().manual_seed(r * 100 + c)
    x = torch.randn(r, h, w, c, generator=g).cuda().requires_grad_(True)
    wt = (torch.randn(n, c, k, k, generator=g) / (c * k * k) ** 0.5).cuda().requires_grad_(True)
    y = split_conv_same(x, wt).view(r, h, w, n)
    gy = torch.randn(r, h, w, n, generator=g).cuda()
    (y * gy).sum().backward()
    xd = x.detach().double().permute(0, 3, 1, 2).requires_grad_(True)
    wd = wt.detach().double().requires_grad_(True)
    ref = F.conv2d(xd, wd, padding=k // 2)
    (ref * gy.double().permute(0, 3, 1, 2)).sum().backward()
    bound = 2e-5 * F.conv2d(xd.detach().abs(), wd.detach().abs(), padding=k // 2) + 1e-6
    assert ((y.detach().double().permute(0, 3, 1, 2) - ref.detach()).abs() <= bound).all()
    assert (x.grad.double().permute(0, 3, 1, 2) - xd.grad).abs().max() <= 2e-5 * xd.grad.abs().max() * 8
    assert (wt.grad.double() - wd.grad).abs().max() <= 2e-5 * wd.grad.abs().max() * 8


def test_im2col_split_layout():
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 3, 4, 8, generator=g).cuda()
    for flip in (False, True):
        rows = _C.im2col_split_bf16x3(x, 3, 3, flip=flip)
        assert rows.shape == (24, 3 * 9 * 8)
        hi = x.to(torch.bfloat16)
        pad = F.pad(hi.permute(0, 3, 1, 2), (1, 1, 1, 1)).permute(0, 2, 3, 1)  # [2, 5, 6, 8]
        for t in range(9):
            ts = 8 - t if flip else t
            want = pad[:, ts // 3: ts // 3 + 3, ts % 3: ts % 3 + 4, :].reshape(24, 8)
            assert torch.equal(rows[:, t * 8:(t + 1) * 8], want)
            assert torch.equal(rows[:, 72 + t * 8: 72 + (t + 1) * 8], want)


def test_res5_head_nhwc_paths_match_conv_path():
    """ResNetHead: the NHWC / GEMM path (fp32 GEMMs and bf16 hi/lo split GEMMs, 3x3 through either layout) against the
    plain per-layer convolution path on the same weights."""
    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.backbone import ResNetHead

    cfg = get_defaults()
    cfg.freeze()
    torch.manual_seed(0)
    head = ResNetHead(cfg).cuda()
    for m in head.modules():  # non-trivial FrozenBN affine so the fold matters
        if hasattr(m, "running_var"):
            m.weight.uniform_(0.5, 1.5)
            m.bias.uniform_(-0.2, 0.2)
    x = torch.randn(24, 1024, 14, 14, device="cuda")

    def run(nhwc, split, c33, sconv=False, pair=False):
        head.nhwc = nhwc
        for b in head.layer4:
            b.split_gemm, b.conv3x3_nchw, b.split_conv, b.pair_gemm = split, c33, sconv, pair
        xx = x.clone().requires_grad_(True)
        y = head(xx)
        head.zero_grad()
        (y * torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)).sum().backward()
        return y.detach(), xx.grad, head.layer4[0].conv1.weight.grad.clone(), head.layer4[2].conv3.weight.grad.clone()

    ref = run(False, False, True)
    for cfg_ in ((True, False, True), (True, False, False), (True, True, True), (True, True, None, True),
                 (True, True, None, True, True)):  # last: pair-layout split GEMM, one autograd node per bottleneck
        got = run(*cfg_)
        assert got[0].shape == ref[0].shape
        assert (got[0] - ref[0]).abs().max().item() <= 2e-4 * ref[0].abs().max().item(), cfg_
        # gradients: a pre-activation within rounding of zero may flip its ReLU gate, which moves isolated entries
        # by O(1) -- compare in the L2 norm
        for a, b in zip(got[1:], ref[1:]):
            assert (a - b).norm().item() <= 5e-3 * b.norm().item(), cfg_  # fp32 NHWC alone: up to 1.1e-3


def test_res5_head_nhwc_empty_and_single_roi():
    """R = 0 (an image without positives) and R = 1 go through the split-GEMM path without special cases."""
    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.backbone import ResNetHead

    cfg = get_defaults()
    cfg.freeze()
    torch.manual_seed(0)
    head = ResNetHead(cfg).cuda()
    y0 = head(torch.zeros(0, 1024, 14, 14, device="cuda", requires_grad=True))
    assert y0.shape == (0, 2048, 7, 7)
    y0.sum().backward()
    x1 = torch.randn(1, 1024, 14, 14, device="cuda")
    head.nhwc = False
    ref = head(x1)
    head.nhwc = True
    got = head(x1)
    assert (got - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
