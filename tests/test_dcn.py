"""Deformable convolution: CPU pins of the fp64 oracle (analytic identities) and GPU parity of the HIP path
(im2col / col2im / col2im_coord kernels + MFMA GEMMs, through the reference's `_C` signatures and autograd
functions) against that oracle."""
import pytest
import torch
import torch.nn.functional as F

from oracle import dcn as O


def _case(g, B=2, C=8, Cout=6, H=9, W=11, k=3, stride=1, pad=1, dil=1, groups=1, dg=1, dtype=torch.float64):
    x = torch.randn(B, C, H, W, generator=g, dtype=dtype)
    w = torch.randn(Cout, C // groups, k, k, generator=g, dtype=dtype) * 0.2
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    off = torch.randn(B, dg * 2 * k * k, Ho, Wo, generator=g, dtype=dtype) * 1.5
    mask = torch.sigmoid(torch.randn(B, dg * k * k, Ho, Wo, generator=g, dtype=dtype))
    bias = torch.randn(Cout, generator=g, dtype=dtype)
    return x, w, off, mask, bias


def test_oracle_zero_offset_is_conv2d():
    g = torch.Generator().manual_seed(0)
    for kw in (dict(), dict(stride=2, pad=2, dil=2), dict(groups=2, dg=2)):
        x, w, off, mask, bias = _case(g, **kw)
        args = dict(stride=(kw.get("stride", 1),) * 2, padding=(kw.get("pad", 1),) * 2, dilation=(kw.get("dil", 1),) * 2,
                    groups=kw.get("groups", 1), deformable_groups=kw.get("dg", 1))
        got = O.deform_conv2d(x, off * 0, w, **args)
        want = F.conv2d(x, w, None, args["stride"], args["padding"], args["dilation"], args["groups"])
        assert torch.allclose(got, want, atol=1e-10)
        # modulated with mask == 1 equals v1; integer offsets equal a shifted convolution tap-wise
        assert torch.allclose(O.deform_conv2d(x, off, w, mask=torch.ones_like(mask), **args),
                              O.deform_conv2d(x, off, w, **args), atol=1e-12)


def test_oracle_far_outside_samples_are_zero():
    g = torch.Generator().manual_seed(1)
    x, w, off, mask, bias = _case(g)
    out = O.deform_conv2d(x, off * 0 + 1000.0, w, padding=(1, 1))
    assert float(out.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(stride=2, pad=2, dil=2), dict(groups=2, dg=2, C=8, Cout=4),
                                dict(B=3, C=4, Cout=5, H=7, W=6, k=1, pad=0), dict(B=4, C=6, Cout=6, H=12, W=10, dg=3)])
def test_deform_conv_v1_v2_vs_oracle(kw):
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import deform_conv, modulated_deform_conv

    g = torch.Generator().manual_seed(42)
    x, w, off, mask, bias = _case(g, **kw)
    st, pd, dl = (kw.get("stride", 1),) * 2, (kw.get("pad", 1),) * 2, (kw.get("dil", 1),) * 2
    gr, dg = kw.get("groups", 1), kw.get("dg", 1)
    for modulated in (False, True):
        xr, wr, orr = x.clone().requires_grad_(True), w.clone().requires_grad_(True), off.clone().requires_grad_(True)
        mr, br = mask.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        want = O.deform_conv2d(xr, orr, wr, mr if modulated else None, br if modulated else None, st, pd, dl, gr, dg)
        go = torch.randn(want.shape, generator=g, dtype=torch.float64)
        want.backward(go)
        xd, wd, od = (t.detach().float().cuda().requires_grad_(True) for t in (x, w, off))
        md, bd = (t.detach().float().cuda().requires_grad_(True) for t in (mask, bias))
        if modulated:
            got = modulated_deform_conv(xd, od, md, wd, bd, st, pd, dl, gr, dg)
        else:
            got = deform_conv(xd, od, wd, st, pd, dl, gr, dg, 2 if x.shape[0] % 2 == 0 else 1)
        got.backward(go.float().cuda())
        scale = want.abs().max().item()
        assert (got.detach().cpu().double() - want.detach()).abs().max().item() <= 1e-5 * scale  # 1e-3 rel: north_star
        pairs = [(xd, xr), (wd, wr), (od, orr)] + ([(md, mr), (bd, br)] if modulated else [])
        for dev_t, ref_t in pairs:
            s = ref_t.grad.abs().max().item() + 1e-12
            assert (dev_t.grad.cpu().double() - ref_t.grad).abs().max().item() <= 2e-5 * s


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(C=64, Cout=12), dict(C=64, Cout=64, dg=2, stride=2, pad=2, dil=2),
                                dict(B=3, C=96, Cout=8, H=7, W=6, k=1, pad=0, dg=3), dict(B=1, C=32, Cout=132, H=21, W=19)])
def test_deform_conv_forward_without_column_buffer_vs_oracle(kw):
    """Channel counts the implicit-GEMM route takes ((C / deformable_group) % 32 == 0): forward of v1 and modulated v2 with
    the A tiles sampled inside the split GEMM kernel -- against the fp64 oracle within the three-term split product's
    3e-5 of the maximum, and against the column route (im2col + fp32 GEMM) of the same call; the backward (column route
    either way) against the oracle as in the test above."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import deform_conv, modulated_deform_conv

    g = torch.Generator().manual_seed(7)
    x, w, off, mask, bias = _case(g, **kw)
    st, pd, dl = (kw.get("stride", 1),) * 2, (kw.get("pad", 1),) * 2, (kw.get("dil", 1),) * 2
    dg = kw.get("dg", 1)
    calls = []
    orig = _C._L.ovis_deform_conv_implicit_f32
    for modulated in (False, True):
        xr, wr, orr = x.clone().requires_grad_(True), w.clone().requires_grad_(True), off.clone().requires_grad_(True)
        want = O.deform_conv2d(xr, orr, wr, mask if modulated else None, bias if modulated else None, st, pd, dl, 1, dg)
        go = torch.randn(want.shape, generator=g, dtype=torch.float64)
        want.backward(go)

        def run():
            xd, wd, od = (t.detach().float().cuda().requires_grad_(True) for t in (x, w, off))
            if modulated:
                y = modulated_deform_conv(xd, od, mask.float().cuda(), wd, bias.float().cuda(), st, pd, dl, 1, dg)
            else:
                y = deform_conv(xd, od, wd, st, pd, dl, 1, dg, 1)
            y.backward(go.float().cuda())
            return y.detach().cpu().double(), [t.grad.cpu().double() for t in (xd, wd, od)]

        n0 = len(calls)
        _C._L.ovis_deform_conv_implicit_f32 = lambda *a: (calls.append(1), orig(*a))[1]
        try:
            got, grads = run()
        finally:
            _C._L.ovis_deform_conv_implicit_f32 = orig
        assert len(calls) == n0 + 1  # the forward really took the implicit route
        _C.dcn_implicit = False
        try:
            col, _ = run()
        finally:
            _C.dcn_implicit = True
        scale = want.abs().max().item()
        assert (got - want.detach()).abs().max().item() <= 3e-5 * scale
        assert (got - col).abs().max().item() <= 3e-5 * scale
        for dev_g, ref_t in zip(grads, (xr, wr, orr)):
            assert (dev_g - ref_t.grad).abs().max().item() <= 2e-5 * (ref_t.grad.abs().max().item() + 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(C=64, Cout=64, dg=2, stride=2, pad=2, dil=2), dict(C=128, Cout=128, H=12, W=10),
                                dict(B=3, C=96, Cout=32, H=7, W=6, k=1, pad=0, dg=3), dict(B=1, C=32, Cout=160, H=21, W=19),
                                dict(B=1, C=256, Cout=32, H=9, W=8), dict(B=2, C=512, Cout=32, H=6, W=7, dg=2, stride=2)])
def test_deform_conv_backward_on_rows_vs_oracle(kw):
    """Channel counts the rows route takes ((C / deformable_group) % 32 == 0 and C_out % 32 == 0): the backward of v1 and
    modulated v2 WITHOUT the reference-layout column buffer -- dcol = dY . W on the split GEMM, one scatter / gather pass
    for dX / dOffset / dMask, the weight gradient as a transpose-read split GEMM (C=128: the kernel; otherwise its library
    fallback) over the sampled pair rows -- against the fp64 oracle within the three-term product's 3e-5 of the maximum,
    and against the column route of the same call."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import deform_conv, modulated_deform_conv

    g = torch.Generator().manual_seed(19)
    x, w, off, mask, bias = _case(g, **kw)
    st, pd, dl = (kw.get("stride", 1),) * 2, (kw.get("pad", 1),) * 2, (kw.get("dil", 1),) * 2
    dg = kw.get("dg", 1)
    calls = {"col2im_rows": 0, "im2col_rows": 0}
    o1, o2 = _C._L.ovis_deform_col2im_rows_f32, _C._L.ovis_deform_im2col_pair_rows_f32
    for modulated in (False, True):
        xr, wr, orr = x.clone().requires_grad_(True), w.clone().requires_grad_(True), off.clone().requires_grad_(True)
        mr, br = mask.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        want = O.deform_conv2d(xr, orr, wr, mr if modulated else None, br if modulated else None, st, pd, dl, 1, dg)
        go = torch.randn(want.shape, generator=g, dtype=torch.float64)
        want.backward(go)

        def run():
            xd, wd, od = (t.detach().float().cuda().requires_grad_(True) for t in (x, w, off))
            md, bd = (t.detach().float().cuda().requires_grad_(True) for t in (mask, bias))
            if modulated:
                y = modulated_deform_conv(xd, od, md, wd, bd, st, pd, dl, 1, dg)
            else:
                y = deform_conv(xd, od, wd, st, pd, dl, 1, dg, 1)
            y.backward(go.float().cuda())
            return [t.grad.cpu().double() for t in ((xd, wd, od) + ((md, bd) if modulated else ()))]

        n0 = dict(calls)
        _C._L.ovis_deform_col2im_rows_f32 = lambda *a: (calls.__setitem__("col2im_rows", calls["col2im_rows"] + 1), o1(*a))[1]
        _C._L.ovis_deform_im2col_pair_rows_f32 = lambda *a: (calls.__setitem__("im2col_rows", calls["im2col_rows"] + 1), o2(*a))[1]
        try:
            grads = run()
        finally:
            _C._L.ovis_deform_col2im_rows_f32, _C._L.ovis_deform_im2col_pair_rows_f32 = o1, o2
        assert calls["col2im_rows"] == n0["col2im_rows"] + 1 and calls["im2col_rows"] == n0["im2col_rows"] + 1
        _C.dcn_implicit = False
        try:
            col_grads = run()
        finally:
            _C.dcn_implicit = True
        refs = [xr, wr, orr] + ([mr, br] if modulated else [])
        for got, col, ref_t in zip(grads, col_grads, refs):
            s_ = ref_t.grad.abs().max().item() + 1e-12
            assert (got - ref_t.grad).abs().max().item() <= 3e-5 * s_
            assert (got - col).abs().max().item() <= 3e-5 * s_


@pytest.mark.gpu
def test_dcn_modules_and_dfconv():
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import DFConv2d, ModulatedDeformConvPack

    torch.manual_seed(0)
    x = torch.randn(2, 8, 10, 12, device="cuda")
    for mod in (True, False):
        layer = DFConv2d(8, 6, with_modulated_dcn=mod).cuda()
        y = layer(x)
        assert y.shape == (2, 6, 10, 12)
        y.sum().backward()
        assert layer.conv.weight.grad is not None and layer.offset.weight.grad is not None
        assert layer(x[:0]).shape == (0, 6, 10, 12)
    # zero-initialised offset/mask conv: Pack == 0.5 * ordinary convolution (+ bias), mask = sigmoid(0)
    pack = ModulatedDeformConvPack(8, 6, 3, padding=1).cuda()
    want = 0.5 * F.conv2d(x, pack.weight, None, 1, 1) + pack.bias.view(1, -1, 1, 1)
    assert torch.allclose(pack(x), want, rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------- deformable position-sensitive RoI pooling
def _psroi_case(seed=0, n=6, od=4, gs=2, P=4, ps=None, H=14, W=18, classes=2, B=2):
    g = torch.Generator().manual_seed(seed)
    C = od * gs * gs
    data = torch.randn(B, C, H, W, generator=g)
    xy = torch.rand(n, 2, generator=g) * torch.tensor([W * 8.0, H * 8.0])
    wh = torch.rand(n, 2, generator=g) * 90 + 20
    rois = torch.cat([torch.randint(0, B, (n, 1), generator=g).float(), xy, xy + wh], 1)
    ps = P if ps is None else ps
    trans = torch.randn(n, 2 * classes, ps, ps, generator=g) * 0.5
    return data, rois, trans


def test_psroi_oracle_constant_and_linear_maps():
    """Analytic pins of the oracle (deform_pool_kernel_cuda.cu:31-139): a constant map pools to the constant wherever a
    sample falls inside; on f = a x + b y + c bilinear interpolation is exact, so an interior bin pools to f at the
    mean sample position, which follows in closed form from the RoI (start = round(x1) * scale - 0.5, ...)."""
    from oracle.dcn import deform_psroi_pool

    P, spp, scale = 3, 4, 0.25
    H, W = 40, 50
    rois = torch.tensor([[0, 16.0, 24.0, 111.0, 95.0], [0, 40.3, 8.2, 80.9, 60.4]])
    const = torch.full((1, 2, H, W), 3.5)
    out, cnt = deform_psroi_pool(const, rois, None, scale, P, 2, True, 1, None, spp, 0.0)
    assert bool((cnt == spp * spp).all()) and torch.allclose(out, torch.full_like(out, 3.5))
    a, b, c0 = 0.7, -1.3, 2.0
    ramp = (a * torch.arange(W).view(1, 1, 1, W) + b * torch.arange(H).view(1, 1, H, 1) + c0).expand(1, 2, H, W).contiguous()
    out, _ = deform_psroi_pool(ramp, rois, None, scale, P, 2, True, 1, None, spp, 0.0)
    for i, (_, x1, y1, x2, y2) in enumerate(rois.tolist()):
        sw, sh = round(x1) * scale - 0.5, round(y1) * scale - 0.5
        rw, rh = (round(x2) + 1) * scale - 0.5 - sw, (round(y2) + 1) * scale - 0.5 - sh
        for ph in range(P):
            for pw in range(P):
                mw = sw + pw * rw / P + (spp - 1) / 2 * rw / P / spp
                mh = sh + ph * rh / P + (spp - 1) / 2 * rh / P / spp
                assert abs(float(out[i, 0, ph, pw]) - (a * mw + b * mh + c0)) < 1e-4


def test_psroi_oracle_offset_equals_shifted_roi_and_finite_differences():
    """trans * trans_std * roi_width is a shift of the sampling window: the same value as pooling a RoI moved by that
    many feature pixels; the autograd gradients (the oracle's backward) agree with central finite differences."""
    from oracle.dcn import deform_psroi_pool

    P, spp, scale, std = 2, 2, 0.5, 0.1
    H, W = 30, 30
    data, _, _ = _psroi_case(3, od=1, gs=1, H=H, W=W, B=1)
    roi = torch.tensor([[0, 10.0, 12.0, 29.0, 31.0]])         # start 4.5 / 5.5, width = height = 10 feature px
    shift_px = 3.0                                            # feature pixels
    t = shift_px / (std * 10.0)
    trans = torch.full((1, 2, P, P), t)
    got, _ = deform_psroi_pool(data, roi, trans, scale, P, 1, False, 1, None, spp, std)
    moved = roi + torch.tensor([[0, 1, 1, 1, 1]]) * (shift_px / scale)
    want, _ = deform_psroi_pool(data, moved, None, scale, P, 1, True, 1, None, spp, 0.0)
    assert torch.allclose(got, want, atol=1e-5)
    # finite differences in fp64 data; offsets perturbed away from cell boundaries
    d = data.double().requires_grad_(True)
    tr = (torch.randn(1, 2, P, P, generator=torch.Generator().manual_seed(1)) * 0.3).requires_grad_(True)
    out, _ = deform_psroi_pool(d, roi, tr, scale, P, 1, False, 1, None, spp, std)
    wgt = torch.randn(out.shape, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    gd, gt = torch.autograd.grad((out * wgt).sum(), [d, tr])
    eps = 1e-3
    for idx in [(0, 0, 0, 1), (0, 1, 1, 0)]:
        tp, tm = tr.detach().clone(), tr.detach().clone()
        tp[idx] += eps
        tm[idx] -= eps
        fp = (deform_psroi_pool(d.detach(), roi, tp, scale, P, 1, False, 1, None, spp, std)[0] * wgt).sum()
        fm = (deform_psroi_pool(d.detach(), roi, tm, scale, P, 1, False, 1, None, spp, std)[0] * wgt).sum()
        assert abs(float((fp - fm) / (2 * eps)) - float(gt[idx])) < 2e-2 * max(1.0, abs(float(gt[idx])))
    k = (0, 0, 9, 9)
    dp = d.detach().clone()
    dp[k] += 1.0  # the op is linear in the data
    f1 = (deform_psroi_pool(dp, roi, tr.detach(), scale, P, 1, False, 1, None, spp, std)[0] * wgt).sum()
    f0 = (deform_psroi_pool(d.detach(), roi, tr.detach(), scale, P, 1, False, 1, None, spp, std)[0] * wgt).sum()
    assert abs(float(f1 - f0) - float(gd[k])) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(no_trans=True, gs=1, P=7, od=8), dict(no_trans=False, gs=2, P=4, od=4, classes=2),
                                dict(no_trans=False, gs=1, P=6, od=6, classes=3, ps=3, spp=2),
                                dict(no_trans=False, gs=3, P=3, od=2, classes=1, std=1.0, far=True)])
def test_deform_psroi_pooling_vs_oracle(kw):
    """HIP deform_roi_pooling (forward, sample counts, both gradients) vs the oracle: values 1e-5, counts exact,
    gradients 1e-4 (fp32 atomics); RoIs partly and wholly outside the map included (count 0 -> output 0, no gradient)."""
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import deform_roi_pooling
    from oracle.dcn import deform_psroi_pool

    no_trans, gs, P, od = kw["no_trans"], kw["gs"], kw["P"], kw["od"]
    classes, ps, spp, std = kw.get("classes", 1), kw.get("ps"), kw.get("spp", 4), kw.get("std", 0.1)
    data, rois, trans = _psroi_case(od + P, n=9, od=od, gs=gs, P=P, ps=ps, classes=classes)
    if kw.get("far"):
        rois[0, 1:] = torch.tensor([900.0, 900.0, 980.0, 990.0])   # wholly outside the 18x14 map (scale 1/8)
        rois[1, 1:] = torch.tensor([-60.0, -40.0, 30.0, 20.0])     # partly outside
    scale = 0.125
    d0 = data.double().requires_grad_(True)
    t0 = trans.clone().requires_grad_(True)
    want, want_cnt = deform_psroi_pool(d0, rois, None if no_trans else t0, scale, P, od, no_trans, gs, ps, spp, std)
    gout = torch.randn(want.shape, generator=torch.Generator().manual_seed(5))
    gw = torch.autograd.grad(want, [d0] if no_trans else [d0, t0], gout.double())
    d1 = data.cuda().requires_grad_(True)
    t1 = (trans.cuda() if not no_trans else torch.empty(0, device="cuda")).requires_grad_(not no_trans)
    got = deform_roi_pooling(d1, rois.cuda(), t1, scale, P, od, no_trans, gs, ps, spp, std)
    assert got.shape == want.shape
    assert torch.allclose(got.cpu().double(), want.detach(), rtol=1e-5, atol=1e-5)
    got.backward(gout.cuda())
    assert (d1.grad.cpu().double() - gw[0]).abs().max().item() <= 1e-4 * max(1.0, gw[0].abs().max().item())
    if not no_trans:
        assert (t1.grad.cpu().double() - gw[1].double()).abs().max().item() <= 2e-4 * max(1.0, gw[1].abs().max().item())
    if kw.get("far"):
        assert float(want_cnt[0].sum()) == 0 and float(got[0].abs().sum()) == 0


@pytest.mark.gpu
def test_deform_roi_pooling_modules_and_empty():
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import (DeformRoIPooling, DeformRoIPoolingPack,
                                                                ModulatedDeformRoIPoolingPack, deform_roi_pooling)

    data, rois, _ = _psroi_case(11, n=5, od=8, gs=1, P=7)
    data, rois = data.cuda(), rois.cuda()
    plain = DeformRoIPooling(0.125, 7, 8, True)(data, rois, None)
    for cls in (DeformRoIPoolingPack, ModulatedDeformRoIPoolingPack):
        m = cls(0.125, 7, 8, False, trans_std=0.1, deform_fc_channels=64).cuda()
        y = m(data, rois)
        assert y.shape == (5, 8, 7, 7)
        # zero-initialised offset (and mask logit) layers: no shift, mask = sigmoid(0) = 0.5
        ref = plain * (0.5 if cls is ModulatedDeformRoIPoolingPack else 1.0)
        assert torch.allclose(y, ref, atol=1e-6)
        y.sum().backward()
        assert m.offset_fc[0].weight.grad is not None
    out = deform_roi_pooling(data, rois[:0], torch.empty(0, device="cuda"), 0.125, 7, 8, True, 1, None, 4, 0.0)
    assert out.shape == (0, 8, 7, 7)
    with pytest.raises(NotImplementedError):
        deform_roi_pooling(data.cpu(), rois.cpu(), torch.empty(0), 0.125, 7, 8, True, 1, None, 4, 0.0)
