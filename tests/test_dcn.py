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
