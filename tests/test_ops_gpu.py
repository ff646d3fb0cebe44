"""GPU parity: HIP kernels (through the C ABI, via cvpr22_cross_modal_pseudo_labeling_amd._C) against
the golden vectors made from the reference and against the CPU oracle on seeded inputs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def C():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    return _C


def _rois(g, r, n_img, img_w, img_h, wmin, wmax):
    b = torch.randint(0, n_img, (r, 1), generator=g).float()
    x1 = torch.rand(r, 1, generator=g) * (img_w * 0.8)
    y1 = torch.rand(r, 1, generator=g) * (img_h * 0.8)
    w = torch.rand(r, 1, generator=g) * (wmax - wmin) + wmin
    h = torch.rand(r, 1, generator=g) * (wmax - wmin) + wmin
    return torch.cat([b, x1, y1, (x1 + w).clamp(max=img_w - 1), (y1 + h).clamp(max=img_h - 1)], 1)


# ---------------------------------------------------------------- RoIAlign forward
def test_roi_align_forward_golden_bit_exact(C, golden_dir):
    z = np.load(os.path.join(golden_dir, "roi_align_forward.npz"))
    x, rois, scale = torch.from_numpy(z["input"]).cuda(), torch.from_numpy(z["rois"]).cuda(), float(z["scale"])
    for key, (ph, pw, sr) in {"out_sr0": (14, 14, 0), "out_sr2": (14, 14, 2), "out_7x7_sr0": (7, 7, 0)}.items():
        got = C.roi_align_forward(x, rois, scale, ph, pw, sr).cpu()
        assert torch.equal(got, torch.from_numpy(z[key])), key  # bit-exact vs the reference CPU kernel


def _close_to_pooled(got, want, x_absmax):
    """Tolerance of the matrix-core forward: operands are split into bf16 hi + lo (16 mantissa bits) and the lo.lo
    product is dropped, so a pooled value is off by at most ~2^-15 of sum |w x| <= max |x| (weights sum to 1)."""
    return (got - want).abs().max().item() <= 4e-5 * x_absmax


def test_roi_align_forward_golden_matrix_core(C, golden_dir):
    z = np.load(os.path.join(golden_dir, "roi_align_forward.npz"))
    x, rois, scale = torch.from_numpy(z["input"]).cuda(), torch.from_numpy(z["rois"]).cuda(), float(z["scale"])
    for key, (ph, pw, sr) in {"out_sr0": (14, 14, 0), "out_sr2": (14, 14, 2), "out_7x7_sr0": (7, 7, 0)}.items():
        got = C.roi_align_forward_mfma(x, rois, scale, ph, pw, sr).cpu()
        assert _close_to_pooled(got, torch.from_numpy(z[key]), x.abs().max().item()), key


@pytest.mark.parametrize("shape", [(2, 70, 50, 84, 300, 14), (1, 33, 13, 17, 40, 7), (3, 16, 100, 168, 64, 14)])
def test_roi_align_forward_vs_oracle(C, oracle_mod, shape):
    n, c, h, w, r, p = shape
    g = torch.Generator().manual_seed(n * 1000 + c)
    x = torch.randn(n, c, h, w, generator=g)
    rois = _rois(g, r, n, w * 16, h * 16, 4, min(w, h) * 16)
    want = oracle_mod.roi_align_forward(x, rois, 1 / 16, p, p, 0)
    got = C.roi_align_forward(x.cuda(), rois.cuda(), 1 / 16, p, p, 0).cpu()
    assert torch.equal(got, want)
    fast = C.roi_align_forward_mfma(x.cuda(), rois.cuda(), 1 / 16, p, p, 0).cpu()  # 13x17 map: falls back to exact
    assert _close_to_pooled(fast, want, x.abs().max().item())


def test_roi_align_forward_matrix_core_edge_rois(C, oracle_mod):
    """Degenerate / clipped / out-of-map / multi-block RoIs and fixed sampling grids through the matrix-core path."""
    rois = torch.tensor([[0, -50.0, -40.0, 30.0, 20.0], [1, 600.0, 300.0, 600.0, 300.0],
                         [0, 650.0, 380.0, 2000.0, 900.0], [1, 5000.0, 5000.0, 5100.0, 5100.0],
                         [0, 0.0, 0.0, 671.0, 399.0], [1, 100.3, 50.7, 101.9, 52.2], [0, 3.0, 200.0, 660.0, 230.0]])
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 70, 25, 42, generator=g)  # 70 channels: a full 64-channel slab + a partial one
    for (p, sr) in [(14, 0), (7, 0), (7, 3), (14, 1)]:
        want = oracle_mod.roi_align_forward(x, rois, 1 / 16, p, p, sr)
        got = C.roi_align_forward_mfma(x.cuda(), rois.cuda(), 1 / 16, p, p, sr).cpu()
        assert _close_to_pooled(got, want, x.abs().max().item()), (p, sr, (got - want).abs().max())
    assert float(got[3].abs().max()) == 0.0  # the out-of-map RoI pools to exact zeros


def test_roi_align_forward_strided_nhwc_bit_exact(C, golden_dir):
    """The pooler fused with the consumer's stride: bins (2i, 2j) only, NHWC -- bit-identical to the same bins of the
    bit-exact forward (all staging paths: 16/8/4-channel LDS batches, one-channel LDS, global gather; 70 channels =
    two full 32-channel tiles + a partial one), and its backward equals the zero-scattered full backward to round-off."""
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import ROIAlign

    z = np.load(os.path.join(golden_dir, "roi_align_forward.npz"))
    x, rois, scale = torch.from_numpy(z["input"]).cuda(), torch.from_numpy(z["rois"]).cuda(), float(z["scale"])
    for (ph, sr, s) in [(14, 0, 2), (14, 2, 2), (7, 0, 2), (14, 0, 3)]:
        full = C.roi_align_forward(x, rois, scale, ph, ph, sr)
        got = C.roi_align_forward_strided_nhwc(x, rois, scale, ph, ph, sr, s)
        assert torch.equal(got, full[:, :, ::s, ::s].permute(0, 2, 3, 1)), (ph, sr, s)
    g = torch.Generator().manual_seed(8)
    xb = torch.randn(2, 70, 50, 84, generator=g).cuda()
    rb = torch.cat([_rois(g, 40, 2, 1333, 800, 8, 900), torch.tensor([[0, 0.0, 0.0, 1332.0, 799.0]])]).cuda()
    full = C.roi_align_forward(xb, rb, 1 / 16, 14, 14, 0)
    assert torch.equal(C.roi_align_forward_strided_nhwc(xb, rb, 1 / 16, 14, 14, 0, 2), full[:, :, ::2, ::2].permute(0, 2, 3, 1))
    big = torch.randn(1, 5, 120, 160, generator=g).cuda()  # whole-map RoI: 19200-cell window -> global gather path
    rbig = torch.tensor([[0, 0.0, 0.0, 2559.0, 1919.0], [0, 30.0, 40.0, 900.0, 700.0]]).cuda()
    assert torch.equal(C.roi_align_forward_strided_nhwc(big, rbig, 1 / 16, 14, 14, 0, 2),
                       C.roi_align_forward(big, rbig, 1 / 16, 14, 14, 0)[:, :, ::2, ::2].permute(0, 2, 3, 1))
    # autograd: same gradient as slicing the full layer's output
    layer = ROIAlign((14, 14), 1 / 16, 0)
    xa = xb[:, :6].clone().requires_grad_(True)
    xc = xb[:, :6].clone().requires_grad_(True)
    go = torch.randn(rb.shape[0], 7, 7, 6, generator=g).cuda()
    (layer.forward_strided_nhwc(xa, rb, 2) * go).sum().backward()
    (layer(xc, rb)[:, :, ::2, ::2].permute(0, 2, 3, 1) * go).sum().backward()
    # the strided backward reads the 7x7 tiles with tables of the strided bins, the full one the zero-scattered 14x14 tile:
    # the same products in a different accumulation order
    assert (xa.grad - xc.grad).abs().max().item() <= 1e-5 * xc.grad.abs().max().item()


@pytest.mark.parametrize("c", [64, 100, 1024])
def test_roi_align_strided_poolers_on_channels_last_map_bit_exact(C, c):
    """A channels-last feature map (NCHW view of NHWC memory: what the trunk hands over) is pooled in place by the
    NHWC-input kernel; bins bit-identical to the NCHW window-staging kernel and to the exact kernel's, fp32 and pair output,
    RoIs of every size incl. one outside the map and non-power-of-two sampling grids."""
    g = torch.Generator().manual_seed(c)
    n, h, w = 2, 50, 84
    x = torch.randn(n, c, h, w, generator=g).cuda()
    x_cl = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)          # same values, NHWC memory
    assert not x_cl.is_contiguous()
    rois = torch.cat([_rois(g, 60, n, 1333, 800, 8, 1200), torch.tensor([[0, 0.0, 0.0, 1332.0, 799.0], [1, 5000.0, 5000.0, 5100.0, 5100.0],
                                                                        [1, 100.0, 50.0, 820.0, 700.0]])]).cuda()
    want = C.roi_align_forward(x, rois, 1 / 16, 14, 14, 0)[:, :, ::2, ::2].permute(0, 2, 3, 1)
    got = C.roi_align_forward_strided_nhwc(x_cl, rois, 1 / 16, 14, 14, 0, 2)
    assert torch.equal(got, want)
    assert torch.equal(got, C.roi_align_forward_strided_nhwc(x, rois, 1 / 16, 14, 14, 0, 2))
    if c % 32 == 0:
        p_cl, shp = C.roi_align_forward_strided_pair(x_cl, rois, 1 / 16, 14, 14, 0, 2)
        p_nc, _ = C.roi_align_forward_strided_pair(x, rois, 1 / 16, 14, 14, 0, 2)
        assert shp == (7, 7) and torch.equal(p_cl, p_nc)
        assert torch.equal(p_cl, C.split_pair(got.reshape(-1, c)))
    for sr, s in ((2, 2), (0, 3), (3, 1)):
        a = C.roi_align_forward_strided_nhwc(x_cl, rois, 1 / 16, 14, 14, sr, s)
        b = C.roi_align_forward(x, rois, 1 / 16, 14, 14, sr)[:, :, ::s, ::s].permute(0, 2, 3, 1)
        assert torch.equal(a, b), (sr, s)


def test_roi_align_forward_large_map_global_path(C, oracle_mod):
    # 120x160 = 19200-cell window for the whole-image RoI: exceeds the LDS window budget
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 3, 120, 160, generator=g)
    rois = torch.tensor([[0, 0.0, 0.0, 2559.0, 1919.0], [0, 100.0, 50.0, 2000.0, 1800.0], [0, 5.0, 5.0, 60.0, 70.0]])
    want = oracle_mod.roi_align_forward(x, rois, 1 / 16, 14, 14, 0)
    got = C.roi_align_forward(x.cuda(), rois.cuda(), 1 / 16, 14, 14, 0).cpu()
    assert torch.equal(got, want)
    fast = C.roi_align_forward_mfma(x.cuda(), rois.cuda(), 1 / 16, 14, 14, 0).cpu()  # 8 x 10 blocks: the generic block loop
    assert _close_to_pooled(fast, want, x.abs().max().item())


def test_roi_align_empty(C):
    x = torch.randn(2, 4, 8, 8, device="cuda")
    out = C.roi_align_forward(x, torch.zeros(0, 5, device="cuda"), 0.25, 7, 7, 2)
    assert C.roi_align_forward_mfma(x, torch.zeros(0, 5, device="cuda"), 0.25, 7, 7, 2).shape == (0, 4, 7, 7)
    assert out.shape == (0, 4, 7, 7)
    gin = C.roi_align_backward(torch.zeros(0, 4, 7, 7, device="cuda"), torch.zeros(0, 5, device="cuda"), 0.25, 7, 7,
                               2, 4, 8, 8, 2)
    assert gin.shape == (2, 4, 8, 8) and float(gin.abs().sum()) == 0.0


# ---------------------------------------------------------------- RoIAlign backward
@pytest.mark.parametrize("shape,sr", [((2, 40, 50, 84, 200, 14), 0), ((2, 8, 25, 42, 64, 14), 2),
                                      ((1, 5, 13, 17, 30, 7), 0)])
def test_roi_align_backward_vs_oracle(C, oracle_mod, shape, sr):
    n, c, h, w, r, p = shape
    g = torch.Generator().manual_seed(17 + c)
    rois = _rois(g, r, n, w * 16, h * 16, 4, min(w, h) * 16)
    go = torch.randn(r, c, p, p, generator=g)
    want = oracle_mod.roi_align_backward(go.double(), rois, 1 / 16, p, p, n, c, h, w, sr, dtype=torch.float64)
    want32 = oracle_mod.roi_align_backward(go, rois, 1 / 16, p, p, n, c, h, w, sr)
    got = C.roi_align_backward(go.cuda(), rois.cuda(), 1 / 16, p, p, n, c, h, w, sr).cpu()
    # fp32 sums in a different order than the reference's atomics: tolerance, stated
    # (north_star: 1e-3 relative); the f32 oracle uses the same f32 weights -> tight.
    assert torch.allclose(got, want32, rtol=1e-4, atol=1e-4)
    scale = want.abs().max().item()
    assert (got.double() - want).abs().max().item() <= 1e-3 * scale


def test_roi_align_backward_golden_rois_adjoint(C, golden_dir):
    z = np.load(os.path.join(golden_dir, "roi_align_forward.npz"))
    rois, scale = torch.from_numpy(z["rois"]).cuda(), float(z["scale"])
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 8, 25, 42, generator=g).cuda()
    go = torch.randn(rois.shape[0], 8, 14, 14, generator=g).cuda()
    f = C.roi_align_forward_mfma(x, rois, scale, 14, 14, 0)
    b = C.roi_align_backward(go, rois, scale, 14, 14, 2, 8, 25, 42, 0)
    lhs = (f.double() * go.double()).sum().item()
    rhs = (x.double() * b.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * (f.double() * go.double()).abs().sum().item()


def test_roi_align_autograd_layer(C, oracle_mod):
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import ROIAlign

    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 6, 20, 30, generator=g)
    rois = _rois(g, 25, 2, 480, 320, 8, 300)
    layer = ROIAlign((14, 14), 1 / 16, 0)
    xd = x.cuda().requires_grad_(True)
    out = layer(xd, rois.cuda())
    go = torch.randn(out.shape, generator=g)
    out.backward(go.cuda())
    want_f = oracle_mod.roi_align_forward(x, rois, 1 / 16, 14, 14, 0)
    assert torch.equal(out.detach().cpu(), want_f)  # the layer's default forward is the bit-exact kernel
    fast = ROIAlign((14, 14), 1 / 16, 0, matrix_core=True)(x.cuda(), rois.cuda())
    assert _close_to_pooled(fast.cpu(), want_f, x.abs().max().item())
    want = oracle_mod.roi_align_backward(go, rois, 1 / 16, 14, 14, 2, 6, 20, 30, 0)
    assert torch.allclose(xd.grad.cpu(), want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(2, 40, 50, 84, 300), (1, 8, 25, 42, 37), (3, 16, 33, 20, 64)])
def test_roi_align_backward_strided_vs_oracle(C, oracle_mod, shape):
    """Backward of the strided pooler on the [R, C, 7, 7] tiles of the computed bins (tables built for bins 2i, 2j) ==
    the oracle's backward of the zero-scattered full [R, C, 14, 14] tile (adjoint-pinned restatement of
    ROIAlign_cuda.cu:178-254), and == the full-tile kernel on the scattered gradient."""
    n, c, h, w, r = shape
    g = torch.Generator().manual_seed(h * w + r)
    rois = _rois(g, r, n, w * 16, h * 16, 16, min(w, h) * 12)
    go = torch.randn(r, c, 7, 7, generator=g)
    full = torch.zeros(r, c, 14, 14)
    full[:, :, ::2, ::2] = go
    got = C.roi_align_backward_strided(go.cuda(), rois.cuda(), 1 / 16, 14, 14, n, c, h, w, 0, 2)
    assert got is not None and got.shape == (n, c, h, w)
    want = oracle_mod.roi_align_backward(full, rois, 1 / 16, 14, 14, n, c, h, w, 0)
    assert torch.allclose(got.cpu(), want, rtol=1e-4, atol=1e-4)
    ref = C.roi_align_backward(full.cuda(), rois.cuda(), 1 / 16, 14, 14, n, c, h, w, 0)
    assert (got - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    # through the layer: forward_strided_nhwc + autograd
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import ROIAlign
    x = torch.randn(n, c, h, w, generator=g).cuda().requires_grad_(True)
    y = ROIAlign((14, 14), 1 / 16, 0).forward_strided_nhwc(x, rois.cuda(), 2)
    y.backward(go.permute(0, 2, 3, 1).contiguous().cuda())
    # the layer hands the NHWC gradient itself to the library (pre-split tiles, two matrix instructions per stage)
    nhwc = C.roi_align_backward_strided_nhwc(go.permute(0, 2, 3, 1).contiguous().cuda(), rois.cuda(), 1 / 16, 14, 14, n, c, h, w, 0, 2)
    assert nhwc is not None and torch.equal(x.grad, nhwc)
    assert torch.allclose(nhwc.cpu(), want, rtol=1e-4, atol=1e-4)
    assert (nhwc - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    again = C.roi_align_backward_strided_nhwc(go.permute(0, 2, 3, 1).contiguous().cuda(), rois.cuda(), 1 / 16, 14, 14, n, c, h, w, 0, 2)
    assert torch.equal(again, nhwc)  # no atomics: bit-reproducible


def test_roi_align_backward_strided_nhwc_edges(C, oracle_mod):
    """The NHWC small-tile form: 8 x 8 tiles (bin stride 1 of an 8 x 8 pooler), odd tile sizes, channel counts that are no
    multiple of 64, empty / foreign RoIs -- and the shapes it refuses (map lower than one block) fall back in the layer."""
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import ROIAlign
    g = torch.Generator().manual_seed(77)
    for (n, c, h, w, r, ph, pw, s) in [(2, 70, 30, 40, 50, 8, 8, 1), (1, 3, 16, 16, 9, 13, 9, 2), (2, 130, 50, 84, 120, 14, 14, 2)]:
        rois = _rois(g, r, n, w * 16, h * 16, 16, min(w, h) * 12)
        rois[0, 3] = rois[0, 1] - 40.0  # a malformed RoI (x2 < x1): contributes like the reference's clamped 1-px box
        th, tw = (ph + s - 1) // s, (pw + s - 1) // s
        go = torch.randn(r, th, tw, c, generator=g)
        full = torch.zeros(r, c, ph, pw)
        full[:, :, ::s, ::s] = go.permute(0, 3, 1, 2)
        got = C.roi_align_backward_strided_nhwc(go.cuda(), rois.cuda(), 1 / 16, ph, pw, n, c, h, w, 0, s)
        assert got is not None
        want = oracle_mod.roi_align_backward(full, rois, 1 / 16, ph, pw, n, c, h, w, 0)
        assert torch.allclose(got.cpu(), want, rtol=1e-4, atol=1e-4), (n, c, h, w, r, ph, pw, s)
    # not covered: tiles above 8 x 8, maps lower than a 16-cell block -> None, and the layer still differentiates
    assert C.roi_align_backward_strided_nhwc(torch.randn(5, 14, 14, 8).cuda(), _rois(g, 5, 1, 640, 320, 16, 200).cuda(),
                                             1 / 16, 14, 14, 1, 8, 20, 40, 0, 1) is None
    rois = _rois(g, 12, 1, 640, 192, 16, 150)
    assert C.roi_align_backward_strided_nhwc(torch.randn(12, 7, 7, 8).cuda(), rois.cuda(), 1 / 16, 14, 14, 1, 8, 12, 40, 0, 2) is None
    x = torch.randn(1, 8, 12, 40, generator=g).cuda().requires_grad_(True)
    y = ROIAlign((14, 14), 1 / 16, 0).forward_strided_nhwc(x, rois.cuda(), 2)
    gy = torch.randn(y.shape, generator=g).cuda()
    y.backward(gy)
    full = torch.zeros(12, 8, 14, 14)
    full[:, :, ::2, ::2] = gy.cpu().permute(0, 3, 1, 2)
    assert torch.allclose(x.grad.cpu(), oracle_mod.roi_align_backward(full, rois, 1 / 16, 14, 14, 1, 8, 12, 40, 0), rtol=1e-4, atol=1e-4)


def test_roi_align_full_size_properties(C):
    # BASELINE shape: [2,1024,50,84], R=1024. Linearity + constant-map invariance (size-independent).
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(2, 1024, 50, 84, generator=g).cuda()
    rois = _rois(g, 1024, 2, 1333, 800, 16, 316).cuda()
    a = C.roi_align_forward_mfma(x, rois, 1 / 16, 14, 14, 0)
    b = C.roi_align_forward_mfma(2 * x, rois, 1 / 16, 14, 14, 0)
    assert torch.equal(b, 2 * a)  # scaling by 2 is exact in fp32 and in the bf16 hi/lo split
    ones = C.roi_align_forward_mfma(torch.ones_like(x), rois, 1 / 16, 14, 14, 0)
    assert torch.allclose(ones, torch.ones_like(ones), atol=1e-6)  # all RoIs inside the map
    go = torch.randn(1024, 1024, 14, 14, generator=g).cuda()
    gi = C.roi_align_backward(go, rois, 1 / 16, 14, 14, 2, 1024, 50, 84, 0)
    # mass conservation: each RoI bin distributes exactly its gradient (weights sum to 1)
    assert torch.allclose(gi.double().sum(), go.double().sum(), rtol=1e-4)


def _rpn_like_rois(g, r, n_img):
    """RoIs with the size mix of RPN proposals (log-uniform areas 32^2..600^2, ratios 1:2..2:1, a few near-full-image)."""
    area = torch.exp(torch.rand(r, generator=g) * (2 * torch.log(torch.tensor(600.0 / 32))) + 2 * torch.log(torch.tensor(32.0)))
    ratio = torch.exp((torch.rand(r, generator=g) - 0.5) * 2 * torch.log(torch.tensor(2.0)))
    w, h = torch.sqrt(area / ratio).clamp(max=1332), torch.sqrt(area * ratio).clamp(max=799)
    x1 = torch.rand(r, generator=g) * (1333 - w)
    y1 = torch.rand(r, generator=g) * (800 - h)
    b = torch.randint(0, n_img, (r,), generator=g).float()
    rois = torch.stack([b, x1, y1, x1 + w - 1, y1 + h - 1], 1)
    rois[:4, 1:] = torch.tensor([[0.0, 0.0, 1332.0, 799.0], [3.5, 2.25, 1320.0, 790.0], [600.0, 0.0, 1332.0, 799.0], [0.0, 300.0, 1332.0, 799.0]])
    return rois


def test_roi_align_step_poolers_full_size_vs_oracle(C, oracle_mod):
    """BASELINE size ([2,1024,50,84], 2000 RPN-like RoIs): the poolers the training step actually runs -- the default
    exact kernel, the strided NHWC form (teacher step) and the strided pair form (student step: small / large window
    launches, bins written as bf16 hi | lo) -- against the oracle (== the reference CPU kernel) on sampled
    (RoI, channel-block) tiles: the oracle pools 8 channels of 96 RoIs, the kernels pool everything.  Bit-exact; the
    pair form is compared as hi + lo with the split's own error bound."""
    g = torch.Generator().manual_seed(77)
    n, c, h, w = 2, 1024, 50, 84
    x = torch.randn(n, c, h, w, generator=g)
    rois = _rpn_like_rois(g, 2000, n)
    xd, rd = x.cuda(), rois.cuda()
    full = C.roi_align_forward(xd, rd, 1 / 16, 14, 14, 0)                              # [R, C, 14, 14]
    nhwc = C.roi_align_forward_strided_nhwc(xd, rd, 1 / 16, 14, 14, 0, 2)              # [R, 7, 7, C]
    pair, (oh, ow) = C.roi_align_forward_strided_pair(xd, rd, 1 / 16, 14, 14, 0, 2)   # [R*49, 2C] bf16
    assert (oh, ow) == (7, 7) and nhwc.shape == (2000, 7, 7, c)
    assert torch.equal(nhwc, full[:, :, ::2, ::2].permute(0, 2, 3, 1))                # strided bins == the exact kernel's
    sel_r = torch.cat([torch.arange(4), torch.randperm(2000, generator=g)[:92]])
    for c0 in (0, 504, 1016):
        want = oracle_mod.roi_align_forward(x[:, c0:c0 + 8].contiguous(), rois[sel_r], 1 / 16, 14, 14, 0)
        assert torch.equal(full[sel_r.cuda(), c0:c0 + 8].cpu(), want), c0
    # pair layout: per 32 channels [hi x32 | lo x32] bf16; hi + lo reproduces the fp32 bin to the split's 2^-17
    pr = pair.view(2000, 49, c // 32, 2, 32).float()
    rec = (pr[:, :, :, 0] + pr[:, :, :, 1]).reshape(2000, 7, 7, c)
    assert torch.equal(pr[:, :, :, 0].reshape(2000, 7, 7, c), nhwc.to(torch.bfloat16).float())
    assert float((rec - nhwc).abs().max()) <= 2.0 ** -16 * float(nhwc.abs().max())


# ---------------------------------------------------------------- NMS
@pytest.mark.parametrize("name", ["rpn_like", "dense", "tiny", "one"])
def test_nms_golden_exact(C, golden_dir, name):
    z = np.load(os.path.join(golden_dir, "nms.npz"))
    boxes, scores = torch.from_numpy(z[f"{name}_boxes"]).cuda(), torch.from_numpy(z[f"{name}_scores"]).cuda()
    keep = C.nms(boxes, scores, float(z[f"{name}_thr"]))
    assert keep.dtype == torch.int64 and keep.is_cuda
    assert torch.equal(keep.cpu(), torch.from_numpy(z[f"{name}_keep"]))


@pytest.mark.parametrize("k,thr", [(6000, 0.7), (12000, 0.7), (4097, 0.5), (63, 0.3), (64, 0.3), (65, 0.3), (12289, 0.7),
                                   (12288, 0.6)])
def test_nms_vs_oracle(C, oracle_mod, k, thr):
    g = torch.Generator().manual_seed(k)
    xy = torch.rand(k, 2, generator=g) * torch.tensor([1200.0, 720.0])
    wh = torch.rand(k, 2, generator=g) * 200 + 8
    boxes = torch.cat([xy, xy + wh], 1)
    scores = torch.rand(k, generator=g)
    want = oracle_mod.nms(boxes, scores, thr)
    got = C.nms(boxes.cuda(), scores.cuda(), thr)
    assert torch.equal(got.cpu(), want)


def test_nms_heavy_overlap_and_ties(C, oracle_mod):
    g = torch.Generator().manual_seed(8)
    k = 3000
    xy = torch.rand(k, 2, generator=g) * 60 + 100   # everything overlaps: long suppression chains
    wh = torch.rand(k, 2, generator=g) * 120 + 40
    boxes = torch.cat([xy, xy + wh], 1)
    scores = (torch.rand(k, generator=g) * 50).floor() / 50  # many exact score ties -> stable order
    want = oracle_mod.nms(boxes, scores, 0.6)
    got = C.nms(boxes.cuda(), scores.cuda(), 0.6)
    assert torch.equal(got.cpu(), want)
    # comparison mode: exact-tie IoU (0.5) kept by `>`, dropped by `>=`
    b2 = torch.tensor([[0.0, 0.0, 9.0, 9.0], [0.0, 0.0, 9.0, 4.0]]).cuda()
    s2 = torch.tensor([0.9, 0.8]).cuda()
    assert C.nms(b2, s2, 0.5).tolist() == [0, 1]
    keep, n = C.nms_padded(b2, s2, 0.5, ge_mode=True)
    assert keep[: int(n)].tolist() == [0]


def test_nms_empty_returns_cpu_like_reference(C):
    out = C.nms(torch.zeros(0, 4, device="cuda"), torch.zeros(0, device="cuda"), 0.5)
    assert out.numel() == 0 and out.dtype == torch.int64 and out.device.type == "cpu"


def test_nms_idempotent_full_size(C):
    g = torch.Generator().manual_seed(1234)
    k = 12000
    xy = torch.rand(k, 2, generator=g) * torch.tensor([1200.0, 720.0])
    wh = torch.rand(k, 2, generator=g) * 200 + 8
    boxes = torch.cat([xy, xy + wh], 1).cuda()
    scores = torch.rand(k, generator=g).cuda()
    keep = C.nms(boxes, scores, 0.7)
    assert bool((keep[1:] > keep[:-1]).all())  # ascending original indices
    again = C.nms(boxes[keep], scores[keep], 0.7)
    assert again.numel() == keep.numel()  # survivors never suppress each other


@pytest.mark.parametrize("k,groups,thr", [(5000, 47, 0.5), (12289, 3, 0.7), (700, 700, 0.5), (64, 1, 0.3)])
def test_nms_grouped_equals_per_class_loop(C, oracle_mod, k, groups, thr):
    """ovis_nms_grouped_f32 == the per-class loop of box_head/inference.py:137-150 (the oracle's NMS on the boxes of
    every class separately), survivors in ascending candidate order; many score ties and heavy overlap."""
    g = torch.Generator().manual_seed(k + groups)
    xy = torch.rand(k, 2, generator=g) * 300
    wh = torch.rand(k, 2, generator=g) * 150 + 20
    boxes = torch.cat([xy, xy + wh], 1)
    scores = (torch.rand(k, generator=g) * 200).floor() / 200
    cls = torch.randint(0, groups, (k,), generator=g)
    want = []
    for j in range(groups):
        idx = (cls == j).nonzero().squeeze(1)
        if idx.numel():
            want.append(idx[oracle_mod.nms(boxes[idx], scores[idx], thr)])
    want = torch.cat(want).sort().values
    got = C.nms_grouped(boxes.cuda(), scores.cuda(), cls.cuda(), thr)
    assert got.dtype == torch.int64 and torch.equal(got.cpu(), want)
    # one group == the plain kernel
    if groups == 1:
        assert torch.equal(got, C.nms(boxes.cuda(), scores.cuda(), thr))


def test_post_processor_grouped_nms_equals_per_class_loop(C):
    """PostProcessor.filter_results on the device (one grouped NMS) returns the detections of the reference's
    per-class loop, in the same order (box_head/inference.py:121-163)."""
    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.roi_heads import PostProcessor
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.structures import BoxList

    cfg = get_defaults()
    cfg.freeze()
    pp = PostProcessor(cfg)
    g = torch.Generator().manual_seed(5)
    n, nc = 1000, 49
    for scale, per_img in ((0.0, 100), (3.0, 100), (3.0, 7)):  # nothing above the threshold / many / more than the cap
        pp.detections_per_img = per_img
        xy = torch.rand(n, nc, 2, generator=g) * 400
        wh = torch.rand(n, nc, 2, generator=g) * 200 + 10
        boxes = torch.cat([xy, xy + wh], 2).reshape(n * nc, 4).cuda()
        scores = torch.softmax(torch.randn(n, nc, generator=g) * scale, 1).reshape(-1).cuda()
        bl = BoxList(boxes, (800, 600))
        bl.add_field("scores", scores)
        got = pp.filter_results(bl, nc)
        want = pp.filter_results_per_class(bl, nc)
        assert len(got) == len(want) and (scale == 0.0 or len(got) > 0)
        assert torch.equal(got.bbox, want.bbox)
        assert torch.equal(got.get_field("scores"), want.get_field("scores"))
        assert torch.equal(got.get_field("labels"), want.get_field("labels"))
        assert got.get_field("labels").dtype == torch.int64


# ---------------------------------------------------------------- sigmoid focal loss
def test_focal_golden(C, golden_dir):
    z = np.load(os.path.join(golden_dir, "sigmoid_focal_loss.npz"))
    for sfx, gamma, alpha in (("", 2.0, 0.25), ("2", 1.5, 0.4)):
        logits, targets = torch.from_numpy(z["logits" + sfx]).cuda(), torch.from_numpy(z["targets" + sfx]).cuda()
        d = torch.from_numpy(z["d_losses" + sfx]).cuda()
        nc = logits.shape[1]
        loss = C.sigmoid_focalloss_forward(logits, targets, nc, gamma, alpha).cpu().double()
        grad = C.sigmoid_focalloss_backward(logits, targets, d, nc, gamma, alpha).cpu().double()
        assert torch.allclose(loss, torch.from_numpy(z["loss" + sfx]), rtol=2e-5, atol=1e-7)
        assert torch.allclose(grad, torch.from_numpy(z["grad" + sfx]), rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("num,c", [(4001, 80), (513, 7), (1, 1)])
def test_focal_vs_oracle_and_module(C, oracle_mod, num, c):
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import SigmoidFocalLoss

    g = torch.Generator().manual_seed(num)
    logits = torch.randn(num, c, generator=g) * 4
    targets = torch.randint(-1, c + 1, (num,), generator=g, dtype=torch.int32)
    want = oracle_mod.sigmoid_focal_loss_forward(logits, targets, 2.0, 0.25)
    ld = logits.cuda().requires_grad_(True)
    got = C.sigmoid_focalloss_forward(ld.detach(), targets.cuda(), c, 2.0, 0.25).cpu()
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-7)
    total = SigmoidFocalLoss(2.0, 0.25)(ld, targets.cuda())
    total.backward()
    wgrad = oracle_mod.sigmoid_focal_loss_backward(logits, targets, torch.ones(num, c), 2.0, 0.25)
    assert torch.allclose(ld.grad.cpu(), wgrad, rtol=1e-5, atol=1e-7)
    assert abs(total.item() - want.double().sum().item()) <= 1e-4 * max(1.0, want.double().sum().item())


@pytest.mark.parametrize("gamma", [2.0, 1.5, 1.0, 0.0, 3.0])
def test_focal_extreme_logits_and_gammas_vs_oracle(C, oracle_mod, gamma):
    """Saturated logits (|x| up to 120: p underflows, the reference's log(max(p, FLT_MIN)) clamp takes over), x = 0,
    every gamma branch of the kernel, positive / negative / ignored rows -- against the C restatement of the CUDA formula
    (SigmoidFocalLoss_cuda.cu:21-101), forward and backward."""
    vals = torch.tensor([-120.0, -95.0, -88.0, -87.0, -60.0, -20.0, -3.0, -1e-3, 0.0, 1e-3, 2.5, 17.0, 40.0, 89.0, 110.0, 0.7])
    num = vals.numel()
    logits = vals[:, None].repeat(1, 4).contiguous()              # [16, 4]: column 1 is the positive class where t == 2
    for t in (2, 0, -1):
        targets = torch.full((num,), t, dtype=torch.int32)
        up = torch.linspace(0.5, 1.5, num * 4).view(num, 4)
        want = oracle_mod.sigmoid_focal_loss_forward(logits, targets, gamma, 0.25)
        wgrad = oracle_mod.sigmoid_focal_loss_backward(logits, targets, up, gamma, 0.25)
        got = C.sigmoid_focalloss_forward(logits.cuda(), targets.cuda(), 4, gamma, 0.25).cpu()
        grad = C.sigmoid_focalloss_backward(logits.cuda(), targets.cuda(), up.cuda(), 4, gamma, 0.25).cpu()
        assert bool(torch.isfinite(got).all()) and bool(torch.isfinite(grad).all())
        assert torch.allclose(got, want, rtol=2e-5, atol=1e-7), (t, (got - want).abs().max())
        assert torch.allclose(grad, wgrad, rtol=2e-5, atol=1e-7), (t, (grad - wgrad).abs().max())


def test_roi_align_backward_plane_kernel_reproducible(C, oracle_mod):
    """The plane-owner matrix-core kernel (default for maps whose plane fits LDS) issues no atomics: two runs are
    bit-identical; RoIs up to 500 px exercise multi-block windows (several 16-cell blocks per axis)."""
    g = torch.Generator().manual_seed(5)
    n, c, h, w, r = 2, 10, 50, 84, 300
    b = torch.randint(0, n, (r, 1), generator=g).float()
    xy = torch.rand(r, 2, generator=g) * torch.tensor([1000.0, 600.0])
    wh = torch.rand(r, 2, generator=g) * 500 + 4
    rois = torch.cat([b, xy, (xy + wh).clamp(max=1300)], 1)
    go = torch.randn(r, c, 14, 14, generator=g)
    a = C.roi_align_backward(go.cuda(), rois.cuda(), 1 / 16, 14, 14, n, c, h, w, 0)
    b2 = C.roi_align_backward(go.cuda(), rois.cuda(), 1 / 16, 14, 14, n, c, h, w, 0)
    want = oracle_mod.roi_align_backward(go, rois, 1 / 16, 14, 14, n, c, h, w, 0)
    assert torch.equal(a, b2), "not reproducible"
    assert torch.allclose(a.cpu(), want, rtol=1e-4, atol=1e-4), (a.cpu() - want).abs().max()


def test_roi_align_backward_edge_rois(C, oracle_mod):
    """Degenerate / clipped / out-of-map RoIs, odd pooled width, fixed sampling grids."""
    rois = torch.tensor([[0, -50.0, -40.0, 30.0, 20.0],       # hangs over the top-left corner
                         [1, 600.0, 300.0, 600.0, 300.0],      # zero size -> clamped to one cell
                         [0, 650.0, 380.0, 2000.0, 900.0],     # runs past the bottom-right corner
                         [1, 5000.0, 5000.0, 5100.0, 5100.0],  # entirely outside: contributes nothing
                         [0, 0.0, 0.0, 671.0, 399.0],          # the whole map
                         [1, 100.3, 50.7, 101.9, 52.2]])       # sub-cell RoI: all bins in one or two cells
    g = torch.Generator().manual_seed(11)
    for (p, sr) in [(14, 0), (7, 0), (7, 3), (14, 1)]:
        go = torch.randn(rois.shape[0], 3, p, p, generator=g)
        want = oracle_mod.roi_align_backward(go, rois, 1 / 16, p, p, 2, 3, 25, 42, sr)
        got = C.roi_align_backward(go.cuda(), rois.cuda(), 1 / 16, p, p, 2, 3, 25, 42, sr).cpu()
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-4), (p, sr, (got - want).abs().max())


def test_roi_align_backward_large_map_fallback(C, oracle_mod):
    """A plane larger than the LDS budget (120 x 100 f32 = 48 KB) takes the window-gather + atomic kernel."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _lib
    L = _lib.load()
    assert L.ovis_roi_align_backward_plane_supported(50, 84, 14, 14) == 1
    assert L.ovis_roi_align_backward_plane_supported(120, 100, 14, 14) == 0
    g = torch.Generator().manual_seed(3)
    rois = _rois(g, 40, 1, 1600, 1900, 8, 600)
    go = torch.randn(40, 4, 14, 14, generator=g)
    want = oracle_mod.roi_align_backward(go, rois, 1 / 16, 14, 14, 1, 4, 120, 100, 0)
    got = C.roi_align_backward(go.cuda(), rois.cuda(), 1 / 16, 14, 14, 1, 4, 120, 100, 0).cpu()
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(2, 8, 25, 42, 40, 7, 7), (1, 3, 10, 12, 9, 3, 3), (2, 16, 50, 84, 64, 14, 14)])
def test_roi_pool_vs_oracle_exact(C, oracle_mod, shape):
    """ROIPool forward (values AND argmax) and backward against the C restatement of ROIPool_cuda.cu; integer outputs
    exact, values exact (a max), backward 1e-6 (atomic summation order)."""
    n, c, h, w, r, ph, pw = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, c, h, w, generator=g)
    xy = torch.rand(r, 2, generator=g) * torch.tensor([w * 16.0 - 40, h * 16.0 - 40])
    wh = torch.rand(r, 2, generator=g) * 300 + 1
    rois = torch.cat([torch.randint(0, n, (r, 1), generator=g).float(), xy, xy + wh], 1)
    rois[0, 1:] = torch.tensor([-30.0, -20.0, 5.0, 8.0])             # clipped
    rois[1, 1:] = torch.tensor([w * 16.0 + 50, 10.0, w * 16.0 + 90, 40.0])  # outside: empty bins
    rois[2, 1:] = torch.tensor([100.0, 100.0, 90.0, 95.0])            # malformed -> 1x1
    out, arg = C.roi_pool_forward(x.cuda(), rois.cuda(), 1 / 16, ph, pw)
    o_ref, a_ref = oracle_mod.roi_pool_forward(x, rois, 1 / 16, ph, pw)
    assert torch.equal(out.cpu(), o_ref) and torch.equal(arg.cpu(), a_ref)
    assert arg.dtype == torch.int32 and (a_ref == -1).any()
    gout = torch.randn(out.shape, generator=g)
    gin = C.roi_pool_backward(gout.cuda(), x.cuda(), rois.cuda(), arg, 1 / 16, ph, pw, n, c, h, w)
    g_ref = oracle_mod.roi_pool_backward(gout, a_ref, rois, n, c, h, w)
    assert (gin.cpu() - g_ref).abs().max().item() <= 1e-5 * max(1.0, g_ref.abs().max().item())


def test_roi_pool_layer_autograd_and_empty(C):
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import ROIPool, roi_pool
    x = torch.randn(1, 4, 20, 30, device="cuda", requires_grad=True)
    rois = torch.tensor([[0, 16.0, 16.0, 200.0, 150.0], [0, 0.0, 0.0, 479.0, 319.0]], device="cuda")
    y = ROIPool((7, 7), 1 / 16)(x, rois)
    assert y.shape == (2, 4, 7, 7)
    y.sum().backward()
    # every pooled element sends gradient 1 to exactly one input element
    assert abs(float(x.grad.sum()) - y.numel()) < 1e-3
    full = roi_pool(x.detach(), rois[1:], (1, 1), 1 / 16)
    assert torch.equal(full.view(4), x.detach().view(4, -1).max(1).values)
    y0 = roi_pool(x.detach(), rois[:0], (7, 7), 1 / 16)
    assert y0.shape == (0, 4, 7, 7)
    with pytest.raises(RuntimeError):
        roi_pool(x.detach().cpu(), rois.cpu(), (7, 7), 1 / 16)
