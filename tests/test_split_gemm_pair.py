"""GPU: pair-layout split GEMM / implicit-GEMM convolution (csrc/split_gemm.hip) and the fused bottleneck node
(layers/pair_bottleneck.py) against fp64 references.  Tolerance: |err| <= 2e-5 * sum_k |a_k||b_k| (the three-term
bf16 hi/lo product drops lo.lo and rounds lo to 8 bits: ~4e-6 relative per term; an fp32 GEMM sits at ~2e-6)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _C():
    from cvpr22_cross_modal_pseudo_labeling_amd import _C as c
    return c


def _unpair(p):
    """pair layout [rows, 2K] bf16 -> (hi, lo) [rows, K] each."""
    rows, k2 = p.shape
    v = p.view(rows, k2 // 64, 2, 32)
    return v[:, :, 0, :].reshape(rows, -1), v[:, :, 1, :].reshape(rows, -1)


def test_split_pair_layout_and_precision():
    C = _C()
    x = torch.randn(37, 96, device="cuda") * 3
    p = C.split_pair(x)
    assert p.shape == (37, 192) and p.dtype == torch.bfloat16
    hi, lo = _unpair(p)
    assert torch.equal(hi, x.to(torch.bfloat16))
    assert torch.equal(lo, (x - hi.float()).to(torch.bfloat16))
    assert ((hi.double() + lo.double()) - x.double()).abs().max().item() <= 2.0 ** -16 * x.abs().max().item()
    # row-strided source view
    big = torch.randn(20, 128, device="cuda")
    assert torch.equal(C.split_pair(big[:, 32:96]), C.split_pair(big[:, 32:96].contiguous()))
    with pytest.raises(RuntimeError):
        C.split_pair(torch.randn(4, 40, device="cuda"))  # cols % 32 != 0


# config: 0 = the launcher's choice, 1 / 2 = one / two LDS stages forced, 8 = no K slices (include/ovis_hip.h)
@pytest.mark.parametrize("config", [0, 1, 2, 8])
@pytest.mark.parametrize("m,k,n,bias,res,relu", [(128, 32, 128, False, False, False), (300, 64, 64, False, False, False),
                                                   (1000, 512, 192, True, True, True), (1813, 1024, 512, True, False, True),
                                                   (513, 2048, 2048, False, True, False), (1, 32, 4, True, False, False),
                                                   (490, 2048, 512, True, True, True), (2100, 160, 64, True, False, True),
                                                   (777, 1024, 76, True, False, False)])
def test_split_gemm_pair_vs_fp64(config, m, k, n, bias, res, relu):
    C = _C()
    g = torch.Generator(device="cuda").manual_seed(m * 7 + k + n)
    a = torch.randn(m, k, device="cuda", generator=g)
    b = torch.randn(n, k, device="cuda", generator=g)
    bi = torch.randn(n, device="cuda", generator=g) if bias else None
    r = torch.randn(m, n, device="cuda", generator=g) if res else None
    c, cp = C.split_gemm_pair(C.split_pair(a), C.split_pair(b), bi, r, relu, True, n % 32 == 0, config=config)
    ref = a.double() @ b.double().t()
    if bias:
        ref += bi.double()
    if res:
        ref += r.double()
    if relu:
        ref = ref.clamp(min=0)
    bound = a.abs().double() @ b.abs().double().t() + 1
    assert ((c.double() - ref).abs() / bound).max().item() < TOL
    if cp is not None:  # the fused epilogue split is bit-identical to splitting the fp32 result
        assert torch.equal(cp, C.split_pair(c))


# config: 4 = shifted-row form even where the halo form applies
@pytest.mark.parametrize("config", [0, 1, 2, 4, 5, 8])
@pytest.mark.parametrize("r,h,w,c,n,kh,kw,flip", [(5, 7, 7, 64, 64, 3, 3, False), (37, 7, 7, 512, 512, 3, 3, False),
                                                    (3, 5, 9, 32, 128, 3, 3, True), (2, 13, 11, 64, 96, 3, 5, False),
                                                    (1, 50, 84, 256, 256, 3, 3, True), (10, 7, 7, 512, 512, 3, 3, True),
                                                    (61, 7, 7, 128, 192, 3, 3, False), (4, 9, 6, 64, 128, 3, 3, True),
                                                    (3, 2, 3, 96, 128, 3, 5, False)])
def test_implicit_conv_vs_fp64(config, r, h, w, c, n, kh, kw, flip):
    C = _C()
    from cvpr22_cross_modal_pseudo_labeling_amd.layers.pair_bottleneck import conv_weight_matrix
    g = torch.Generator(device="cuda").manual_seed(r + h * 3 + w * 5 + c)
    x = torch.randn(r, h, w, c, device="cuda", generator=g)
    wt = torch.randn(n, c, kh, kw, device="cuda", generator=g)
    xp = C.split_pair(x.view(-1, c))
    wp = C.split_pair(conv_weight_matrix(wt).contiguous())
    y, _ = C.split_gemm_pair(xp, wp, conv=(h, w, kh, kw, flip), config=config)
    wref = wt.flip(2, 3) if flip else wt
    pad = (kh // 2, kw // 2)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wref.double(), padding=pad).permute(0, 2, 3, 1).reshape(-1, n)
    bound = F.conv2d(x.permute(0, 3, 1, 2).abs().double(), wref.abs().double(), padding=pad).permute(0, 2, 3, 1).reshape(-1, n) + 1
    assert ((y.double() - ref).abs() / bound).max().item() < TOL
    if not flip:  # the materialised pair im2col rows give the same product through the plain GEMM
        y2, _ = C.split_gemm_pair(C.im2col_pair(xp, h, w, kh, kw), wp, config=config & 3)
        assert (y - y2).abs().max().item() <= 1e-5 * y.abs().max().item()


def test_halo_form_equals_shifted_form_bitwise():
    """The halo-tile form of the implicit 3x3 (7x7 maps) runs the same products in a different k order than the
    shifted-row form: same values to fp32 summation order; the K-sliced small-M grid equals the un-sliced one likewise."""
    C = _C()
    from cvpr22_cross_modal_pseudo_labeling_amd.layers.pair_bottleneck import conv_weight_matrix
    g = torch.Generator(device="cuda").manual_seed(5)
    for r in (10, 300):
        x = torch.randn(r, 7, 7, 256, device="cuda", generator=g)
        wt = torch.randn(384, 256, 3, 3, device="cuda", generator=g) * 0.05
        b = torch.randn(384, device="cuda", generator=g)
        xp, wp = C.split_pair(x.view(-1, 256)), C.split_pair(conv_weight_matrix(wt).contiguous())
        outs = [C.split_gemm_pair(xp, wp, b, None, True, True, True, conv=(7, 7, 3, 3, False), config=cfg)
                for cfg in (0, 4, 8, 12, 5)]
        ref = outs[3][0]
        for y, yp in outs:
            assert (y - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
            assert torch.equal(yp, C.split_pair(y))


def test_k_concatenated_second_operand():
    """a2_pair: [A | A2] x [B | B2]^T as one product == the two products summed (conv3 + projection shortcut)."""
    C = _C()
    g = torch.Generator(device="cuda").manual_seed(9)
    for m in (490, 3000):
        a, a2 = torch.randn(m, 512, device="cuda", generator=g), torch.randn(m, 1024, device="cuda", generator=g)
        b, b2 = torch.randn(256, 512, device="cuda", generator=g), torch.randn(256, 1024, device="cuda", generator=g)
        bias = torch.randn(256, device="cuda", generator=g)
        bcat = C.split_pair(torch.cat([b, b2], 1).contiguous())
        for cfg in (0, 1, 2, 8):
            y, yp = C.split_gemm_pair(C.split_pair(a), bcat, bias, None, True, True, True, a2_pair=C.split_pair(a2), config=cfg)
            ref = (a.double() @ b.double().t() + a2.double() @ b2.double().t() + bias.double()).clamp(min=0)
            bound = a.abs().double() @ b.abs().double().t() + a2.abs().double() @ b2.abs().double().t() + 1
            assert ((y.double() - ref).abs() / bound).max().item() < TOL
            assert torch.equal(yp, C.split_pair(y))
    with pytest.raises(RuntimeError):
        C.split_gemm_pair(C.split_pair(a), bcat, a2_pair=C.split_pair(a2[:-1]))


def test_im2col_pair_layout():
    C = _C()
    x = torch.randn(2, 3, 4, 32, device="cuda")
    xp = C.split_pair(x.view(-1, 32))
    rows = C.im2col_pair(xp, 3, 4, 3, 3)
    assert rows.shape == (24, 9 * 64)
    padp = F.pad(xp.view(2, 3, 4, 64), (0, 0, 1, 1, 1, 1))
    for t in range(9):
        want = padp[:, t // 3: t // 3 + 3, t % 3: t % 3 + 4, :].reshape(24, 64)
        assert torch.equal(rows[:, t * 64:(t + 1) * 64], want)


def test_gate_split_pair():
    C = _C()
    dy = torch.randn(50, 64, device="cuda")
    y = torch.randn(50, 64, device="cuda").clamp(min=0)
    want = dy * (y > 0)
    for gate in (y, C.split_pair(y)):
        gp, g32 = C.gate_split_pair(dy, gate, want_f32=True)
        assert torch.equal(g32, want)
        assert torch.equal(gp, C.split_pair(want))
    gp, g32 = C.gate_split_pair(dy)
    assert g32 is None and torch.equal(gp, C.split_pair(dy))


def test_dw_pair_vs_fp64():
    C = _C()
    from cvpr22_cross_modal_pseudo_labeling_amd.layers.pair_bottleneck import dw_pair
    dy = torch.randn(777, 64, device="cuda")
    x = torch.randn(777, 96, device="cuda")
    dw = dw_pair(C.split_pair(dy), C.split_pair(x))
    ref = dy.double().t() @ x.double()
    bound = dy.abs().double().t() @ x.abs().double() + 1
    assert ((dw.double() - ref).abs() / bound).max().item() < TOL


@pytest.mark.parametrize("m,n,ch,conv", [(64, 128, 128, None), (1000, 128, 256, None), (49 * 37, 256, 128, (7, 7, 3, 3)),
                                         (5 * 9 * 13, 128, 128, (9, 13, 3, 5)), (49 * 300, 512, 512, (7, 7, 3, 3)),
                                         (2 * 100 * 167, 128, 128, (100, 167, 3, 3)), (18 * 3 * 3, 128, 128, (3, 3, 3, 3)),
                                         (31, 128, 128, None)])
def test_split_gemm_pair_tn_vs_fp64(m, n, ch, conv):
    """The transpose-read weight-gradient kernel (row slices + slab sum), plain and with shifted tap reads."""
    C = _C()
    g = torch.Generator(device="cuda").manual_seed(m + n + ch)
    dy = torch.randn(m, n, device="cuda", generator=g)
    x = torch.randn(m, ch, device="cuda", generator=g)
    assert C.split_gemm_pair_tn_supported(n, ch, conv)
    dw = C.split_gemm_pair_tn(C.split_pair(dy), C.split_pair(x), conv)
    if conv is None:
        cols = x.double()
    else:
        h, w, kh, kw = conv
        r = m // (h * w)
        cols = F.unfold(x.view(r, h, w, ch).permute(0, 3, 1, 2).double(), (kh, kw), padding=(kh // 2, kw // 2))
        cols = cols.view(r, ch, kh * kw, h * w).permute(0, 3, 2, 1).reshape(m, kh * kw * ch)  # tap-major rows
    ref = dy.double().t() @ cols
    bound = dy.abs().double().t() @ cols.abs() + 1
    assert ((dw.double() - ref).abs() / bound).max().item() < TOL
    # the library route on the same operands agrees
    from cvpr22_cross_modal_pseudo_labeling_amd.layers.pair_bottleneck import dw_pair
    xp = C.split_pair(x) if conv is None else C.im2col_pair(C.split_pair(x), *conv)
    q = torch.mm(C.split_pair(dy).t(), xp, out_dtype=torch.float32)
    k = xp.shape[1] // 2
    lib = q.view(n // 32, 2, 32, k // 32, 2, 32).sum(dim=(1, 4)).reshape(n, k)
    assert (dw - lib).abs().max().item() <= 1e-4 * lib.abs().max().item() + 1e-5


@pytest.mark.parametrize("proj", [True, False])
def test_bottleneck_pair_node_vs_fp64_autograd(proj):
    """Forward, input gradient and all weight gradients of the fused node vs an fp64 autograd bottleneck."""
    from cvpr22_cross_modal_pseudo_labeling_amd.layers.pair_bottleneck import bottleneck_pair
    torch.manual_seed(3)
    r, h, w = 6, 7, 7
    cin, cb, cout = (128, 128, 256) if proj else (256, 128, 256)  # multiples of 128: the transpose-read dW kernel
    dev = "cuda"
    x = torch.randn(r * h * w, cin, device=dev, requires_grad=True)
    w1 = (torch.randn(cb, cin, 1, 1, device=dev) / cin ** 0.5).requires_grad_(True)
    w2 = (torch.randn(cb, cb, 3, 3, device=dev) / (9 * cb) ** 0.5).requires_grad_(True)
    w3 = (torch.randn(cout, cb, 1, 1, device=dev) / cb ** 0.5).requires_grad_(True)
    wd = (torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5).requires_grad_(True) if proj else None
    b1, b2, b3 = (torch.randn(n, device=dev) * 0.1 for n in (cb, cb, cout))
    out, outp = bottleneck_pair(x, None, (h, w), w1, b1, w2, b2, w3, b3, wd, True)
    from cvpr22_cross_modal_pseudo_labeling_amd import _C as C
    assert torch.equal(outp, C.split_pair(out.detach()))
    gout = torch.randn_like(out)
    params = [x, w1, w2, w3] + ([wd] if proj else [])
    grads = torch.autograd.grad(out, params, gout)

    xd = x.detach().double().view(r, h, w, cin).permute(0, 3, 1, 2).requires_grad_(True)
    pd = [p.detach().double().requires_grad_(True) for p in params[1:]]
    o = F.relu(F.conv2d(xd, pd[0], b1.double()))
    o = F.relu(F.conv2d(o, pd[1], b2.double(), padding=1))
    o = F.conv2d(o, pd[2], b3.double())
    o = F.relu(o + (F.conv2d(xd, pd[3]) if proj else xd))
    ref = o.permute(0, 2, 3, 1).reshape(-1, cout)
    assert (out.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    rgrads = torch.autograd.grad(ref, [xd] + pd, gout.double())
    rgrads = [rgrads[0].permute(0, 2, 3, 1).reshape(-1, cin)] + list(rgrads[1:])
    for gname, a, b in zip(("dx", "dw1", "dw2", "dw3", "dwd"), grads, rgrads):
        assert a.shape == b.shape, gname
        assert (a.double() - b).norm().item() <= 1e-4 * b.norm().item(), gname


@pytest.mark.parametrize("m,k,n", [(777, 128, 256), (3000, 512, 2048), (64, 64, 32)])
def test_split_gemm_pair_residual_in_pair_layout(m, k, n):
    """The shortcut operand given as pair rows: act(A B^T + bias + (hi + lo)) -- bit-identical to passing the fp32 tensor
    hi + lo as the residual, for fp32 and pair outputs, with and without K slices."""
    C = _C()
    g = torch.Generator(device="cuda").manual_seed(m + n)
    a = C.split_pair(torch.randn(m, k, device="cuda", generator=g))
    b = C.split_pair(torch.randn(n, k, device="cuda", generator=g) * 0.1)
    bias = torch.randn(n, device="cuda", generator=g)
    r = torch.randn(m, n, device="cuda", generator=g)
    rp = C.split_pair(r)
    pr = rp.view(m, n // 32, 2, 32).float()
    r_rec = (pr[:, :, 0] + pr[:, :, 1]).reshape(m, n).contiguous()   # hi + lo: exact in fp32
    for config in (0, 8):
        want, want_p = C.split_gemm_pair(a, b, bias, r_rec, True, True, True, config=config)
        got, got_p = C.split_gemm_pair(a, b, bias, None, True, True, True, config=config, residual_pair=rp)
        assert torch.equal(got, want) and torch.equal(got_p, want_p)
        none, only_p = C.split_gemm_pair(a, b, bias, None, True, False, True, config=config, residual_pair=rp)
        assert none is None and torch.equal(only_p, want_p)
    assert float((r_rec - r).abs().max()) <= 2.0 ** -16 * float(r.abs().max())


@pytest.mark.parametrize("m,k,n", [(1000, 512, 2048), (77, 128, 128), (4100, 2048, 512)])
def test_split_gemm_pair_rp_gated_equals_two_step_form(m, k, n):
    """(A B^T + (hi + lo of the pair shortcut)) * (x > 0), pair and fp32 outputs, in ONE launch -- bit-identical to the
    two-step form it replaces (product with the fp32 shortcut hi + lo, then gate_split_pair with the gate's pair form),
    and within the split product's error of the fp64 value."""
    C = _C()
    g = torch.Generator(device="cuda").manual_seed(m + k)
    a32, b32 = torch.randn(m, k, device="cuda", generator=g), torch.randn(n, k, device="cuda", generator=g) * 0.1
    a, b = C.split_pair(a32), C.split_pair(b32)
    r = torch.randn(m, n, device="cuda", generator=g)
    rp = C.split_pair(r)
    pr = rp.view(m, n // 32, 2, 32).float()
    r_rec = (pr[:, :, 0] + pr[:, :, 1]).reshape(m, n).contiguous()
    x = torch.randn(m, n, device="cuda", generator=g)
    x[::3] = 0.0  # exact zeros gate off (relu'(0) = 0)
    xp = C.split_pair(x)
    full, _ = C.split_gemm_pair(a, b, None, r_rec, False, True, False, config=8)
    want_p, want = C.gate_split_pair(full, xp, want_f32=True)
    got, got_p = C.split_gemm_pair_rp_gated(a, b, rp, xp, out_f32=True, out_pair=True)
    assert torch.equal(got, want) and torch.equal(got_p, want_p)
    none, only_p = C.split_gemm_pair_rp_gated(a, b, rp, xp)
    assert none is None and torch.equal(only_p, want_p)
    ref = (a32.double() @ b32.double().t() + r_rec.double()) * (x > 0)
    assert float((got.double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max())
    with pytest.raises(RuntimeError):  # the narrow-tile kernels have no such form
        C.split_gemm_pair_rp_gated(a, C.split_pair(b32[:64]), C.split_pair(r[:, :64].contiguous()),
                                   C.split_pair(x[:, :64].contiguous()))


@pytest.mark.parametrize("maps,n,k", [(600, 256, 512), (2000, 2048, 512), (301, 128, 64)])
def test_split_gemm_pair_rp_pool_epilogue(maps, n, k):
    """conv3 + pair shortcut + ReLU of the last res5 block with the 7 x 7 average pooling in the GEMM epilogue
    (``ovis_split_gemm_pair_rp_pool``): the fp32 result is bit-identical to the plain launch, the pooled rows equal the mean
    over every 49-row group of that result to fp32 summation accuracy, the pool-only form (no result written) gives the
    same pooled bits, two runs are bit-identical (at most two atomic addends per pooled value), and shapes whose launch plan
    is not the plain un-split form are refused up front."""
    C = _C()
    g = torch.Generator(device="cuda").manual_seed(maps + n)
    m = maps * 49                                   # ragged last row tile (m % 128 != 0) and maps straddling tiles
    a = C.split_pair(torch.randn(m, k, device="cuda", generator=g))
    b = C.split_pair(torch.randn(n, k, device="cuda", generator=g) * 0.05)
    r = C.split_pair(torch.randn(m, n, device="cuda", generator=g))
    bias = torch.randn(n, device="cuda", generator=g)
    assert C.split_gemm_pair_pool_supported(m, n, k, 49)
    want, _ = C.split_gemm_pair(a, b, bias, None, True, True, False, residual_pair=r)
    got, gotp, pooled = C.split_gemm_pair_rp_pool(a, b, bias, r, True, True, True, 49)
    assert torch.equal(got, want) and torch.equal(gotp, C.split_pair(want))
    ref = want.view(maps, 49, n).double().mean(1)
    assert pooled.shape == (maps, n)
    assert float((pooled.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    none_c, none_p, pooled_only = C.split_gemm_pair_rp_pool(a, b, bias, r, True, False, False, 49)
    assert none_c is None and none_p is None and torch.equal(pooled_only, pooled)
    _, _, again = C.split_gemm_pair_rp_pool(a, b, bias, r, True, False, False, 49)
    assert torch.equal(again, pooled)
    # no shortcut operand
    w2, _ = C.split_gemm_pair(a, b, bias, None, True, True, False)
    _, _, p2 = C.split_gemm_pair_rp_pool(a, b, bias, None, True, False, False, 49)
    assert float((p2.double() - w2.view(maps, 49, n).double().mean(1)).abs().max()) <= 2e-6 * float(w2.abs().max())
    assert not C.split_gemm_pair_pool_supported(490, 2048, 512, 49)     # a handful of row tiles: they keep their split-K plan
    assert not C.split_gemm_pair_pool_supported(m, 64, k, 49)           # narrow tiles
    assert not C.split_gemm_pair_pool_supported(m, n, k, 16)            # a 4 x 4 map: more than three maps per 64 rows


@pytest.mark.parametrize("train", [False, True])
def test_bottleneck_chain_pair_only_vs_fp32_chain(train):
    """A res5-like chain (projection block + two identity blocks) with the intermediate results carried in PAIR layout
    only (identity shortcut = hi + lo of the block input, zero-stride placeholders as fp32 handles) against the same chain
    with fp32 + pair intermediates: outputs within 3e-5 of the maximum, every gradient within 5e-3 in the L2 norm (a
    pre-activation within rounding of zero may flip its ReLU gate between the two arithmetics, which moves isolated
    entries by O(1): the bound of the other cross-path gradient tests)."""
    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.layers.pair_bottleneck import is_placeholder
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.backbone import ResNetHead
    cfg = get_defaults()
    cfg.freeze()
    torch.manual_seed(11)
    head = ResNetHead(cfg).cuda()
    for m in head.modules():
        if hasattr(m, "running_var"):
            m.weight.uniform_(0.5, 1.5)
            m.bias.uniform_(-0.2, 0.2)
    for p in head.parameters():
        p.requires_grad_(train)
    x = torch.randn(20, 7, 7, 1024, device="cuda")
    gy = torch.randn(20, 2048, 7, 7, device="cuda")

    def run(pair_only):
        for b in head.layer4:
            b.pair_only_chain = pair_only
        xx = x.clone().requires_grad_(train)
        with torch.set_grad_enabled(train):
            y = head.forward_pooled_nhwc(xx)
            if train:
                head.zero_grad()
                y.backward(gy)
        return [y.detach()] + ([xx.grad] + [p.grad.clone() for p in head.parameters()] if train else [])

    # the chained blocks really exchange placeholders
    b0 = head.layer4[0]
    b0.pair_only_chain = True
    with torch.no_grad():
        mid, midp = b0.forward_nhwc(x, prestrided=True, want_pair=True, pair_only=True)
        both, bothp = b0.forward_nhwc(x, prestrided=True, want_pair=True)  # no look-ahead given: both forms are written
    assert is_placeholder(mid) and midp.shape == (20 * 49, 2 * 2048)
    assert bool(torch.isnan(mid).all())  # a misuse of the placeholder shows up as NaN, not as plausible garbage
    assert not is_placeholder(both) and torch.equal(bothp, midp)
    from cvpr22_cross_modal_pseudo_labeling_amd import _C as C
    calls = {"rp_gated": 0, "gate_split": 0}
    rp_gated, gate_split, one_call = C.split_gemm_pair_rp_gated, C.gate_split_pair, C.bottleneck_identity_backward

    def count(name, fn):
        def wrapped(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        return wrapped

    ref = run(False)
    C.split_gemm_pair_rp_gated, C.gate_split_pair = count("rp_gated", rp_gated), count("gate_split", gate_split)
    C.bottleneck_identity_backward = count("rp_gated", one_call)  # (the one-call backward of a linked identity block ends in it)
    try:
        got = run(True)
    finally:
        C.split_gemm_pair_rp_gated, C.gate_split_pair, C.bottleneck_identity_backward = rp_gated, gate_split, one_call
    if train:
        # the two identity blocks hand their input gradient down gated and split (GradLink): one gate + split pass (the
        # gradient entering the chain) instead of three
        assert calls == {"rp_gated": 2, "gate_split": 1}
    assert (got[0] - ref[0]).abs().max().item() <= 3e-5 * ref[0].abs().max().item()
    for a, b in zip(got[1:], ref[1:]):
        assert (a - b).norm().item() <= 5e-3 * b.norm().item() + 1e-12


def test_gate_split_pair_with_gathered_row_groups():
    """The gradient of a gather of whole row groups (the mask head reads the positive RoIs' 7x7 maps) handed over as dense
    maps + a group -> slot table: same result as scattering it into a zero [rows, cols] tensor and adding that first --
    bit for bit when it is the only dense term next to the pooled gradient (a + b = b + a), to rounding otherwise."""
    C = _C()
    g = torch.Generator(device="cuda").manual_seed(9)
    groups, pr, cols = 37, 49, 96
    rows = groups * pr
    y = torch.randn(rows, cols, device="cuda", generator=g)
    yp = C.split_pair(y)
    pooled = torch.randn(groups, cols, device="cuda", generator=g)
    sel = torch.tensor([3, 4, 17, 36, 0], device="cuda")
    gsel = torch.randn(sel.numel() * pr, cols, device="cuda", generator=g)
    slot = torch.full((groups,), -1, dtype=torch.int32, device="cuda")
    slot[sel] = torch.arange(sel.numel(), dtype=torch.int32, device="cuda")
    dense = torch.zeros(groups, pr, cols, device="cuda")
    dense[sel] = gsel.view(-1, pr, cols)
    dense = dense.view(rows, cols)
    for gate in (y, yp):
        want_p, want = C.gate_split_pair(dense, gate, want_f32=True, pooled=pooled, pool_rows=pr)
        got_p, got = C.gate_split_pair(None, gate, want_f32=True, pooled=pooled, pool_rows=pr, selected=gsel, group_slot=slot)
        assert torch.equal(got, want) and torch.equal(got_p, want_p)
    dy = torch.randn(rows, cols, device="cuda", generator=g)
    _, want = C.gate_split_pair(dy + dense, y, want_f32=True, pooled=pooled, pool_rows=pr)
    _, got = C.gate_split_pair(dy, y, want_f32=True, pooled=pooled, pool_rows=pr, selected=gsel, group_slot=slot)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)
    with pytest.raises(RuntimeError):
        C.gate_split_pair(dy, y, pooled=pooled, pool_rows=pr, selected=gsel)  # the slot table is missing


def test_pair_only_output_feeds_exactly_one_block():
    """A block output kept in pair layout only has a placeholder as its fp32 handle: a second consumer's gradient could
    not be summed into it, so the second ``bottleneck_pair`` on the same pair tensor is refused (loudly, in the forward)."""
    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.backbone import ResNetHead
    cfg = get_defaults()
    cfg.freeze()
    torch.manual_seed(3)
    head = ResNetHead(cfg).cuda()
    b0, b1 = head.layer4[0], head.layer4[1]
    x = torch.randn(4, 7, 7, 1024, device="cuda", requires_grad=True)
    mid, midp = b0.forward_nhwc(x, prestrided=True, want_pair=True, pair_only=True)
    b1.forward_nhwc(mid, xp=midp)
    with pytest.raises(RuntimeError, match="can feed one block"):
        b1.forward_nhwc(mid, xp=midp)
    with torch.no_grad():  # without autograd nothing has to be summed: allowed
        b1.forward_nhwc(mid, xp=midp)


def test_frozen_trunk_with_stride_in_3x3_keeps_fp32_between_routes():
    """MODEL.RESNETS.STRIDE_IN_1X1 False: the first block of layer2 / layer3 has a strided 3x3 and is not on the pair-GEMM
    route, so the block in front of it must NOT drop the fp32 copy of its output (look-ahead in ``chain_nhwc``; a pair-only
    hand-over would feed that block a placeholder).  Frozen trunk, NHWC chain vs the per-layer NCHW convolutions."""
    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.layers.pair_bottleneck import is_placeholder
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.backbone import ResNetC4, chain_nhwc
    cfg = get_defaults()
    cfg.merge_from_list(["MODEL.RESNETS.STRIDE_IN_1X1", False])
    cfg.freeze()
    torch.manual_seed(21)
    body = ResNetC4(cfg).cuda()
    for m in body.modules():
        if hasattr(m, "running_var"):
            m.weight.uniform_(0.5, 1.0)
            m.bias.uniform_(-0.1, 0.1)
    for p in body.parameters():
        p.requires_grad_(False)
    assert not body.layer2[0].takes_pair_only_input() and body.layer2[1].takes_pair_only_input()
    x = torch.randn(1, 3, 128, 160, device="cuda") * 40
    with torch.no_grad():
        got = body(x)[0]
        body.nhwc = False
        want = body(x)[0]
        body.nhwc = True
        # the hand-over in front of the strided block carries real fp32 values; inside a stage it is pair-only
        y = body.stem.forward_gemm(x).permute(0, 2, 3, 1).contiguous()
        mid, midp = body.layer1[0].forward_nhwc(y, want_pair=True, pair_only=body.layer1[1].takes_pair_only_input())
        assert is_placeholder(mid) and midp is not None
        last1 = chain_nhwc(list(body.layer1), y)
        m2, m2p = body.layer1[1].forward_nhwc(mid, xp=midp, want_pair=True, pair_only=body.layer1[2].takes_pair_only_input())
        out, outp = body.layer1[2].forward_nhwc(m2, xp=m2p, want_pair=True, pair_only=body.layer2[0].takes_pair_only_input())
        assert not is_placeholder(out) and torch.equal(out, last1)
        with pytest.raises(RuntimeError, match="pair layout only"):
            body.layer2[0].forward_nhwc(mid.expand(1, 32, 40, 256), xp=None)
    assert torch.isfinite(got).all()
    assert (got - want).abs().max().item() <= 2e-4 * want.abs().max().item()


def test_stem_gemm_matches_convolution():
    """7x7 / stride-2 stem as im2col-pair + split GEMM (+ folded FrozenBN, ReLU, max-pool) vs the fp64 convolution."""
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.backbone import Stem
    torch.manual_seed(5)
    stem = Stem(64).cuda()
    stem.bn1.weight.uniform_(0.5, 1.5)
    stem.bn1.bias.uniform_(-0.2, 0.2)
    stem.bn1.running_mean.uniform_(-1, 1)
    stem.bn1.running_var.uniform_(0.5, 2.0)
    for p in stem.parameters():
        p.requires_grad = False
    x = torch.randn(2, 3, 67, 93, device="cuda") * 50
    assert stem.gemm_supported(x)
    got = stem.forward_gemm(x)
    scale, shift = stem.bn1.fold()
    y = F.conv2d(x.double(), stem.conv1.weight.double() * scale.double().view(-1, 1, 1, 1), shift.double(), 2, 3)
    ref = F.max_pool2d(F.relu(y), 3, 2, 1)
    assert got.shape == ref.shape
    assert (got.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    # patch rows: k = (ky*KW + kx)*C + c, zero-padded
    C = _C()
    rows, (ho, wo) = C.im2col_nchw_pair(x, 7, 7, 2, 3)
    hi, lo = _unpair(rows)
    cols = F.unfold(x, (7, 7), padding=3, stride=2).view(2, 3, 49, ho * wo).permute(0, 3, 2, 1).reshape(2 * ho * wo, 147)
    assert torch.equal(hi[:, :147], cols.to(torch.bfloat16)) and not hi[:, 147:].any() and not lo[:, 147:].any()


def test_rpn_head_gemm_matches_convolutions():
    """Frozen RPN head (3x3 + the two 1x1 predictors) on the split GEMM vs fp64 convolutions."""
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.rpn import RPNHead
    torch.manual_seed(9)
    head = RPNHead(64, 15).cuda()
    for p in head.parameters():
        p.data.normal_(0, 0.05)
        p.requires_grad = False
    x = torch.randn(2, 64, 13, 21, device="cuda")
    assert head._gemm_ok(x)
    obj, reg = head(x)
    t = F.relu(F.conv2d(x.double(), head.conv.weight.double(), head.conv.bias.double(), padding=1))
    obj_ref = F.conv2d(t, head.cls_logits.weight.double(), head.cls_logits.bias.double())
    reg_ref = F.conv2d(t, head.bbox_pred.weight.double(), head.bbox_pred.bias.double())
    assert obj.shape == obj_ref.shape and reg.shape == reg_ref.shape
    assert (obj.double() - obj_ref).abs().max().item() <= 1e-5 * obj_ref.abs().max().item()
    assert (reg.double() - reg_ref).abs().max().item() <= 1e-5 * reg_ref.abs().max().item()


def test_bottleneck_pair_pooled_output_and_fused_pool_gradient():
    """pool=True: the node also returns the mean over the map; its gradient is broadcast inside gate_split_pair and must
    equal the tensor-op route (mean of the fp32 output -> autograd)."""
    from cvpr22_cross_modal_pseudo_labeling_amd.layers.pair_bottleneck import bottleneck_pair
    torch.manual_seed(4)
    r, h, w, cin, cb, cout = 5, 7, 7, 128, 128, 128
    dev = "cuda"
    ws = [(torch.randn(cb, cin, 1, 1, device=dev) / cin ** 0.5), (torch.randn(cb, cb, 3, 3, device=dev) / (9 * cb) ** 0.5),
          (torch.randn(cout, cb, 1, 1, device=dev) / cb ** 0.5)]
    bs = [torch.randn(n, device=dev) * 0.1 for n in (cb, cb, cout)]
    gp = torch.randn(r, cout, device=dev)
    gd = torch.randn(r * h * w, cout, device=dev)

    def run(pool, dense):
        x = torch.randn(r * h * w, cin, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).requires_grad_(True)
        wl = [t.clone().requires_grad_(True) for t in ws]
        res = bottleneck_pair(x, None, (h, w), wl[0], bs[0], wl[1], bs[1], wl[2], bs[2], None, False, None, pool)
        out = res[0]
        pooled = res[2] if pool else out.view(r, h * w, cout).mean(1)
        loss = (pooled * gp).sum() + ((out * gd).sum() if dense else 0.0)
        grads = torch.autograd.grad(loss, [x] + wl)
        return pooled.detach(), grads

    for dense in (True, False):
        p_ref, g_ref = run(False, dense)
        p_got, g_got = run(True, dense)
        assert torch.equal(p_ref, p_got)
        for a, b in zip(g_got, g_ref):
            assert (a - b).norm().item() <= 1e-5 * b.norm().item()
    from cvpr22_cross_modal_pseudo_labeling_amd import _C as C
    dy = torch.randn(14, 32, device=dev)
    y = torch.randn(14, 32, device=dev)
    pl = torch.randn(2, 32, device=dev)
    want = (dy + (pl * (1.0 / 7)).repeat_interleave(7, 0)) * (y > 0)
    g_pair, g32 = C.gate_split_pair(dy, y, want_f32=True, pooled=pl, pool_rows=7)
    assert torch.equal(g32, want) and torch.equal(g_pair, C.split_pair(want))
    g_pair, g32 = C.gate_split_pair(None, y, want_f32=True, pooled=pl, pool_rows=7)
    assert torch.equal(g32, (pl * (1.0 / 7)).repeat_interleave(7, 0) * (y > 0))


def test_strided_pooler_pair_output_is_split_of_fp32_output():
    """roi_align_forward_strided_pair == split_pair(roi_align_forward_strided_nhwc) bit for bit (small, large, clipped and
    empty RoIs), and the res5 head gives identical results from either hand-over."""
    C = _C()
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 64, 50, 84, generator=g).cuda()
    rois = torch.tensor([[0, 10.0, 20.0, 200.0, 150.0], [1, 0.0, 0.0, 1332.0, 799.0], [0, 500.5, 300.25, 503.0, 301.0],
                         [1, -40.0, -30.0, 90.0, 60.0], [0, 1300.0, 780.0, 1400.0, 900.0], [1, 64.0, 64.0, 64.0, 64.0]]).cuda()
    more = torch.cat([torch.randint(0, 2, (40, 1), generator=g).float(), torch.rand(40, 2, generator=g) * 600], 1)
    more = torch.cat([more, more[:, 1:] + torch.rand(40, 2, generator=g) * 500 + 4], 1).cuda()
    rois = torch.cat([rois, more], 0)
    f32 = C.roi_align_forward_strided_nhwc(x, rois, 1 / 16, 14, 14, 0, 2)
    pair, (oh, ow) = C.roi_align_forward_strided_pair(x, rois, 1 / 16, 14, 14, 0, 2)
    assert (oh, ow) == (7, 7) and pair.shape == (rois.shape[0] * 49, 128)
    assert torch.equal(pair, C.split_pair(f32.view(-1, 64)))


def test_conv_same_pair_node_vs_fp64_autograd():
    """Trainable 3x3 "same" convolution node (implicit GEMM fwd / dX, transpose-read dW, bias, ReLU) vs fp64 autograd."""
    from cvpr22_cross_modal_pseudo_labeling_amd.layers.pair_bottleneck import conv_same_pair
    torch.manual_seed(8)
    n, h, w, c, co = 2, 9, 11, 128, 128
    x = torch.randn(n * h * w, c, device="cuda", requires_grad=True)
    wt = (torch.randn(co, c, 3, 3, device="cuda") / (9 * c) ** 0.5).requires_grad_(True)
    b = (torch.randn(co, device="cuda") * 0.1).requires_grad_(True)
    y = conv_same_pair(x, (h, w), wt, b, True)
    gy = torch.randn_like(y)
    gx, gw, gb = torch.autograd.grad(y, [x, wt, b], gy)
    xd = x.detach().double().view(n, h, w, c).permute(0, 3, 1, 2).requires_grad_(True)
    wd, bd = wt.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    ref = F.relu(F.conv2d(xd, wd, bd, padding=1)).permute(0, 2, 3, 1).reshape(-1, co)
    rx, rw, rb = torch.autograd.grad(ref, [xd, wd, bd], gy.double())
    assert (y.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert (gx.double() - rx.permute(0, 2, 3, 1).reshape(-1, c)).norm().item() <= 1e-4 * rx.norm().item()
    assert (gw.double() - rw).norm().item() <= 1e-4 * rw.norm().item()
    assert (gb.double() - rb).norm().item() <= 1e-4 * rb.norm().item()


def test_weight_prep_pair_and_scaled_slab_reduce():
    """weight_prep_pair == split_pair of the folded tap-major matrix (and of its data-gradient transpose); the fused slab
    reduce of split_gemm_pair_tn returns scale * dW in the weight's own layout."""
    C = _C()
    from cvpr22_cross_modal_pseudo_labeling_amd.layers.pair_bottleneck import conv_weight_matrix, conv_weight_matrix_t
    torch.manual_seed(12)
    for shape in ((128, 64, 3, 3), (256, 128, 1, 1), (32, 96, 3, 5)):
        w = torch.randn(*shape, device="cuda")
        sc = torch.rand(shape[0], device="cuda") + 0.5
        fwd, bwd = C.weight_prep_pair(w, sc, True)
        fw = w * sc.view(-1, 1, 1, 1)
        assert torch.equal(fwd, C.split_pair(conv_weight_matrix(fw).contiguous()))
        assert torch.equal(bwd, C.split_pair(conv_weight_matrix_t(fw).contiguous()))
        fwd2, none = C.weight_prep_pair(w)
        assert none is None and torch.equal(fwd2, C.split_pair(conv_weight_matrix(w).contiguous()))
    m, n, ch = 49 * 20, 128, 128
    dy, x = torch.randn(m, n, device="cuda"), torch.randn(m, ch, device="cuda")
    sc = torch.rand(n, device="cuda") + 0.5
    gp, xp = C.split_pair(dy), C.split_pair(x)
    plain = C.split_gemm_pair_tn(gp, xp, (7, 7, 3, 3))                                   # [n, 9*ch] tap-major
    got = C.split_gemm_pair_tn(gp, xp, (7, 7, 3, 3), scale=sc, weight_shape=(n, ch, 3, 3))
    want = plain.view(n, 3, 3, ch).permute(0, 3, 1, 2) * sc.view(-1, 1, 1, 1)
    assert got.shape == (n, ch, 3, 3) and (got - want).abs().max().item() <= 1e-6 * want.abs().max().item()


def test_mask_predictor_gemm_path_matches_convolutions():
    """MaskRCNNC4Predictor: transposed conv as a split GEMM + pixel shuffle, 1x1 heads as fp32 GEMMs, vs the module's own
    convolution path (outputs and all parameter / input gradients)."""
    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.roi_heads import MaskRCNNC4Predictor
    cfg = get_defaults()
    cfg.merge_from_list(["MODEL.UNCERTAINTY", True, "MODEL.CLS_AGNOSTIC_MASK", True])
    cfg.freeze()
    torch.manual_seed(2)
    m = MaskRCNNC4Predictor(cfg, 2048).cuda()
    x = torch.randn(5, 7, 7, 2048, device="cuda").permute(0, 3, 1, 2)   # NCHW view of NHWC memory, as the head hands over
    gm = torch.randn(5, 2, 14, 14, device="cuda")
    gs = torch.randn(5, 1, 14, 14, device="cuda")

    def run(gemm):
        m.split_gemm = gemm
        xx = x.clone().requires_grad_(True)
        m.zero_grad()
        mu, sigma = m.forward_parts(xx)
        ((mu * gm).sum() + (sigma * gs).sum()).backward()
        return [mu.detach(), sigma.detach(), xx.grad] + [p.grad.clone() for p in m.parameters()]

    ref, got = run(False), run(True)
    assert m._gemm_ok(x)
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        assert (a - b).norm().item() <= 2e-5 * b.norm().item() + 1e-7
    mu0, s0 = m.forward_parts(x[:0])
    assert mu0.shape == (0, 2, 14, 14) and s0.shape == (0, 1, 14, 14)


def test_split_gemm_pair_gated_epilogue():
    """Gated data-gradient GEMM == gate_split_pair(split_gemm_pair(...), y) bit for bit (plain and implicit 3x3)."""
    C = _C()
    torch.manual_seed(6)
    m, k, n = 49 * 11, 256, 128
    a, b = torch.randn(m, k, device="cuda"), torch.randn(n, k, device="cuda")
    y = torch.randn(m, n, device="cuda").clamp(min=0)
    ap, bp, yp = C.split_pair(a), C.split_pair(b), C.split_pair(y)
    # the gated form follows the same launch plan as the plain product (K slices on under-filled grids, the gate applied by
    # the slab reduction): same summation order, so the comparison is exact for every config
    for cfg in (0, 8):
        d, _ = C.split_gemm_pair(ap, bp, config=cfg)
        want_p, want = C.gate_split_pair(d, yp, want_f32=True)
        got, got_p = C.split_gemm_pair_gated(ap, bp, yp, out_f32=True, out_pair=True, config=cfg)
        assert torch.equal(got, want) and torch.equal(got_p, want_p)
    wm = torch.randn(n, 9 * k, device="cuda")
    wp = C.split_pair(wm)
    for cfg in (0, 8):
        d, _ = C.split_gemm_pair(ap, wp, conv=(7, 7, 3, 3, True), config=cfg)
        want_p, _ = C.gate_split_pair(d, yp)
        _, got_p = C.split_gemm_pair_gated(ap, wp, yp, conv=(7, 7, 3, 3, True), config=cfg)
        assert torch.equal(got_p, want_p)
    # the teacher step's layer3 data gradient (50 x 84 maps, 132 tiles walking 72 k-steps: four K slices): sliced == plain
    # product + gate bit for bit, and within the split product's error of the un-split order
    m2, c2 = 2 * 50 * 84, 256
    a2, y2 = torch.randn(m2, c2, device="cuda"), torch.randn(m2, c2, device="cuda").clamp(min=0)
    w2 = torch.randn(c2, 9 * c2, device="cuda") / (9 * c2) ** 0.5
    a2p, y2p, w2p = C.split_pair(a2), C.split_pair(y2), C.split_pair(w2)
    d, _ = C.split_gemm_pair(a2p, w2p, conv=(50, 84, 3, 3, True))
    want_p, want = C.gate_split_pair(d, y2p, want_f32=True)
    got, got_p = C.split_gemm_pair_gated(a2p, w2p, y2p, conv=(50, 84, 3, 3, True), out_f32=True)
    assert torch.equal(got, want) and torch.equal(got_p, want_p)
    unsplit, _ = C.split_gemm_pair_gated(a2p, w2p, y2p, conv=(50, 84, 3, 3, True), out_f32=True, out_pair=False, config=8)
    assert (got - unsplit).abs().max().item() <= 1e-5 * unsplit.abs().max().item()


def test_split_gemm_full_size_properties():
    """BASELINE-size res5 shapes (2048 RoIs x 49 positions = 100352 rows) through size-independent properties: a block of
    rows / maps computed alone is bit-identical to the same rows of the full launch (tile and XCD renumbering, 64-bit
    addressing), sampled entries agree with fp64 dot products, and the weight gradient over all rows equals the sum
    over two halves."""
    C = _C()
    g = torch.Generator(device="cuda").manual_seed(77)
    m, k, n = 2048 * 49, 1024, 2048
    a = torch.randn(m, k, device="cuda", generator=g)
    b = torch.randn(n, k, device="cuda", generator=g) / k ** 0.5
    bias = torch.randn(n, device="cuda", generator=g)
    ap, bp = C.split_pair(a), C.split_pair(b)
    full, full_p = C.split_gemm_pair(ap, bp, bias, None, True, True, True)
    assert torch.equal(full_p, C.split_pair(full))
    for r0, r1 in ((0, 128), (50000, 50999), (m - 777, m)):
        part, _ = C.split_gemm_pair(C.split_pair(a[r0:r1]), bp, bias, None, True, True, False, config=8)  # same k order: no K slices
        assert torch.equal(part, full[r0:r1]), (r0, r1)
    rows = torch.randint(0, m, (64,), device="cuda", generator=g)
    cols = torch.randint(0, n, (64,), device="cuda", generator=g)
    ref = ((a[rows].double() * b[cols].double()).sum(1) + bias[cols].double()).clamp(min=0)
    bound = (a[rows].abs().double() * b[cols].abs().double()).sum(1) + 1
    assert ((full[rows, cols].double() - ref).abs() / bound).max().item() < TOL
    del full, full_p
    # implicit 3x3 on [2048, 7, 7, 512] -> 512: maps 1000..1010 alone == the same maps of the full launch
    r, h, w, c, n2 = 2048, 7, 7, 512, 512
    x = torch.randn(r * h * w, c, device="cuda", generator=g)
    wm = torch.randn(n2, 9 * c, device="cuda", generator=g) / (9 * c) ** 0.5
    xp, wp = C.split_pair(x), C.split_pair(wm)
    y, _ = C.split_gemm_pair(xp, wp, conv=(h, w, 3, 3, False))
    sub, _ = C.split_gemm_pair(C.split_pair(x[1000 * 49:1010 * 49]), wp, conv=(h, w, 3, 3, False), config=8)
    assert torch.equal(sub, y[1000 * 49:1010 * 49])
    # weight gradient: all rows == first half + second half (fp32 slab sums: 1e-5 relative), sampled entries vs fp64
    gy = torch.randn(r * h * w, n2, device="cuda", generator=g)
    gp = C.split_pair(gy)
    dw = C.split_gemm_pair_tn(gp, xp, (h, w, 3, 3))
    half = (r // 2) * h * w
    dw2 = C.split_gemm_pair_tn(C.split_pair(gy[:half]), C.split_pair(x[:half]), (h, w, 3, 3)) + \
        C.split_gemm_pair_tn(C.split_pair(gy[half:]), C.split_pair(x[half:]), (h, w, 3, 3))
    assert (dw - dw2).abs().max().item() <= 1e-5 * dw.abs().max().item()
    center = dw.view(n2, 9, c)[:, 4]                      # the centre tap is the plain product G^T X
    i, j = torch.randint(0, n2, (32,), device="cuda", generator=g), torch.randint(0, c, (32,), device="cuda", generator=g)
    ref = (gy[:, i].double() * x[:, j].double()).sum(0)
    bound = (gy[:, i].abs().double() * x[:, j].abs().double()).sum(0) + 1
    assert ((center[i, j].double() - ref).abs() / bound).max().item() < TOL


def test_weight_prep_plan_equals_per_weight_preparation():
    """``WeightPrepPlan`` (every trainable bottleneck weight of a model in ONE launch, csrc/weight_prep_multi.hip) writes
    exactly the bytes ``_C.weight_prep_pair`` writes per convolution: forward forms, transposed forms, the conv3 | shortcut
    matrix of a projection block in place; scales folded or absent; and it notices every way a weight can change."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    from cvpr22_cross_modal_pseudo_labeling_amd.layers.pair_bottleneck import WeightPrepPlan, note_weights_written

    g = torch.Generator().manual_seed(5)

    def conv(n, c, k, scaled=True):
        w = (torch.randn(n, c, k, k, generator=g) * 0.1).cuda().requires_grad_()
        return w, ((torch.rand(n, generator=g) + 0.5).cuda() if scaled else None)

    proj = [conv(64, 256, 1), conv(64, 64, 3), conv(256, 64, 1), conv(256, 256, 1)]           # with a projection shortcut
    plain = [conv(128, 512, 1, False), conv(128, 128, 3), conv(512, 128, 1, False)]           # identity shortcut, mixed scales
    big = [conv(512, 2048, 1), conv(512, 512, 3), conv(2048, 512, 1)]                        # res5 sizes: many chunks per item
    plan = WeightPrepPlan([("a", proj), ("b", plain), ("c", big)])
    assert plan.lookup("a", tuple(s for _, s in proj)) is None                                # nothing prepared yet
    plan.run()
    for key, convs in (("a", proj), ("b", plain), ("c", big)):
        wp = plan.lookup(key, tuple(s for _, s in convs))
        assert wp is not None
        want = [_C.weight_prep_pair(w, s, True) for w, s in convs]
        for name, (f, _) in zip(("w1", "w2", "w3", "wd"), want):
            assert torch.equal(wp[name].contiguous().view(torch.int16), f.view(torch.int16)), (key, name)
        for i, (_, t) in enumerate(want):
            assert torch.equal(wp["wts"][i].view(torch.int16), t.view(torch.int16)), (key, i)
        if len(convs) == 4:
            assert torch.equal(wp["w3d"].view(torch.int16), torch.cat([want[2][0], want[3][0]], 1).view(torch.int16))
            assert wp["w3"].data_ptr() == wp["w3d"].data_ptr()
        else:
            assert wp["wd"] is None and wp["wts"][3] is None and "w3d" not in wp
    scales_a = tuple(s for _, s in proj)
    # (1) a raw-pointer writer announced itself, (2) an in-place update moved a version counter, (3) other scale tensors
    note_weights_written()
    assert plan.lookup("a", scales_a) is None
    plan.run()
    assert plan.lookup("a", scales_a) is not None
    with torch.no_grad():
        proj[1][0].mul_(2.0)
    assert plan.lookup("a", scales_a) is None and plan.lookup("b", tuple(s for _, s in plain)) is not None
    plan.run()
    wp = plan.lookup("a", scales_a)
    assert torch.equal(wp["w2"].view(torch.int16), _C.weight_prep_pair(proj[1][0], proj[1][1], True)[0].view(torch.int16))
    assert plan.describes()
    assert plan.lookup("a", (scales_a[0].clone(),) + scales_a[1:]) is None and not plan.describes()


@pytest.mark.parametrize("m_hw,cin,mid", [((3, 20, 24), 256, 128), ((2, 50, 84), 1024, 256), ((1, 7, 9), 128, 128)])
def test_bottleneck_identity_backward_in_one_call_equals_the_separate_launches(m_hw, cin, mid):
    """``_C.bottleneck_identity_backward`` (one native call: the data-gradient chain on the current stream, the three weight
    gradients on a second stream) returns exactly what the nine separate wrapper calls of ``_BottleneckPair.backward`` return:
    same entry points underneath, same plans -- bit for bit, with and without a second stream, with and without scales."""
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    r, h, w = m_hw
    m = r * h * w
    g = torch.Generator().manual_seed(m + cin)

    def pair(rows, cols, scale=1.0, relu=False):
        t = torch.randn(rows, cols, generator=g) * scale
        return _C.split_pair((t.relu() if relu else t).cuda())

    g3p, xp, o1p, o2p = pair(m, cin, 0.1), pair(m, cin, 1.0, True), pair(m, mid, 1.0, True), pair(m, mid, 1.0, True)
    w1 = (torch.randn(mid, cin, 1, 1, generator=g) * 0.05).cuda()
    w2 = (torch.randn(mid, mid, 3, 3, generator=g) * 0.05).cuda()
    w3 = (torch.randn(cin, mid, 1, 1, generator=g) * 0.05).cuda()
    side = torch.cuda.Stream()
    for scaled in (True, False):
        ss = tuple((torch.rand(n, generator=g) + 0.5).cuda() if scaled else None for n in (mid, mid, cin))
        t1, t2, t3 = (_C.weight_prep_pair(wt, sc, True)[1] for wt, sc in zip((w1, w2, w3), ss))
        # the separate launches, in the order of _BottleneckPair.backward
        dw3 = _C.split_gemm_pair_tn(g3p, o2p, None, scale=ss[2], weight_shape=tuple(w3.shape))
        _, g2p = _C.split_gemm_pair_gated(g3p, t3, o2p)
        dw2 = _C.split_gemm_pair_tn(g2p, o1p, (h, w, 3, 3), scale=ss[1], weight_shape=tuple(w2.shape))
        _, g1p = _C.split_gemm_pair_gated(g2p, t2, o1p, conv=(h, w, 3, 3, True))
        dw1 = _C.split_gemm_pair_tn(g1p, xp, None, scale=ss[0], weight_shape=tuple(w1.shape))
        _, gx = _C.split_gemm_pair_rp_gated(g1p, t1, g3p, xp)
        for second in (side, None):
            got = _C.bottleneck_identity_backward(g3p, xp, o1p, o2p, t1, t2, t3, ss, (h, w, 3, 3),
                                                  (w1.shape, w2.shape, w3.shape), second)
            torch.cuda.synchronize()
            assert torch.equal(got[0].view(torch.int16), gx.view(torch.int16))
            for a, b, name in zip(got[1:], (dw1, dw2, dw3), ("dw1", "dw2", "dw3")):
                assert a.shape == b.shape and torch.equal(a, b), (name, scaled, second is not None)
    with pytest.raises(RuntimeError):
        _C.bottleneck_identity_backward(g3p, xp, o1p, o2p[:, :-2], t1, t2, t3, (None, None, None), (h, w, 3, 3),
                                        (w1.shape, w2.shape, w3.shape))

