"""TEST INFRASTRUCTURE ONLY -- fp64 CPU oracle of deformable convolution v1 / v2.

Restates the forward sampling rule of maskrcnn_benchmark/csrc/cuda/deform_conv_kernel_cuda.cu:92-122,198-250
(bilinear sample, taps outside the map read zero, samples outside (-1,H)x(-1,W) are zero; modulated: x mask,
:475-575,578-640) as vectorised torch ops, followed by the grouped product with the weight
(deform_conv_cuda.cu:232-245).  Backward is obtained by autograd of this forward, i.e. the exact gradient of the
same function -- what the reference's hand-written col2im / col2im_coord kernels (:287-443, :643-774) compute.
PARITY UNPINNED by reference tests: the reference has no CPU implementation of these ops (csrc/deform_conv.h:41)
and no tests; the pins are analytic (zero offsets == F.conv2d, mask == 1 == v1; see tests/test_dcn*.py).
"""
import torch


def deform_conv2d(x, offset, weight, mask=None, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1,
                  deformable_groups=1):
    B, C, H, W = x.shape
    Cout, _, KH, KW = weight.shape
    sh, sw = stride
    ph, pw = padding
    dh, dw = dilation
    Ho = (H + 2 * ph - (dh * (KH - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (KW - 1) + 1)) // sw + 1
    K = KH * KW
    dg = deformable_groups
    cpg = C // dg
    dev, dt = x.device, x.dtype
    hs = (torch.arange(Ho, device=dev, dtype=dt) * sh - ph).view(1, 1, Ho, 1)
    ws = (torch.arange(Wo, device=dev, dtype=dt) * sw - pw).view(1, 1, 1, Wo)
    off = offset.view(B, dg, K, 2, Ho, Wo)
    cols = []
    xg = x.view(B, dg, cpg, H * W)
    for k in range(K):
        i, j = divmod(k, KW)
        hi = hs + i * dh + off[:, :, k, 0]          # [B, dg, Ho, Wo]
        wi = ws + j * dw + off[:, :, k, 1]
        inside = (hi > -1) & (wi > -1) & (hi < H) & (wi < W)
        hl, wl = torch.floor(hi), torch.floor(wi)
        lh, lw = hi - hl, wi - wl
        val = 0
        for dy, wy in ((0, 1 - lh), (1, lh)):
            for dx, wx in ((0, 1 - lw), (1, lw)):
                yy, xx = hl + dy, wl + dx
                ok = inside & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
                idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long().view(B, dg, 1, Ho * Wo).expand(-1, -1, cpg, -1)
                g = torch.gather(xg, 3, idx).view(B, dg, cpg, Ho, Wo)
                val = val + (wy * wx * ok).unsqueeze(2) * g
        if mask is not None:
            val = val * mask.view(B, dg, K, Ho, Wo)[:, :, k].unsqueeze(2)
        cols.append(val.reshape(B, C, Ho, Wo))
    col = torch.stack(cols, 2)                        # [B, C, K, Ho, Wo]
    col = col.view(B, groups, (C // groups) * K, Ho * Wo)
    wg = weight.reshape(groups, Cout // groups, (C // groups) * K)
    out = torch.einsum("gok,bgkn->bgon", wg, col).reshape(B, Cout, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def deform_psroi_pool(data, rois, trans, spatial_scale, out_size, output_dim, no_trans, group_size=1, part_size=None,
                      sample_per_part=4, trans_std=0.0):
    """TEST INFRASTRUCTURE ONLY -- deformable position-sensitive RoI pooling, restating
    maskrcnn_benchmark/csrc/cuda/deform_pool_kernel_cuda.cu:31-139 as vectorised torch ops: returns (out, count).

    The bin geometry is evaluated in float32 in the reference's expression order, with its promotions to double (the
    ``- 0.5``, ``max(.., 0.1)``, border tests and clamp use double literals there), so that floor / ceil cells, the
    inside-the-map test and the sample count are the reference's; sample values and their mean are fp64.  The backward
    of the reference (:141-263) is autograd of this function: the clamp is straight-through, which is exactly the
    reference's offset gradient (it differentiates the bilinear weights at the clamped position and ignores the clamp).
    PARITY UNPINNED by reference tests (the reference has no CPU kernel and no tests for this op); pins are analytic
    (constant and linear maps, offset == shifted RoI, finite differences: tests/test_dcn.py)."""
    f32, f64 = torch.float32, torch.float64
    P = out_size
    part_size = P if part_size is None else part_size
    B, C, H, W = data.shape
    n = rois.shape[0]
    spp = sample_per_part
    dev = data.device
    r = rois.detach().to(f32)
    sc = torch.tensor(spatial_scale, dtype=f32, device=dev)

    def rnd(t):  # roundf: halves away from zero
        return torch.where(t >= 0, torch.floor(t + 0.5), torch.ceil(t - 0.5))

    def lo(col):
        return ((rnd(r[:, col]) * sc).to(f64) - 0.5).to(f32)

    def hi(col):
        return (((rnd(r[:, col]).to(f64) + 1.0).to(f32) * sc).to(f64) - 0.5).to(f32)

    start_w, start_h, end_w, end_h = lo(1), lo(2), hi(3), hi(4)
    roi_w = (end_w - start_w).to(f64).clamp(min=0.1).to(f32)
    roi_h = (end_h - start_h).to(f64).clamp(min=0.1).to(f32)
    Pf, sf = torch.tensor(float(P), dtype=f32, device=dev), torch.tensor(float(spp), dtype=f32, device=dev)
    bin_w, bin_h = roi_w / Pf, roi_h / Pf
    sub_w, sub_h = bin_w / sf, bin_h / sf
    pidx = torch.arange(P, dtype=f32, device=dev)
    part = torch.floor(pidx / Pf * float(part_size)).long()
    grp = torch.floor(pidx * float(group_size) / Pf).long().clamp(0, group_size - 1)
    num_classes = 1 if no_trans else trans.shape[1] // 2
    cec = output_dim if no_trans else output_dim // num_classes
    class_id = torch.arange(output_dim, device=dev) // cec
    if no_trans:
        tx = ty = torch.zeros(n, output_dim, P, P, dtype=f32, device=dev)
    else:
        t = trans.to(f32).view(n, num_classes, 2, part_size, part_size)[:, class_id]      # [n, od, 2, ps, ps]
        t = t[:, :, :, part][:, :, :, :, part]                                             # [n, od, 2, P(ph), P(pw)]
        std = torch.tensor(trans_std, dtype=f32, device=dev)
        tx, ty = t[:, :, 0] * std, t[:, :, 1] * std
    v = lambda a: a.view(n, 1, 1, 1)
    wstart = pidx.view(1, 1, 1, P) * v(bin_w) + v(start_w)
    wstart = wstart + tx * v(roi_w)                                                        # [n, od, P, P]
    hstart = pidx.view(1, 1, P, 1) * v(bin_h) + v(start_h)
    hstart = hstart + ty * v(roi_h)
    sidx = torch.arange(spp, dtype=f32, device=dev)
    w = wstart[..., None, None] + sidx.view(1, 1, 1, 1, 1, spp) * sub_w.view(n, 1, 1, 1, 1, 1)   # [n, od, P, P, 1, spp]
    h = hstart[..., None, None] + sidx.view(1, 1, 1, 1, spp, 1) * sub_h.view(n, 1, 1, 1, 1, 1)   # [n, od, P, P, spp, 1]
    w, h = w.expand(n, output_dim, P, P, spp, spp), h.expand(n, output_dim, P, P, spp, spp)
    wd, hd = w.detach().to(f64), h.detach().to(f64)
    inside = ~((wd < -0.5) | (wd > W - 0.5) | (hd < -0.5) | (hd > H - 0.5))
    wc = w + (wd.clamp(0.0, W - 1.0).to(f32) - w.detach())     # straight-through clamp
    hc = h + (hd.clamp(0.0, H - 1.0).to(f32) - h.detach())
    x1, x2 = torch.floor(wc.detach()).long(), torch.ceil(wc.detach()).long()
    y1, y2 = torch.floor(hc.detach()).long(), torch.ceil(hc.detach()).long()
    dx, dy = (wc - x1.to(f32)).to(f64), (hc - y1.to(f32)).to(f64)
    c = (torch.arange(output_dim, device=dev).view(-1, 1, 1) * group_size + grp.view(1, P, 1)) * group_size + grp.view(1, 1, P)
    b = r[:, 0].long().view(n, 1, 1, 1, 1, 1).expand_as(x1)
    cc = c.view(1, output_dim, P, P, 1, 1).expand_as(x1)
    d = data.to(f64)
    ok = inside
    xs = lambda t_: t_.clamp(0, W - 1)
    ys = lambda t_: t_.clamp(0, H - 1)
    v11, v12 = d[b, cc, ys(y1), xs(x1)], d[b, cc, ys(y2), xs(x1)]
    v21, v22 = d[b, cc, ys(y1), xs(x2)], d[b, cc, ys(y2), xs(x2)]
    val = (1 - dx) * (1 - dy) * v11 + (1 - dx) * dy * v12 + dx * (1 - dy) * v21 + dx * dy * v22
    val = torch.where(ok, val, torch.zeros_like(val))
    count = ok.sum(dim=(4, 5))
    total = val.sum(dim=(4, 5))
    out = torch.where(count > 0, total / count.clamp(min=1).to(f64), torch.zeros_like(total))
    return out, count.to(f64)
