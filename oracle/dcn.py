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
