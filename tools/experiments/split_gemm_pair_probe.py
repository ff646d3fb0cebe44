"""Pair-layout split GEMM (csrc/split_gemm.hip): correctness vs fp64 and speed vs the K-concatenated hipBLASLt route."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cvpr22_cross_modal_pseudo_labeling_amd import _C


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(True)
    b = torch.cuda.Event(True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def check_plain(m, k, n, bias=False, res=False, relu=False, tile_m=0):
    g = torch.Generator(device="cuda").manual_seed(m * 7 + k + n)
    a = torch.randn(m, k, device="cuda", generator=g)
    b = torch.randn(n, k, device="cuda", generator=g)
    bi = torch.randn(n, device="cuda", generator=g) if bias else None
    r = torch.randn(m, n, device="cuda", generator=g) if res else None
    c, cp = _C.split_gemm_pair(_C.split_pair(a), _C.split_pair(b), bi, r, relu, True, n % 32 == 0, tile_m=tile_m)
    ref = a.double() @ b.double().t()
    if bias:
        ref += bi.double()
    if res:
        ref += r.double()
    if relu:
        ref = ref.clamp(min=0)
    bound = (a.abs().double() @ b.abs().double().t()) + 1
    err = ((c.double() - ref).abs() / bound).max().item()
    msg = f"plain m={m} k={k} n={n} bias={bias} res={res} relu={relu} tile={tile_m}: err/bound {err:.2e}"
    if cp is not None:
        # the pair output must equal the pair split of the fp32 output bit for bit
        same = torch.equal(cp, _C.split_pair(c))
        msg += f" pair_out_equal={same}"
        assert same
    print(msg)
    assert err < 2e-5, msg


def check_conv(r, h, w, c, n, kh=3, kw=3, flip=False, tile_m=0):
    g = torch.Generator(device="cuda").manual_seed(r + h * 3 + w * 5 + c)
    x = torch.randn(r, h, w, c, device="cuda", generator=g)
    wt = torch.randn(n, c, kh, kw, device="cuda", generator=g)
    wm = wt.permute(0, 2, 3, 1).reshape(n, kh * kw * c).contiguous()
    y, _ = _C.split_gemm_pair(_C.split_pair(x.view(-1, c)), _C.split_pair(wm), conv=(h, w, kh, kw, flip), tile_m=tile_m)
    wref = wt.flip(2, 3) if flip else wt
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wref.double(), padding=(kh // 2, kw // 2)).permute(0, 2, 3, 1).reshape(-1, n)
    bound = F.conv2d(x.permute(0, 3, 1, 2).abs().double(), wref.abs().double(), padding=(kh // 2, kw // 2)).permute(0, 2, 3, 1).reshape(-1, n) + 1
    err = ((y.double() - ref).abs() / bound).max().item()
    print(f"conv r={r} {h}x{w} c={c} n={n} k={kh}x{kw} flip={flip} tile={tile_m}: err/bound {err:.2e}")
    assert err < 2e-5
    # pair im2col against the GEMM of its rows
    rows = _C.im2col_pair(_C.split_pair(x.view(-1, c)), h, w, kh, kw)
    y2, _ = _C.split_gemm_pair(rows, _C.split_pair(wm))
    if not flip:
        assert torch.equal(y, y2) or (y - y2).abs().max().item() < 1e-4 * y.abs().max().item(), "im2col_pair mismatch"


def bench_plain(m, k, n, tag):
    a = torch.randn(m, k, device="cuda")
    b = torch.randn(n, k, device="cuda")
    ap, bp = _C.split_pair(a), _C.split_pair(b)
    a3, b3 = _C.split_bf16x3(a, 0), _C.split_bf16x3(b, 1)
    fl = 2.0 * m * n * k
    out = []
    for tm in (128, 1128, 71128):
        ms = t(lambda: _C.split_gemm_pair(ap, bp, tile_m=tm))
        out.append(f"tile{tm} {ms:.3f} ms {fl / ms / 1e9:.0f} TF")
    def padded(x, pad):
        buf = torch.empty(x.shape[0], x.shape[1] + pad, dtype=x.dtype, device=x.device)
        buf[:, :x.shape[1]] = x
        return buf[:, :x.shape[1]]
    for pad in ():
        app, bpp = padded(ap, pad), padded(bp, pad)
        ms = t(lambda: _C.split_gemm_pair(app, bpp, tile_m=128))
        out.append(f"pad{2 * pad}B {ms:.3f} ms {fl / ms / 1e9:.0f} TF")
        ms = t(lambda: _C.split_gemm_pair(app, bpp, tile_m=20128))
        out.append(f"pad{2 * pad}B-noMFMA {ms:.3f} ms")
    ms_l = t(lambda: torch.mm(a3, b3.t(), out_dtype=torch.float32))
    ms_s3 = t(lambda: _C.split_bf16x3(a, 0))
    ms_sp = t(lambda: _C.split_pair(a))
    print(f"{tag} [{m}x{k}]x[{n}x{k}]: " + " | ".join(out) + f" | hipBLASLt-3K {ms_l:.3f} ms {fl / ms_l / 1e9:.0f} TF"
          f" | split3 {ms_s3:.3f} ms, split_pair {ms_sp:.3f} ms ({8 * a.numel() / ms_sp / 1e6:.0f} GB/s)")


def bench_conv(r, h, w, c, n, tag):
    x = torch.randn(r, h, w, c, device="cuda")
    wm = torch.randn(n, 9 * c, device="cuda")
    xp, wp = _C.split_pair(x.view(-1, c)), _C.split_pair(wm)
    w3 = _C.split_bf16x3(wm, 1)
    fl = 2.0 * r * h * w * n * 9 * c
    out = []
    for tm in (128, 1128, 71128):
        ms = t(lambda: _C.split_gemm_pair(xp, wp, conv=(h, w, 3, 3, False), tile_m=tm))
        out.append(f"tile{tm} {ms:.3f} ms {fl / ms / 1e9:.0f} TF")
    ms_i = t(lambda: _C.im2col_split_bf16x3(x, 3, 3))
    rows = _C.im2col_split_bf16x3(x, 3, 3)
    ms_l = t(lambda: torch.mm(rows, w3.t(), out_dtype=torch.float32))
    print(f"{tag} conv3x3 [{r},{h},{w},{c}]->{n}: " + " | ".join(out) + f" | im2col {ms_i:.3f} + hipBLASLt {ms_l:.3f} ms "
          f"({fl / (ms_i + ms_l) / 1e9:.0f} TF)")


def check_tn(m, n, ch, conv=None):
    g = torch.Generator(device="cuda").manual_seed(m + n + ch)
    dy = torch.randn(m, n, device="cuda", generator=g)
    x = torch.randn(m, ch, device="cuda", generator=g)
    dw = _C.split_gemm_pair_tn(_C.split_pair(dy), _C.split_pair(x), conv)
    if conv is None:
        ref = dy.double().t() @ x.double()
        bound = dy.abs().double().t() @ x.abs().double() + 1
    else:
        h, w, kh, kw = conv
        r = m // (h * w)
        cols = F.unfold(x.view(r, h, w, ch).permute(0, 3, 1, 2).double(), (kh, kw), padding=(kh // 2, kw // 2))  # [r, ch*T, hw]
        cols = cols.view(r, ch, kh * kw, h * w).permute(0, 3, 2, 1).reshape(m, kh * kw * ch)                      # tap-major
        ref = dy.double().t() @ cols
        bound = dy.abs().double().t() @ cols.abs() + 1
    err = ((dw.double() - ref).abs() / bound).max().item()
    print(f"tn m={m} n={n} ch={ch} conv={conv}: err/bound {err:.2e}")
    assert err < 2e-5


def bench_tn(m, n, ch, conv, tag):
    from cvpr22_cross_modal_pseudo_labeling_amd.layers.pair_bottleneck import dw_pair
    gp = _C.split_pair(torch.randn(m, n, device="cuda"))
    xp = _C.split_pair(torch.randn(m, ch, device="cuda"))
    taps = 1 if conv is None else conv[2] * conv[3]
    fl = 2.0 * m * n * ch * taps
    ms = t(lambda: _C.split_gemm_pair_tn(gp, xp, conv))
    if conv is None:
        ms_l = t(lambda: dw_pair(gp, xp))
    else:
        ms_l = t(lambda: dw_pair(gp, _C.im2col_pair(xp, *conv)))
    print(f"{tag} dW m={m} n={n} ch={ch} conv={conv}: tn {ms:.3f} ms {fl / ms / 1e9:.0f} TF | hipBLASLt quadrant route {ms_l:.3f} ms {fl / ms_l / 1e9:.0f} TF")


if __name__ == "__main__":
    torch.manual_seed(0)
    check_tn(64, 128, 128)
    check_tn(1000, 128, 256)
    check_tn(49 * 37, 256, 128, (7, 7, 3, 3))
    check_tn(5 * 9 * 13, 128, 128, (9, 13, 3, 5))
    check_tn(49 * 300, 512, 512, (7, 7, 3, 3))
    R = int(os.environ.get("PROBE_R", "1024"))
    bench_tn(R * 49, 512, 1024, None, "b0 conv1")
    bench_tn(R * 49, 2048, 1024, None, "b0 shortcut")
    bench_tn(R * 49, 2048, 512, None, "conv3")
    bench_tn(R * 49, 512, 2048, None, "b1 conv1")
    bench_tn(R * 49, 512, 512, (7, 7, 3, 3), "conv2")
    for tm in (128, 1128, 71128):
        check_plain(128, 32, 128, tile_m=tm)
        check_plain(300, 64, 64, tile_m=tm)
        check_plain(1000, 512, 192, bias=True, res=True, relu=True, tile_m=tm)
        check_plain(49 * 37, 1024, 512, bias=True, relu=True, tile_m=tm)
        check_plain(513, 2048, 2048, res=True, tile_m=tm)
        check_conv(5, 7, 7, 64, 64, tile_m=tm)
        check_conv(37, 7, 7, 512, 512, tile_m=tm)
        check_conv(3, 5, 9, 32, 128, flip=True, tile_m=tm)
        check_conv(2, 13, 11, 64, 96, kh=3, kw=5, tile_m=tm)
        check_conv(1, 50, 84, 256, 256, flip=True, tile_m=tm)
    print("correctness ok")
    R = int(os.environ.get("PROBE_R", "1024"))
    bench_plain(R * 49, 1024, 512, "res5 b0 conv1")
    bench_plain(R * 49, 1024, 2048, "res5 b0 shortcut")
    bench_plain(R * 49, 512, 2048, "res5 conv3")
    bench_plain(R * 49, 2048, 512, "res5 b1 conv1")
    bench_conv(R, 7, 7, 512, 512, "res5")
    bench_plain(2 * 200 * 334, 64, 64, "layer1 conv1")
    bench_plain(2 * 200 * 334, 64, 256, "layer1 conv3")
    bench_plain(2 * 200 * 334, 256, 64, "layer1 b1 conv1")
    bench_conv(2, 200, 334, 64, 64, "layer1")
    bench_plain(2 * 100 * 167, 128, 512, "layer2 conv3")
    bench_conv(2, 100, 167, 128, 128, "layer2")
    bench_plain(2 * 50 * 84, 1024, 256, "layer3 conv1")
    bench_plain(2 * 50 * 84, 256, 1024, "layer3 conv3")
    bench_conv(2, 50, 84, 256, 256, "layer3")
