"""Go / no-go numbers for a Winograd F(2x2, 3x3) form of the res5 head's 3x3 ([R*49, 512] -> 512 on 7x7 maps; 42 % of the step's
GEMM time): what the products alone would cost on the existing split-GEMM kernels, what the transforms would move if they were
separate passes, and the arithmetic error of the scheme (emulated on the host in fp32 with three-term bf16 products).

    python tools/experiments/winograd_probe.py            (GPU box; prints the table profiles/r4_winograd_probe.txt quotes)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def emulated_error():
    """F(2x2, 3x3) on 7x7 maps (4 x 4 tiles of 2 x 2 outputs over the 8 x 8 padded output), transforms in fp32, the 16 per-position
    products as three-term bf16 hi/lo products (emulated: (hi + lo)(hi + lo) - lo * lo in fp64, rounded to fp32) -> max error
    relative to the maximum of an fp64 direct convolution."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    r, c, k = 8, 64, 32
    x = torch.randn(r, c, 7, 7, generator=g)
    w = torch.randn(k, c, 3, 3, generator=g) * 0.05
    ref = F.conv2d(x.double(), w.double(), padding=1)
    bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
    gm = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
    at = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
    xp = F.pad(x, (1, 2, 1, 2))                                   # 10 x 10: tiles of 4 x 4 input at stride 2 -> 4 x 4 tiles
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                    # [r, c, 4, 4, 4, 4]
    v = torch.einsum("ij,rcabjk,lk->rcabil", bt, tiles, bt)      # input transform (adds only)
    u = torch.einsum("ij,kcjl,ml->kcim", gm, w, gm)               # weight transform (once per weight)

    def split(t):
        hi = t.to(torch.bfloat16).float()
        return hi, (t - hi).to(torch.bfloat16).float()

    vh, vl = split(v)
    uh, ul = split(u)
    m = (torch.einsum("rcabil,kcil->rkabil", vh.double(), uh.double()) + torch.einsum("rcabil,kcil->rkabil", vh.double(), ul.double())
         + torch.einsum("rcabil,kcil->rkabil", vl.double(), uh.double())).float()
    y = torch.einsum("ij,rkabjl,ml->rkabim", at, m, at)           # [r, k, 4, 4, 2, 2]
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(r, k, 8, 8)[:, :, :7, :7]
    direct_split = None
    return float((y.double() - ref).abs().max() / ref.abs().max())


def main():
    print(f"arithmetic: F(2x2, 3x3) with fp32 transforms + three-term bf16 products, max error / max |fp64 direct| = {emulated_error():.2e}")
    if not torch.cuda.is_available():
        return
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    def t(fn, it=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / it * 1e3

    r, c, n = 2048, 512, 512
    m = r * 49
    x = _C.split_pair(torch.randn(m, c, device="cuda"))
    w9 = _C.split_pair(torch.randn(n, 9 * c, device="cuda") * 0.02)
    bias = torch.randn(n, device="cuda")
    direct = t(lambda: _C.split_gemm_pair(x, w9, bias, None, True, False, True, conv=(7, 7, 3, 3, False)))
    print(f"direct implicit 3x3 (halo form), [{m} x {c}] 3x3 -> {n}, pair epilogue: {direct:7.1f} us "
          f"({6.0 * m * n * 9 * c / direct / 1e6:5.0f} TFLOP/s issued)")
    mw = r * 16 * 16                                              # 16 tiles per map x 16 transformed positions
    v = _C.split_pair(torch.randn(mw, c, device="cuda"))
    w1 = _C.split_pair(torch.randn(n, c, device="cuda") * 0.02)
    prod = t(lambda: _C.split_gemm_pair(v, w1, None, None, False, True, False))
    print(f"the 16 per-position products as ONE 1x1 product of the same flops, [{mw} x {c}] -> {n}, fp32 result (the output "
          f"transform needs fp32): {prod:7.1f} us ({6.0 * mw * n * c / prod / 1e6:5.0f} TFLOP/s issued; "
          f"{(mw * 1.0) / (m * 9):.2f} of the direct form's MFMAs)")
    vin = torch.randn(mw, c, device="cuda")
    tin = t(lambda: _C.split_pair(vin))
    yout = torch.randn(mw, n, device="cuda")
    tout = t(lambda: _C.bias_act_(yout, bias, None, True))
    print(f"separate transform passes, streaming stand-ins of the same bytes: input transform writes V in pair layout "
          f"([{mw} x {c}] x 4 B = {mw * c * 4 / 1e6:.0f} MB; split_pair over it: {tin:6.1f} us), output transform reads M in fp32 "
          f"([{mw} x {n}] x 4 B = {mw * n * 4 / 1e6:.0f} MB; one pass over it: {tout:6.1f} us)")
    print(f"products + separate transforms >= {prod + tin + tout:7.1f} us vs direct {direct:7.1f} us")


if __name__ == "__main__":
    main()
