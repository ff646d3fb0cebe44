"""Probe: fp32-accurate GEMM on the bf16 matrix pipe via hi/lo split and K-concatenation, through hipBLASLt
(torch.mm with out_dtype) -- speed and error vs the fp32 GEMM for the res5 1x1-conv shapes."""
import torch, time
dev = "cuda"
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
def split3(x):  # [M,K] f32 -> [M,3K] bf16 = [hi | hi | lo]
    hi = x.to(torch.bfloat16); lo = (x - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, hi, lo], 1)
def split3b(w):  # [N,K] f32 -> [N,3K] bf16 = [hi | lo | hi]
    hi = w.to(torch.bfloat16); lo = (w - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo, hi], 1)
for (M, K, N) in [(100352, 1024, 512), (100352, 512, 2048), (100352, 2048, 512), (100352, 1024, 2048), (50176, 512, 2048)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
    ref = (x.double() @ w.double().t())
    f32 = x @ w.t()
    ms32 = t(lambda: x @ w.t())
    try:
        xs, ws = split3(x), split3b(w)
        got = torch.mm(xs, ws.t(), out_dtype=torch.float32)
        msb = t(lambda: torch.mm(xs, ws.t(), out_dtype=torch.float32))
        msprep = t(lambda: split3(x))
        e32 = ((f32.double() - ref).abs().max() / ref.abs().max()).item(); eb = ((got.double() - ref).abs().max() / ref.abs().max()).item()
        print(f"M={M} K={K} N={N}: f32 {ms32:.3f} ms ({2*M*K*N/ms32/1e9:.0f} TF)  bf16x3 gemm {msb:.3f} ms ({2*M*K*N/msb/1e9:.0f} TF-equiv) prep {msprep:.3f} ms  err f32 {e32:.2e} bf16x3 {eb:.2e}")
    except Exception as e:
        print("bf16x3 failed:", repr(e)[:300])
