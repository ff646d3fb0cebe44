import torch
from torch.profiler import profile, ProfilerActivity
out = torch.randn(1024 * 49, 2048, device="cuda", requires_grad=True)
x = out.view(1024, 7, 7, 2048).permute(0, 3, 1, 2)
p = x.mean(dim=(2, 3))
g = torch.randn_like(p)
for _ in range(2):
    out.grad = None
    p = out.view(1024, 7, 7, 2048).permute(0, 3, 1, 2).mean(dim=(2, 3))
    p.backward(g)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    out.grad = None
    p = out.view(1024, 7, 7, 2048).permute(0, 3, 1, 2).mean(dim=(2, 3))
    p.backward(g)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=70))
print(out.grad.is_contiguous(), out.grad.stride())
