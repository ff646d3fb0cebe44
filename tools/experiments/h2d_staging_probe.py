"""Which way of getting a DataLoader batch (tensors in shared memory, first touched by this process) onto the device is cheapest
for the staging thread: per-tensor pin_memory().to(), copy into a reused pinned buffer then to(), or to() straight from the
pageable tensor.  python tools/experiments/h2d_staging_probe.py [images_per_batch]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import SyntheticBatches  # noqa: E402

ims = int(sys.argv[1]) if len(sys.argv) > 1 else 2
loader = torch.utils.data.DataLoader(SyntheticBatches(ims, seed0=1, rank=0), batch_size=None, num_workers=4, prefetch_factor=2,
                                     persistent_workers=True)
it = iter(loader)


def tensors(b):
    out = [b[0]]
    for t in b[1]:
        out.append(t.bbox)
        out += [v for v in t.extra_fields.values() if torch.is_tensor(v)]
    return out


for _ in range(6):
    next(it)
big = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
stream = torch.cuda.Stream()
res = {}
for name in ("touch only (sum of first bytes per page)", "pin_memory().to()", "copy_ into reused pinned, to()", "pageable .to()"):
    ts = []
    for _ in range(8):
        b = next(it)
        tl = tensors(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            if name.startswith("touch"):
                for t in tl:
                    t.view(-1).view(torch.uint8)[::4096].sum()
            elif name.startswith("pin_memory"):
                d = [t.pin_memory().to("cuda", non_blocking=True) for t in tl]
            elif name.startswith("copy_"):
                off, d = 0, []
                for t in tl:
                    n = t.numel() * t.element_size()
                    v = big[off:off + n].view(t.dtype).view(t.shape)
                    v.copy_(t)
                    d.append(v.to("cuda", non_blocking=True))
                    off += (n + 63) // 64 * 64
            else:
                d = [t.to("cuda", non_blocking=True) for t in tl]
        t1 = time.perf_counter()
        stream.synchronize()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    res[name] = ts
    print(f"{name:45s} host-side {1e3 * sorted(x[0] for x in ts)[len(ts) // 2]:6.1f} ms   until on device {1e3 * sorted(x[1] for x in ts)[len(ts) // 2]:6.1f} ms"
          f"   ({sum(t.numel() * t.element_size() for t in tl) / 1e6:.1f} MB in {len(tl)} tensors)")
