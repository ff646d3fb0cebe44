"""How many batches per second the CLI's input path delivers on its own (SyntheticBatches workers -> DataLoader ->
DevicePrefetcher), and where a batch's time goes (worker generation, the main process' hand-over, pinning + copy):
    python tools/experiments/data_path_rate.py [images_per_batch] [workers]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd.data.prefetch import DevicePrefetcher  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import SyntheticBatches, make_batch  # noqa: E402

ims = int(sys.argv[1]) if len(sys.argv) > 1 else 2
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
t0 = time.perf_counter()
for i in range(5):
    make_batch(ims, device="cpu", seed=i)
print(f"make_batch({ims}) in one process: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms")
loader = torch.utils.data.DataLoader(SyntheticBatches(ims, seed0=1234, rank=0), batch_size=None, num_workers=workers,
                                     prefetch_factor=2, persistent_workers=True)
it = iter(loader)
for _ in range(8):
    next(it)
t0 = time.perf_counter()
for _ in range(40):
    b = next(it)
print(f"DataLoader alone ({workers} workers): {(time.perf_counter() - t0) / 40 * 1e3:.1f} ms per batch")
t0 = time.perf_counter()
for _ in range(10):
    pinned = b[0].pin_memory()
print(f"pin_memory of the image tensor ({b[0].numel() * 4 / 1e6:.1f} MB): {(time.perf_counter() - t0) / 10 * 1e3:.1f} ms")
del it
data = DevicePrefetcher(loader, "cuda", depth=2)
for _ in range(8):
    next(data)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    next(data)
torch.cuda.synchronize()
print(f"DataLoader + DevicePrefetcher: {(time.perf_counter() - t0) / 40 * 1e3:.1f} ms per batch")
data.close()
