"""DataLoader worker start method when the training process has ALREADY initialised the GPU (it always has: the model is on the
device before the loader starts): forked workers inherit the parent's GPU mappings and their memory operations crawl.
    python tools/experiments/loader_context_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import SyntheticBatches  # noqa: E402


def main():
    torch.zeros(1, device="cuda")
    for ctx in ("fork", "spawn", "forkserver"):
        loader = torch.utils.data.DataLoader(SyntheticBatches(2, seed0=1234, rank=0), batch_size=None, num_workers=4, prefetch_factor=2,
                                             persistent_workers=True, multiprocessing_context=ctx)
        it = iter(loader)
        for _ in range(8):
            next(it)
        t0 = time.perf_counter()
        for _ in range(40):
            next(it)
        print(f"{ctx:10s} workers started after GPU init: {(time.perf_counter() - t0) / 40 * 1e3:.1f} ms per 2-image batch", flush=True)
        del it, loader


if __name__ == "__main__":
    main()
