"""Why does next(loader) take 39 ms per 2-image batch once the consumer stages the batch, when it takes 6.5 ms if the batch is
dropped unread?  One loader, four consumers in turn: drop / read the bytes / copy to the device / sleep 30 ms."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import SyntheticBatches  # noqa: E402


def main():
    torch.zeros(1, device="cuda")
    loader = torch.utils.data.DataLoader(SyntheticBatches(2, seed0=1234, rank=0), batch_size=None, num_workers=4, prefetch_factor=2,
                                         persistent_workers=True)
    it = iter(loader)
    for _ in range(8):
        next(it)
    for mode in ("drop", "read", "to_device", "sleep30", "drop"):
        tn = tc = 0.0
        for _ in range(30):
            t0 = time.perf_counter()
            b = next(it)
            t1 = time.perf_counter()
            if mode == "read":
                b[0].sum()
            elif mode == "to_device":
                b[0].to("cuda", non_blocking=True)
                torch.cuda.synchronize()
            elif mode == "sleep30":
                time.sleep(0.03)
            t2 = time.perf_counter()
            tn += t1 - t0
            tc += t2 - t1
        print(f"{mode:10s} next(loader) {tn / 30 * 1e3:6.1f} ms   consumer {tc / 30 * 1e3:6.1f} ms", flush=True)


if __name__ == "__main__":
    main()
