"""Per-launch HBM traffic of the hand-written kernel families from two rocprofv3 counter_collection CSVs (FETCH_SIZE,
WRITE_SIZE; KiB units).  gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies the 128-byte requests of
wide (16 B / lane) coalesced reads at 64 B, so the read side is doubled: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvpr22_cross_modal_pseudo_labeling_amd.utils import provenance  # noqa: E402
OURS = set()  # __global__ functions of csrc/*.hip: the hand-written kernels (torch / rocPRIM kernels are left out)
for src in glob.glob(os.path.join(ROOT, "cvpr22_cross_modal_pseudo_labeling_amd", "csrc", "*.hip")):
    OURS.update(re.findall(r"__global__[^;{]*?void\s+(\w+)\s*\(", open(src).read(), re.S))


INSTANCES = set()  # template instances the profiled run launched, as the profiler printed them


def load(path):
    agg = collections.defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name", "")
            if "anonymous namespace" not in name:
                continue
            m = re.search(r"(?:\(anonymous namespace\)::)?(\w+)(?:<[^>]*>)?\(", name.replace("void ", ""))
            if m and m.group(1) in OURS:
                agg[m.group(1)].append(float(r["Counter_Value"]))
                full = re.match(r"(\w+(?:<[^(]*>)?)\(", name.replace("void ", "").replace("(anonymous namespace)::", ""))
                INSTANCES.add(full.group(1) if full else m.group(1))
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {"workload": sys.argv[3] if len(sys.argv) > 3 else "student",
       "command": "bash tools/pmc_step.sh (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py "
                  "--steps 2 --warmup 2 --no-pipeline), all launches of the run",
       "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: wide reads are tallied at half their size)",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, []), write.get(k, [])
    n = max(len(f), len(w))
    if n == 0:
        continue
    fb, wb = 2 * 1024 * sum(f) / max(len(f), 1), 1024 * sum(w) / max(len(w), 1)
    out["kernels"][k] = {"launches": n, "read_MB_per_launch": round(fb / 1e6, 3), "write_MB_per_launch": round(wb / 1e6, 3),
                         "hbm_MB_per_launch": round((fb + wb) / 1e6, 3)}
out["provenance"] = provenance.stamp(INSTANCES)
print(json.dumps(out, indent=1))
