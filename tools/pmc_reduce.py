"""Reduce a rocprofv3 counter_collection CSV to per-kernel averages for the ovis kernels."""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name", "")
        if "anonymous namespace" not in name and "ovis" not in name:
            continue
        m = re.search(r"(?:\(anonymous namespace\)::)?(\w+(?:<[^>]*>)?)\(", name.replace("void ", ""))
        short = m.group(1) if m else name[:40]
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    for c, v in cs.items():
        print(f"{k:40s} {c:32s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
