"""Kernel launches of ONE steady training step from a rocprofv3 --kernel-trace CSV: the launches between two consecutive
batched-NMS reduce kernels (one per step), by kernel name.  python tools/step_launch_histogram.py trace.csv [top]"""
import collections
import csv
import re
import sys

rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
marks = [i for i, r in enumerate(rows) if "nms_reduce" in r[2]]
spans = sorted((b - a, a, b) for a, b in zip(marks[:-1], marks[1:]))
n, a, b = spans[len(spans) // 4]  # a steady step (the lower quartile: no warm-up extras)
cnt, dur = collections.Counter(), collections.Counter()
for s, e, name in rows[a:b]:
    name = re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", name)
    name = re.sub(r"<.*", "", name)[:60]
    cnt[name] += 1
    dur[name] += (e - s) / 1e3
print(f"{n} launches in the step, {sum(dur.values()) / 1e3:.2f} ms of kernel time")
for name, c in cnt.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    print(f"{c:5d}  {dur[name]:9.1f} us  {name}")
