"""Idle gaps of the GPU (no kernel of any stream running) from a rocprofv3 --kernel-trace CSV, aggregated by the kernels
that end / start them, over a window of the trace: python tools/gap_report.py trace.csv [lo hi] (fractions of the trace,
or negative numbers = milliseconds before its end)."""
import collections
import csv
import re
import sys

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.35, 0.65)
t0, t1 = rows[0][0], max(e for _, e, _ in rows)
if lo < 0:  # negative arguments: milliseconds before the end of the trace
    a, b = t1 + lo * 1e6, t1 + hi * 1e6
else:
    a, b = t0 + (t1 - t0) * lo, t0 + (t1 - t0) * hi


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([\w:]+(?:<[^(]{0,60})?)", n)
    return (m.group(1) if m else n)[:70]


gaps = collections.defaultdict(lambda: [0, 0.0])
cur_end, cur_name = None, None
total_gap = 0.0
for s, e, n in rows:
    if s < a or s > b:
        if cur_end is not None and s > b:
            break
        if s < a:
            cur_end, cur_name = max(cur_end or 0, e), n if (cur_end is None or e >= cur_end) else cur_name
            continue
    if cur_end is not None and s > cur_end:
        g = (s - cur_end) / 1e3
        if g > 5.0:
            k = (short(cur_name), short(n))
            gaps[k][0] += 1
            gaps[k][1] += g
        total_gap += g
    if cur_end is None or e > cur_end:
        cur_end, cur_name = e, n
print(f"window {(b - a) / 1e6:.1f} ms, idle {total_gap / 1e3:.2f} ms ({100 * total_gap * 1e3 / (b - a):.1f} %)")
for (p, n), (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{g / 1e3:7.2f} ms  x{c:4d}  {p}  ->  {n}")
