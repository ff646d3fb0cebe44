"""GPU busy fraction over time from a rocprofv3 --kernel-trace CSV: union of all kernel intervals per window."""
import csv
import sys

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 100e6
t0, t1 = rows[0][0], max(e for _, e in rows)
# merge intervals
merged = []
for s, e in rows:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
n = int((t1 - t0) / win) + 1
busy = [0.0] * n
for s, e in merged:
    i = int((s - t0) / win)
    while s < e:
        edge = t0 + (i + 1) * win
        seg = min(e, edge) - s
        busy[i] += seg
        s += seg
        i += 1
tail = busy[-40:]
print("window ms:", win / 1e6, "busy fraction of the last windows:", " ".join(f"{b / win:.2f}" for b in tail))
