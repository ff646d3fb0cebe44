"""Summarise a rocprofv3 --kernel-trace CSV over its LAST `--last-ms` milliseconds (i.e. the timed,
steady-state region of bench.py, excluding MIOpen's find/tuning launches during warm-up).

    python tools/trace_summary.py <kernel_trace.csv> --last-ms 800 [--top 40] > summary.csv
"""
import argparse
import csv
import collections
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--last-ms", type=float, default=1e12)
    ap.add_argument("--top", type=int, default=50)
    args = ap.parse_args()
    rows = []
    with open(args.trace, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    if not rows:
        sys.exit("empty trace")
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(args.last_ms * 1e6)
    agg = collections.defaultdict(lambda: [0, 0, 1 << 62, 0])
    total = 0
    for s, e, n in rows:
        if s < t0:
            continue
        a = agg[n]
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
        total += d
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[: args.top]:
        w.writerow([n[:160], a[0], a[1], round(a[1] / a[0], 1), round(100.0 * a[1] / total, 3), a[2], a[3]])
    w.writerow(["TOTAL_KERNEL_TIME", sum(a[0] for a in agg.values()), total, "", 100.0, "", ""])


if __name__ == "__main__":
    main()
