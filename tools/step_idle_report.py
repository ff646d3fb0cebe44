"""Per-step GPU idle time from a rocprofv3 --kernel-trace CSV of a bench.py run: steps are cut at the batched NMS reduce
(one launch per training step); per step: wall time, time with at least one kernel running (union over all streams), time
with kernels of two streams overlapping, idle time and the largest idle gaps with the kernels around them.
    python tools/step_idle_report.py trace.csv [first_step last_step]"""
import csv
import re
import sys

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([\w:]+(?:<[^(]{0,40})?)", n)
    return (m.group(1) if m else n)[:48]


marks = [i for i, (_, _, n) in enumerate(rows) if "nms_reduce" in n]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) - 1
print(f"{len(marks)} steps in the trace; step = from one batched NMS reduce to the next")
print(" step   wall_ms  busy_ms  overlap_ms  idle_ms  idle_%   largest gaps (ms: kernel before -> kernel after)")
for si in range(lo, min(hi, len(marks) - 1)):
    seg = rows[marks[si]:marks[si + 1] + 1]
    t0, t1 = seg[0][0], seg[-1][0]
    busy = overlap = 0
    cur_s, cur_e, cur_n = seg[0]
    gaps = []
    for s, e, n in seg[1:]:
        if s > t1:
            break
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, short(cur_n), short(n)))
            cur_s, cur_e, cur_n = s, e, n
        else:
            overlap += min(e, cur_e) - s
            if e > cur_e:
                cur_e, cur_n = e, n
    busy += min(cur_e, t1) - cur_s
    wall = t1 - t0
    idle = wall - busy
    gaps.sort(reverse=True)
    top = "; ".join(f"{g / 1e6:.2f}: {a} -> {b}" for g, a, b in gaps[:3])
    print(f"{si:5d}  {wall / 1e6:8.2f} {busy / 1e6:8.2f} {overlap / 1e6:10.2f} {idle / 1e6:8.2f} {100.0 * idle / max(wall, 1):6.1f}   {top}")
