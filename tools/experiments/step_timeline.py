"""Timeline of ONE steady-state step from a rocprofv3 --kernel-trace CSV: python tools/experiments/step_timeline.py trace.csv [small_us]
Steps are cut at the fused SGD launch.  Kernels of >= small_us (default 20) print one line each (start offset, duration,
queue, name); runs of shorter kernels collapse into one line (count, summed duration, wall span, the names that dominate) --
the spans are where a chain of small launches holds the stream while the machine is mostly empty."""
import collections
import csv
import re
import sys

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
small = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
    m = re.match(r"([\w:]+(?:<[^(]{0,40})?)", n)
    return (m.group(1) if m else n)[:60]


cuts = [i for i, r in enumerate(rows) if "sgd_momentum_multi_kernel" in r[2]]
# a step ends with its last SGD launch: cut where the next SGD launch is > 5 ms away
ends = [i for k, i in enumerate(cuts) if k + 1 == len(cuts) or rows[cuts[k + 1]][0] - rows[i][0] > 5e6]
if len(ends) < 3:
    raise SystemExit("fewer than three steps in the trace")
a, b = ends[-3] + 1, ends[-2] + 1  # the step before the last one
step = rows[a:b]
t0 = step[0][0]
print(f"step of {len(step)} launches, {(max(r[1] for r in step) - t0) / 1e3:.1f} us wall, "
      f"{sum(r[1] - r[0] for r in step) / 1e3:.1f} us summed")
queues = {q: i for i, q in enumerate(sorted({r[3] for r in step}))}
run = []


def flush():
    global run
    if not run:
        return
    names = collections.Counter()
    for s, e, n, q in run:
        names[short(n)] += (e - s) / 1e3
    top = ", ".join(f"{n} {v:.0f}" for n, v in names.most_common(4))
    span = (max(r[1] for r in run) - run[0][0]) / 1e3
    print(f"{(run[0][0] - t0) / 1e3:9.1f}  [{len(run):3d} small: {sum(r[1] - r[0] for r in run) / 1e3:7.1f} us in a span of {span:7.1f}]  {top}")
    run = []


for s, e, n, q in step:
    d = (e - s) / 1e3
    if d < small:
        run.append((s, e, n, q))
        continue
    flush()
    print(f"{(s - t0) / 1e3:9.1f}  {d:8.1f}  q{queues[q]}  {short(n)}")
flush()
