#!/bin/bash
# GPU idle gaps of the sequential (un-pipelined) training step: bash tools/gap_step.sh [workload] [outdir]
# rocprofv3 kernel trace of a short bench.py --no-pipeline run -> tools/gap_report.py over the steady state + launches per step
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
export TMPDIR=/tmp
WL=${1:-student}; OUT=${2:-gpurun_out/gap_$WL}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac   # relative paths are relative to the repository root, whatever the caller's cwd
mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --workload $WL --no-pipeline --no-cpu-baseline --secondary-steps 0 --steps 8 --warmup 3 --burn-seconds 1 > $OUT/bench.log 2>&1 || { echo "rocprofv3 / bench.py failed:"; tail -20 $OUT/bench.log; exit 1; }
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] || { echo "no *kernel_trace.csv under $OUT/trace"; tail -20 $OUT/bench.log; exit 1; }
python tools/gap_report.py $f -250 -50 | cut -c1-170
python - $f <<PY
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
# the batched NMS reduce runs exactly once per training step: the kernels between two consecutive ones are one step
marks = [i for i, (t, n) in enumerate(rows) if "nms_reduce" in n]
per_step = sorted(b - a for a, b in zip(marks[:-1], marks[1:]))[: max(len(marks) - 1, 1)]
steady = per_step[: max(1, len(per_step) // 2)]  # the smaller half: steps without warm-up / calibration extras
print(f"launches per step: {steady[len(steady) // 2]} (median over the steady steps; {len(marks)} steps in the trace)")
PY
python tools/step_launch_histogram.py $f 45
python tools/step_big_small_kernels.py $f 40
rm -rf $OUT/trace
