#!/bin/bash
# Per-step GPU idle of the TEACHER training step (no frozen half, no pipeline): bash tools/gap_teacher.sh [outdir]
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/gap_teacher}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --workload teacher --no-cpu-baseline --steps 12 --warmup 4 --burn-seconds 1 > $OUT/bench.log 2>&1 || { echo "rocprofv3 / bench.py failed:"; tail -20 $OUT/bench.log; exit 1; }
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] || { echo "no *kernel_trace.csv under $OUT/trace"; tail -20 $OUT/bench.log; exit 1; }
python tools/step_idle_report.py $f | cut -c1-260
python tools/gap_report.py $f 0.5 0.8 | cut -c1-170 | head -30
rm -rf $OUT/trace
