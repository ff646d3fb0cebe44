#!/bin/bash
# Fill of the machine by the kernels of the SEQUENTIAL step: bash tools/fill_step.sh [workload] [outdir]
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
export TMPDIR=/tmp
WL=${1:-student}; OUT=${2:-gpurun_out/fill_$WL}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --workload $WL --no-pipeline --no-cpu-baseline --secondary-steps 0 --steps 8 --warmup 3 --burn-seconds 1 > $OUT/bench.log 2>&1 || { echo "rocprofv3 / bench.py failed:"; tail -20 $OUT/bench.log; exit 1; }
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] || { echo "no *kernel_trace.csv under $OUT/trace"; tail -20 $OUT/bench.log; exit 1; }
python tools/step_fill_report.py $f | tee $OUT/fill_report.txt | cut -c1-200
rm -rf $OUT/trace
