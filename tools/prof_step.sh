#!/bin/bash
# rocprofv3 kernel-trace summary of the training-step benchmark: bash tools/prof_step.sh [workload] [outdir]
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
export TMPDIR=/tmp
WL=${1:-student}; OUT=${2:-gpurun_out/prof_step_$WL}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$(dirname "$OUT")"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python bench.py --workload $WL --steps 6 --warmup 3 --no-cpu-baseline --secondary-steps 0 > $OUT.log 2>&1 || { echo "rocprofv3 / bench.py failed:"; tail -20 $OUT.log; exit 1; }
grep "^{\"metric\"" $OUT.log | tail -1 | cut -c1-400
python tools/trace_summary.py $OUT/t_kernel_trace.csv --last-ms 700 --top 45 | cut -c1-150
