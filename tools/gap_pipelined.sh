#!/bin/bash
# GPU idle (no kernel of any stream running) of the PIPELINED training step: bash tools/gap_pipelined.sh [outdir]
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/gap_pipelined}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --no-cpu-baseline --secondary-steps 0 --steps 12 --warmup 4 --burn-seconds 1 > $OUT/bench.log 2>&1 || { echo "rocprofv3 / bench.py failed:"; tail -20 $OUT/bench.log; exit 1; }
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] || { echo "no *kernel_trace.csv under $OUT/trace"; tail -20 $OUT/bench.log; exit 1; }
python tools/step_idle_report.py $f | cut -c1-230
# (bench.py replays a few SEQUENTIAL steps after the timed region for its per-kernel figures: the last steps of the trace
#  are those; the pipelined steps are the ones with overlap_ms > 0)
rm -rf $OUT/trace
