#!/bin/bash
# rocprofv3 kernel-trace summary of one op micro-benchmark: bash tools/prof_op.sh <op> [outdir]
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
export TMPDIR=/tmp
OP=${1:-roi_bwd}; OUT=${2:-gpurun_out/prof_$OP}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$(dirname "$OUT")"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python tools/bench_ops.py --ops $OP --iters 10 > $OUT.log 2>&1 || { echo "rocprofv3 / bench_ops.py failed:"; tail -20 $OUT.log; exit 1; }
python - <<PY
import csv, glob
for f in glob.glob('$OUT/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage'])
PY
