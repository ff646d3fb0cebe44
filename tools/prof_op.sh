#!/bin/bash
# rocprofv3 kernel-trace summary of one op micro-benchmark: bash tools/prof_op.sh <op> [outdir]
OP=${1:-roi_bwd}; OUT=${2:-gpurun_out/prof_$OP}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python tools/bench_ops.py --ops $OP --iters 10 > /dev/null 2>&1
python - <<PY
import csv, glob
for f in glob.glob('$OUT/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage'])
PY
