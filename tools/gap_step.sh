#!/bin/bash
# GPU idle gaps of the sequential (un-pipelined) training step: bash tools/gap_step.sh [workload] [outdir]
# rocprofv3 kernel trace of a short bench.py --no-pipeline run -> tools/gap_report.py over the steady state + launches per step
WL=${1:-student}; OUT=${2:-gpurun_out/gap_$WL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --workload $WL --no-pipeline --no-cpu-baseline --steps 8 --warmup 3 --burn-seconds 1 > $OUT/bench.log 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/gap_report.py $f -250 -50 | cut -c1-170
python - $f $OUT/bench.log <<PY
import csv, json, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1])))
t1 = max(e for _, e in rows)
sel = [r for r in rows if t1 - 250e6 < r[0] < t1 - 50e6]
ms = json.loads([l for l in open(sys.argv[2]) if l.startswith('{"metric')][-1])["ms_per_step"]
print(f"launches per step: {len(sel) * ms / 200:.0f} ({len(sel)} kernels in 200 ms of steady state, {ms:.1f} ms per step under the profiler)")
PY
rm -rf $OUT/trace
