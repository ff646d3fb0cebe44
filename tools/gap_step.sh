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
python - $f <<PY
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
t1 = rows[-1][0]
sel = [n for t, n in rows if t1 - 250e6 < t < t1 - 50e6]
# the batched NMS reduce runs exactly once per training step: its count = the number of steps inside the window
steps = sum(1 for n in sel if "nms_reduce" in n)
print(f"launches per step: {len(sel) / max(steps, 1):.0f} ({len(sel)} kernels and {steps} steps in 200 ms of steady state)")
PY
rm -rf $OUT/trace
