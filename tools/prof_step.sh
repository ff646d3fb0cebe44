#!/bin/bash
# rocprofv3 kernel-trace summary of the training-step benchmark: bash tools/prof_step.sh [workload] [outdir]
WL=${1:-student}; OUT=${2:-gpurun_out/prof_step_$WL}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python bench.py --workload $WL --steps 6 --warmup 3 --no-cpu-baseline > $OUT.log 2>&1
grep "^{\"metric\"" $OUT.log | tail -1 | cut -c1-400
python tools/trace_summary.py $OUT/t_kernel_trace.csv --last-ms 700 --top 45 | cut -c1-150
