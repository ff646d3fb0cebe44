#!/bin/bash
# GPU idle (no kernel of any stream running) of the PIPELINED training step: bash tools/gap_pipelined.sh [outdir]
OUT=${1:-gpurun_out/gap_pipelined}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --no-cpu-baseline --steps 12 --warmup 4 --burn-seconds 1 > $OUT/bench.log 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/gap_report.py $f -300 -60 | cut -c1-170 | head -24
rm -rf $OUT/trace
