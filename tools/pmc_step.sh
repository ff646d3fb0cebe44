#!/bin/bash
# HBM traffic of the hand-written kernels inside the benchmark step: one rocprofv3 --pmc pass per counter (FETCH_SIZE and
# WRITE_SIZE do not fit one pass; --pmc is only ever combined with --kernel-trace) over a short sequential run of
# bench.py, reduced per kernel family by tools/pmc_step_reduce.py.  Usage on the GPU box:
#   bash tools/pmc_step.sh <outdir> [student|teacher]
set -u
OUT=${1:-gpurun_out/pmc_step}; WL=${2:-student}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcstep_$c -o p -- python3 $ROOT/bench.py --workload $WL --steps 2 --warmup 2 --no-cpu-baseline --no-pipeline --burn-seconds 0 --secondary-steps 0 > /tmp/pmcstep_$c.log 2>&1
  f=$(find /tmp/pmcstep_$c -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pass $c: no output"; tail -5 /tmp/pmcstep_$c.log; exit 1; fi
  cp $f $ROOT/$OUT/$c.csv
done
python3 $ROOT/tools/pmc_step_reduce.py $ROOT/$OUT/FETCH_SIZE.csv $ROOT/$OUT/WRITE_SIZE.csv $WL > $ROOT/$OUT/hbm_traffic_$WL.json
rm -f $ROOT/$OUT/FETCH_SIZE.csv $ROOT/$OUT/WRITE_SIZE.csv
cat $ROOT/$OUT/hbm_traffic_$WL.json
