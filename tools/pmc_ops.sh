#!/bin/bash
# Collect PMC counters for the hand-written kernels (one rocprofv3 pass per counter group; --pmc is never
# combined with any trace domain other than --kernel-trace).  Usage on the GPU box:
#   bash tools/pmc_ops.sh <outdir> <ops> [iters]
set -u
OUT=${1:-gpurun_out/pmc}; OPS=${2:-roi_fwd}; ITERS=${3:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_$i -o p -- python3 $ROOT/tools/bench_ops.py --ops $OPS --iters $ITERS > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $ROOT/tools/pmc_reduce.py $f >> $ROOT/$OUT/counters.txt; else echo "pass $i ($grp): no output" >> $ROOT/$OUT/counters.txt; tail -3 /tmp/pmc_$i.log >> $ROOT/$OUT/counters.txt; fi
done
cat $ROOT/$OUT/counters.txt
