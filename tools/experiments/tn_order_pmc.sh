#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of split_gemm_tn_kernel for the shipped tile order and the forced-order variants (separate --pmc passes,
# the program directly after `--`):  bash tools/experiments/tn_order_pmc.sh <outdir>
set -u
OUT=${1:-gpurun_out/tn_order}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT; cd /tmp && export TMPDIR=/tmp
for v in shipped tn_gmajor; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tn_${v}_$c
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/tn_${v}_$c -o p -- python3 $ROOT/tools/experiments/tn_order_ab.py child $v > /tmp/tn_${v}_$c.log 2>&1
    f=$(find /tmp/tn_${v}_$c -name "*counter_collection.csv" | head -1)
    echo "== $v $c" >> $ROOT/$OUT/counters.txt
    if [ -n "$f" ]; then python3 $ROOT/tools/pmc_reduce.py $f | grep -i "tn_kernel\|kernel " >> $ROOT/$OUT/counters.txt; else tail -3 /tmp/tn_${v}_$c.log >> $ROOT/$OUT/counters.txt; fi
  done
done
cat $ROOT/$OUT/counters.txt
