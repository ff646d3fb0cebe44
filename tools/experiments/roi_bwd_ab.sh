#!/bin/bash
# Same-box A/B of RoIAlign-backward builds on the 856.5 MB micro-benchmark: every variant library under
# tools/experiments/variants named on the command line, interleaved three times (box clocks drift).
#   bash tools/experiments/roi_bwd_ab.sh out.txt r5 s1f32 ...      ("shipped" = the in-tree library)
set -u
out="$1"; shift
: > "$out"
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = shipped ]; then lib=""; else lib="--lib tools/experiments/variants/libovis_hip_$v.so"; fi
    python tools/bench_ops.py --ops ${OPS:-roi_bwd,roi_bwd_strided} --iters 300 $lib 2>/dev/null | sed "s/^/round $round $v /" >> "$out"
  done
done
