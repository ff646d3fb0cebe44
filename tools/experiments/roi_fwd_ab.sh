#!/bin/bash
# Same-box A/B of RoIAlign-forward builds on the 856.5 MB micro-benchmark (see roi_bwd_ab.sh).
#   bash tools/experiments/roi_fwd_ab.sh out.txt fwd_r5 shipped
set -u
out="$1"; shift
: > "$out"
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = shipped ]; then lib=""; else lib="--lib tools/experiments/variants/libovis_hip_$v.so"; fi
    python tools/bench_ops.py --ops roi_fwd --iters 200 $lib 2>/dev/null | grep '"roi_align_forward"' | sed "s/^/round $round $v /" >> "$out"
  done
done
