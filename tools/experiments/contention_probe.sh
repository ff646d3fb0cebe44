#!/bin/bash
# Upper bound on the gradient exchange's cost to the backward it overlaps, on ONE GPU: the step beside
# device-to-device copies of the all-reduce payload (student 75 MB, teacher 139 MB; x2: RCCL reads and writes the buckets,
# and a ring moves 2 (N-1)/N of the payload) on a separate stream, alternated with the plain step on the same box.
#   bash tools/experiments/contention_probe.sh out.txt
set -u
out="$1"; : > "$out"
for round in 1 2; do
  for spec in "student 0" "student 75" "student 150" "teacher 0" "teacher 139" "teacher 278"; do
    set -- $spec
    extra=""; [ "$1" = student ] && extra="--secondary-steps 0"
    python bench.py --workload $1 --steps 40 --warmup 8 --no-cpu-baseline --no-roi-micro $extra --contention-copy-mb $2 2>/dev/null \
      | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $round $1 copy_MB_per_step $2 ms_per_step', round(d['ms_per_step'], 3), d['ms_per_step_spread'])" >> "$out"
  done
done
