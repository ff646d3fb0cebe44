#!/bin/bash
# 500-step loss trajectories, product path vs fp32 convolutions vs perturbed fp32:
#   bash tools/loss_trajectories.sh <outdir> [steps]
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
OUT=${1:-gpurun_out/loss_traj}; STEPS=${2:-500}
mkdir -p "$OUT"
for wl in student teacher; do
  for path in split fp32 fp32_eps; do
    timeout 1500 python tools/experiments/loss_trajectory.py $STEPS $path $wl > "$OUT/${wl}_${path}.json" 2> "$OUT/${wl}_${path}.err" || echo "FAILED $wl $path"
  done
  python tools/experiments/loss_trajectory.py compare "$OUT/${wl}_split.json" "$OUT/${wl}_fp32.json" "$OUT/${wl}_fp32_eps.json" > "$OUT/loss_trajectory_${wl}.txt" || echo "compare failed $wl"
done
tail -5 "$OUT"/loss_trajectory_*.txt
