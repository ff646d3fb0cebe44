#!/bin/bash
# Same-box comparison of several environment settings: bash tools/experiments/ab_envs.sh [--workload teacher] "A=1" "B=2" ...
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"
cd "$ROOT"
WL=()
if [ "${1:-}" = "--workload" ]; then WL=(--workload "$2"); shift 2; fi
one() { timeout 600 env "$@" python bench.py "${WL[@]}" --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('%.3f ms/step  median chunk %.3f' % (d['ms_per_step'], d.get('ms_per_step_spread', {}).get('median', 0)))"; }
for i in 0 1; do
  for sw in "$@"; do echo "round $i  $sw: $(one "$sw")"; done
done
