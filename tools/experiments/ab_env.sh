#!/bin/bash
# Same-box A/B of an environment switch: bash tools/experiments/ab_env.sh "VAR=value" [bench args...]
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"
cd "$ROOT"
SW=$1; shift
ARGS=("$@"); [ ${#ARGS[@]} -gt 0 ] || ARGS=(--steps 40 --warmup 8 --no-cpu-baseline)
one() { timeout 600 env "$@" python bench.py "${ARGS[@]}" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('%.3f ms/step  median chunk %.3f' % (d['ms_per_step'], d.get('ms_per_step_spread', {}).get('median', 0)))"; }
for i in 0 1 2; do
  echo "round $i  shipped : $(one OVIS_AB_NONE=1)"
  echo "round $i  $SW: $(one "$SW")"
done
