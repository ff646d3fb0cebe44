#!/bin/bash
# Same-box comparison of bench.py in several source trees (sub-directories of the repository, each built):
#   bash tools/experiments/ab_trees.sh "<tree> <tree> ..." [bench args...]      ("." = this tree)
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"
TREES=$1; shift
ARGS=("$@"); [ ${#ARGS[@]} -gt 0 ] || ARGS=(--steps 40 --warmup 8 --no-cpu-baseline)
one() { (cd "$ROOT/$1" && timeout 600 python bench.py "${ARGS[@]}" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('%.3f ms/step  median chunk %.3f' % (d['ms_per_step'], d.get('ms_per_step_spread', {}).get('median', 0)))"); }
for i in 0 1; do
  for t in $TREES; do echo "round $i  $t: $(one "$t")"; done
done
