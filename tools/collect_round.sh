#!/bin/bash
# Collects the measurement artifacts of a round on the GPU box into gpurun_out/<tag>/ (copy what is to be judged into
# profiles/): bash tools/collect_round.sh <tag>
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
TAG=${1:-r4}; OUT=gpurun_out/$TAG
mkdir -p $OUT
FAILED=""
step() { "$@" || { FAILED="$FAILED [$*]"; echo "collect_round: FAILED: $*" >&2; }; }   # keep collecting, report at the end
step bash -c 'python bench.py --steps 20 --warmup 5 > "$0"/bench_student_default.json 2> "$0"/bench_student_default.err' "$OUT"
step bash -c 'python bench.py --secondary-steps 0 --steps 60 --warmup 5 --min-seconds 5 --per-shape-csv "$0"/per_shape_student.csv > "$0"/bench_student.json 2> "$0"/bench_student.err' "$OUT"
step bash -c 'python bench.py --workload teacher --steps 60 --warmup 5 --min-seconds 5 --per-shape-csv "$0"/per_shape_teacher.csv > "$0"/bench_teacher.json 2> "$0"/bench_teacher.err' "$OUT"
step bash -c 'python bench.py --no-pipeline --no-cpu-baseline --secondary-steps 0 --steps 30 > "$0"/bench_student_nopipe.json 2>/dev/null' "$OUT"
# the rounds-1..4 protocol (one device batch throughout, 20 steps) beside the current default, so that lines stay comparable
step bash -c 'python bench.py --resident-input --steps 20 --warmup 5 --no-cpu-baseline --no-roi-micro --secondary-steps 0 > "$0"/bench_student_resident20.json 2>/dev/null' "$OUT"
step bash -c 'python tools/bench_ops.py > "$0"/bench_ops.txt 2>&1' "$OUT"
step bash -c 'python tools/experiments/op_count.py > "$0"/op_count.txt 2>&1' "$OUT"
step bash -c 'bash tools/prof_step.sh student "$0"/prof_student > "$0"/prof_student.txt 2>&1' "$OUT"
step bash -c 'bash tools/prof_step.sh teacher "$0"/prof_teacher > "$0"/prof_teacher.txt 2>&1' "$OUT"
step bash -c 'bash tools/gap_step.sh student "$0"/gap_student > "$0"/gap_student.txt 2>&1' "$OUT"
step bash -c 'bash tools/fill_step.sh student "$0"/fill_student > "$0"/fill_student.txt 2>&1' "$OUT"
step bash -c 'bash tools/fill_step.sh teacher "$0"/fill_teacher > "$0"/fill_teacher.txt 2>&1' "$OUT"
step bash -c 'bash tools/prof_op.sh roi_bwd "$0"/prof_roi_bwd > "$0"/prof_roi_bwd.txt 2>&1' "$OUT"
step bash -c 'bash tools/pmc_step.sh "$0"/pmc_teacher teacher > "$0"/pmc_teacher.log 2>&1' "$OUT"
step bash -c 'bash tools/pmc_step.sh "$0"/pmc_student student > "$0"/pmc_student.log 2>&1' "$OUT"
step bash -c 'bash tools/pmc_mfma.sh "$0"/pmc_mfma_student student > "$0"/pmc_mfma_student.log 2>&1' "$OUT"
step bash -c 'bash tools/pmc_mfma.sh "$0"/pmc_mfma_teacher teacher > "$0"/pmc_mfma_teacher.log 2>&1' "$OUT"
find $OUT -name "*kernel_trace.csv" -delete
ls $OUT
[ -z "$FAILED" ] || { echo "collect_round: steps that failed:$FAILED" >&2; exit 1; }
