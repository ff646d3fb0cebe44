#!/bin/bash
# Collects the measurement artifacts of a round on the GPU box into gpurun_out/<tag>/ (copy what is to be judged into
# profiles/): bash tools/collect_round.sh <tag>
TAG=${1:-r2}; OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/bench_student_default.json 2> $OUT/bench_student_default.err
python bench.py --steps 60 --warmup 5 --min-seconds 5 --per-shape-csv $OUT/per_shape_student.csv > $OUT/bench_student.json 2> $OUT/bench_student.err
python bench.py --workload teacher --steps 60 --warmup 5 --min-seconds 5 --per-shape-csv $OUT/per_shape_teacher.csv > $OUT/bench_teacher.json 2> $OUT/bench_teacher.err
python bench.py --no-pipeline --no-cpu-baseline --steps 30 > $OUT/bench_student_nopipe.json 2>/dev/null
python tools/bench_ops.py > $OUT/bench_ops.txt 2>&1
python tools/experiments/op_count.py > $OUT/op_count.txt 2>&1
bash tools/prof_step.sh student $OUT/prof_student > $OUT/prof_student.txt 2>&1
bash tools/prof_step.sh teacher $OUT/prof_teacher > $OUT/prof_teacher.txt 2>&1
bash tools/gap_step.sh student $OUT/gap_student > $OUT/gap_student.txt 2>&1
bash tools/prof_op.sh roi_bwd $OUT/prof_roi_bwd > $OUT/prof_roi_bwd.txt 2>&1
bash tools/pmc_step.sh $OUT/pmc_teacher teacher > $OUT/pmc_teacher.log 2>&1
bash tools/pmc_step.sh $OUT/pmc_student student > $OUT/pmc_student.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
ls $OUT
