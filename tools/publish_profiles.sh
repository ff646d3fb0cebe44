#!/bin/bash
# Copies the artifacts tools/collect_round.sh wrote to gpurun_out/<tag>/ into profiles/ under their round-2 names:
#   bash tools/publish_profiles.sh <tag> [prefix]          (here, after the gpurun call has merged gpurun_out/)
TAG=${1:?tag}; P=${2:-r2}; S=gpurun_out/$TAG; D=profiles
cpf() { [ -f "$1" ] && cp "$1" "$2" && echo "$2"; }
cpf $S/bench_student_default.json $D/${P}_bench_student_default.json
cpf $S/bench_student.json $D/${P}_bench_student.json
cpf $S/bench_teacher.json $D/${P}_bench_teacher.json
cpf $S/bench_student_nopipe.json $D/${P}_bench_student_nopipe.json
cpf $S/bench_student_resident20.json $D/${P}_bench_student_resident20.json
cpf $S/per_shape_student.csv $D/${P}_split_gemm_per_shape_student.csv
cpf $S/per_shape_teacher.csv $D/${P}_split_gemm_per_shape_teacher.csv
cpf $S/bench_ops.txt $D/${P}_bench_ops.txt
cpf $S/op_count.txt $D/${P}_op_count.txt
cpf $S/prof_student.txt $D/${P}_step_student_top_kernels.txt
cpf $S/prof_teacher.txt $D/${P}_step_teacher_top_kernels.txt
cpf $S/prof_student/t_kernel_stats.csv $D/${P}_step_student_kernel_stats_incl_warmup.csv
cpf $S/prof_teacher/t_kernel_stats.csv $D/${P}_step_teacher_kernel_stats_incl_warmup.csv
cpf $S/gap_student.txt $D/${P}_gap_report_student_nopipe.txt
cpf $S/prof_roi_bwd/t_kernel_stats.csv $D/${P}_roi_bwd_kernel_stats.csv
cpf $S/pmc_teacher/hbm_traffic_teacher.json $D/${P}_pmc_step_hbm_traffic_teacher.json
cpf $S/pmc_student/hbm_traffic_student.json $D/${P}_pmc_step_hbm_traffic_student.json
cpf $S/pmc_mfma_student/mfma_busy_student.json $D/${P}_pmc_mfma_busy_student.json
cpf $S/pmc_mfma_teacher/mfma_busy_teacher.json $D/${P}_pmc_mfma_busy_teacher.json
cpf $S/fill_student.txt $D/${P}_step_fill_report_student.txt
cpf $S/fill_teacher.txt $D/${P}_step_fill_report_teacher.txt
