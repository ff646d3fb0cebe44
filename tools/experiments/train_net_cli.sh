set -u
mkdir -p gpurun_out/r5/cli
OUT=gpurun_out/r5/cli/train_net_cli.txt
: > $OUT
for spec in "student_teacher_mask_rcnn_uncertainty::SOLVER.CHECKPOINT_PERIOD 30" "zeroshot_mask:SOLVER.IMS_PER_BATCH 2:"; do
  cfg=${spec%%:*}; rest=${spec#*:}; o1=${rest%%:*}; o2=${rest#*:}
  echo "# python tools/train_net.py --config-file configs/coco_cap_det/$cfg.yaml --max-iter 60 OUTPUT_DIR <dir> SOLVER.LOG_PERIOD 20 $o1 $o2 SOLVER.BASE_LR 0.0001" >> $OUT
  timeout 300 python tools/train_net.py --config-file configs/coco_cap_det/$cfg.yaml --max-iter 60 OUTPUT_DIR gpurun_out/r5/cli/run_$cfg SOLVER.LOG_PERIOD 20 $o1 $o2 SOLVER.BASE_LR 0.0001 2>&1 | grep -v amdgpu.ids | grep "INFO\|Error\|error\|Traceback" | cut -c1-400 >> $OUT
  echo >> $OUT
done
rm -rf gpurun_out/r5/cli/run_*
