#!/bin/bash
# SQ counters of the RoIAlign backward micro-benchmark for experiment builds: bash tools/experiments/roi_bwd_pmc.sh <outdir> variant ...
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"
OUT=$1; shift
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > "$OUT/sq_counters_available.txt"
for v in "$@"; do
  LIB="$ROOT/tools/experiments/variants/libovis_hip_$v.so"; [ "$v" = base ] && LIB="$ROOT/cvpr22_cross_modal_pseudo_labeling_amd/libovis_hip.so"
  : > "$OUT/$v.txt"
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SMEM" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SMEM SQ_WAVE32_INSTS" \
             "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT"; do
    i=$((i+1))
    have=""; for c in $grp; do grep -qx "$c" "$OUT/sq_counters_available.txt" || [ "$c" = GRBM_GUI_ACTIVE ] && have="$have $c"; done
    rm -rf /tmp/rbp_$i
    timeout 300 rocprofv3 --kernel-trace --pmc $have --output-format csv -d /tmp/rbp_$i -o p -- python3 $ROOT/tools/bench_ops.py --ops roi_bwd --roi-kinds uniform --iters 3 --lib $LIB > /tmp/rbp_$i.log 2>&1
    f=$(find /tmp/rbp_$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python3 $ROOT/tools/pmc_reduce.py $f | grep roi_bwd_mfma >> "$OUT/$v.txt"; else echo "pass $i: no output" >> "$OUT/$v.txt"; tail -3 /tmp/rbp_$i.log >> "$OUT/$v.txt"; fi
  done
done
